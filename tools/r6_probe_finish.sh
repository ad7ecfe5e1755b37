#!/bin/bash
# upper bound of removing split-K finish launches (timing only, wrong numerics): a probe build of the library
# (-DSG2IM_PROBE_SKIP_FINISH, lib/libsg2im_hip_probe.so) with SG2IM_SKIP_FINISH_PROBE = 0 | 2 (weight gradients') | 1 (all)
cd $GRAFT_REPO_ROOT
export SG2IM_LIB=$GRAFT_REPO_ROOT/sg2im_amd/lib/libsg2im_hip_probe.so
for rep in 1 2; do for m in 0 2 1; do
  SG2IM_SKIP_FINISH_PROBE=$m python bench.py --steps 60 --warmup 20 --cpu_baseline_steps 0 --no_roofline 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('skip-finish mode $m:', d['ms_per_step'], 'ms/step')"
done; done
