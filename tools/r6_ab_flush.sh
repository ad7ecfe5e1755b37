#!/bin/bash
# early release of the queued refinement weight gradients at module i (SG2IM_WGRAD_FLUSH_AT), A/B in ONE call
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in -1 3 2 1; do
  for dt in f32 bf16; do
    SG2IM_WGRAD_FLUSH_AT=$v python bench.py --steps 50 --warmup 10 --cpu_baseline_steps 0 --no_roofline --dtype $dt 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[flush_at=$v] $dt coco', d['ms_per_step'])"
  done
done; done
for v in -1 2; do for dt in bf16; do
  echo "== marks flush_at=$v $dt"
  SG2IM_WGRAD_FLUSH_AT=$v SG2IM_MARKS=1 python bench.py --steps 20 --warmup 5 --cpu_baseline_steps 0 --no_roofline --dtype $dt 2>&1 | grep "^\[mark\]"
done; done
