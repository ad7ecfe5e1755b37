#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
C=gpurun_out/c6_conv.log
: > $C
for v in _old ""; do
  echo "== variant '$v' (old = no scheduling fence / register laundering)" >> $C
  SG2IM_LIB=$PWD/sg2im_amd/lib/libsg2im_hip$v.so timeout 300 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids >> $C
done
cat $C
timeout 600 python -m pytest tests -m gpu -x -q -k "conv_forward or linear_layers or golden_coco or trainer_two_steps" 2>&1 | tail -5
for v in _old ""; do
  echo "== bench variant '$v'"
  SG2IM_LIB=$PWD/sg2im_amd/lib/libsg2im_hip$v.so timeout 300 python bench.py --steps 48 --warmup 16 --no_roofline --cpu_baseline_steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done 2>&1 | tee gpurun_out/c6_bench.log
