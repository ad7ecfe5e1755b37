#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
SG2IM_V2_MIN=0 timeout 300 python tools/gpu_check.py sec_conv 2>&1 | grep -v amdgpu.ids | grep "v2\|====\|BAD\|raised\|Error" | tee gpurun_out/c11_v2_conv.log | tail -40
timeout 900 python -m pytest tests -m gpu -x -q -k "direct_to_lds or trainer_two_steps or golden or conv_forward" 2>&1 | tail -6
C=gpurun_out/c11_conv.log
: > $C
for v in 0 1; do
  echo "== SG2IM_V2=$v" >> $C
  SG2IM_V2=$v timeout 300 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids >> $C
done
cat $C
for v in 0 1; do
  echo "== bench SG2IM_V2=$v"
  SG2IM_V2=$v timeout 300 python bench.py --steps 48 --warmup 16 --no_roofline --cpu_baseline_steps 0 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done 2>&1 | tee gpurun_out/c11_bench.log
