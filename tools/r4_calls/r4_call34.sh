#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_call34.log
: > $L
SG2IM_WGRAD_HALO_ROWS=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "weight_gradient_halo or conv_forward_dgrad or fused_batchnorm_reductions" 2>&1 | grep -v "^  File" | tail -4 >> $L
for v in 0 1; do
  echo "=== SG2IM_WGRAD_HALO_ROWS=$v" >> $L
  SG2IM_WGRAD_HALO_ROWS=$v timeout 300 python tools/bench_conv.py --only=m0,m1,m2,m3,m4,out,mask 2>&1 | grep -v amdgpu.ids >> $L
done
b() { python bench.py --steps 40 --warmup 10 --cpu_baseline_steps 0 --no_roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'])" >> $L; }
SG2IM_WGRAD_HALO_ROWS=0 b "all taps per workgroup"
SG2IM_WGRAD_HALO_ROWS=1 b "split by kernel row   "
SG2IM_WGRAD_HALO_ROWS=0 b "all taps per workgroup"
SG2IM_WGRAD_HALO_ROWS=1 b "split by kernel row   "
cat $L
