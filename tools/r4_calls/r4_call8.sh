#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 0 1; do
  echo "=== SG2IM_HALO_B2=$v"
  SG2IM_HALO_B2=$v timeout 300 python tools/bench_conv.py --only=m0,m1,m2,m3,m4,out,mask 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r4_halo_b2_layers.log 2>&1
SG2IM_HALO_B2=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "conv_forward_dgrad or fused_batchnorm" 2>&1 | tail -5 >> gpurun_out/r4_halo_b2_layers.log
for v in 0 1; do
  SG2IM_HALO_B2=$v timeout 600 python bench.py --steps 40 --warmup 10 --cpu_baseline_steps 0 --no_roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HALO_B2=$v', d['ms_per_step'], d['value'])" >> gpurun_out/r4_halo_b2_layers.log
done
cat gpurun_out/r4_halo_b2_layers.log
