#!/bin/bash
# A/B of two library builds: parity of the current one, layer table and step for both (one box)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
L=$PWD/sg2im_amd/lib
timeout 300 python tools/gpu_check.py sec_conv sec_linear sec_gconv sec_golden_coco sec_golden_vg 2>&1 | grep -v amdgpu.ids | tail -2 | tee gpurun_out/c31_parity.log
for v in _v2 ""; do
  echo "== layers [$v]"; SG2IM_LIB=$L/libsg2im_hip$v.so timeout 300 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids | tail -27
done > gpurun_out/c31_layers.log 2>&1
grep "==\|TOTAL" gpurun_out/c31_layers.log
b() { timeout 300 python bench.py --steps 48 --warmup 16 --no_roofline --cpu_baseline_steps 0 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['value'])"; }
for rep in 1 2; do
  SG2IM_LIB=$L/libsg2im_hip_v2.so b "previous build"
  b "current build"
done | tee gpurun_out/c31_bench.log
