#!/bin/bash
# paired-chunk loop (SG2IM_X2) A/B: parity, layer table, step - all variants on one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
L=$PWD/sg2im_amd/lib
timeout 300 python -m pytest tests -m gpu -x -q -k "without_any_triples" 2>&1 | tail -3 | tee gpurun_out/c14_pytest_edge.log
for v in _x2_3 _x2_7; do
  echo "== parity $v"; SG2IM_LIB=$L/libsg2im_hip$v.so timeout 300 python tools/gpu_check.py sec_conv sec_linear sec_gconv 2>&1 | grep -v amdgpu.ids | tail -4
done 2>&1 | tee gpurun_out/c14_parity.log
for v in "" _x2_1 _x2_2 _x2_3 _x2_7; do
  echo "== layers [$v]"; SG2IM_LIB=$L/libsg2im_hip$v.so timeout 300 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids | tail -32
done > gpurun_out/c14_layers.log 2>&1
grep "==\|TOTAL" gpurun_out/c14_layers.log | tail -20
for rep in 1 2; do
for v in "" _x2_1 _x2_2 _x2_3 _x2_7; do
  SG2IM_LIB=$L/libsg2im_hip$v.so timeout 300 python bench.py --steps 48 --warmup 16 --no_roofline --cpu_baseline_steps 0 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant [$v]', d['ms_per_step'], d['value'])"
done
done | tee gpurun_out/c14_bench.log
