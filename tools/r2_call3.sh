#!/bin/bash
# round 2, GPU call 3: ping-pong GEMM kernels A/B + parity, align_corners, config 0, full suite, bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "ping_pong or align_corners or config0 or conv_forward or linear_layers" 2>&1 | tail -15 > gpurun_out/c3_pytest_new.log
cat gpurun_out/c3_pytest_new.log
C=gpurun_out/c3_conv.log
: > $C
for pp in 0 1; do
  for mn in 1.0 0.4; do
    [ $pp = 0 ] && [ $mn != 1.0 ] && continue
    echo "== SG2IM_PP=$pp SG2IM_PP_MIN=$mn" >> $C
    SG2IM_PP=$pp SG2IM_PP_MIN=$mn timeout 300 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids >> $C
  done
done
cat $C
for pp in 0 1; do
  echo "== bench SG2IM_PP=$pp"
  SG2IM_PP=$pp timeout 300 python bench.py --steps 48 --warmup 16 --no_roofline --cpu_baseline_steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done 2>&1 | tee gpurun_out/c3_bench.log
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/c3_pytest_all.log
cat gpurun_out/c3_pytest_all.log
