#!/bin/bash
# round 2, GPU call 2: which round-1 behaviour produced the replay fault + staggered-start GEMM variants
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/c2_fault.log
echo "== probes with round-1 behaviours re-enabled" > $L
for env in "SG2IM_PROBE_LANES_IN_CAPTURE=1" "SG2IM_PROBE_NO_INIT=1" "SG2IM_PROBE_EAGER_WARMUP=1" "SG2IM_PROBE_LANES_IN_CAPTURE=1 SG2IM_PROBE_NO_INIT=1 SG2IM_PROBE_EAGER_WARMUP=1"; do
  for m in sigmoid step; do
    echo "-- $env mode=$m" >> $L
    env $env timeout 180 python tools/graph_fault_probe.py $m 2>&1 | grep -v "^$" | tail -3 >> $L; echo "   rc=${PIPESTATUS[0]}" >> $L
  done
done
cat $L
C=gpurun_out/c2_conv.log
: > $C
for v in "" _st16 _st32 _st48 _st32hw; do
  echo "== variant '$v'" >> $C
  SG2IM_LIB=$PWD/sg2im_amd/lib/libsg2im_hip$v.so timeout 300 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids >> $C
done
cat $C
for v in "" _st32 _st48; do
  echo "== bench variant '$v'"
  SG2IM_LIB=$PWD/sg2im_amd/lib/libsg2im_hip$v.so timeout 300 python bench.py --steps 48 --warmup 16 --no_roofline --cpu_baseline_steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done 2>&1 | tee gpurun_out/c2_bench.log
