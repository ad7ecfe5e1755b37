#!/bin/bash
# round 4, GPU call 1: the persistent GraphTripleConv forward + the new trust tests, then an A/B of the step
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "=== stack test"; timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "stack_persistent or graph_triple_conv_layer or empty_and_ragged" 2>&1 | tail -25
echo "=== exchange tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "in_graph_exchange or rccl_path" 2>&1 | tail -25
echo "=== trainer tests"; timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "trainer_two_steps or padded_batch_step" 2>&1 | tail -15
} > gpurun_out/r4_call1_tests.log 2>&1
for p in 1 0; do
  SG2IM_GCN_PERSIST=$p SG2IM_MARKS=1 timeout 600 python bench.py --steps 40 --warmup 10 --cpu_baseline_steps 0 --no_roofline > gpurun_out/r4_call1_bench_persist$p.json 2> gpurun_out/r4_call1_bench_persist$p.err
done
tail -3 gpurun_out/r4_call1_bench_persist*.json
grep -h "mark" gpurun_out/r4_call1_bench_persist1.err | head -40
cat gpurun_out/r4_call1_tests.log
