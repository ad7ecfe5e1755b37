#!/bin/bash
# round 2, GPU call 1: graph-fault root cause probes, new bucketing tests, full GPU suite, bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== C repro (torch-free)" > gpurun_out/c1_fault.log
for m in none lib_null lib_same lib_other lib_conv own_null; do
  for i in "" init; do
    timeout 60 tools/_bin/graph_fault_repro $m $i >> gpurun_out/c1_fault.log 2>&1; echo "   rc=$? ($m $i)" >> gpurun_out/c1_fault.log
  done
done
echo "== python probe" >> gpurun_out/c1_fault.log
for m in control sigmoid sigmoid_s conv torchop step; do
  timeout 180 python tools/graph_fault_probe.py $m 2>&1 | grep -v "^$" | tail -4 >> gpurun_out/c1_fault.log; echo "   rc=${PIPESTATUS[0]} ($m)" >> gpurun_out/c1_fault.log
done
cat gpurun_out/c1_fault.log
timeout 900 python -m pytest tests -m gpu -x -q -k "padded or bucketed or graph or reproducible or rccl or train_script" 2>&1 | tail -25 > gpurun_out/c1_pytest_new.log
cat gpurun_out/c1_pytest_new.log
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/c1_pytest_all.log
cat gpurun_out/c1_pytest_all.log
timeout 600 python bench.py --steps 48 --warmup 16 > gpurun_out/c1_bench.json 2> gpurun_out/c1_bench.err
tail -c 6000 gpurun_out/c1_bench.json; tail -5 gpurun_out/c1_bench.err
timeout 300 python bench.py --steps 48 --warmup 16 --n_batches 1 --no_roofline --cpu_baseline_steps 0 > gpurun_out/c1_bench_1batch.json 2>> gpurun_out/c1_bench.err
tail -c 1500 gpurun_out/c1_bench_1batch.json
