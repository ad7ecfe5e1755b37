#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/gcn_stack_probe.py > gpurun_out/r4_gcn_probe2.log 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "stack_persistent" > gpurun_out/r4_call3_tests.log 2>&1
SG2IM_MARKS=1 timeout 600 python bench.py --steps 40 --warmup 10 --cpu_baseline_steps 0 --no_roofline > gpurun_out/r4_call3_bench.json 2> gpurun_out/r4_call3_bench.err
cat gpurun_out/r4_gcn_probe2.log; tail -n 5 gpurun_out/r4_call3_tests.log; grep -h "mark\|gcn-stamps" gpurun_out/r4_call3_bench.err | head -30; python -c "
import json; d=json.loads(open('gpurun_out/r4_call3_bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
