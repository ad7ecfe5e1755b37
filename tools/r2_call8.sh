#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/gpu_check.py sec_conv_bf16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c8_conv_bf16.log | tail -50
timeout 900 python -m pytest tests -m gpu -x -q -k "bf16" -s 2>&1 | grep -v "^$" | tail -25 | tee gpurun_out/c8_pytest_bf16.log
for st in coco vg; do
timeout 300 python bench.py --steps 48 --warmup 16 --dtype bf16 --style $st --cpu_baseline_steps 0 2>/dev/null | tail -1 > gpurun_out/c8_bench_bf16_$st.json; python -c "
import json; d=json.load(open('gpurun_out/c8_bench_bf16_$st.json')); r=d['roofline']; print('$st', d['ms_per_step'], d['value'], r['achieved'], r['frac'], r['crn_only'], r['by_kind'])"
done
