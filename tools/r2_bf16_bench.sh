#!/bin/bash
# secondary bench lines: bf16 operand path, COCO-64 and VG-64 workloads (profiles/r2_bench_bf16_*.json)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for st in coco vg; do
timeout 300 python bench.py --steps 48 --warmup 16 --dtype bf16 --style $st --cpu_baseline_steps 0 2>/dev/null | grep '^{"metric' > gpurun_out/bf16_bench_$st.json; python -c "
import json; d=json.load(open('gpurun_out/bf16_bench_$st.json')); r=d['roofline']; print('$st', d['ms_per_step'], d['value'], r['achieved'], r['frac'], r['crn_only'])"
done
timeout 300 python bench.py --steps 48 --warmup 16 --style vg --cpu_baseline_steps 0 --no_roofline 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('vg f32', d['ms_per_step'], d['value'])"
