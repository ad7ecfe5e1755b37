#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
for fl in "--force_dist" "--eval_generator" "--no_graphs --steps 10 --warmup 3" "--force_dist --dtype bf16"; do
  timeout 600 python bench.py --steps 30 --warmup 8 --cpu_baseline_steps 0 --no_roofline $fl 2>gpurun_out/r4_call18.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$fl', d['ms_per_step'], d['value'], d.get('gradient_exchange', {}).get('allreduce_ms'), d.get('gradient_exchange', {}).get('dp_schedule'))" || tail -5 gpurun_out/r4_call18.err
done 2>&1 | tee gpurun_out/r4_bench_variants.log
