#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
for pl in f32 bf16; do
  SG2IM_GRAD_PAYLOAD=$pl SG2IM_MARKS=1 timeout 600 python bench.py --steps 30 --warmup 8 --cpu_baseline_steps 0 --no_roofline --force_dist --dtype bf16 2>gpurun_out/r4_call19_$pl.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('payload $pl', d['ms_per_step'], d['value'], d.get('gradient_exchange'))"
  grep -h "mark" gpurun_out/r4_call19_$pl.err | awk '{printf "%s %s | ", $2, $3} END{print ""}'
done
