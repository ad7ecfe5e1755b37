#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_call28.log
: > $L
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "graph_replay or bit_reproducible or trainer_two_steps" 2>&1 | tail -3 >> $L
b() { python bench.py --steps 40 --warmup 10 --cpu_baseline_steps 0 --no_roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'])" >> $L; }
SG2IM_ADAM_FIRST_BUCKET=0 b "first-bucket-on-lane"
SG2IM_ADAM_FIRST_BUCKET=1 b "first-bucket-on-main"
SG2IM_ADAM_FIRST_BUCKET=0 b "first-bucket-on-lane"
SG2IM_ADAM_FIRST_BUCKET=1 b "first-bucket-on-main"
SG2IM_MARKS=1 python bench.py --steps 40 --warmup 10 --cpu_baseline_steps 0 --no_roofline 2>&1 >/dev/null | grep '\[mark\]' >> $L
cat $L
