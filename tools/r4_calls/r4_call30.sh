#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_call30.log
: > $L
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "two_linear_heads or lanes_of_a_trainer or golden_coco or golden_vg or trainer_two_steps or graph_replay" 2>&1 | grep -v "^  File" | tail -6 >> $L
b() { python bench.py --steps 40 --warmup 10 --cpu_baseline_steps 0 --no_roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'])" >> $L; }
SG2IM_TWO_HEADS=0 b "two GEMM heads "
SG2IM_TWO_HEADS=1 b "one-launch heads"
SG2IM_TWO_HEADS=0 b "two GEMM heads "
SG2IM_TWO_HEADS=1 b "one-launch heads"
SG2IM_MARKS=1 python bench.py --steps 40 --warmup 10 --cpu_baseline_steps 0 --no_roofline 2>&1 >/dev/null | grep '\[mark\]' >> $L
cat $L
