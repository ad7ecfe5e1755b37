#!/bin/bash
# shared discriminator pass over the generated images + four-lane few-channel data gradient + embedding CSRs ahead of time
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_call21.log
: > $L
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "shared_pass or conv_forward_dgrad or trainer_two_steps or graph_replay or golden or bit_reproducible or in_graph_exchange or without_a_discriminator or build_cnn_arch" 2>&1 | tail -8 >> $L
b() { python bench.py --steps 40 --warmup 10 --cpu_baseline_steps 0 --no_roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$1', d['ms_per_step'], d['value'], r.get('launches_per_step_all_kernels'), r.get('launches_per_step'), r.get('frac'))" >> $L; }
SG2IM_SHARE_FAKE_PASS=0 SG2IM_FEWC_SCALAR=1 b "share0 scalar"
SG2IM_SHARE_FAKE_PASS=1 SG2IM_FEWC_SCALAR=1 b "share1 scalar"
SG2IM_SHARE_FAKE_PASS=1 b "share1 quad  "
SG2IM_SHARE_FAKE_PASS=0 SG2IM_FEWC_SCALAR=1 b "share0 scalar"
SG2IM_SHARE_FAKE_PASS=1 b "share1 quad  "
python bench.py --steps 40 --warmup 10 --cpu_baseline_steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('full line', d['ms_per_step'], d['value'], r.get('launches_per_step_all_kernels'), r.get('launches_per_step'), r.get('frac'))" >> $L
SG2IM_MARKS=1 python bench.py --steps 40 --warmup 10 --cpu_baseline_steps 0 --no_roofline 2>&1 >/dev/null | grep '\[mark\]' >> $L
timeout 300 python tools/bench_conv.py --only=d_img,d_obj 2>&1 | grep -v amdgpu.ids >> $L
cat $L
