#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_call24.log
: > $L
K="shared_pass or conv_forward_dgrad or trainer_two_steps or graph_replay or golden or bit_reproducible or in_graph_exchange or without_a_discriminator or build_cnn_arch"
run() { echo "=== [$1]" >> $L; env $1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "$K" > gpurun_out/r4_call24_last.txt 2>&1; rc=$?; echo "rc=$rc" >> $L; grep -v "^  File\|^$" gpurun_out/r4_call24_last.txt | tail -4 >> $L; return $rc; }
run "A=1" || run "SG2IM_SHARE_FAKE_PASS=0" || run "SG2IM_EMB_CSR_AHEAD=0" || run "SG2IM_FEWC_SCALAR=1"
cat $L
