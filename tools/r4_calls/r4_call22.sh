#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_call22.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -v -k "shared_pass or conv_forward_dgrad or trainer_two_steps or graph_replay or golden or bit_reproducible or in_graph_exchange or without_a_discriminator or build_cnn_arch" > $L 2>&1
grep -n "PASSED\|FAILED\|ERROR\|Fatal\|fault\|Abort\|Segmentation" $L | head -40
grep -n "Fatal Python error" -A25 $L | head -60
