#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_call23.log
: > $L
run() { echo "=== $1 [$2]" >> $L; env $2 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "$1" 2>&1 | grep -v "^  File\|^$" | tail -6 >> $L; }
run "bit_reproducible" "A=1"
run "in_graph_exchange or bit_reproducible" "A=1"
run "in_graph_exchange or bit_reproducible" "SG2IM_SHARE_FAKE_PASS=0"
run "in_graph_exchange or bit_reproducible" "A=1"
cat $L
