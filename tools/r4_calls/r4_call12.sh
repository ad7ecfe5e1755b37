#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in "SG2IM_LATE_HEADS=0 SG2IM_LAZY_LAYOUT_GRAD=1" "SG2IM_LATE_HEADS=1 SG2IM_LAZY_LAYOUT_GRAD=0" "SG2IM_LATE_HEADS=0 SG2IM_LAZY_LAYOUT_GRAD=0 SG2IM_GCN_PERSIST=0"; do
  echo "== $cfg"; env $cfg timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "graph_replay_matches" 2>&1 | grep "AssertionError\|passed\|failed" | tail -3
done
