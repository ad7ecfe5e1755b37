#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_dp_rccl.py -x -q -m gpu -k "in_graph_exchange or rccl_path or two_ranks_on_one_gpu" > gpurun_out/r4_call20.log 2>&1
grep -E "passed|failed|Error" gpurun_out/r4_call20.log | tail -5
