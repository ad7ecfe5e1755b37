#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_data_loaders.py tests/test_gpu_parity.py -x -q -m gpu -k "real_coco_loaders or in_graph_exchange" > gpurun_out/r4_call10_tests.log 2>&1
tail -n 12 gpurun_out/r4_call10_tests.log
