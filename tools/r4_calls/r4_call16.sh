#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "layout" 2>&1 | tail -3
cd /tmp
rm -rf /tmp/lk
rocprofv3 --kernel-trace --stats -d /tmp/lk -o lk -- python $GRAFT_REPO_ROOT/tools/bench_layout.py > /tmp/lk.log 2>&1
grep -v amdgpu /tmp/lk.log | tail -4
DB=$(find /tmp/lk -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/prof_summary.py $DB 1 2>&1 | head -9
