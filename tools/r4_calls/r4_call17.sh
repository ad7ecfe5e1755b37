#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/lk
rocprofv3 --kernel-trace --stats -d /tmp/lk -o lk -- python $R/tools/bench_layout.py > /tmp/lk.log 2>&1
DB=$(find /tmp/lk -name "*.db" | head -1)
python $R/tools/prof_summary.py $DB 1 2>&1 | head -14
