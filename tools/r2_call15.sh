#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
rm -rf /tmp/kt
rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/bench.py --steps 20 --warmup 4 --cpu_baseline_steps 0 --no_roofline > /tmp/kt.log 2>&1
DB=$(find /tmp/kt -name "*.db" | head -1)
python - <<PY
import sqlite3
c=sqlite3.connect("$DB")
print([r[1] for r in c.execute("pragma table_info(kernels)").fetchall()])
PY
python $R/tools/prof_timeline.py $DB | tee $O/c15_timeline.txt
