#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
rm -rf /tmp/kt
rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/bench.py --steps 12 --warmup 4 --cpu_baseline_steps 0 --no_roofline > /tmp/kt.log 2>&1
DB=$(find /tmp/kt -name "*.db" | head -1)
python $R/tools/prof_step_dump.py $DB 3 > $O/c28_step_dump.txt
head -3 $O/c28_step_dump.txt; wc -l $O/c28_step_dump.txt
