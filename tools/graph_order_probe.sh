#!/bin/bash
# tools/graph_order_probe.sh OUT
out=$GRAFT_REPO_ROOT/$1
cd /tmp && export TMPDIR=/tmp
{
rm -rf /tmp/gp; PROBE_LONG=mm rocprofv3 --kernel-trace -d /tmp/gp -o gp -- python $GRAFT_REPO_ROOT/tools/graph_order_probe.py 300
python $GRAFT_REPO_ROOT/tools/prof_queues.py $(find /tmp/gp -name "*.db" | head -1)
} > $out 2>&1
grep -v "^\[" $out | tail -60
