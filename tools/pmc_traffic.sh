#!/bin/bash
# usage: tools/pmc_traffic.sh LAYER -> FETCH_SIZE / WRITE_SIZE per launch of the conv kernels of one layer
# (separate --pmc passes, as MI355X_MICROARCH.md prescribes; FETCH_SIZE is doubled per its gfx950 note)
L=${1:-m4.conv0}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -o p -- python $R/tools/bench_one.py $L 6 > /tmp/pmc_$C.log 2>&1 || tail -5 /tmp/pmc_$C.log
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob('/tmp/pmc_*/**/*counter_collection.csv', recursive=True):
  for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].replace('void sg2im::', '')[:48]
    if 'conv_' not in k and 'splitk' not in k: continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
print('layer $L: per-launch memory-side traffic (rocprofv3 units: KB)')
for k, d in acc.items():
  f = d.get('FETCH_SIZE', 0) / max(1, n[(k, 'FETCH_SIZE')]); w = d.get('WRITE_SIZE', 0) / max(1, n[(k, 'WRITE_SIZE')])
  print('%-48s FETCH_SIZE %10.0f KB  (x2 per the gfx950 note: %8.1f MB)   WRITE_SIZE %10.0f KB (%7.1f MB)' % (k, f, 2 * f / 1024, w, w / 1024))
PY
