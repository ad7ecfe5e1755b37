#!/bin/bash
# usage: tools/pmc_one.sh LAYER  -> per-kernel PMC sums for one conv layer (forward pass), three passes
set -e
L=${1:-m4.conv0}
X=${2:-}          # extra bench_one.py flags, e.g. bf16
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_LEVEL_WAVES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
         "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$i -o p -- python $R/tools/bench_one.py $L 6 $X > /tmp/pmc_$i.log 2>&1 || tail -5 /tmp/pmc_$i.log
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('/tmp/pmc_*/**/*counter_collection.csv', recursive=True):
  for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0][:60]
    if 'conv_' not in k: continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] in ('SQ_WAVES', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_ANY', 'SQ_INSTS_LDS'): cnt[(k, r['Counter_Name'])] += 1
for k, d in acc.items():
  print(k)
  for c, v in sorted(d.items()):
    print('   %-28s %16.0f' % (c, v))
PY
