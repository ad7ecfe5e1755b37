#!/bin/bash
# Memory-side traffic of the implicit-GEMM family over whole training steps: separate
# `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (as MI355X_MICROARCH.md prescribes; no tracing
# domains besides the kernel trace), eager launches, 8 steps.  Writes gpurun_out/pmc_step.json,
# which is copied to profiles/ and read by bench.py for roofline.traffic.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
STEPS=6; WARM=2
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcs_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmcs_$C -o p -- \
    python $R/bench.py --steps $STEPS --warmup $WARM --no_graphs --cpu_baseline_steps 0 --no_roofline > /tmp/pmcs_$C.log 2>&1 || tail -5 /tmp/pmcs_$C.log
done
python - <<PY
import csv, glob, collections, json
steps = $STEPS + $WARM
tot = collections.defaultdict(float); launches = collections.Counter()
for f in glob.glob('/tmp/pmcs_*/**/*counter_collection.csv', recursive=True):
  for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'conv_fwd' not in k and 'conv_dgrad' not in k and 'conv_wgrad' not in k and 'conv_halo' not in k and 'splitk_finish' not in k and 'gcn_stack' not in k and 'two_heads' not in k and 'conv1x1_fewout' not in k: continue
    tot[r['Counter_Name']] += float(r['Counter_Value']); launches[r['Counter_Name']] += 1
n = launches['FETCH_SIZE'] / steps
fetch_mb = 2 * tot['FETCH_SIZE'] / 1024 / steps        # KB units; x2: the guide's gfx950 FETCH_SIZE correction
write_mb = tot['WRITE_SIZE'] / 1024 / steps
out = {'steps': steps, 'launches_per_step': n, 'fetch_mb_per_step': round(fetch_mb, 1), 'write_mb_per_step': round(write_mb, 1),
       'traffic_mb_per_launch': round((fetch_mb + write_mb) / max(n, 1), 2),
       'method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (KB units, FETCH_SIZE x2 on gfx950), '
                 'summed over conv_fwd/dgrad/wgrad/halo/1x1-fewout + splitk_finish + gcn_stack + two_heads kernels of %d eager steps' % steps}
print(json.dumps(out))
open('$R/gpurun_out/pmc_step.json', 'w').write(json.dumps(out, indent=1))
PY
