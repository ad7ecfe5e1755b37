#!/bin/bash
# Round 6: the configurations of tools/r6_configs.sh that were missing after its first run: 256 x 256 (the lane workspace
# had to grow) and VG-128 with the reference's default 5-module CRN (seed map 8 x 8; 75 GF/image).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6cfg; mkdir -p $O
run() { # name, args...
  n=$1; shift
  timeout 900 python bench.py --cpu_baseline_steps 0 "$@" > $O/$n.out 2> $O/$n.err
  grep '^{"metric' $O/$n.out > $O/$n.json
  python - <<PY
import json
try:
  d = json.load(open('$O/$n.json'))
  r = d['roofline'] or {}
  print('$n', d['ms_per_step'], 'ms/step', d['value'], 'img/s', 'family', r.get('achieved'), r.get('frac'), 'crn', (r.get('crn_only') or {}).get('tflops'), (r.get('crn_only') or {}).get('frac'))
except Exception as e:
  print('$n FAILED', e); print(open('$O/$n.err').read()[-3000:])
PY
}
D6=1024,512,256,128,64,64
for dt in f32 bf16; do
  run bench_s256_$dt --style vg --dtype $dt --image_size 256 --refinement_dims $D6 --min_objs 10 --max_objs 29 --extra_rels 60 --steps 10 --warmup 3 --n_batches 4
  run bench_vg128_5mod_$dt --style vg --dtype $dt --image_size 128 --steps 20 --warmup 5 --n_batches 8
done
