#!/bin/bash
# Round 6, item 1 of VERDICT r5: BASELINE configs[2..4] at their real per-GPU sizes - bench lines (fp32 and bf16) with
# roofline, and the large-batch parity cases.  Outputs under gpurun_out/r6cfg/.
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
  run bench_vg64_$dt --style vg --dtype $dt --steps 40 --warmup 10
  run bench_vg128_$dt --style vg --dtype $dt --image_size 128 --refinement_dims $D6 --steps 20 --warmup 5 --n_batches 8
  run bench_s256_$dt --style vg --dtype $dt --image_size 256 --refinement_dims $D6 --min_objs 10 --max_objs 29 --extra_rels 60 --steps 10 --warmup 3 --n_batches 4
done
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "other_baseline_shapes" 2>&1 | tail -15 > $O/pytest_shapes.log; cat $O/pytest_shapes.log
cp gpurun_out/grad_parity.log $O/ 2>/dev/null
