#!/bin/bash
# A/B of an environment knob inside ONE gpurun call (box-to-box variance is +-5 %): tools/ab_bench.sh OUTDIR "VAR=a" "VAR=b" ...
out=$1; shift
mkdir -p $out
for rep in 1 2; do
  for kv in "$@"; do
    tag=$(echo "$kv" | tr '= /' '___')
    env $kv python bench.py --steps ${STEPS:-100} --warmup 20 --cpu_baseline_steps 0 --no_roofline > $out/bench_${tag}_$rep.json 2> $out/bench_${tag}_$rep.err
    python - <<PY
import json
d = json.load(open('$out/bench_${tag}_$rep.json'))
print('$kv rep $rep: %.3f ms/step  %.1f img/s' % (d['ms_per_step'], d['value']))
PY
  done
done
