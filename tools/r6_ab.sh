#!/bin/bash
# generic step A/B: tools/r6_ab.sh "ENV=a ENV2=b" "ENV=c" ...   (each config: fp32 coco, bf16 coco, fp32 vg; two rounds)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "$@"; do
  for w in "f32 coco" "bf16 coco" "f32 vg"; do set -- $w
    env $cfg timeout 300 python bench.py --steps 50 --warmup 10 --cpu_baseline_steps 0 --no_roofline --dtype $1 --style $2 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$cfg] $1 $2', d['ms_per_step'], 'ms/step', d['value'], 'img/s')"
  done
done
done
