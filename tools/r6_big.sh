#!/bin/bash
# big bf16 configurations A/B: tools/r6_big.sh "ENV=a" "ENV=b" ...
cd $GRAFT_REPO_ROOT
D6=1024,512,256,128,64,64
for cfg in "$@"; do
  env $cfg timeout 600 python bench.py --cpu_baseline_steps 0 --no_roofline --style vg --dtype bf16 --image_size 128 --steps 20 --warmup 5 --n_batches 8 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$cfg] bf16 vg128 5mod', d['ms_per_step'], 'ms/step', d['value'], 'img/s')"
  env $cfg timeout 600 python bench.py --cpu_baseline_steps 0 --no_roofline --style vg --dtype bf16 --image_size 256 --refinement_dims $D6 --min_objs 10 --max_objs 29 --extra_rels 60 --steps 10 --warmup 3 --n_batches 4 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$cfg] bf16 s256', d['ms_per_step'], 'ms/step', d['value'], 'img/s')"
done
