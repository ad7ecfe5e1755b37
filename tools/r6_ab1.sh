#!/bin/bash
# one-dtype step A/B: tools/r6_ab1.sh DTYPE "ENV=a" "ENV=b" ...   (COCO-64, two rounds)
cd $GRAFT_REPO_ROOT
DT=$1; shift
for rep in 1 2; do for cfg in "$@"; do
  env $cfg python bench.py --steps 50 --warmup 10 --cpu_baseline_steps 0 --no_roofline --dtype $DT 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$cfg] $DT', d['ms_per_step'])"
done; done
