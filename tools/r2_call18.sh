#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
b() { timeout 300 python bench.py --steps 48 --warmup 16 --no_roofline --cpu_baseline_steps 0 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['value'])"; }
for rep in 1 2 3; do
  for m in 0 1 2 3; do SG2IM_WGRAD_SIDE_SMALL=$m b "small=$m"; done
done | tee gpurun_out/c18_bench.log
