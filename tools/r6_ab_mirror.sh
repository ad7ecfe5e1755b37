#!/bin/bash
# bf16 weight-mirror refresh behind the head of the step (SG2IM_MIRROR_LATE=1, default) vs first thing, A/B in ONE call
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bf16 or mirror" 2>&1 | tail -3
for rep in 1 2 3; do for v in 1 0; do
  SG2IM_MIRROR_LATE=$v python bench.py --steps 50 --warmup 10 --cpu_baseline_steps 0 --no_roofline --dtype bf16 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[late=$v] bf16 coco', d['ms_per_step'])"
  SG2IM_MIRROR_LATE=$v python bench.py --steps 50 --warmup 10 --cpu_baseline_steps 0 --no_roofline --dtype bf16 --style vg 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[late=$v] bf16 vg64', d['ms_per_step'])"
done; done
for v in 1 0; do
  echo "== marks late=$v bf16"
  SG2IM_MIRROR_LATE=$v SG2IM_MARKS=1 python bench.py --steps 20 --warmup 5 --cpu_baseline_steps 0 --no_roofline --dtype bf16 2>&1 | grep "^\[mark\]" | head -6
done
