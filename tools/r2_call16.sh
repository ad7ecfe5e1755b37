#!/bin/bash
# weight gradients of the small networks on the side lane: parity + step A/B (one box)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "graph or bucket or padded or trainer" 2>&1 | tail -4 | tee gpurun_out/c16_pytest.log
b() { timeout 300 python bench.py --steps 48 --warmup 16 --no_roofline --cpu_baseline_steps 0 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['value'])"; }
for rep in 1 2; do
  SG2IM_WGRAD_SIDE_SMALL=0 b "small=0"
  b "small=1"
  SG2IM_TAIL_AT=0 b "small=1 tail_at=0"
  SG2IM_TAIL_AT=1 b "small=1 tail_at=1"
  SG2IM_TAIL_AT=2 b "small=1 tail_at=2"
done | tee gpurun_out/c16_bench.log
