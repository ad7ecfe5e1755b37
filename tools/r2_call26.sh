#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee gpurun_out/c26_pytest_all.log
for dt in f32 bf16; do
  timeout 300 python bench.py --steps 48 --warmup 16 --no_roofline --cpu_baseline_steps 0 --dtype $dt 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$dt', d['ms_per_step'], d['value'])"
done | tee gpurun_out/c26_bench.log
SG2IM_DEFER_WGRAD=0 SG2IM_SCHEDULE=2 timeout 300 python bench.py --steps 48 --warmup 16 --no_roofline --cpu_baseline_steps 0 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f32 old schedule', d['ms_per_step'], d['value'])" | tee -a gpurun_out/c26_bench.log
