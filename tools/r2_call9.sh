#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "pool_csr or graph_triple or padded or bucketed or golden_coco or empty_and" 2>&1 | tail -4
for f in 3800 9000 18000 36000 72000; do
  echo "== SG2IM_FIN0=$f"
  SG2IM_FIN0=$f timeout 300 python bench.py --steps 48 --warmup 16 --no_roofline --cpu_baseline_steps 0 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
  SG2IM_FIN0=$f timeout 300 python bench.py --steps 48 --warmup 16 --no_roofline --cpu_baseline_steps 0 --dtype bf16 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16', d['ms_per_step'], d['value'])"
done 2>&1 | tee gpurun_out/c9_fin0.log
