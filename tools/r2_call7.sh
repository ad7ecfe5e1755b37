#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 60 tools/_bin/tr_probe > gpurun_out/c7_tr_probe.log 2>&1; head -140 gpurun_out/c7_tr_probe.log
timeout 600 python -m pytest tests -m gpu -x -q -k "rccl or graph_replay or reproducible or dp_" 2>&1 | tail -6
timeout 300 python bench.py --steps 24 --warmup 8 --no_roofline --cpu_baseline_steps 0 --force_dist 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('gradient_exchange'))"
