#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python tools/gpu_check.py sec_layout sec_layout_align_corners sec_golden_coco 2>&1 | grep -v amdgpu.ids | tail -2
timeout 600 python -m pytest tests -m gpu -x -q -k "layout or padded or bucketed" 2>&1 | tail -2
timeout 300 python bench.py --steps 48 --warmup 16 --cpu_baseline_steps 0 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['hbm_bound']['layout_fwd'])"
