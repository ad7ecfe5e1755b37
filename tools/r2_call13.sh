#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for rep in 1 2; do
for v in "" _rb512 _rb1024; do
  SG2IM_LIB=$PWD/sg2im_amd/lib/libsg2im_hip$v.so timeout 300 python bench.py --steps 48 --warmup 16 --no_roofline --cpu_baseline_steps 0 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant [$v]', d['ms_per_step'], d['value'])"
done
done | tee gpurun_out/c13_redblocks.log
