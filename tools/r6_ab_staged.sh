#!/bin/bash
# GraphTripleConv backward: staged launches of the persistent kernel's stages (SG2IM_GCN_PERSIST_BWD=staged) vs auto, A/B in ONE call
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gconv_stack" 2>&1 | tail -3
for rep in 1 2 3; do for v in auto staged staged_full; do
  for dt in f32 bf16; do
    SG2IM_GCN_PERSIST_BWD=$v python bench.py --steps 50 --warmup 10 --cpu_baseline_steps 0 --no_roofline --dtype $dt 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[gcn_bwd=$v] $dt coco', d['ms_per_step'])"
    SG2IM_GCN_PERSIST_BWD=$v python bench.py --steps 50 --warmup 10 --cpu_baseline_steps 0 --no_roofline --dtype $dt --style vg 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[gcn_bwd=$v] $dt vg64', d['ms_per_step'])"
  done
done; done
for v in auto staged; do for dt in f32 bf16; do
  echo "== marks gcn_bwd=$v $dt"
  SG2IM_GCN_PERSIST_BWD=$v SG2IM_MARKS=1 python bench.py --steps 20 --warmup 5 --cpu_baseline_steps 0 --no_roofline --dtype $dt 2>&1 | grep "^\[mark\]" | tail -9
done; done
