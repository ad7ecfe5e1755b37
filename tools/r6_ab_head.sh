#!/bin/bash
# head of the step: the one-launch triples CSR and the split layout launch, A/B in ONE call
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "csr or layout or pyramid or empty_and_ragged or trainer_two_steps or graph_replay" 2>&1 | tail -4
for rep in 1 2 3; do for v in "1 1" "0 1" "1 0" "0 0"; do
  set -- $v
  for dt in f32 bf16; do
    SG2IM_TRIPLES_CSR=$1 SG2IM_LAYOUT_SPLIT=$2 python bench.py --steps 50 --warmup 10 --cpu_baseline_steps 0 --no_roofline --dtype $dt 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[csr=$1 split=$2] $dt coco', d['ms_per_step'])"
  done
done; done
for v in "1 1" "0 0"; do
  set -- $v
  for dt in f32 bf16; do
    echo "== marks csr=$1 split=$2 $dt"
    SG2IM_TRIPLES_CSR=$1 SG2IM_LAYOUT_SPLIT=$2 SG2IM_MARKS=1 python bench.py --steps 20 --warmup 5 --cpu_baseline_steps 0 --no_roofline --dtype $dt 2>&1 | grep "^\[mark\]" | head -8
  done
done
