#!/bin/bash
# large VG-style shapes: GraphTripleConv backward layer by layer (0) / one co-resident launch (auto = low) / staged, A/B in ONE call
cd $GRAFT_REPO_ROOT
D6=1024,512,256,128,64,64
for rep in 1 2; do for v in auto 0 staged; do
  for dt in f32 bf16; do
  SG2IM_GCN_PERSIST_BWD=$v python bench.py --steps 10 --warmup 3 --n_batches 4 --cpu_baseline_steps 0 --no_roofline --dtype $dt --style vg --image_size 256 --refinement_dims $D6 --min_objs 10 --max_objs 29 --extra_rels 60 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[gcn_bwd=$v] $dt s256', d['ms_per_step'])"
  done
  SG2IM_GCN_PERSIST_BWD=$v python bench.py --steps 20 --warmup 5 --n_batches 8 --cpu_baseline_steps 0 --no_roofline --dtype bf16 --style vg --image_size 128 --refinement_dims $D6 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[gcn_bwd=$v] bf16 vg128', d['ms_per_step'])"
done; done
