#!/bin/bash
# six-level refinement networks (BASELINE configs[3..4]): the layout gradient straight from the level gradients (<= 6 levels now) vs materialised (SG2IM_LAZY_LAYOUT_GRAD=0), A/B in ONE call
cd $GRAFT_REPO_ROOT
D6=1024,512,256,128,64,64
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "level_gradients or other_baseline_shapes" 2>&1 | tail -3
for rep in 1 2; do for v in 1 0; do for dt in f32 bf16; do
  SG2IM_LAZY_LAYOUT_GRAD=$v python bench.py --steps 20 --warmup 5 --n_batches 8 --cpu_baseline_steps 0 --no_roofline --dtype $dt --style vg --image_size 128 --refinement_dims $D6 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[from_levels=$v] $dt vg128', d['ms_per_step'])"
  SG2IM_LAZY_LAYOUT_GRAD=$v python bench.py --steps 10 --warmup 3 --n_batches 4 --cpu_baseline_steps 0 --no_roofline --dtype $dt --style vg --image_size 256 --refinement_dims $D6 --min_objs 10 --max_objs 29 --extra_rels 60 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[from_levels=$v] $dt s256', d['ms_per_step'])"
done; done; done
