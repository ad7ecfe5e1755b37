#!/bin/bash
# A/B of the bf16 halo'd kernels' staging group (SG2IM_HALO_TG = 1 | 3 | 9 taps per barrier pair) with the weight mirror
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6h9; mkdir -p $O
for tg in 3 9; do
SG2IM_HALO_TG=$tg timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "weight_mirror" 2>&1 | tail -3
done
for rep in 1 2; do
for tg in 1 3 9; do
  for st in coco vg; do
  SG2IM_HALO_TG=$tg timeout 300 python bench.py --steps 50 --warmup 10 --cpu_baseline_steps 0 --no_roofline --dtype bf16 --style $st 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('TG=$tg bf16 $st', d['ms_per_step'], 'ms/step', d['value'], 'img/s')"
  done
done
done 2>&1 | tee $O/step_ab_tg.log
for tg in 1 3; do
SG2IM_HALO_TG=$tg timeout 300 python bench.py --cpu_baseline_steps 0 --no_roofline --style vg --dtype bf16 --image_size 256 --refinement_dims 1024,512,256,128,64,64 --min_objs 10 --max_objs 29 --extra_rels 60 --steps 10 --warmup 3 --n_batches 4 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('TG=$tg bf16 s256', d['ms_per_step'], 'ms/step', d['value'], 'img/s')"
done 2>&1 | tee -a $O/step_ab_tg.log
