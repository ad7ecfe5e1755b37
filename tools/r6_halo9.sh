#!/bin/bash
# A/B of the bf16 halo'd kernels' weight mirror (SG2IM_WEIGHT_MIRROR / SG2IM_HALO_WB) and nine-tap staging (SG2IM_HALO9)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6h9; mkdir -p $O
for h9 in 0 1; do
SG2IM_HALO9=$h9 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "bf16_operand_conv or weight_mirror" 2>&1 | tail -5 > $O/pytest_bf16_conv_h9_$h9.log; cat $O/pytest_bf16_conv_h9_$h9.log
done
for rep in 1 2; do
for cfg in "0 0" "1 0" "1 1"; do set -- $cfg
  for st in coco vg; do
  SG2IM_WEIGHT_MIRROR=$1 SG2IM_HALO9=$2 timeout 300 python bench.py --steps 50 --warmup 10 --cpu_baseline_steps 0 --no_roofline --dtype bf16 --style $st 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('MIRROR=$1 HALO9=$2 bf16 $st', d['ms_per_step'], 'ms/step', d['value'], 'img/s')"
  done
done
done 2>&1 | tee $O/step_ab.log
SG2IM_WEIGHT_MIRROR=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "bf16_training_step" 2>&1 | tail -5
