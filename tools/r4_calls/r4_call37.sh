#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_call37.log
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -s -k "one_persistent_launch" > $L 2>&1
grep "rel \|passed\|failed" $L | cut -c1-110 | tail -70
python - <<'PY' >> $L 2>&1
import torch, time
from sg2im_amd import ops, functional as HF
from sg2im_amd.layers import build_cnn
D = torch.device('cuda:0')
for name, shape in (('d_obj crops 224', (224, 32, 32, 3)), ('d_img 32 images', (32, 64, 64, 3))):
  cnn, _ = build_cnn('I3,C4-64-2,C4-128-2,C4-256-2', normalization='batch', activation='leakyrelu-0.2', padding='valid', pooling='avg')
  cnn = cnn.to(D).train()
  x = torch.randn(*shape, device=D)
  convs = [m for m in cnn if isinstance(m, torch.nn.Conv2d)]
  bns = [m for m in cnn if isinstance(m, torch.nn.BatchNorm2d)]
  params = [(HF._cl_weight(cv.weight), cv.bias) for cv in convs]
  p2 = [convs[0].weight, convs[0].bias]
  for bn, cv in zip(bns, convs[1:]):
    p2 += [bn.weight, bn.bias, cv.weight, cv.bias]
  def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
  a = t(lambda: HF.DiscCnnFn.apply(x, bns, cnn.specs, cnn.slope, True, None, None, *p2))
  b = t(lambda: ops.disc_stack_forward(x, cnn.specs, params, bns, cnn.slope, 1, HF.BN_EPS, HF.BN_MOMENTUM))
  st = ops.gconv_stack_stamps(D)
  print('%s: launch path %.1f us (eager, back to back), one persistent launch %.1f us; stamps %s' % (name, a, b, ' '.join('%.0f' % v for v in st)))
PY
tail -3 $L
