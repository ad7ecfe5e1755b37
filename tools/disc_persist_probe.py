"""The discriminator CNN forward as one persistent launch (csrc/disc_persist.hip) against the launch path, isolated:
eager back-to-back time of both and the persistent kernel's device-clock stamps (start, then before / after every grid
barrier, end - microseconds).  profiles/r4_disc_persistent_forward.log is this script's output.

  python tools/disc_persist_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sg2im_amd import functional as HF  # noqa: E402
from sg2im_amd import ops  # noqa: E402
from sg2im_amd.layers import build_cnn  # noqa: E402


def timed(fn, n=30):
  for _ in range(5):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3


def main():
  dev = torch.device('cuda:0')
  for name, shape in (('d_obj: 224 crops of 32 x 32', (224, 32, 32, 3)), ('d_img: 32 images of 64 x 64', (32, 64, 64, 3))):
    cnn, _ = build_cnn('I3,C4-64-2,C4-128-2,C4-256-2', normalization='batch', activation='leakyrelu-0.2', padding='valid',
                       pooling='avg')
    cnn = cnn.to(dev).train()
    x = torch.randn(*shape, device=dev)
    convs = [m for m in cnn if isinstance(m, torch.nn.Conv2d)]
    bns = [m for m in cnn if isinstance(m, torch.nn.BatchNorm2d)]
    params = [(HF._cl_weight(cv.weight), cv.bias) for cv in convs]
    flat = [convs[0].weight, convs[0].bias]
    for bn, cv in zip(bns, convs[1:]):
      flat += [bn.weight, bn.bias, cv.weight, cv.bias]
    a = timed(lambda: HF.DiscCnnFn.apply(x, bns, cnn.specs, cnn.slope, True, None, None, *flat))
    b = timed(lambda: ops.disc_stack_forward(x, cnn.specs, params, bns, cnn.slope, 1, HF.BN_EPS, HF.BN_MOMENTUM))
    st = ops.gconv_stack_stamps(dev)
    print('%s: launch path %.1f us (eager launches, host bound), one persistent launch %.1f us; stamps %s'
          % (name, a, b, ' '.join('%.0f' % v for v in st)))


if __name__ == '__main__':
  main()
