"""Isolated timing of the BatchNorm kernels at the refinement network's shapes (batch 32):
statistics (forward) and the three-launch backward (partial sums, final, apply), against the HBM
bytes each has to move."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from sg2im_amd import ops

D = torch.device('cuda', 0)
SHAPES = [(64, 64, False), (64, 64, True), (32, 128, False), (32, 128, True), (16, 256, True), (8, 512, True), (4, 1024, True)]


def timeit(fn, iters=20):
  fn(); fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(iters):
    fn()
  b.record()
  torch.cuda.synchronize()
  return a.elapsed_time(b) / iters * 1e3


def main():
  NB = 32
  print('%-22s %9s | %9s %7s | %9s %7s' % ('shape', 'MB', 'stats us', 'TB/s', 'bwd us', 'TB/s'))
  for H, C, pool in SHAPES:
    rows = NB * H * H
    y = torch.randn(rows, C, device=D)
    bn = torch.nn.BatchNorm2d(C).to(D)
    dy = torch.empty_like(y)
    dgamma, dbeta = torch.zeros(C, device=D), torch.zeros(C, device=D)
    if pool:
      g = torch.randn(NB, 2 * H, 2 * H, C, device=D)
    else:
      g = torch.randn(rows, C, device=D)
    st = ops.bn_stats(y, rows, C, C, bn, True)
    t1 = timeit(lambda: ops.bn_stats(y, rows, C, C, bn, True))
    t2 = timeit(lambda: ops.bn_act_backward(ops._f(g), C, int(pool), NB, H, H, y, C, C, bn.weight, st, 0.01, True, dy, dgamma, dbeta))
    mb = rows * C * 4 / 1e6
    b1 = mb
    b2 = mb * ((4 if pool else 1) + 1) * 2 + mb        # partial: g + y; apply: g + y -> dy
    print('%-22s %9.1f | %9.1f %7.2f | %9.1f %7.2f' % ('%dx%dx%d%s' % (H, H, C, ' pool2' if pool else ''), mb, t1, b1 / t1, t2, b2 / t2), flush=True)


if __name__ == '__main__':
  main()
