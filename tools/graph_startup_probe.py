"""Is there a fixed start-up cost per hipGraph replay?  Captures chains of N small dependent library launches
(the first graph-convolution layer's GEMMs, cycling through a few distinct kernels) and times back-to-back
replays WITHOUT a profiler: time per replay = start-up + N x per-launch cost.  (Under rocprofv3 the first ~10
library kernels of every replayed training step are ~100 us apart, profiles/r2_step_kernel_sequence.txt.)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from sg2im_amd import _lib, ops

D = torch.device('cuda', 0)


def chain(n, bufs):
  x, W1, b1, h, W2, b2, y = bufs
  for i in range(n // 2):
    ops.conv2d_forward(ops.conv_desc([ops.rows_src(x)], x.size(0), 1, 1), W1, W1.size(0), b1, h, W1.size(0), 0.0)
    ops.conv2d_forward(ops.conv_desc([ops.rows_src(h)], h.size(0), 1, 1), W2, W2.size(0), b2, y, W2.size(0), 0.0)


def main():
  _lib.init()
  T = 384
  bufs = (torch.randn(T, 384, device=D), torch.randn(512, 384, device=D) * 0.05, torch.zeros(512, device=D),
          torch.empty(T, 512, device=D), torch.randn(384, 512, device=D) * 0.05, torch.zeros(384, device=D),
          torch.empty(T, 384, device=D))
  s = torch.cuda.Stream()
  with torch.cuda.stream(s):
    ops.workspace(D); ops.scratch(D, 1 << 20)
    chain(4, bufs)
  torch.cuda.synchronize()
  print('%6s %12s %14s' % ('N', 'us/replay', 'us/launch-pair'))
  res = []
  for n in (4, 16, 32, 64, 128):
    g = torch.cuda.CUDAGraph()
    _lib.CAPTURING = True
    try:
      with torch.cuda.graph(g, stream=s):
        chain(n, bufs)
    finally:
      _lib.CAPTURING = False
    for _ in range(5):
      g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 200
    for _ in range(R):
      g.replay()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / R * 1e6
    res.append((n, us))
    print('%6d %12.1f %14.2f' % (n, us, us / (n / 2)), flush=True)
  (n0, u0), (n1, u1) = res[1], res[-1]
  per = (u1 - u0) / (n1 - n0)
  print('per launch %.2f us, start-up per replay %.1f us' % (per, u0 - per * n0))


if __name__ == '__main__':
  main()
