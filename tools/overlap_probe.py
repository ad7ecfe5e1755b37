"""Do tiny kernels on a second stream hide under a machine-filling GEMM kernel on this stack?
Times (a) one big conv + N tiny launches back to back on one stream, (b) the same with the tiny
chain on a second stream, eagerly and as a captured graph.  usage: python tools/overlap_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sg2im_amd import ops

D = torch.device('cuda', 0)
N, H, C0, C1, Cout = 32, 64, 160, 128, 64
srcs = [ops.nhwc_src(torch.randn(N, H, H, C0, device=D)), ops.nhwc_src(torch.randn(N, H // 2, H // 2, C1, device=D), 1)]
d = ops.conv_desc(srcs, N, H, H, 3, 3, 1, 1)
W = torch.randn(Cout, 3, 3, C0 + C1, device=D) * 0.01
b = torch.randn(Cout, device=D)
y = torch.empty(N, H, H, Cout, device=D)
small = torch.randn(64, 512, device=D)
tmp = torch.empty_like(small)
NT = 40
side = torch.cuda.Stream()


def big():
  ops.conv2d_forward(d, W, Cout, b, y, Cout)


def tiny():
  for _ in range(NT):
    ops.leaky_forward(small, 0.2, tmp)


def serial():
  big(); tiny()


def forked():
  main = torch.cuda.current_stream()
  side.wait_stream(main)
  with torch.cuda.stream(side):
    tiny()
  big()
  main.wait_stream(side)


def timeit(fn, iters=20):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(iters):
    fn()
  e.record()
  torch.cuda.synchronize()
  return a.elapsed_time(e) / iters * 1e3


def graphed(fn):
  s = torch.cuda.Stream()
  with torch.cuda.stream(s):
    fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s, capture_error_mode='thread_local'):
      fn()
  return g.replay


print('big alone            %8.1f us' % timeit(big))
print('tiny x%d alone       %8.1f us' % (NT, timeit(tiny)))
print('eager  serial        %8.1f us' % timeit(serial))
print('eager  two streams   %8.1f us' % timeit(forked))
gs, gf = graphed(serial), graphed(forked)
print('graph  serial        %8.1f us' % timeit(gs))
print('graph  two branches  %8.1f us' % timeit(gf))
