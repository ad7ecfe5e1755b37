"""How does a hipGraph replay issue independent branches?  Two branches with no dependency between them - a chain of
N tiny dependent kernels on a side stream and ONE long kernel on the capture stream - captured in either order;
if a replay ran the branches as soon as their dependencies allow, both orders would cost max(chain, long).
Usage: python tools/graph_order_probe.py [N]   (environment: the clr knobs under test)"""
import os
import sys
import time

import torch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device('cuda', 0)
tiny = torch.zeros(64, device=dev)
big = torch.ones(1 << 29, device=dev)          # 2 GB: one in-place pass = 4 GB of traffic, ~1 ms
side = torch.cuda.Stream()
cap = torch.cuda.Stream()


def chain():
  for _ in range(N):
    tiny.add_(1.0)


A = torch.randn(128, 1 << 18, device=dev)
B = torch.randn(1 << 18, 128, device=dev)
C = torch.empty(128, 128, device=dev)
tiny2 = torch.zeros(64, device=dev)
side2 = torch.cuda.Stream()


def long_kernel():
  if os.environ.get('PROBE_LONG', 'bw') == 'bw':
    big.mul_(1.0)               # fills every CU (bandwidth bound)
  else:
    torch.mm(A, B, out=C)       # a handful of workgroups with a long K loop: most CUs stay free


def chain2():
  for _ in range(N):
    tiny2.add_(1.0)


def capture(order):
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g, stream=cap):
    main = torch.cuda.current_stream()
    tiny.add_(0.0)                # common root
    for what in order:
      if what == 'chain':
        side.wait_stream(main)
        with torch.cuda.stream(side):
          chain()
      elif what == 'long':
        long_kernel()
      elif what == 'chain_main':
        chain()
      elif what == 'chain2':
        side2.wait_stream(main)
        with torch.cuda.stream(side2):
          chain2()
    main.wait_stream(side)
    main.wait_stream(side2)
    tiny.add_(0.0)                # join
  return g


def timed(g, reps=20):
  for _ in range(3):
    g.replay()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(reps):
    g.replay()
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / reps * 1e3


long_kernel(); chain(); chain2()      # (library initialisation outside any capture)
torch.cuda.synchronize()
knobs = {k: v for k, v in os.environ.items() if k.startswith(('DEBUG_', 'GPU_MAX', 'HIP_FORCE', 'ROC_'))}
print('knobs', knobs, 'N', N)
for name, order in (('chain alone (side)', ['chain']), ('long alone', ['long']), ('chain then long', ['chain', 'long']),
                    ('long then chain', ['long', 'chain']), ('serial, one stream', ['chain_main', 'long']),
                    ('two chains, two streams', ['chain', 'chain2'])):
  g = capture(order)
  print('%-22s %.3f ms/replay' % (name, timed(g)))
  del g
