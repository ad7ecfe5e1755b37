"""Where the persistent GraphTripleConv kernel spends its time: per stage compute / barrier wait of workgroup 0 (device
clock stamps), the whole launch (HIP events), and the layer-by-layer launches beside it.  COCO-64 batch-32 shape."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from sg2im_amd import functional as HF, ops
from sg2im_amd.synthetic import synthetic_batch

D = torch.device('cuda', 0)
g = torch.Generator().manual_seed(0)
batch = synthetic_batch(32, seed=3)
objs, triples = batch[1], batch[4]
O, T = objs.numel(), triples.size(0)
O, T = (O + 32) // 32 * 32, (T + 63) // 64 * 64           # the bucket-padded sizes of the training graph
s = torch.cat([triples[:, 0], torch.full((T - triples.size(0),), O - 1, dtype=torch.long)]).to(D)
o = torch.cat([triples[:, 2], torch.full((T - triples.size(0),), O - 1, dtype=torch.long)]).to(D)
live = torch.tensor([triples.size(0)], dtype=torch.int32, device=D)
csr = ops.Csr(s, o, O, live=(live, 1))
nl, Din, H = 5, 128, 512
W = []
for l in range(nl):
  for (a, b) in ((H, 3 * Din), (2 * H + Din, H), (H, H), (Din, H)):
    W += [torch.randn(a, b, device=D) * (2.0 / b) ** 0.5, torch.randn(a, device=D) * 0.1]
ov, pv = torch.randn(O, Din, device=D), torch.randn(T, Din, device=D)
print('O %d T %d' % (O, T))


def timed(fn, n=30):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3


with torch.no_grad():
  us = timed(lambda: HF.GraphTripleConvStackFn.apply(ov, pv, s, o, csr, True, *W))
  ops.gconv_stack_check(D)
  st = ops.gconv_stack_stamps(D)
  print('persistent stack forward: %.1f us per launch (events, back to back); stamps: %d' % (us, len(st)))
  names = []
  for l in range(nl):
    names += ['L%d A' % l, 'L%d B' % l, 'L%d C' % l, 'L%d D' % l, 'L%d E' % l]
  prev = 0.0
  k = 0
  for i in range(1, len(st) - 1, 2):
    print('  %-6s compute %6.2f us   barrier wait %6.2f us' % (names[k], st[i] - prev, st[i + 1] - st[i]))
    prev = st[i + 1]
    k += 1
  print('  %-6s compute %6.2f us   (end)   total %.2f us' % (names[k] if k < len(names) else 'tail', st[-1] - prev, st[-1]))

  # backward: one launch vs layer by layer
  def stamps_report(names, st):
    prev, k = 0.0, 0
    for i in range(1, len(st) - 1, 2):
      print('  %-6s compute %6.2f us   barrier wait %6.2f us' % (names[k] if k < len(names) else '?', st[i] - prev, st[i + 1] - st[i]))
      prev = st[i + 1]
      k += 1
    print('  %-6s compute %6.2f us   (end)   total %.2f us' % (names[k] if k < len(names) else 'tail', st[-1] - prev, st[-1]))


def bwd_bench(flag):
  ops.GCN_PERSISTENT_BACKWARD = flag
  Wg = [w.clone().requires_grad_(True) for w in W]
  ovg, pvg = ov.clone().requires_grad_(True), pv.clone().requires_grad_(True)
  go, gp = torch.randn(O, Din, device=D), torch.randn(T, Din, device=D)

  def run():
    x, p = HF.GraphTripleConvStackFn.apply(ovg, pvg, s, o, csr, True, *Wg)
    torch.autograd.backward([x, p], [go, gp])
  us = timed(run, 20)
  return us


names = []
for l in range(nl - 1, -1, -1):
  names += ['L%d P1' % l, 'L%d P2' % l, 'L%d P3' % l, 'L%d P4' % l, 'L%d P5' % l]
us_off = bwd_bench(False)
for mode in ('full', 'low'):            # sg2im_gconv_stack_grads.low_footprint = 0 / 1
  us_on = bwd_bench(mode)
  ops.gconv_stack_check(D)
  stb = ops.gconv_stack_stamps(D)
  print('forward + backward, one launch each (%s footprint): %.1f us; forward one launch + backward layer by layer: %.1f us '
        '(eager, incl. host)' % (mode, us_on, us_off))
  print('persistent backward (%s footprint), workgroup 0: kernel %.1f us' % (mode, stb[-1]))
  stamps_report(names, stb)
ops.GCN_PERSISTENT_BACKWARD = False

with torch.no_grad():
  def per_layer():
    x, p = ov, pv
    for l in range(nl):
      x, p = HF.GraphTripleConvFn.apply(x, p, s, o, csr, True, *W[8 * l:8 * l + 8])
  print('layer-by-layer launches: %.1f us per stack' % timed(per_layer))
