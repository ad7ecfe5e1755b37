"""Isolated timing of the HBM-bound layout kernels at the bench shape (COCO-64, batch 32): layout + noise + pyramid
forward (sg2im_layout_pyramid_forward) and the vector gradient straight from the per-level gradients
(sg2im_layout_backward_vecs_levels) against pyramid_backward + layout_backward."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from sg2im_amd import ops
from sg2im_amd.synthetic import synthetic_batch

D = torch.device('cuda', 0)
batch = synthetic_batch(32, seed=3)
imgs, objs, boxes, masks, triples, o2i = [t.to(D) for t in batch[:6]]
N, H, Dv, L = 32, 64, 128, 5
O = objs.numel()
vecs = torch.randn(O, Dv, device=D)
img_csr = ops.Csr(o2i, None, N)
levels = [torch.randn(N, H >> i, H >> i, Dv, device=D) for i in range(L)]
factors = [1 << i for i in range(L)]


def timeit(fn, n=20):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(n):
    fn()
  b.record()
  torch.cuda.synchronize()
  return a.elapsed_time(b) / n * 1e3


dv = torch.empty(O, Dv, device=D)
t = timeit(lambda: ops.layout_backward_vecs_levels(levels, factors, vecs, boxes, masks, img_csr, N, H, H, False, dv))
nbytes = 4.0 * (sum(x.numel() for x in levels) + 2 * O * Dv)
print('layout_bwd_vecs_levels (+ reduce): %.1f us for %.1f MB of level gradients = %.2f TB/s' % (t, nbytes / 1e6, nbytes / t / 1e6))
summed = torch.empty(N, H, H, Dv, device=D)


def old():
  ops.pyramid_backward(levels, factors, [Dv] * L, N, H, H, Dv, summed)
  ops.layout_backward(summed, vecs, boxes, masks, o2i, img_csr, N, H, H, False, dv, None, None)
t2 = timeit(old)
print('pyramid_backward + layout_backward (d_vecs only): %.1f us' % t2)
noise = torch.randn(N, 32, H, H, device=D)
lv = [torch.empty(N, H >> i, H >> i, Dv + 32, device=D) for i in range(L)]
t3 = timeit(lambda: ops.layout_pyramid_forward(vecs, boxes, masks, img_csr, N, H, H, False, noise, lv))
nb = 4.0 * (sum(x.numel() for x in lv) + noise.numel())
print('layout + noise + pyramid forward: %.1f us for %.1f MB = %.2f TB/s' % (t3, nb / 1e6, nb / t3 / 1e6))
