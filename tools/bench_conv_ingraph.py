import os, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import torch
from sg2im_amd import ops
from bench_conv import LAYERS, NB, D
want = sys.argv[1:]
tot = {'fwd':0,'dgrad':0,'wgrad':0}
for L in LAYERS:
  name, H, C0, C1, Cout, k, s, p = L[:8]
  if want and not any(name.startswith(w) for w in want): continue
  N = L[8] if len(L) > 8 else NB
  srcs = []
  if C0: srcs.append(ops.nhwc_src(torch.randn(N, H, H, C0, device=D)))
  if C1: srcs.append(ops.nhwc_src(torch.randn(N, H // 2, H // 2, C1, device=D), 1))
  d = ops.conv_desc(srcs, N, H, H, k, k, s, p)
  Ct = C0 + C1
  W = torch.randn(Cout, k, k, Ct, device=D) * 0.01
  b = torch.randn(Cout, device=D)
  y = torch.empty(N, d.out_h, d.out_w, Cout, device=D); gy = torch.randn_like(y)
  dx = torch.empty(N, H, H, Ct, device=D); dw = torch.empty_like(W)
  fns = {'fwd': lambda: ops.conv2d_forward(d, W, Cout, b, y, Cout),
         'dgrad': lambda: ops.conv2d_backward_data(d, W, Cout, gy, Cout, 0, Ct, dx, Ct),
         'wgrad': lambda: ops.conv2d_backward_weight(d, gy, Cout, Cout, dw)}
  row = []
  for what, fn in fns.items():
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
      for _ in range(10): fn()
    g.replay(); torch.cuda.synchronize()
    a, bb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); g.replay(); bb.record(); torch.cuda.synchronize()
    us = a.elapsed_time(bb) / 20 * 1e3
    tot[what] += us
    row.append('%s %7.1f' % (what, us))
  print('%-10s %s' % (name, '  '.join(row)), flush=True)
print('TOTAL us', {k: round(v, 1) for k, v in tot.items()})
