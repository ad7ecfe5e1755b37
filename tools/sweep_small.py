import os, sys
os.environ['SG2IM_PLAN_TUNE']='1'
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import torch
from sg2im_amd import ops
from bench_conv import LAYERS, NB, D, timeit
want = sys.argv[1:]
for L in LAYERS:
  name, H, C0, C1, Cout, k, s, p = L[:8]
  if not any(name.startswith(w) for w in want): continue
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
  # time inside a captured graph of 20 back-to-back launches: no CPU launch overhead
  for what, fn in fns.items():
    row = []
    for t in (2,):
      for ns in (1, 2, 3, 4, 6, 8, 12, 16, 24, 36):
        os.environ['SG2IM_FORCE_PLAN'] = '%d,%d' % (t, ns)
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
          for _ in range(20): fn()
        g.replay(); torch.cuda.synchronize()
        a, bb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); g.replay(); bb.record(); torch.cuda.synchronize()
        row.append('x%d=%.1f' % (ns, a.elapsed_time(bb) / 40 * 1e3))
    print('%-10s %-5s us/launch(+finish) in-graph: %s' % (name, what, ' '.join(row)), flush=True)
