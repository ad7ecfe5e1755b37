"""Empirical check of the launch planner: for every COCO-64 conv layer and pass, time each
(tile, split-K) candidate (forced through SG2IM_FORCE_PLAN) and compare the best with the
planner's own choice.  usage: python tools/plan_sweep.py"""
import os
import sys

os.environ['SG2IM_PLAN_TUNE'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
from sg2im_amd import ops
from bench_conv import LAYERS, NB, D, timeit

TILES = ['128x128', '128x64', '64x64', '64x128']
NS = [1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 13, 14, 15, 16, 20, 24, 28, 32, 36, 40, 48, 56, 64]
tot_model = {'fwd': 0.0, 'dgrad': 0.0, 'wgrad': 0.0}
tot_best = dict(tot_model)
ALL = []
for L in LAYERS:
  name, H, C0, C1, Cout, k, s, p = L[:8]
  N = L[8] if len(L) > 8 else NB
  srcs = []
  if C0:
    srcs.append(ops.nhwc_src(torch.randn(N, H, H, C0, device=D)))
  if C1:
    srcs.append(ops.nhwc_src(torch.randn(N, H // 2, H // 2, C1, device=D), 1))
  d = ops.conv_desc(srcs, N, H, H, k, k, s, p)
  Ct = C0 + C1
  W = torch.randn(Cout, k, k, Ct, device=D) * 0.01
  b = torch.randn(Cout, device=D)
  y = torch.empty(N, d.out_h, d.out_w, Cout, device=D)
  gy = torch.randn_like(y)
  dx = torch.empty(N, H, H, Ct, device=D)
  dw = torch.empty_like(W)
  fns = {'fwd': lambda: ops.conv2d_forward(d, W, Cout, b, y, Cout),
         'dgrad': lambda: ops.conv2d_backward_data(d, W, Cout, gy, Cout, 0, Ct, dx, Ct),
         'wgrad': lambda: ops.conv2d_backward_weight(d, gy, Cout, Cout, dw)}
  for what, fn in fns.items():
    os.environ.pop('SG2IM_FORCE_PLAN', None)
    t_model = timeit(fn, 6)
    res = []
    for t in range(4):
      for ns in NS:
        os.environ['SG2IM_FORCE_PLAN'] = '%d,%d' % (t, ns)
        res.append((timeit(fn, 6), t, ns))
    ALL.append({'layer': name, 'pass': what, 'model': t_model, 'res': res[:]})
    res.sort()
    tot_model[what] += t_model
    tot_best[what] += min(res[0][0], t_model)
    top = ' '.join('%s/x%d=%.3f' % (TILES[t], ns, tt) for tt, t, ns in res[:4])
    print('%-9s %-5s model %.3f | best %s' % (name, what, t_model, top), flush=True)
import json
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(ALL, open(os.path.join(ROOT, 'gpurun_out', 'plan_sweep.json'), 'w'))
print('TOTAL model', tot_model, 'best', tot_best)
