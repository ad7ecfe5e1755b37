"""Print the launch plans (tile, split-K) the library chooses for the COCO-64 / batch-32 conv layers.
Runs without a GPU: pointers are dummies and the launches themselves fail, the planner output
(SG2IM_PLAN_DEBUG=1, stderr) is what this is for."""
import ctypes
import os
import sys

os.environ['SG2IM_PLAN_DEBUG'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sg2im_amd import _lib
from bench_conv import LAYERS, NB

lib = _lib.load()
P = ctypes.c_void_p
FAKE = 1 << 30
WS = 256 << 20
for L in LAYERS:
  name, H, C0, C1, Cout, k, s, p = L[:8]
  N = L[8] if len(L) > 8 else NB
  d = _lib.ConvDesc()
  d.nsrc = 0
  if C0:
    d.src[d.nsrc] = _lib.Src(FAKE, None, None, None, 1.0, C0, C0, 0)
    d.nsrc += 1
  if C1:
    d.src[d.nsrc] = _lib.Src(FAKE, None, None, None, 1.0, C1, C1, 1)
    d.nsrc += 1
  Ho = (H + 2 * p - k) // s + 1
  d.batch, d.in_h, d.in_w, d.out_h, d.out_w, d.kh, d.kw, d.stride, d.pad = N, H, H, Ho, Ho, k, k, s, p
  Ct = C0 + C1
  for what in ('fwd', 'dgrad', 'wgrad'):
    sys.stderr.write('== %s %s\n' % (name, what)); sys.stderr.flush()
    if what == 'fwd':
      lib.sg2im_conv2d_forward(ctypes.byref(d), P(FAKE), Cout, P(FAKE), 1.0, P(FAKE), Cout, 0, P(FAKE), WS, None)
    elif what == 'dgrad':
      lib.sg2im_conv2d_backward_data(ctypes.byref(d), P(FAKE), Cout, P(FAKE), Cout, 0, Ct, P(FAKE), Ct, 0, P(FAKE), WS, None)
    else:
      lib.sg2im_conv2d_backward_weight(ctypes.byref(d), P(FAKE), Cout, Cout, P(FAKE), None, 0, P(FAKE), WS, None)
