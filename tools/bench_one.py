"""Run one conv layer shape repeatedly (for rocprofv3 --pmc passes).
usage: bench_one.py NAME [iters]   with NAME from tools/bench_conv.py LAYERS"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sg2im_amd import ops
from tools.bench_conv import LAYERS, NB as N, D
name = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bn = len(sys.argv) > 3 and 'bn' in sys.argv[3:]
BF16 = 'bf16' in sys.argv[3:]              # bf16 operands + weight mirror (the kernels a bf16 Trainer step runs)
if BF16:
  ops.CONV_COMPUTE = 1
for L in LAYERS:
  if L[0] == name:
    _, H, C0, C1, Cout, k, s, p = L[:8]
x0 = torch.randn(N, H, H, C0, device=D)
sc = torch.rand(C0, device=D) + 0.5 if bn else None
sh = torch.randn(C0, device=D) if bn else None
srcs = [ops.nhwc_src(x0, 0, sc, sh, 0.2 if bn else 1.0)]
if C1:      # (as in the network: the previous module's BatchNorm + LeakyReLU pending on the upsampled feature source)
  srcs.append(ops.nhwc_src(torch.randn(N, H // 2, H // 2, C1, device=D), 1, torch.rand(C1, device=D) + 0.5,
                           torch.randn(C1, device=D) * 0.1, 0.2))
d = ops.conv_desc(srcs, N, H, H, k, k, s, p)
Ct = C0 + C1
W = torch.randn(Cout, k, k, Ct, device=D) * 0.01
if BF16 and k == 3:
  _mir = torch.cat([W.reshape(-1).to(torch.bfloat16), torch.zeros(16, dtype=torch.bfloat16, device=D)])
  ops.WEIGHT_MIRROR, ops.WEIGHT_MIRROR_LOOKUP = True, (lambda w: _mir.data_ptr())
b = torch.randn(Cout, device=D)
y = torch.empty(N, d.out_h, d.out_w, Cout, device=D)
gy = torch.randn_like(y)
dx = torch.empty(N, H, H, Ct, device=D)
dw = torch.empty_like(W)
for _ in range(iters):
  ops.conv2d_forward(d, W, Cout, b, y, Cout)
  ops.conv2d_backward_data(d, W, Cout, gy, Cout, 0, Ct, dx, Ct)
  ops.conv2d_backward_weight(d, gy, Cout, Cout, dw)
torch.cuda.synchronize()
