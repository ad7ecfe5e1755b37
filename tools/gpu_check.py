"""Diagnostic sweep: run every HIP op against the CPU oracle / torch CPU reference and
print an error table (never raises).  Usage on the GPU box:
   python tools/gpu_check.py > gpurun_out/check.log 2>&1
"""
import ctypes
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

from oracle import sg2im_oracle as orc
from sg2im_amd import ops
from sg2im_amd._lib import call
from sg2im_amd import functional as HF
from sg2im_amd.synthetic import make_vocab, synthetic_batch
from tests import hip_harness as hh
from tests.util import load_golden, clone_params

D = torch.device('cuda', 0)
RESULTS = []


def err(a, b):
  a, b = a.detach().double().cpu(), b.detach().double().cpu()
  if a.shape != b.shape:
    return float('nan'), 'shape %s vs %s' % (tuple(a.shape), tuple(b.shape)), float('nan'), 0.0
  d = (a - b).abs().max().item() if a.numel() else 0.0
  s = b.abs().max().item() if b.numel() else 0.0
  return d / max(s, 1e-30), 'abs %.3e scale %.3e' % (d, s), d, s


def report(name, a, b, exact=False):
  """RESULTS rows: (name, rel-to-max error, info, abs error, scale)"""
  if exact:
    ok = torch.equal(a.detach().cpu(), b.detach().cpu())
    RESULTS.append((name, 0.0 if ok else 1.0, 'bit-exact' if ok else 'NOT bit-exact', 0.0 if ok else 1.0, 1.0))
  else:
    RESULTS.append((name,) + err(a, b))
  print('%-58s rel %.3e  %s' % (RESULTS[-1][0], RESULTS[-1][1], RESULTS[-1][2]), flush=True)


def section(fn):
  t0 = time.time()
  try:
    fn()
  except Exception:
    print('!!! %s raised' % fn.__name__)
    traceback.print_exc()
    RESULTS.append((fn.__name__, float('inf'), 'EXCEPTION', float('inf'), 0.0))
  torch.cuda.synchronize()
  print('--- %s done in %.1fs' % (fn.__name__, time.time() - t0), flush=True)


def conv_case(name, N, H, W, C0, C1, up1, Cout, k, stride, pad, bnact=False, seed=0):
  g = torch.Generator().manual_seed(seed)
  x0 = torch.randn(N, C0, H, W, generator=g)
  xs = [x0]
  if C1:
    h1, w1 = (H // 2, W // 2) if up1 else (H, W)
    x1 = torch.randn(N, C1, h1, w1, generator=g)
    xs.append(x1)
  Ct = C0 + C1
  Wt = torch.randn(Cout, Ct, k, k, generator=g) / (Ct * k * k) ** 0.5
  b = torch.randn(Cout, generator=g)
  sc = sh = None
  # CPU reference
  xr = [t.clone().requires_grad_(True) for t in xs]
  parts = [xr[0]]
  if bnact:
    sc = torch.rand(C0, generator=g) + 0.5
    sh = torch.randn(C0, generator=g) * 0.3
    parts[0] = F.leaky_relu(xr[0] * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1), 0.2)
  if C1:
    parts.append(F.interpolate(xr[1], scale_factor=2, mode='nearest') if up1 else xr[1])
  Wr, br = Wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
  y = F.conv2d(torch.cat(parts, 1), Wr, br, stride=stride, padding=pad)
  gy = torch.randn(y.shape, generator=g)
  y.backward(gy)
  # HIP
  nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(D)
  srcs = [ops.nhwc_src(nhwc(xs[0]), 0, sc.to(D) if bnact else None, sh.to(D) if bnact else None, 0.2 if bnact else 1.0)]
  if C1:
    srcs.append(ops.nhwc_src(nhwc(xs[1]), 1 if up1 else 0))
  d = ops.conv_desc(srcs, N, H, W, k, k, stride, pad)
  Wp = Wt.permute(0, 2, 3, 1).contiguous().to(D)
  out = torch.empty(N, d.out_h, d.out_w, Cout, device=D)
  ops.conv2d_forward(d, Wp, Cout, b.to(D), out, Cout)
  report(name + ' fwd', out.permute(0, 3, 1, 2), y)
  gyd = nhwc(gy)
  dw = torch.empty(Cout, k, k, Ct, device=D)
  db = torch.empty(Cout, device=D)
  ops.conv2d_backward_weight(d, gyd, Cout, Cout, dw, dbias=db)
  report(name + ' wgrad', dw.permute(0, 3, 1, 2), Wr.grad)
  report(name + ' bias grad (fused in wgrad)', db, br.grad)
  # accumulate mode: both gradients are added onto the existing buffers
  ops.conv2d_backward_weight(d, gyd, Cout, Cout, dw, accumulate=True, dbias=db)
  report(name + ' wgrad accumulate', dw.permute(0, 3, 1, 2), 2 * Wr.grad)
  report(name + ' bias grad accumulate', db, 2 * br.grad)
  if not bnact:
    dx0 = torch.empty(N, H, W, C0, device=D)
    ops.conv2d_backward_data(d, Wp, Cout, gyd, Cout, 0, C0, dx0, C0)
    report(name + ' dgrad0', dx0.permute(0, 3, 1, 2), xr[0].grad)
  if C1:
    dx1 = torch.empty(N, H, W, C1, device=D)
    ops.conv2d_backward_data(d, Wp, Cout, gyd, Cout, C0, C1, dx1, C1)
    ref = xr[1].grad
    got = dx1.permute(0, 3, 1, 2)
    if up1:
      got = F.avg_pool2d(got.cpu(), 2) * 4
    report(name + ' dgrad1', got, ref)


def conv_case_wide_weight(name, N, H, W, C0, Wch, Cout, k, pad, seed=0):
  """sg2im_conv_desc.weight_channels: weight rows with Wch > C0 channels per tap, sources supplying the first
  C0 - the reference computes with the missing channels fed zeros (crn.py:105's all-zero feature channel):
  same forward, same input gradient, and the extra channels' weight gradient is exactly zero."""
  g = torch.Generator().manual_seed(seed)
  x0 = torch.randn(N, C0, H, W, generator=g)
  Wt = torch.randn(Cout, Wch, k, k, generator=g) / (Wch * k * k) ** 0.5
  b = torch.randn(Cout, generator=g)
  xr = x0.clone().requires_grad_(True)
  Wr, br = Wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
  y = F.conv2d(torch.cat([xr, torch.zeros(N, Wch - C0, H, W)], 1), Wr, br, padding=pad)
  gy = torch.randn(y.shape, generator=g)
  y.backward(gy)
  nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(D)
  d = ops.conv_desc([ops.nhwc_src(nhwc(x0))], N, H, W, k, k, 1, pad, weight_channels=Wch)
  Wp = Wt.permute(0, 2, 3, 1).contiguous().to(D)
  out = torch.empty(N, d.out_h, d.out_w, Cout, device=D)
  ops.conv2d_forward(d, Wp, Cout, b.to(D), out, Cout)
  report(name + ' fwd', out.permute(0, 3, 1, 2), y)
  gyd = nhwc(gy)
  dw = torch.full((Cout, k, k, Wch), 7.0, device=D)          # the kernel must leave the extra columns alone
  db = torch.empty(Cout, device=D)
  ops.conv2d_backward_weight(d, gyd, Cout, Cout, dw, dbias=db)
  report(name + ' wgrad (supplied channels)', dw[..., :C0].permute(0, 3, 1, 2), Wr.grad[:, :C0])
  report(name + ' wgrad leaves the other columns untouched', dw[..., C0:], torch.full((Cout, k, k, Wch - C0), 7.0))
  report(name + ' bias grad', db, br.grad)
  ops.conv2d_backward_weight(d, gyd, Cout, Cout, dw, accumulate=True, dbias=db)
  report(name + ' wgrad accumulate', dw[..., :C0].permute(0, 3, 1, 2), 2 * Wr.grad[:, :C0])
  dx0 = torch.empty(N, H, W, C0, device=D)
  ops.conv2d_backward_data(d, Wp, Cout, gyd, Cout, 0, C0, dx0, C0)
  report(name + ' dgrad', dx0.permute(0, 3, 1, 2), xr.grad)


def conv_case_bf16(name, N, H, W, C0, C1, up1, Cout, k, stride, pad, bnact=False, seed=0):
  """compute_dtype 1: the kernels round both operands to bf16 and accumulate in fp32; the reference
  does the same with torch (operands through .bfloat16().float(), fp32 convolution), so the two
  agree to fp32 summation-order accuracy - an exact emulation, op by op."""
  g = torch.Generator().manual_seed(seed)
  rb = lambda t: t.bfloat16().float()
  x0 = torch.randn(N, C0, H, W, generator=g)
  xs = [x0]
  if C1:
    h1, w1 = (H // 2, W // 2) if up1 else (H, W)
    xs.append(torch.randn(N, C1, h1, w1, generator=g))
  Ct = C0 + C1
  Wt = torch.randn(Cout, Ct, k, k, generator=g) / (Ct * k * k) ** 0.5
  b = torch.randn(Cout, generator=g)
  sc = sh = None
  parts = [xs[0]]
  if bnact:
    sc = torch.rand(C0, generator=g) + 0.5
    sh = torch.randn(C0, generator=g) * 0.3
    parts[0] = F.leaky_relu(xs[0] * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1), 0.2)
  if C1:
    parts.append(F.interpolate(xs[1], scale_factor=2, mode='nearest') if up1 else xs[1])
  xcat = torch.cat(parts, 1)
  xb = rb(xcat)
  Wb = rb(Wt)
  y = F.conv2d(xb, Wb, b, stride=stride, padding=pad)
  gy = torch.randn(y.shape, generator=g)
  gyb = rb(gy)
  xr = xcat.clone().requires_grad_(True)
  dx_ref = torch.autograd.grad(F.conv2d(xr, Wb, None, stride=stride, padding=pad), xr, gyb)[0]
  Wr = Wt.clone().requires_grad_(True)
  dw_ref = torch.autograd.grad(F.conv2d(xb, Wr, None, stride=stride, padding=pad), Wr, gyb)[0]
  nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(D)
  srcs = [ops.nhwc_src(nhwc(xs[0]), 0, sc.to(D) if bnact else None, sh.to(D) if bnact else None, 0.2 if bnact else 1.0)]
  if C1:
    srcs.append(ops.nhwc_src(nhwc(xs[1]), 1 if up1 else 0))
  d = ops.conv_desc(srcs, N, H, W, k, k, stride, pad, compute=1)
  Wp = Wt.permute(0, 2, 3, 1).contiguous().to(D)
  out = torch.empty(N, d.out_h, d.out_w, Cout, device=D)
  ops.conv2d_forward(d, Wp, Cout, b.to(D), out, Cout)
  report(name + ' bf16 fwd', out.permute(0, 3, 1, 2), y)
  gyd = nhwc(gy)
  dw = torch.empty(Cout, k, k, Ct, device=D)
  db = torch.empty(Cout, device=D)
  ops.conv2d_backward_weight(d, gyd, Cout, Cout, dw, dbias=db)
  report(name + ' bf16 wgrad', dw.permute(0, 3, 1, 2), dw_ref)
  report(name + ' bf16 bias grad (fp32 sums)', db, gy.sum((0, 2, 3)))
  if not bnact:
    dx0 = torch.empty(N, H, W, C0, device=D)
    ops.conv2d_backward_data(d, Wp, Cout, gyd, Cout, 0, C0, dx0, C0)
    report(name + ' bf16 dgrad0', dx0.permute(0, 3, 1, 2), dx_ref[:, :C0])
  if C1:
    dx1 = torch.empty(N, H, W, C1, device=D)
    ops.conv2d_backward_data(d, Wp, Cout, gyd, Cout, C0, C1, dx1, C1)
    report(name + ' bf16 dgrad1', dx1.permute(0, 3, 1, 2), dx_ref[:, C0:])


def sec_conv_bf16():
  conv_case_bf16('conv3x3 64->64 16x16 (tile64)', 2, 16, 16, 64, 0, 0, 64, 3, 1, 1)
  conv_case_bf16('conv3x3 160+128up->128 16x16', 4, 16, 16, 160, 128, 1, 128, 3, 1, 1)
  conv_case_bf16('conv3x3 bnact 128->256 8x8 splitK', 4, 8, 8, 128, 0, 0, 256, 3, 1, 1, bnact=True)
  conv_case_bf16('conv3x3 32->64 64x64 (128x64 tile)', 4, 64, 64, 32, 0, 0, 64, 3, 1, 1)
  conv_case_bf16('conv3x3 96->192 32x32 (128x128 tile)', 8, 32, 32, 96, 0, 0, 192, 3, 1, 1)
  conv_case_bf16('conv4x4s2 64->128 valid 31x31 (parity dgrad)', 4, 31, 31, 64, 0, 0, 128, 4, 2, 0)
  conv_case_bf16('conv4x4s2 128->256 valid 14x14', 4, 14, 14, 128, 0, 0, 256, 4, 2, 0)
  conv_case_bf16('conv3x3 1184->512 8x8 (m1.conv0 shape)', 4, 8, 8, 160, 1024, 1, 512, 3, 1, 1)
  conv_case_bf16('conv3x3 36->20 pad1 9x11 (ragged tiles)', 3, 9, 11, 36, 0, 0, 20, 3, 1, 1)
  # halo'd-tile kernels with bf16 operands: every patch form, pending affine, ragged chunks, split-K
  conv_case_bf16('halo conv3x3 bnact 64->96 32x32', 4, 32, 32, 64, 0, 0, 96, 3, 1, 1, bnact=True)
  conv_case_bf16('halo conv3x3 80+48up->48 12x32 (4x32 patches)', 3, 12, 32, 80, 48, 1, 48, 3, 1, 1)
  conv_case_bf16('halo conv3x3 64+64up->64 6x64 (2x64 patches)', 2, 6, 64, 64, 64, 1, 64, 3, 1, 1)
  conv_case_bf16('halo conv3x3 32->256 16x16 (split-K)', 16, 16, 16, 32, 0, 0, 256, 3, 1, 1)


class _Bn(object):
  """parameter container with the attributes of nn.BatchNorm2d the ops read (device tensors)"""

  def __init__(self, C, g):
    self.weight = (torch.rand(C, generator=g) + 0.5).to(D)
    self.bias = (torch.randn(C, generator=g) * 0.3).to(D)
    self.running_mean = torch.zeros(C, device=D)
    self.running_var = torch.ones(C, device=D)
    self.num_batches_tracked = torch.zeros((), dtype=torch.long, device=D)


def conv_bn_case(name, N, H, W, Cin, Cout, k, stride, pad, out_slope=1.0, live=None, unbiased_mult=1, compute=0, seed=0):
  """sg2im_conv2d_forward_bn: conv output AND the BatchNorm statistics / folded affine / running statistics of that
  output (the reductions ride in the conv epilogue or the split-K finish) against torch; `live`: only the first
  `live` batch entries are real (padded row batch)."""
  g = torch.Generator().manual_seed(seed)
  x = torch.randn(N, Cin, H, W, generator=g) + 0.7
  Wt = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
  b = torch.randn(Cout, generator=g) * 2.0 + 1.0           # (a large mean / std ratio: the pivot-shifted sums matter)
  bn = _Bn(Cout, g)
  nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(D)
  d = ops.conv_desc([ops.nhwc_src(nhwc(x))], N, H, W, k, k, stride, pad, compute=compute)
  Wp = Wt.permute(0, 2, 3, 1).contiguous().to(D)
  out = torch.empty(N, d.out_h, d.out_w, Cout, device=D)
  cnt = None
  if live is not None:
    cnt = (torch.tensor([live], dtype=torch.int32, device=D), d.out_h * d.out_w)
  rows = N * d.out_h * d.out_w
  st = ops.conv2d_forward_bn(d, Wp, Cout, b.to(D), out, Cout, bn, True, 1e-5, 0.1, out_slope=out_slope,
                             unbiased_rows=unbiased_mult * rows if unbiased_mult != 1 else 0, count=cnt)
  if compute == 0:
    y = F.leaky_relu(F.conv2d(x, Wt, b, stride=stride, padding=pad), out_slope) if out_slope != 1.0 else \
        F.conv2d(x, Wt, b, stride=stride, padding=pad)
    report(name + ' out', out.permute(0, 3, 1, 2), y)
  # the statistics are checked against the tensor the kernel itself wrote (exact up to summation order)
  o = out.detach().cpu().double()[:live if live is not None else N].reshape(-1, Cout)
  mean, var = o.mean(0), o.var(0, unbiased=False)
  invstd = 1.0 / torch.sqrt(var + 1e-5)
  gam, bet = bn.weight.cpu().double(), bn.bias.cpu().double()
  report(name + ' mean', st.mean, mean)
  report(name + ' invstd', st.invstd, invstd)
  report(name + ' folded scale', st.scale, gam * invstd)
  report(name + ' folded shift', st.shift, bet - mean * gam * invstd)
  n = o.size(0) * unbiased_mult
  report(name + ' running_mean', bn.running_mean, 0.1 * mean)
  report(name + ' running_var', bn.running_var, 0.9 + 0.1 * var * n / (n - 1))
  report(name + ' num_batches_tracked', bn.num_batches_tracked.float(), torch.ones(()))


def dgrad_bn_case(name, N, h, w, C, Cout2, k, pad, pool2, slope=0.2, live=None, seed=0):
  """sg2im_conv2d_backward_data_bn + sg2im_bn_backward_apply against torch autograd through
  conv(upsample?(leaky(batch_norm(y)))): the data gradient w.r.t. the activated BatchNorm output, the
  BatchNorm's dgamma / dbeta and dy."""
  g = torch.Generator().manual_seed(seed)
  y = torch.randn(N, C, h, w, generator=g) * 1.5 + 0.4
  bn = _Bn(C, g)
  f = 2 if pool2 else 1
  H2, W2 = h * f, w * f
  Wt = torch.randn(Cout2, C, k, k, generator=g) / (C * k * k) ** 0.5
  nl = live if live is not None else N
  # torch reference on the real entries only (a padded batch's dummy rows see a zero upstream gradient)
  yr = y[:nl].clone().requires_grad_(True)
  gam = bn.weight.cpu().clone().requires_grad_(True)
  bet = bn.bias.cpu().clone().requires_grad_(True)
  z = F.leaky_relu(F.batch_norm(yr, None, None, gam, bet, True, 0.1, 1e-5), slope)
  zin = F.interpolate(z, scale_factor=2, mode='nearest') if pool2 else z
  o2 = F.conv2d(zin, Wt, None, padding=pad)
  gy2 = torch.randn((N,) + tuple(o2.shape[1:]), generator=g)
  gy2[nl:] = 0
  o2.backward(gy2[:nl])
  nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(D)
  yd = nhwc(y)
  cnt = None if live is None else (torch.tensor([live], dtype=torch.int32, device=D), h * w)
  st = ops.bn_stats(yd, N * h * w, C, C, bn, True, 1e-5, 0.1, count=cnt)
  d2 = ops.conv_desc([ops.nhwc_src(yd, 1 if pool2 else 0, st.scale, st.shift, slope)], N, H2, W2, k, k, 1, pad)
  Wp = Wt.permute(0, 2, 3, 1).contiguous().to(D)
  gz = torch.empty(N, H2, W2, C, device=D)
  dgam, dbet = torch.empty(C, device=D), torch.empty(C, device=D)
  coef = ops.conv2d_backward_data_bn(d2, Wp, Cout2, nhwc(gy2), Cout2, 0, C, gz, C, yd, C, 1 if pool2 else 0, bn.weight, st,
                                     slope, True, dgam, dbet, False, count=cnt)
  dy = torch.empty(N, h, w, C, device=D)
  ops.bn_backward_apply(HF._fptr(gz), C, 1 if pool2 else 0, N, h, w, yd, C, C, st, slope, coef, dy, count=cnt)
  report(name + ' gz (data gradient)', gz.permute(0, 3, 1, 2)[:nl], _dgrad_ref(zin, Wt, pad, gy2[:nl]))
  report(name + ' dgamma', dgam, gam.grad)
  report(name + ' dbeta', dbet, bet.grad)
  report(name + ' dy', dy.permute(0, 3, 1, 2)[:nl], yr.grad)
  if live is not None:
    report(name + ' dy of padding rows is zero', dy[nl:], torch.zeros_like(dy[nl:].cpu()))
  # accumulate mode of the parameter gradients
  ops.conv2d_backward_data_bn(d2, Wp, Cout2, nhwc(gy2), Cout2, 0, C, gz, C, yd, C, 1 if pool2 else 0, bn.weight, st,
                              slope, True, dgam, dbet, True, count=cnt)
  report(name + ' dgamma accumulate', dgam, 2 * gam.grad)


def _dgrad_ref(zin, Wt, pad, gy):
  zi = zin.detach().clone().requires_grad_(True)
  F.conv2d(zi, Wt, None, padding=pad).backward(gy)
  return zi.grad


def sec_conv_bn():
  """convolution + BatchNorm reductions in one set of launches; with SG2IM_PLAN_TUNE=1 in the environment every
  tile shape and the split-K finish form are forced in turn (SG2IM_FORCE_PLAN is re-read per launch)"""
  tune = os.environ.get('SG2IM_PLAN_TUNE') == '1'
  plans = ['0,1', '1,1', '2,1', '3,1', '2,3', '0,2'] if tune else [None]
  for pl in plans:
    if pl is not None:
      os.environ['SG2IM_FORCE_PLAN'] = pl
    tag = ' [plan %s]' % pl if pl else ''
    conv_bn_case('conv_bn 3x3 32->64 32x32' + tag, 4, 32, 32, 32, 64, 3, 1, 1)
    conv_bn_case('conv_bn 3x3 64->192 16x16 relu' + tag, 5, 16, 16, 64, 192, 3, 1, 1, out_slope=0.0, unbiased_mult=4)
    conv_bn_case('conv_bn 4x4s2 64->128 valid 15x15' + tag, 6, 15, 15, 64, 128, 4, 2, 0)
    conv_bn_case('conv_bn 3x3 128->128 8x8 padded batch' + tag, 12, 8, 8, 128, 128, 3, 1, 1, out_slope=0.0, live=7, unbiased_mult=4)
    conv_bn_case('conv_bn 3x3 64->64 16x16 bf16 operands' + tag, 8, 16, 16, 64, 64, 3, 1, 1, compute=1)
    dgrad_bn_case('dgrad_bn 3x3 C64 <- 96 16x16' + tag, 4, 16, 16, 64, 96, 3, 1, False)
    dgrad_bn_case('dgrad_bn 3x3 C128 <- 64 8x8 upsampled (pool2)' + tag, 4, 8, 8, 128, 64, 3, 1, True)
    dgrad_bn_case('dgrad_bn 3x3 C128 <- 128 4x4 pool2 padded batch relu' + tag, 9, 4, 4, 128, 128, 3, 1, True, slope=0.0, live=5)
  if tune:
    os.environ.pop('SG2IM_FORCE_PLAN', None)
  # shapes whose default plan takes the epilogue form (no split-K) and the finish form
  conv_bn_case('conv_bn 3x3 64->64 64x64 batch 8 (epilogue form)', 8, 64, 64, 64, 64, 3, 1, 1)
  conv_bn_case('conv_bn 3x3 160->1024 4x4 batch 32 (split-K finish form)', 32, 4, 4, 160, 1024, 3, 1, 1)
  conv_bn_case('conv_bn 3x3 6->20 9x11 (scalar loaders: standalone fallback)', 3, 9, 11, 6, 20, 3, 1, 1)
  dgrad_bn_case('dgrad_bn 3x3 C64 <- 64 64x64 batch 8 (epilogue form)', 8, 64, 64, 64, 64, 3, 1, False)
  dgrad_bn_case('dgrad_bn 3x3 C512 <- 256 8x8 pool2 (finish form)', 8, 4, 4, 512, 256, 3, 1, True)
  dgrad_bn_case('dgrad_bn 3x3 C6 <- 20 9x11 (scalar loaders: standalone fallback)', 3, 9, 11, 6, 20, 3, 1, False)
  # round 5: small maps with enough rows for the epilogue forms (no split-K)
  conv_bn_case('conv_bn 3x3 32->128 4x4 batch 2048 (epilogue form)', 2048, 4, 4, 32, 128, 3, 1, 1)
  conv_bn_case('conv_bn 3x3 64->128 8x8 batch 512 (epilogue form)', 512, 8, 8, 64, 128, 3, 1, 1)
  dgrad_bn_case('dgrad_bn 3x3 C128 <- 32 4x4 batch 2048 (epilogue form)', 2048, 4, 4, 128, 32, 3, 1, False)
  dgrad_bn_case('dgrad_bn 3x3 C128 <- 32 4x4 -> 8x8 pool2 batch 512 (epilogue form)', 512, 4, 4, 128, 32, 3, 1, True)


def sec_conv():
  conv_case('conv3x3 64->64 16x16 (tile64)', 2, 16, 16, 64, 0, 0, 64, 3, 1, 1)
  conv_case('conv3x3 160+128up->128 16x16', 4, 16, 16, 160, 128, 1, 128, 3, 1, 1)
  conv_case('conv3x3 bnact 128->256 8x8 splitK', 4, 8, 8, 128, 0, 0, 256, 3, 1, 1, bnact=True)
  conv_case('conv3x3 32->64 64x64 (128x64 tile)', 4, 64, 64, 32, 0, 0, 64, 3, 1, 1)
  conv_case('conv3x3 96->192 32x32 (128x128 tile)', 8, 32, 32, 96, 0, 0, 192, 3, 1, 1)
  conv_case('conv4x4s2 3->64 valid 64x64 (VEC1)', 4, 64, 64, 3, 0, 0, 64, 4, 2, 0)
  conv_case('conv4x4s2 64->128 valid 31x31', 4, 31, 31, 64, 0, 0, 128, 4, 2, 0)
  conv_case('conv4x4s2 128->256 valid 14x14', 4, 14, 14, 128, 0, 0, 256, 4, 2, 0)
  conv_case('conv3x3 160+1up->96 4x4 (VEC1 concat)', 4, 4, 4, 160, 1, 1, 96, 3, 1, 1)
  conv_case('conv3x3 3->20 pad1 9x11 (few-channel dgrad)', 3, 9, 11, 3, 0, 0, 20, 3, 1, 1)
  conv_case('conv4x4s2 2->8 pad1 10x10 (few-channel dgrad)', 2, 10, 10, 2, 0, 0, 8, 4, 2, 1)
  conv_case('conv4x4s2 3->64 valid 32x32 batch 37 (four lanes per pixel, ragged last workgroup)', 37, 32, 32, 3, 0, 0, 64, 4, 2, 0)
  conv_case('conv3x3s2 4->80 pad1 13x9 (four lanes per pixel, two channel rounds)', 3, 13, 9, 4, 0, 0, 80, 3, 2, 1)
  conv_case('conv2x2 1->12 valid 7x7 (four lanes per pixel, stride 1)', 2, 7, 7, 1, 0, 0, 12, 2, 1, 0)
  conv_case('conv1x1 64->3 32x32', 2, 32, 32, 64, 0, 0, 3, 1, 1, 0)
  conv_case('conv1x1 128->1 16x16', 8, 16, 16, 128, 0, 0, 1, 1, 1, 0)
  # (csrc/conv_fewout.h: ragged last row block, pending affine in the weight gradient's loader, 2 and 4 output channels,
  #  a channel count the row-group layout does not take -> implicit GEMM)
  conv_case('conv1x1 bnact 64->3 40x40 batch 3 (few outputs, ragged blocks)', 3, 40, 40, 64, 0, 0, 3, 1, 1, 0, bnact=True)
  conv_case('conv1x1 32->4 20x20 (few outputs)', 2, 20, 20, 32, 0, 0, 4, 1, 1, 0)
  conv_case('conv1x1 16->2 8x8 (few outputs)', 5, 8, 8, 16, 0, 0, 2, 1, 1, 0)
  conv_case('conv1x1 48->3 12x12 (12 float4 lanes: not a power of two)', 2, 12, 12, 48, 0, 0, 3, 1, 1, 0)
  conv_case('conv3x3 1184->512 8x8 (m1.conv0 shape)', 4, 8, 8, 160, 1024, 1, 512, 3, 1, 1)
  # plain sources, stride 1, channels % 32 == 0
  conv_case('v2 conv3x3 160+128up->64 32x32 (256x64 tile)', 16, 32, 32, 160, 128, 1, 64, 3, 1, 1)
  conv_case('v2 conv3x3 160+512up->256 16x16 (split-K)', 8, 16, 16, 160, 512, 1, 256, 3, 1, 1)
  conv_case('v2 conv3x3 96->80 19x21 (ragged rows / columns)', 5, 19, 21, 96, 0, 0, 80, 3, 1, 1)
  conv_case('v2 conv5x5 64->96 24x24 pad2', 6, 24, 24, 64, 0, 0, 96, 5, 1, 2)
  conv_case('v2 conv1x1 128->128 32x32', 8, 32, 32, 128, 0, 0, 128, 1, 1, 0)
  # halo'd-tile kernels (csrc/conv_halo.h): ragged channel chunks, pending affine, all three patch shapes
  conv_case('halo conv3x3 80+48up->48 12x32 (4x32 patches, ragged chunks)', 3, 12, 32, 80, 48, 1, 48, 3, 1, 1)
  conv_case('halo conv3x3 bnact 64->96 32x32', 4, 32, 32, 64, 0, 0, 96, 3, 1, 1, bnact=True)
  conv_case('halo conv3x3 bnact 36->40 8x16 (8x16 patches)', 5, 8, 16, 36, 0, 0, 40, 3, 1, 1, bnact=True)
  conv_case('halo conv3x3 64+64up->64 6x64 (2x64 patches)', 2, 6, 64, 64, 64, 1, 64, 3, 1, 1)
  conv_case('halo conv3x3 64+64up->64 64x64 (8x16 patches)', 2, 64, 64, 64, 64, 1, 64, 3, 1, 1)
  conv_case('halo conv3x3 32->256 16x16 (128-wide tiles / split-K)', 16, 16, 16, 32, 0, 0, 256, 3, 1, 1)
  # round 5: the 4 x 4 / 8 x 8 levels at several batch sizes (a multi-image-patch form of the halo'd kernel was probed on
  # these shapes and removed: profiles/r5_halo_small_maps_ab.txt)
  conv_case('conv3x3 bnact 64->96 4x4 batch 16 (split-K)', 16, 4, 4, 64, 0, 0, 96, 3, 1, 1, bnact=True)
  conv_case('conv3x3 64->64 4x4 batch 8', 8, 4, 4, 64, 0, 0, 64, 3, 1, 1)
  conv_case('conv3x3 160+128up->64 8x8 batch 6', 6, 8, 8, 160, 128, 1, 64, 3, 1, 1)
  conv_case('conv3x3 bnact 36->40 8x8 batch 2 (ragged chunks)', 2, 8, 8, 36, 0, 0, 40, 3, 1, 1, bnact=True)
  conv_case('conv3x3 64->64 4x4 batch 12', 12, 4, 4, 64, 0, 0, 64, 3, 1, 1)
  # weight gradients of <= 64 output channels over rows of 32 k pixels (pending affine, ragged blocks, two sources)
  conv_case('conv3x3 bnact 64->64 32x32', 4, 32, 32, 64, 0, 0, 64, 3, 1, 1, bnact=True)
  conv_case('conv3x3 96+40up->48 8x32 (ragged channel block, 48 outputs)', 3, 8, 32, 96, 40, 1, 48, 3, 1, 1)
  conv_case('conv3x3 160+128up->64 64x64 batch 4 (m4.conv0 shape)', 4, 64, 64, 160, 128, 1, 64, 3, 1, 1)
  # weight rows wider than the sources (sg2im_conv_desc.weight_channels): the first refinement module
  conv_case_wide_weight('conv3x3 160 of 161 -> 1024 4x4 (m0.conv0, split-K)', 32, 4, 4, 160, 161, 1024, 3, 1)
  conv_case_wide_weight('conv3x3 128 of 129 -> 96 8x8', 4, 8, 8, 128, 129, 96, 3, 1)
  conv_case_wide_weight('conv3x3 6 of 9 -> 20 5x7 (scalar loaders)', 3, 5, 7, 6, 9, 20, 3, 1)
  conv_case_wide_weight('conv1x1 64 of 72 -> 64 16x16', 4, 16, 16, 64, 72, 64, 1, 0)


def sec_dgrad_act():
  """sg2im_conv2d_backward_data_act - the LeakyReLU / ReLU mask of the layer a data gradient flows into, applied by
  the data gradient's own launches - against the two launches it replaces (sg2im_conv2d_backward_data +
  sg2im_act_backward), BIT FOR BIT: linear layers (epilogue and split-K finish, float4 and scalar finish forms),
  3x3 convolutions on the halo'd-tile kernel (with and without split-K) and on the per-tap kernel, the 1x1 output
  convolution with 3 output channels (scalar loaders), and the forms that fall back to the extra launch inside the
  entry point (few input channels, the stride-2 parity form); zeros in `act` take the slope."""
  from sg2im_amd.ops import conv_desc, nhwc_src, rows_src
  g = torch.Generator().manual_seed(21)
  cases = [  # (tag, batch, h, w, cin, cout, k, stride, pad, slope)
    ('linear 203x512->128 (GraphTripleConv net2, split-K)', 203, 1, 1, 512, 128, 1, 1, 0, 0.0),
    ('linear 342x1152->512 (net1)', 342, 1, 1, 512, 1152, 1, 1, 0, 0.0),
    ('linear 7x20->12 (scalar loaders)', 7, 1, 1, 20, 12, 1, 1, 0, 0.0),
    ('linear 4000x64->32 (no split)', 4000, 1, 1, 64, 32, 1, 1, 0, 0.2),
    ('conv1x1 64->3 64x64 (output_conv[2])', 4, 64, 64, 64, 3, 1, 1, 0, 0.2),
    ('conv3x3 64->64 64x64 (halo, 8x16 patches)', 2, 64, 64, 64, 64, 3, 1, 1, 0.2),
    ('conv3x3 256->32 16x16 (halo, split-K)', 4, 16, 16, 32, 256, 3, 1, 1, 0.2),
    ('conv3x3 96->80 19x21 (per-tap)', 3, 19, 21, 80, 96, 3, 1, 1, 0.01),
    ('conv3x3 3->20 9x11 (few-channel fallback)', 3, 9, 11, 3, 20, 3, 1, 1, 0.2),
    ('conv4x4s2 64->128 valid 31x31 (parity fallback)', 2, 31, 31, 64, 128, 4, 2, 0, 0.2)]
  for tag, B, h, w, cin, cout, k, stride, pad, slope in cases:
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    W = (torch.randn(cout, k, k, cin, generator=g) / (k * k * cin) ** 0.5).to(D)
    dy = torch.randn(B, ho, wo, cout, generator=g).to(D)
    act = torch.randn(B, h, w, cin, generator=g)
    act[act.abs() < 0.3] = 0.0                                   # exact zeros: the ReLU'd entries of a real activation
    act = act.to(D)
    x = torch.zeros(B, h, w, cin, device=D)                      # (the sources are not dereferenced by a data gradient)
    desc = conv_desc([nhwc_src(x)], B, h, w, k, k, stride, pad)
    want = torch.empty(B, h, w, cin, device=D)
    ops.conv2d_backward_data(desc, W, cout, dy, cout, 0, cin, want, cin)
    ops.act_backward(ctypes.c_void_p(want.data_ptr()), cin, 0, B, h, w, act, cin, cin, slope, want)
    got = torch.full((B, h, w, cin), float('nan'), device=D)
    keep, ops.FUSE_ACT_BWD = ops.FUSE_ACT_BWD, True
    try:
      ops.conv2d_backward_data_act(desc, W, cout, dy, cout, 0, cin, got, cin, act, cin, slope)
    finally:
      ops.FUSE_ACT_BWD = keep
    report('dgrad+act ' + tag, got, want, exact=True)
  # a column slice of a wider activation as the mask (row stride > channels) and a channel sub-range of the input
  B, h, w, cin, cout = 3, 16, 16, 96, 64
  W = (torch.randn(cout, 3, 3, cin, generator=g) / (9 * cin) ** 0.5).to(D)
  dy = torch.randn(B, h, w, cout, generator=g).to(D)
  wide = torch.randn(B, h, w, 80, generator=g).to(D)
  desc = conv_desc([nhwc_src(torch.zeros(B, h, w, cin, device=D))], B, h, w, 3, 3, 1, 1)
  want = torch.empty(B, h, w, 64, device=D)
  ops.conv2d_backward_data(desc, W, cout, dy, cout, 32, 64, want, 64)
  ops.act_backward(ctypes.c_void_p(want.data_ptr()), 64, 0, B, h, w, wide[..., 8:72], 80, 64, 0.2, want)
  got = torch.empty(B, h, w, 64, device=D)
  call('sg2im_conv2d_backward_data_act', ctypes.byref(desc), ctypes.c_void_p(W.data_ptr()), cout, ctypes.c_void_p(dy.data_ptr()),
       cout, 32, 64, ctypes.c_void_p(got.data_ptr()), 64, ctypes.c_void_p(wide.data_ptr() + 4 * 8), 80, 0.2,
       ctypes.c_void_p(ops.workspace(D).data_ptr()), ops.workspace(D).numel() * 4, ops._stream())
  report('dgrad+act channel range [32, 96) with a strided mask', got, want, exact=True)


def sec_linear():
  g = torch.Generator().manual_seed(1)
  for (M, K, N) in ((203, 128, 512), (342, 512, 1152), (7, 20, 12), (300, 264, 512)):
    x = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    xr, Wr, br = [t.clone().requires_grad_(True) for t in (x, W, b)]
    y = F.relu(F.linear(xr, Wr, br))
    gy = torch.randn(M, N, generator=g)
    y.backward(gy)
    xd, Wd, bd = [t.clone().to(D).requires_grad_(True) for t in (x, W, b)]
    yd = HF.LinearAct.apply(xd, Wd, bd, 0.0)
    yd.backward(gy.to(D))
    tag = 'linear %dx%d->%d' % (M, K, N)
    report(tag + ' fwd', yd, y)
    report(tag + ' dx', xd.grad, xr.grad)
    report(tag + ' dW', Wd.grad, Wr.grad)
    report(tag + ' db', bd.grad, br.grad)


def sec_two_heads():
  """two nn.Linear heads on one input (AcDiscriminator's real / class heads, discriminators.py:66-75) in one launch
  per direction vs torch; ragged row counts (the last workgroup of 4 rows), a column view with a row stride as x"""
  g = torch.Generator().manual_seed(12)
  for (M, K, n1, n2, wide) in ((203, 1024, 1, 184, False), (7, 20, 3, 5, False), (224, 256, 1, 10, True), (1, 1536, 2, 1, False)):
    xw = torch.randn(M, K + (12 if wide else 0), generator=g)
    x = xw[:, :K]
    W1 = torch.randn(n1, K, generator=g) / K ** 0.5
    W2 = torch.randn(n2, K, generator=g) / K ** 0.5
    b1, b2 = torch.randn(n1, generator=g), torch.randn(n2, generator=g)
    ref = [t.clone().requires_grad_(True) for t in (x, W1, b1, W2, b2)]
    y1, y2 = F.linear(ref[0], ref[1], ref[2]), F.linear(ref[0], ref[3], ref[4])
    g1, g2 = torch.randn(M, n1, generator=g), torch.randn(M, n2, generator=g)
    torch.autograd.backward([y1, y2], [g1, g2])
    assert ops.two_heads_supported(K, n1, n2)
    xd = xw.to(D)[:, :K].detach().requires_grad_(True) if wide else x.clone().to(D).requires_grad_(True)
    dev = [xd] + [t.clone().to(D).requires_grad_(True) for t in (W1, b1, W2, b2)]
    z1, z2 = HF.TwoHeads.apply(*dev)
    torch.autograd.backward([z1, z2], [g1.to(D), g2.to(D)])
    tag = 'two heads %dx%d -> %d + %d%s' % (M, K, n1, n2, ' (strided x)' if wide else '')
    report(tag + ' y1', z1, y1)
    report(tag + ' y2', z2, y2)
    for name, a, b in zip(('dx', 'dW1', 'db1', 'dW2', 'db2'), dev, ref):
      report(tag + ' ' + name, a.grad, b.grad)
  assert not ops.two_heads_supported(1022, 1, 4) and not ops.two_heads_supported(2048, 1, 4)


def sec_pool():
  g = torch.Generator().manual_seed(3)
  for (T, O, H, Dd) in ((700, 9, 24, 8), (342, 203, 512, 128), (5, 4, 6, 3)):
    new_t = torch.randn(T, 2 * H + Dd, generator=g) * 100
    s = torch.randint(0, O, (T,), generator=g)
    o = torch.randint(0, max(O - 1, 1), (T,), generator=g)
    csr = ops.Csr(s.to(D), o.to(D), O)
    nt = new_t.to(D)
    for pooling in ('sum', 'avg'):
      want, _ = orc.gconv_pool(new_t, s, o, O, H, Dd, pooling)
      out = torch.empty(O, H, device=D)
      ops.segment_sum(nt[:, :H], nt[:, H + Dd:], csr, H, pooling == 'avg', out)
      report('pool %s T=%d O=%d H=%d' % (pooling, T, O, H), out, want, exact=True)
    rp = csr.row_ptr.cpu()
    cnt = torch.bincount(torch.cat([s, o]), minlength=O)
    report('csr counts T=%d' % T, (rp[1:] - rp[:-1]).long(), cnt, exact=True)
  tab = torch.randn(50, 16, generator=g)
  idx = torch.randint(0, 50, (77,), generator=g)
  report('gather rows', ops.gather_rows(tab.to(D), idx.to(D), torch.empty(77, 16, device=D)), tab[idx], exact=True)


def sec_gconv():
  g = torch.Generator().manual_seed(5)
  batch = synthetic_batch(8, seed=2)
  objs, triples = batch[1], batch[4]
  O, T = objs.numel(), triples.size(0)
  Din, H, Dout = 128, 512, 128
  P = {}
  orc._lin(P, 'g.net1.0', H, 3 * Din, g, True); orc._lin(P, 'g.net1.2', 2 * H + Dout, H, g, True)
  orc._lin(P, 'g.net2.0', H, H, g, True); orc._lin(P, 'g.net2.2', Dout, H, g, True)
  ov, pv = torch.randn(O, Din, generator=g), torch.randn(T, Din, generator=g)
  edges = torch.stack([triples[:, 0], triples[:, 2]], 1)
  Pl = hh.oracle_leafs(P)
  ovr, pvr = ov.clone().requires_grad_(True), pv.clone().requires_grad_(True)
  no, npd = orc.graph_triple_conv(Pl, 'g', ovr, pvr, edges, H, Dout, 'avg')
  go, gp = torch.randn(O, Dout, generator=g), torch.randn(T, Dout, generator=g)
  (no * go).sum().add((npd * gp).sum()).backward()
  from sg2im_amd.graph import GraphTripleConv
  m = GraphTripleConv(Din, Dout, H, 'avg')
  hh.load_params(m, {k[2:]: v for k, v in P.items()})
  m = m.to(D)
  ovd, pvd = ov.to(D).requires_grad_(True), pv.to(D).requires_grad_(True)
  nod, npdd = m(ovd, pvd, edges.to(D))
  (nod * go.to(D)).sum().add((npdd * gp.to(D)).sum()).backward()
  report('gconv obj out', nod, no); report('gconv pred out', npdd, npd)
  report('gconv d obj', ovd.grad, ovr.grad); report('gconv d pred', pvd.grad, pvr.grad)
  for k, p in m.named_parameters():
    report('gconv d ' + k, p.grad, Pl['g.' + k].grad)


def _masked_stack_reference(Wcpu, dims, ov, pv, s, o, pooling, acts):
  """float64 CPU restatement of the layer stack (sg2im/graph.py:56-120 per layer) in which every ReLU is a
  multiplication by the 0/1 mask of the HIP kernel's OWN activation: the gradient of exactly the piecewise-linear
  branch the kernel took.  (Against the plain oracle a pre-activation within rounding distance of the kink takes the
  other branch on one side and moves a whole row's worth of gradient: measured 1e-2 of a tensor's max with the 3.9 M
  activations of the training shape.)"""
  x, pr = ov.double().requires_grad_(True), pv.double().requires_grad_(True)
  Wd = [w.detach().cpu().double().requires_grad_(True) for w in Wcpu]
  O, T = x.size(0), pr.size(0)
  cnt = torch.bincount(torch.cat([s, o]), minlength=O).clamp(min=1).double().view(O, 1)
  xin, pin = x, pr
  for l, (din, H, dout) in enumerate(dims):
    W1a, b1a, W1b, b1b, W2a, b2a, W2b, b2b = Wd[8 * l:8 * l + 8]
    h1m, ntm, h2m, nom = [(acts[5 * l + i].detach().cpu() > 0).double() for i in (0, 1, 3, 4)]
    tin = torch.cat([xin[s], pin, xin[o]], 1)
    h1 = (tin @ W1a.t() + b1a) * h1m
    nt = (h1 @ W1b.t() + b1b) * ntm
    pooled = torch.zeros(O, H, dtype=torch.float64).index_add(0, s, nt[:, :H]).index_add(0, o, nt[:, H + dout:])
    if pooling == 'avg':
      pooled = pooled / cnt
    h2 = (pooled @ W2a.t() + b2a) * h2m
    xin = (h2 @ W2b.t() + b2b) * nom
    pin = nt[:, H:H + dout]
  return xin, pin, x, pr, Wd


def _gconv_stack_case(tag, O, T, dims, pooling, g, s=None, o=None, mode='full'):
  """dims: [(din, H, dout)] per layer.  The persistent stack kernels (HF.GraphTripleConvStackFn): outputs against the
  oracle's layer-by-layer composition; the saved `pooled` against sg2im_segment_sum over its OWN new_t (bit-exact: the
  loader-side pool follows the reference's accumulation order); every gradient against the float64 reference of the
  same piecewise-linear branch (_masked_stack_reference); the barrier status word."""
  P = {}
  for l, (din, H, dout) in enumerate(dims):
    orc._lin(P, 'g%d.net1.0' % l, H, 3 * din, g, True); orc._lin(P, 'g%d.net1.2' % l, 2 * H + dout, H, g, True)
    orc._lin(P, 'g%d.net2.0' % l, H, H, g, True); orc._lin(P, 'g%d.net2.2' % l, dout, H, g, True)
  din0, dl = dims[0][0], dims[-1][2]
  ov, pv = torch.randn(O, din0, generator=g), torch.randn(T, din0, generator=g)
  if s is None:
    s = torch.randint(0, O, (T,), generator=g)
    o = torch.randint(0, O, (T,), generator=g)
  edges = torch.stack([s, o], 1)
  with torch.no_grad():
    x, pr = ov, pv
    for l, (din, H, dout) in enumerate(dims):
      x, pr = orc.graph_triple_conv(P, 'g%d' % l, x, pr, edges, H, dout, pooling)
  go, gp = torch.randn(O, dl, generator=g), torch.randn(T, dl, generator=g)
  W, names = [], []
  for l in range(len(dims)):
    for k in ('net1.0', 'net1.2', 'net2.0', 'net2.2'):
      W += [P['g%d.%s.weight' % (l, k)].to(D).requires_grad_(True), P['g%d.%s.bias' % (l, k)].to(D).requires_grad_(True)]
      names += ['g%d.%s.weight' % (l, k), 'g%d.%s.bias' % (l, k)]
  sd, od = s.to(D), o.to(D)
  csr = ops.Csr(sd, od, O)
  ovd, pvd = ov.to(D).requires_grad_(True), pv.to(D).requires_grad_(True)
  assert ops.gconv_stack_supported(dims), dims
  tag = '%s [%s]' % (tag, mode)
  keep_bwd, ops.GCN_PERSISTENT_BACKWARD = ops.GCN_PERSISTENT_BACKWARD, mode      # (the one-launch backward is what is checked: 'full' / 'low' footprint)
  xd, prd = HF.GraphTripleConvStackFn.apply(ovd, pvd, sd, od, csr, pooling == 'avg', *W)
  ops.gconv_stack_check(D)
  report(tag + ' obj out', xd, x); report(tag + ' pred out', prd, pr)
  # the saved activations of every layer: pooled == segment_sum(new_t) bit for bit
  saved = xd.grad_fn.saved_tensors
  acts = saved[4 + 8 * len(dims):]
  for l, (din, H, dout) in enumerate(dims):
    new_t, pooled = acts[5 * l + 1], acts[5 * l + 2]
    want = torch.empty(O, H, device=D)
    if T > 0:
      ops.segment_sum(new_t[:, :H], new_t[:, H + dout:], csr, H, pooling == 'avg', want)
    else:
      want.zero_()
    report(tag + ' layer %d pooled == segment_sum(new_t)' % l, pooled, want, exact=True)
  xr, prr, ovr, pvr, Wr = _masked_stack_reference(W, dims, ov, pv, s, o, pooling, acts)
  report(tag + ' obj out (float64, same branches)', xd, xr); report(tag + ' pred out (float64, same branches)', prd, prr)
  (xr * go.double()).sum().add((prr * gp.double()).sum()).backward()
  (xd * go.to(D)).sum().add((prd * gp.to(D)).sum()).backward()
  report(tag + ' d obj', ovd.grad, ovr.grad)
  if T > 0:
    report(tag + ' d pred', pvd.grad, pvr.grad)
  for i, n in enumerate(names):
    if T > 0 or '.net2.' in n:
      report('%s d %s' % (tag, n), W[i].grad, Wr[i].grad)
  ops.gconv_stack_check(D)
  if T > 0:
    # the layer-by-layer launches (sg2im_gconv_layer_backward) on identical activations against the one-launch form
    W2 = [w.detach().clone().requires_grad_(True) for w in W]
    ov2, pv2 = ov.to(D).requires_grad_(True), pv.to(D).requires_grad_(True)
    ops.GCN_PERSISTENT_BACKWARD = False
    try:
      x2, pr2 = HF.GraphTripleConvStackFn.apply(ov2, pv2, sd, od, csr, pooling == 'avg', *W2)
      (x2 * go.to(D)).sum().add((pr2 * gp.to(D)).sum()).backward()
    finally:
      ops.GCN_PERSISTENT_BACKWARD = keep_bwd
    report(tag + ' one-launch vs per-layer backward: d obj', ovd.grad, ov2.grad)
    report(tag + ' one-launch vs per-layer backward: d pred', pvd.grad, pv2.grad)
    worst = max(((err(W[i].grad, W2[i].grad)[0], names[i]) for i in range(len(W))), key=lambda r: r[0])
    j = names.index(worst[1])
    report('%s one-launch vs per-layer backward: worst parameter gradient (%s)' % (tag, worst[1]), W[j].grad, W2[j].grad)
  ops.GCN_PERSISTENT_BACKWARD = keep_bwd


def _plain_stack_reference(Wcpu, dims, ov, pv, s, o, pooling):
  """float64 CPU restatement of the layer stack with PLAIN ReLUs (sg2im/graph.py:56-120, nothing taken from the
  kernel) -> outputs, leaves and the smallest |pre-activation| over every ReLU of the stack"""
  x, pr = ov.double().requires_grad_(True), pv.double().requires_grad_(True)
  Wd = [w.detach().cpu().double().requires_grad_(True) for w in Wcpu]
  O = x.size(0)
  cnt = torch.bincount(torch.cat([s, o]), minlength=O).clamp(min=1).double().view(O, 1)
  xin, pin, margin = x, pr, float('inf')
  for l, (din, H, dout) in enumerate(dims):
    W1a, b1a, W1b, b1b, W2a, b2a, W2b, b2b = Wd[8 * l:8 * l + 8]
    pre = torch.cat([xin[s], pin, xin[o]], 1) @ W1a.t() + b1a
    margin = min(margin, float(pre.detach().abs().min())); h1 = pre.relu()
    pre = h1 @ W1b.t() + b1b
    margin = min(margin, float(pre.detach().abs().min())); nt = pre.relu()
    pooled = torch.zeros(O, H, dtype=torch.float64).index_add(0, s, nt[:, :H]).index_add(0, o, nt[:, H + dout:])
    if pooling == 'avg':
      pooled = pooled / cnt
    pre = pooled @ W2a.t() + b2a
    margin = min(margin, float(pre.detach().abs().min())); h2 = pre.relu()
    pre = h2 @ W2b.t() + b2b
    margin = min(margin, float(pre.detach().abs().min())); xin = pre.relu()
    pin = nt[:, H:H + dout]
  return xin, pin, x, pr, Wd, margin


def _gconv_stack_plain_case(tag, O, T, dims, pooling):
  """The one-launch backward AND the layer-by-layer backward against the PLAIN float64 oracle (VERDICT r4 weak #1b:
  no self-reference to the kernel's own activation masks): a small shape whose every pre-activation is at least 1e-4
  away from the ReLU kink in float64 (checked; seeds are tried in order until one qualifies), so fp32 and float64
  take the same branch everywhere and the gradients must agree to fp32 rounding."""
  for seed in range(100, 140):
    g = torch.Generator().manual_seed(seed)
    P = {}
    for l, (din, H, dout) in enumerate(dims):
      orc._lin(P, 'g%d.net1.0' % l, H, 3 * din, g, True); orc._lin(P, 'g%d.net1.2' % l, 2 * H + dout, H, g, True)
      orc._lin(P, 'g%d.net2.0' % l, H, H, g, True); orc._lin(P, 'g%d.net2.2' % l, dout, H, g, True)
    ov, pv = torch.randn(O, dims[0][0], generator=g), torch.randn(T, dims[0][0], generator=g)
    s, o = torch.randint(0, O, (T,), generator=g), torch.randint(0, O, (T,), generator=g)
    go, gp = torch.randn(O, dims[-1][2], generator=g), torch.randn(T, dims[-1][2], generator=g)
    Wc, names = [], []
    for l in range(len(dims)):
      for k in ('net1.0', 'net1.2', 'net2.0', 'net2.2'):
        Wc += [P['g%d.%s.weight' % (l, k)], P['g%d.%s.bias' % (l, k)]]
        names += ['g%d.%s.weight' % (l, k), 'g%d.%s.bias' % (l, k)]
    xr, prr, ovr, pvr, Wr, margin = _plain_stack_reference(Wc, dims, ov, pv, s, o, pooling)
    if margin >= 1e-4:
      break
  else:
    raise AssertionError('no seed with every pre-activation >= 1e-4 away from the kink')
  (xr * go.double()).sum().add((prr * gp.double()).sum()).backward()
  sd, od = s.to(D), o.to(D)
  csr = ops.Csr(sd, od, O)
  keep_bwd = ops.GCN_PERSISTENT_BACKWARD
  try:
    for one_launch in ('full', 'low', False):
      ops.GCN_PERSISTENT_BACKWARD = one_launch
      W = [w.to(D).requires_grad_(True) for w in Wc]
      ovd, pvd = ov.to(D).requires_grad_(True), pv.to(D).requires_grad_(True)
      xd, prd = HF.GraphTripleConvStackFn.apply(ovd, pvd, sd, od, csr, pooling == 'avg', *W)
      (xd * go.to(D)).sum().add((prd * gp.to(D)).sum()).backward()
      ops.gconv_stack_check(D)
      t = '%s (seed %d, margin %.1e, %s backward) vs the plain float64 oracle: ' % (
        tag, seed, margin, ('one-launch ' + one_launch) if one_launch else 'per-layer')
      report(t + 'obj out', xd, xr); report(t + 'pred out', prd, prr)
      report(t + 'd obj', ovd.grad, ovr.grad); report(t + 'd pred', pvd.grad, pvr.grad)
      for i, n in enumerate(names):
        report(t + 'd ' + n, W[i].grad, Wr[i].grad)
  finally:
    ops.GCN_PERSISTENT_BACKWARD = keep_bwd


def sec_gconv_stack():
  _gconv_stack_plain_case('stack small plain', 8, 12, [(32, 32, 32)] * 2, 'avg')
  for mode in ('full', 'low', 'staged', 'staged_full'):   # the one-launch backward in both footprints and its staged form (sg2im_gconv_stack_grads.low_footprint)
    g = torch.Generator().manual_seed(7)
    batch = synthetic_batch(32, seed=3)
    objs, triples = batch[1], batch[4]
    # the training shape: 5 layers 128 -> 512 -> 128 on a COCO-style batch of 32 images
    _gconv_stack_case('stack coco b32', objs.numel(), triples.size(0), [(128, 512, 128)] * 5, 'avg', g,
                      triples[:, 0].contiguous(), triples[:, 2].contiguous(), mode=mode)
    # ragged sizes (rows not multiples of 32), a narrower first layer, 'sum' pooling, rows with many entries
    _gconv_stack_case('stack ragged sum', 45, 333, [(64, 128, 96), (96, 128, 96)], 'sum', g, mode=mode)
    # one object collects most entries (a long CSR row), several isolated objects
    s = torch.randint(0, 3, (150,), generator=g); o = torch.randint(0, 3, (150,), generator=g)
    _gconv_stack_case('stack long rows', 40, 150, [(32, 64, 32)] * 3, 'avg', g, s, o, mode=mode)
    # no triples at all: every object pools to zero (net2(0))
    _gconv_stack_case('stack no triples', 37, 0, [(32, 64, 32)] * 2, 'avg', g, mode=mode)
    # one layer, more than 256 tiles per stage (several rounds over the resident grid)
    _gconv_stack_case('stack large', 700, 2100, [(128, 512, 128)], 'avg', g, mode=mode)


def sec_layout():
  sec_layout_mode(False)


def sec_layout_align_corners():
  """the torch-0.4 sampling convention (F.grid_sample(align_corners=True)): layout fwd / bwd and crop
  fwd / bwd against the oracle in that mode"""
  sec_layout_mode(True)


def sec_layout_mode(AC):
  g = torch.Generator().manual_seed(7)
  for (N, S, Dd, M, tag) in ((4, 64, 128, 16, 'coco64' + ('-ac' if AC else '')), (2, 32, 20, 8, 'odd' + ('-ac' if AC else ''))):
    batch = synthetic_batch(N, image_size=(S, S), mask_size=M, seed=9)
    imgs, objs, boxes, masks, triples, o2i, _ = batch
    O = objs.numel()
    vecs = torch.randn(O, Dd, generator=g)
    soft = torch.rand(O, M, M, generator=g)
    from sg2im_amd.layout import layout_nhwc
    for name, mk in (('gtmask', masks), ('softmask', soft), ('boxes', None)):
      vr = vecs.clone().requires_grad_(True)
      br = boxes.clone().requires_grad_(True)
      mr = mk.clone().requires_grad_(True) if (mk is not None and mk.is_floating_point()) else mk
      want = (orc.masks_to_layout(vr, br, mr, o2i, S, align_corners=AC) if mk is not None
              else orc.boxes_to_layout(vr, br, o2i, S, align_corners=AC))
      gl = torch.randn(want.shape, generator=g)
      want.backward(gl)
      vd = vecs.to(D).requires_grad_(True)
      bd = boxes.to(D).requires_grad_(True)
      md = mk.to(D) if mk is not None else None
      if md is not None and md.is_floating_point():
        md.requires_grad_(True)
      got = layout_nhwc(vd, bd, md, o2i.to(D), S, n_images=N, align_corners=AC)
      got.backward(gl.permute(0, 2, 3, 1).contiguous().to(D))
      report('layout %s %s fwd' % (tag, name), got.permute(0, 3, 1, 2), want)
      report('layout %s %s dvecs' % (tag, name), vd.grad, vr.grad)
      report('layout %s %s dboxes' % (tag, name), bd.grad, br.grad)
      if md is not None and md.is_floating_point():
        report('layout %s %s dmasks' % (tag, name), md.grad, mr.grad)
    # every object in ONE image: more objects than one LDS pass of the per-image kernels holds (layout.hip GOB / LO)
    o2c = torch.full_like(o2i, N - 1)           # (the last image: the oracle takes the image count from obj_to_img)
    vr, br, mr = vecs.clone().requires_grad_(True), boxes.clone().requires_grad_(True), soft.clone().requires_grad_(True)
    want = orc.masks_to_layout(vr, br, mr, o2c, S, align_corners=AC)
    gl = torch.randn(want.shape, generator=g)
    want.backward(gl)
    vd, bd, md = vecs.to(D).requires_grad_(True), boxes.to(D).requires_grad_(True), soft.to(D).requires_grad_(True)
    got = layout_nhwc(vd, bd, md, o2c.to(D), S, n_images=N, align_corners=AC)
    got.backward(gl.permute(0, 2, 3, 1).contiguous().to(D))
    report('layout %s crowded (%d objects in one image) fwd' % (tag, O), got.permute(0, 3, 1, 2), want)
    for nm, a, b in (('dvecs', vd.grad, vr.grad), ('dboxes', bd.grad, br.grad), ('dmasks', md.grad, mr.grad)):
      report('layout %s crowded %s' % (tag, nm), a, b)
    # crops
    ir = imgs.clone().requires_grad_(True)
    want = orc.crop_bbox_batch(ir, boxes, o2i, 32, align_corners=AC)
    gc = torch.randn(want.shape, generator=g)
    want.backward(gc)
    from sg2im_amd.bilinear import crop_bbox_batch
    idv = imgs.to(D).requires_grad_(True)
    got = crop_bbox_batch(idv, boxes.to(D), o2i.to(D), 32, align_corners=AC)
    got.backward(gc.to(D))
    report('crop %s fwd' % tag, got, want)
    report('crop %s dimg' % tag, idv.grad, ir.grad)
    # stress: boxes partly outside the image, one-pixel and zero-width boxes, unsorted obj_to_img
    Ob = 23
    bx = torch.rand(Ob, 4, generator=g) * 1.4 - 0.2
    sb = torch.stack([torch.minimum(bx[:, 0], bx[:, 2]), torch.minimum(bx[:, 1], bx[:, 3]),
                      torch.maximum(bx[:, 0], bx[:, 2]), torch.maximum(bx[:, 1], bx[:, 3])], 1)
    sb[0] = torch.tensor([0.3, 0.3, 0.3 + 1.0 / S, 0.3 + 1.0 / S]); sb[1] = torch.tensor([0.5, 0.2, 0.5, 0.9])
    o2 = torch.randint(0, N, (Ob,), generator=g)
    ir = imgs.clone().requires_grad_(True)
    want = orc.crop_bbox_batch(ir, sb, o2, 8, align_corners=AC)
    gc = torch.randn(want.shape, generator=g)
    want.backward(gc)
    idv = imgs.to(D).requires_grad_(True)
    got = crop_bbox_batch(idv, sb.to(D), o2.to(D), 8, align_corners=AC)
    got.backward(gc.to(D))
    report('crop %s stress fwd' % tag, got, want)
    report('crop %s stress dimg' % tag, idv.grad, ir.grad)
    first = idv.grad.clone()
    idv.grad = None
    got = crop_bbox_batch(idv, sb.to(D), o2.to(D), 8, align_corners=AC)
    got.backward(gc.to(D))
    report('crop %s backward is reproducible' % tag, idv.grad, first, exact=True)


def sec_losses():
  g = torch.Generator().manual_seed(11)
  from sg2im_amd import losses as L
  a, b = torch.randn(3, 5, 7, generator=g), torch.randn(3, 5, 7, generator=g)
  for name, fn, ref in (('l1', L.l1_loss, F.l1_loss), ('mse', L.mse_loss, F.mse_loss)):
    ar = a.clone().requires_grad_(True)
    want = ref(ar, b) * 2.5
    want.backward()
    ad = a.to(D).requires_grad_(True)
    got = fn(ad, b.to(D), 2.5)
    got.backward()
    report('loss %s' % name, got, want); report('loss %s grad' % name, ad.grad, ar.grad)
  for t in (0.0, 1.0):
    ar = a.clone().requires_grad_(True)
    want = orc.bce_loss(ar.view(-1), torch.full((a.numel(),), t)) * 3
    want.backward()
    ad = a.to(D).requires_grad_(True)
    got = L.bce_loss(ad, t) * 3
    got.backward()
    report('loss bce t=%g' % t, got, want); report('loss bce grad t=%g' % t, ad.grad, ar.grad)
  s = torch.randn(203, 184, generator=g) * 3
  y = torch.randint(0, 184, (203,), generator=g)
  sr = s.clone().requires_grad_(True)
  want = F.cross_entropy(sr, y)
  want.backward()
  sd = s.to(D).requires_grad_(True)
  got = L.cross_entropy(sd, y.to(D))
  got.backward()
  report('loss ce', got, want); report('loss ce grad', sd.grad, sr.grad)
  # wgan / lsgan score terms (sg2im/losses.py:106-145) and the mask BCE on probabilities
  for kind in ('wgan', 'lsgan'):
    og, od = orc.get_gan_losses(kind)
    hg, hd = L.get_gan_losses(kind)
    fr, rr = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    want = og(fr) * 1.5 + od(rr, fr)
    want.backward()
    fd, rd = a.to(D).requires_grad_(True), b.to(D).requires_grad_(True)
    got = hg(fd) * 1.5 + hd(rd, fd)
    got.backward()
    report('loss %s g+d' % kind, got, want)
    report('loss %s grad fake' % kind, fd.grad, fr.grad); report('loss %s grad real' % kind, rd.grad, rr.grad)
  pm = torch.rand(9, 16, 16, generator=g).clamp(1e-4, 1 - 1e-4)
  pm[0, 0, 0], pm[0, 0, 1] = 0.0, 1.0                 # saturated probabilities hit the log clamp
  ym = (torch.rand(9, 16, 16, generator=g) > 0.5).long()
  pr_ = pm.clone().requires_grad_(True)
  want = F.binary_cross_entropy(pr_, ym.float()) * 0.7
  want.backward()
  pd_ = pm.to(D).requires_grad_(True)
  got = L.binary_cross_entropy(pd_, ym.to(D), 0.7)
  got.backward()
  report('loss bce on probabilities', got, want); report('loss bce on probabilities grad', pd_.grad, pr_.grad)
  # adam
  p, gr = torch.randn(1000, generator=g), torch.randn(1000, generator=g)
  pr = p.clone().requires_grad_(True)
  opt = torch.optim.Adam([pr], lr=1e-2)
  pd, m, v = p.to(D), torch.zeros(1000, device=D), torch.zeros(1000, device=D)
  for step in (1, 2, 3):
    pr.grad = gr * step
    opt.step()
    ops.adam_step(pd, (gr * step).to(D), m, v, 1e-2, 0.9, 0.999, 1e-8, step)
  report('adam 3 steps', pd, pr)


def golden_case(name):
  fix = load_golden(name)
  cfg = fix['config']
  gcfg = dict(cfg['g'], vocab=fix['vocab'])
  docfg = dict(cfg['d_obj'], vocab=fix['vocab'])
  dicfg = dict(cfg['d_img'])
  sd = fix['state_before']
  G = hh.build_generator(gcfg, sd['G']); Do = hh.build_d_obj(docfg, sd['Do']); Di = hh.build_d_img(dicfg, sd['Di'])
  for m in (G, Do, Di):
    m.train()
  imgs, objs, boxes, masks, triples, o2i = [hh.to_dev(t) for t in fix['batch'][:6]]
  from sg2im_amd import losses as L
  w = fix['weights']
  with hh.fixed_noise(fix['noise']):
    out = G(objs, triples, o2i, boxes_gt=boxes, masks_gt=masks, num_images=imgs.size(0))
  ip, bp, mp, rs = out
  W = fix['outputs']
  report(name + ' imgs_pred', ip, W['imgs_pred']); report(name + ' boxes_pred', bp, W['boxes_pred'])
  report(name + ' masks_pred', mp, W['masks_pred']); report(name + ' rel_scores', rs, W['rel_scores'])
  l1 = L.l1_loss(ip, imgs, w['l1']); lb = L.mse_loss(bp, boxes, w['bbox'])
  sf, ac = Do(ip, objs, boxes, o2i)
  report(name + ' d_obj scores', sf, W['d_obj_scores_fake'])
  sfi = Di(ip)
  report(name + ' d_img scores', sfi, W['d_img_scores_fake'])
  total = l1 + lb + ac * w['ac'] + L.gan_g_loss(sf) * (w['d'] * w['d_obj']) + L.gan_g_loss(sfi) * (w['d'] * w['d_img'])
  report(name + ' total loss', total, torch.tensor(fix['losses']['total']))
  total.backward()
  for k, gref in fix['grads']['G'].items():
    got = dict(G.named_parameters())[k].grad
    if gref is None:
      e0 = 0.0 if got is None else float(got.abs().max()); RESULTS.append((name + ' G.grad ' + k, e0, 'expect None', e0, 0.0))
      continue
    if got is None:
      RESULTS.append((name + ' G.grad ' + k, float('inf'), 'MISSING', float('inf'), 0.0)); print('MISSING grad', k); continue
    report(name + ' G.grad ' + k, got, gref)
  for k, v in fix['state_after_g_forward']['G'].items():
    report(name + ' G.buf ' + k, G.state_dict()[k].float(), v.float())
  for tag, net in (('Do', Do), ('Di', Di)):            # after their single pass inside the generator loss
    for k, v in fix['state_after_g_forward'].get(tag, {}).items():
      report(name + ' %s.buf %s' % (tag, k), net.state_dict()[k].float(), v.float())
  fake = ip.detach()
  Do.zero_grad(); Di.zero_grad()
  sf, acf = Do(fake, objs, boxes, o2i); sr, acr = Do(imgs, objs, boxes, o2i)
  ld = L.gan_d_loss(sr, sf) + acr + acf
  report(name + ' d_obj loss', ld, torch.tensor(fix['losses']['d_obj']))
  ld.backward()
  for k, gref in fix['grads']['Do'].items():
    report(name + ' Do.grad ' + k, dict(Do.named_parameters())[k].grad, gref)
  li = L.gan_d_loss(Di(imgs), Di(fake))
  report(name + ' d_img loss', li, torch.tensor(fix['losses']['d_img']))
  li.backward()
  for k, gref in fix['grads']['Di'].items():
    if gref is not None:
      report(name + ' Di.grad ' + k, dict(Di.named_parameters())[k].grad, gref)


def golden_eval_case(name):
  """generator in eval() mode (running BN statistics, scripts/train.py:509-512) against the
  <name>_eval.pt vectors of the imported reference; the discriminators stay in train mode"""
  fix = load_golden(name + '_eval')
  cfg = fix['config']
  gcfg = dict(cfg['g'], vocab=fix['vocab'])
  sd = fix['state_before']
  G = hh.build_generator(gcfg, sd['G'])
  Do = hh.build_d_obj(dict(cfg['d_obj'], vocab=fix['vocab']), sd['Do'])
  Di = hh.build_d_img(dict(cfg['d_img']), sd['Di'])
  G.eval(); Do.train(); Di.train()
  imgs, objs, boxes, masks, triples, o2i = [hh.to_dev(t) for t in fix['batch'][:6]]
  from sg2im_amd import losses as L
  w = fix['weights']
  name = name + ' eval'
  with hh.fixed_noise(fix['noise']):
    ip, bp, mp, rs = G(objs, triples, o2i, boxes_gt=boxes, masks_gt=masks, num_images=imgs.size(0))
  W = fix['outputs']
  report(name + ' imgs_pred', ip, W['imgs_pred']); report(name + ' boxes_pred', bp, W['boxes_pred'])
  report(name + ' masks_pred', mp, W['masks_pred']); report(name + ' rel_scores', rs, W['rel_scores'])
  sf, ac = Do(ip, objs, boxes, o2i)
  sfi = Di(ip)
  report(name + ' d_obj scores', sf, W['d_obj_scores_fake']); report(name + ' d_img scores', sfi, W['d_img_scores_fake'])
  total = (L.l1_loss(ip, imgs, w['l1']) + L.mse_loss(bp, boxes, w['bbox']) + ac * w['ac']
           + L.gan_g_loss(sf) * (w['d'] * w['d_obj']) + L.gan_g_loss(sfi) * (w['d'] * w['d_img']))
  report(name + ' total loss', total, torch.tensor(fix['losses']['total']))
  total.backward()
  for k, gref in fix['grads']['G'].items():
    got = dict(G.named_parameters())[k].grad
    if gref is None:
      e0 = 0.0 if got is None else float(got.abs().max()); RESULTS.append((name + ' G.grad ' + k, e0, 'expect None', e0, 0.0))
      continue
    if got is None:
      RESULTS.append((name + ' G.grad ' + k, float('inf'), 'MISSING', float('inf'), 0.0)); print('MISSING grad', k); continue
    report(name + ' G.grad ' + k, got, gref)
  for k, v in fix['state_after_g_forward']['G'].items():       # untouched by an eval-mode forward
    same = torch.equal(G.state_dict()[k].cpu(), v)
    RESULTS.append((name + ' G.buf unchanged ' + k, 0.0 if same else float('inf'), 'bit-exact' if same else 'CHANGED', 0.0 if same else float('inf'), 0.0))


def sec_golden_eval():
  golden_eval_case('tiny_coco')
  golden_eval_case('tiny_vg')


def sec_golden_nonorm():
  golden_case('tiny_coco_nonorm')       # refinement network with --normalization none


def sec_golden_instnorm():
  golden_case('tiny_coco_instnorm')     # InstanceNorm2d in the refinement network and both discriminators


def sec_golden_archtokens():
  golden_case('tiny_coco_archtokens')   # discriminators from R / P / U / FC build_cnn tokens


def sec_golden_mlpbn():
  golden_case('tiny_coco_mlpbn')        # BatchNorm1d in every MLP (--mlp_normalization batch)


def sec_golden_coco():
  golden_case('tiny_coco')


def sec_golden_vg():
  golden_case('tiny_vg')


if __name__ == '__main__':
  print(torch.cuda.get_device_name(0))
  only = sys.argv[1:]
  for fn in (sec_pool, sec_linear, sec_conv, sec_conv_bn, sec_conv_bf16, sec_gconv, sec_layout, sec_layout_align_corners, sec_losses, sec_golden_coco, sec_golden_vg, sec_golden_eval, sec_golden_nonorm, sec_golden_mlpbn, sec_golden_instnorm, sec_golden_archtokens):
    if not only or fn.__name__ in only:
      section(fn)
  bad = [r for r in RESULTS if not (r[1] <= 1e-4 or (r[3] <= 1e-6 and r[4] < 1e-6))]
  print('\n==== %d checks, %d above rel 1e-4 (and abs 1e-6) ====' % (len(RESULTS), len(bad)))
  for r in bad:
    print('BAD %-58s %.3e %s' % r[:3])
