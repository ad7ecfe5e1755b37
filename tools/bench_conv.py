"""Per-layer timing of the implicit-GEMM kernels at the COCO-64 / batch-32 shapes
(SURVEY.md section 8a table T1).  Prints ms and achieved TFLOP/s per layer and pass."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from sg2im_amd import ops

D = torch.device('cuda', 0)
NB = 32
PLAIN = '--plain' in sys.argv
WGRAD_ONLY = '--wgrad' in sys.argv          # time the weight gradients only
BF16 = '--bf16' in sys.argv                  # bfloat16 operands on the 32x32x16 MFMA (Trainer(compute_dtype='bf16'))
MIRROR = '--mirror' in sys.argv              # (with --bf16) weights read from a bfloat16 mirror (sg2im_conv_desc.weight_bf16)
STORAGE = '--storage' in sys.argv            # (with --bf16) feature sources, outputs and gradients in bfloat16 STORAGE where the
                                             # halo'd kernels run the layer (3x3 convolutions on maps >= 16 x 16); implies --mirror
ONLY = [a[7:].split(',') for a in sys.argv if a.startswith('--only=')]      # --only=m3,m4,out: layer name prefixes
LAYERS = [  # name, H, C0, C1(up), Cout, k, stride, pad
  ('m0.conv0', 4, 160, 1, 1024, 3, 1, 1), ('m0.conv1', 4, 1024, 0, 1024, 3, 1, 1),
  ('m1.conv0', 8, 160, 1024, 512, 3, 1, 1), ('m1.conv1', 8, 512, 0, 512, 3, 1, 1),
  ('m2.conv0', 16, 160, 512, 256, 3, 1, 1), ('m2.conv1', 16, 256, 0, 256, 3, 1, 1),
  ('m3.conv0', 32, 160, 256, 128, 3, 1, 1), ('m3.conv1', 32, 128, 0, 128, 3, 1, 1),
  ('m4.conv0', 64, 160, 128, 64, 3, 1, 1), ('m4.conv1', 64, 64, 0, 64, 3, 1, 1),
  ('out.conv0', 64, 64, 0, 64, 3, 1, 1), ('out.conv1', 64, 64, 0, 3, 1, 1, 0),
  ('d_img.c0', 64, 3, 0, 64, 4, 2, 0), ('d_img.c1', 31, 64, 0, 128, 4, 2, 0), ('d_img.c2', 14, 128, 0, 256, 4, 2, 0),
  # per-object layers (optional 9th field = batch): ~216 objects and ~640 triples per 32 images
  ('d_obj.c0', 32, 3, 0, 64, 4, 2, 0, 216), ('d_obj.c1', 15, 64, 0, 128, 4, 2, 0, 216),
  ('d_obj.c2', 6, 128, 0, 256, 4, 2, 0, 216),
  ('mask.c1', 4, 0, 128, 128, 3, 1, 1, 216), ('mask.c2', 8, 0, 128, 128, 3, 1, 1, 216),
  ('mask.c3', 16, 0, 128, 128, 3, 1, 1, 216),
  ('gconv.n1a', 1, 384, 0, 512, 1, 1, 0, 640), ('gconv.n1b', 1, 512, 0, 1152, 1, 1, 0, 640),
  ('gconv.n2a', 1, 512, 0, 512, 1, 1, 0, 216), ('gconv.n2b', 1, 512, 0, 128, 1, 1, 0, 216),
]


# --size=128 --dims=1024,512,256,128,64,64: the refinement network of another configuration (BASELINE configs[3..4]) instead
# of the COCO-64 table (the discriminator / mask / graph layers stay out)
_size = [a[7:] for a in sys.argv if a.startswith('--size=')]
_dims = [a[7:] for a in sys.argv if a.startswith('--dims=')]
if _size:
  S = int(_size[0])
  dims = [int(v) for v in (_dims[0] if _dims else '1024,512,256,128,64').split(',')]
  LAYERS, prev = [], 0
  for i, c in enumerate(dims):
    h = S >> (len(dims) - 1 - i)
    LAYERS.append(('m%d.conv0' % i, h, 160, prev if prev else 1, c, 3, 1, 1))
    LAYERS.append(('m%d.conv1' % i, h, c, 0, c, 3, 1, 1))
    prev = c
  LAYERS += [('out.conv0', S, dims[-1], 0, dims[-1], 3, 1, 1), ('out.conv1', S, dims[-1], 0, 3, 1, 1, 0)]


def timeit(fn, iters=10):
  fn(); fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(iters):
    fn()
  b.record()
  torch.cuda.synchronize()
  return a.elapsed_time(b) / iters


def _st(t, dt):
  """t in storage type dt (bfloat16: with the 16 bytes of slack sg2im_src.dtype asks for)"""
  if dt != torch.bfloat16:
    return t
  buf = torch.empty(t.numel() + 8, dtype=dt, device=t.device)
  v = buf[:t.numel()].view(t.shape)
  v.copy_(t)
  return v


def main():
  if BF16:
    ops.CONV_COMPUTE = 1
  mirrors = {}
  if BF16 and (MIRROR or STORAGE):
    ops.WEIGHT_MIRROR = True
    ops.WEIGHT_MIRROR_LOOKUP = lambda w: mirrors[w.data_ptr()].data_ptr() if w.data_ptr() in mirrors else None
  tot = {'fwd': 0.0, 'dgrad': 0.0, 'wgrad': 0.0}
  totf = 0.0
  print('%-10s %9s | %8s %7s | %8s %7s | %8s %7s' % ('layer', 'GFLOP', 'fwd ms', 'TF/s', 'dgrad ms', 'TF/s', 'wgrad ms', 'TF/s'))
  for L in LAYERS:
    name, H, C0, C1, Cout, k, s, p = L[:8]
    if ONLY and not any(name.startswith(pre) for pre in ONLY[0]):
      continue
    N = L[8] if len(L) > 8 else NB
    srcs = []
    # as in the network: the previous layer's BatchNorm + LeakyReLU is PENDING on the feature source (applied by the
    # operand loader); the layout levels are plain.  --plain: no pending affine anywhere
    aff = lambda C: (None, None, 1.0) if PLAIN else (torch.rand(C, device=D) + 0.5, torch.randn(C, device=D) * 0.1, 0.2)
    # --storage: the tensors a bf16-storage step keeps in bfloat16 (3x3 layers of the refinement network on maps the
    # halo'd kernels tile; the layout levels - a first source next to an upsampled one - stay float32)
    st_layer = BF16 and STORAGE and k == 3 and s == 1 and H >= 16 and name[0] in 'mo'
    sdt = torch.bfloat16 if st_layer else torch.float32
    if C0:
      sc, sh, sl = aff(C0) if (C1 == 0 and k == 3 and s == 1) else (None, None, 1.0)
      srcs.append(ops.nhwc_src(_st(torch.randn(N, H, H, C0, device=D), sdt if C1 == 0 else torch.float32), 0, sc, sh, sl))
    if C1 > 1:
      sc, sh, sl = aff(C1)
      srcs.append(ops.nhwc_src(_st(torch.randn(N, H // 2, H // 2, C1, device=D), sdt), 1, sc, sh, sl))
    # (C1 == 1: the first refinement module - the all-zero feature channel is left out, the weight rows keep it)
    d = ops.conv_desc(srcs, N, H, H, k, k, s, p, weight_channels=C0 + C1 if C1 == 1 else 0)
    Ct = C0 + C1
    W = torch.randn(Cout, k, k, Ct, device=D) * 0.01
    if BF16 and (MIRROR or STORAGE) and k == 3:
      mirrors[W.data_ptr()] = torch.cat([W.reshape(-1).to(torch.bfloat16), torch.zeros(16, dtype=torch.bfloat16, device=D)])
    b = torch.randn(Cout, device=D)
    y = torch.empty(N, d.out_h, d.out_w, Cout, device=D, dtype=sdt)
    gy = torch.randn(N, d.out_h, d.out_w, Cout, device=D).to(sdt)
    Cx = C0 if C1 == 1 else Ct
    dx = torch.empty(N, H, H, Cx, device=D, dtype=sdt)
    dw = torch.empty_like(W)
    gf = 2.0 * N * d.out_h * d.out_w * Cout * Ct * k * k / 1e9
    t1 = 1e-9 if WGRAD_ONLY else timeit(lambda: ops.conv2d_forward(d, W, Cout, b, y, Cout))
    if WGRAD_ONLY:
      t2 = 1e-9
    elif C1 > 1:       # as the network issues it: the layout channels (128 of 160 need gradients) and the feature channels
      dl, dz = torch.empty(N, H, H, 128, device=D), torch.empty(N, H, H, C1, device=D, dtype=sdt)
      t2 = timeit(lambda: (ops.conv2d_backward_data(d, W, Cout, gy, Cout, 0, 128, dl, 128),
                           ops.conv2d_backward_data(d, W, Cout, gy, Cout, C0, C1, dz, C1)))
      t2 *= Ct / float(128 + C1)       # (per FLOP of the full layer, so that the TF/s column stays comparable)
    else:
      t2 = timeit(lambda: ops.conv2d_backward_data(d, W, Cout, gy, Cout, 0, Cx, dx, Cx))
    t3 = timeit(lambda: ops.conv2d_backward_weight(d, gy, Cout, Cout, dw))
    print('%-10s %9.2f | %8.3f %7.1f | %8.3f %7.1f | %8.3f %7.1f' % (name, gf, t1, gf / t1, t2, gf / t2, t3, gf / t3), flush=True)
    tot['fwd'] += t1; tot['dgrad'] += t2; tot['wgrad'] += t3; totf += gf
  print('TOTAL %.1f GFLOP/pass: fwd %.2f ms (%.1f TF/s)  dgrad %.2f ms (%.1f)  wgrad %.2f ms (%.1f)' %
        (totf, tot['fwd'], totf / tot['fwd'], tot['dgrad'], totf / tot['dgrad'], tot['wgrad'], totf / tot['wgrad']))


if __name__ == '__main__':
  main()
