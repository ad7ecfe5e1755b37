"""Prototype check: the direct-to-LDS ("v2") forward conv (tools/_src/conv_v2.hip) against
sg2im_conv2d_forward - results and TFLOP/s on refinement-network shapes (plain single-source inputs)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from sg2im_amd import ops

lib = ctypes.CDLL(os.path.join(ROOT, 'tools', '_bin', 'libconvv2.so'))
P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
lib.conv_fwd_v2.argtypes = [P, I, I, P, I, P, P, I, I, I, I, I, F, I, I, P]
D = torch.device('cuda', 0)


def timeit(fn, iters=20):
  fn(); fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(iters):
    fn()
  b.record()
  torch.cuda.synchronize()
  return a.elapsed_time(b) / iters


def main():
  print('%-28s %8s | %9s %7s | %7s %7s %7s %7s | %s' % ('shape', 'GFLOP', 'old ms', 'TF/s', '128x64', '128x128', '256x64', '256x128', 'max rel err'))
  for name, NB, H, C, Cout in (('m4.conv1 64->64 @64', 32, 64, 64, 64), ('m4.conv0-like 288->64 @64', 32, 64, 288, 64),
                               ('m3.conv1 128->128 @32', 32, 32, 128, 128), ('m3.conv0-like 416->128 @32', 32, 32, 416, 128),
                               ('m2.conv1 256->256 @16', 32, 16, 256, 256), ('m2.conv0-like 672->256 @16', 32, 16, 672, 256),
                               ('ragged 96->80 @19x19 b3', 3, 19, 96, 80)):
    x = torch.randn(NB, H, H, C, device=D)
    W = torch.randn(Cout, 3, 3, C, device=D) * 0.05
    b = torch.randn(Cout, device=D)
    y0 = torch.empty(NB, H, H, Cout, device=D)
    d = ops.conv_desc([ops.nhwc_src(x)], NB, H, H, 3, 3, 1, 1)
    old = lambda: ops.conv2d_forward(d, W, Cout, b, y0, Cout, 0.2)
    outs = []
    res = []
    dbgres = []
    for dbg in (1, 2):
      y = torch.zeros(NB, H, H, Cout, device=D)
      st = torch.cuda.current_stream().cuda_stream
      best = 0.0
      for tile in (0, 2):
        fn = lambda: lib.conv_fwd_v2(x.data_ptr(), C, C, W.data_ptr(), Cout, b.data_ptr(), y.data_ptr(), NB, H, H, 3, 1, 0.2, tile, dbg, st)
        fn()
        best = max(best, 2.0 * NB * H * H * Cout * C * 9 / 1e9 / timeit(fn))
      dbgres.append(best)
    for tile in (0, 1, 2, 3):
      y = torch.zeros(NB, H, H, Cout, device=D)
      st = torch.cuda.current_stream().cuda_stream
      fn = lambda: lib.conv_fwd_v2(x.data_ptr(), C, C, W.data_ptr(), Cout, b.data_ptr(), y.data_ptr(), NB, H, H, 3, 1, 0.2, tile, 0, st)
      rc = fn()
      assert rc == 0, rc
      res.append(timeit(fn))
      outs.append(y)
    t0 = timeit(old)
    gf = 2.0 * NB * H * H * Cout * C * 9 / 1e9
    err = max(float((o - y0).abs().max() / y0.abs().max()) for o in outs)
    print('%-28s %8.2f | %9.3f %7.1f | %7.1f %7.1f %7.1f %7.1f | %.2e | no-DMA %6.1f  same-line DMA %6.1f' % (
      name, gf, t0, gf / t0, gf / res[0], gf / res[1], gf / res[2], gf / res[3], err, dbgres[0], dbgres[1]), flush=True)


if __name__ == '__main__':
  main()
