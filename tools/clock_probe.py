"""Engine clock the MI355X sustains under the implicit-GEMM kernels (tools/_src/clockprobe.hip):
a one-wave probe on a second stream measures shader cycles per 100 MHz reference tick while the load
runs.  Loads: idle, a pure fp32-MFMA burn (no memory traffic), and the forward / data-gradient /
weight-gradient launches of a few refinement-network layers.  Also reports each load's TFLOP/s, so
`TFLOP/s / (157.3 * clock / 2.4 GHz)` = the fraction of the peak AT THE SUSTAINED CLOCK."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from sg2im_amd import ops

lib = ctypes.CDLL(os.path.join(ROOT, 'tools', '_bin', 'libclockprobe.so'))
lib.clockprobe_launch.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p]
lib.mfma_burn_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
D = torch.device('cuda', 0)
side = torch.cuda.Stream()
out = torch.zeros(2, dtype=torch.int64, device=D)
sink = torch.zeros(4, device=D)


def measure(name, launch, flops_per_launch, reps, probe_us=1500):
  """enqueue `reps` launches on the current stream, the probe on the side stream in their middle"""
  for _ in range(3):
    launch()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for i in range(reps):
    launch()
    if i == reps // 4:
      ev = torch.cuda.Event(); ev.record()
      side.wait_event(ev)
      lib.clockprobe_launch(out.data_ptr(), int(probe_us * 100), side.cuda_stream)
  e1.record()
  torch.cuda.synchronize()
  ms = e0.elapsed_time(e1)
  dc, dw = [int(v) for v in out.tolist()]
  ghz = dc / dw * 0.1
  tf = flops_per_launch * reps / (ms * 1e-3) / 1e12 if flops_per_launch else 0.0
  peak_at_clock = 157.3 * ghz / 2.4
  print('%-28s load %8.2f ms  probe %6.0f us  clock %.3f GHz  %7.1f TFLOP/s  = %.3f of 157.3, %.3f of the peak at this clock (%.1f)'
        % (name, ms, dw / 100.0, ghz, tf, tf / 157.3, tf / peak_at_clock if tf else 0.0, peak_at_clock), flush=True)


def conv_loads(name, NB, H, C0, C1, Cout):
  srcs = []
  if C0:
    srcs.append(ops.nhwc_src(torch.randn(NB, H, H, C0, device=D)))
  if C1:
    srcs.append(ops.nhwc_src(torch.randn(NB, H // 2, H // 2, C1, device=D), 1))
  d = ops.conv_desc(srcs, NB, H, H, 3, 3, 1, 1)
  Ct = C0 + C1
  W = torch.randn(Cout, 3, 3, Ct, device=D) * 0.01
  b = torch.randn(Cout, device=D)
  y = torch.empty(NB, H, H, Cout, device=D)
  gy = torch.randn_like(y)
  dx = torch.empty(NB, H, H, Ct, device=D)
  dw = torch.empty_like(W)
  fl = 2.0 * NB * H * H * Cout * Ct * 9
  return [(name + ' fwd', lambda: ops.conv2d_forward(d, W, Cout, b, y, Cout), fl),
          (name + ' dgrad', lambda: ops.conv2d_backward_data(d, W, Cout, gy, Cout, 0, Ct, dx, Ct), fl),
          (name + ' wgrad', lambda: ops.conv2d_backward_weight(d, gy, Cout, Cout, dw), fl)]


def main():
  ops.workspace(D)
  measure('idle', lambda: None, 0, 1)
  # 4 x 256 blocks of 4 waves: 4 waves per SIMD, each 2000 x 32 MFMAs
  burn = lambda: lib.mfma_burn_launch(1024, 2000, sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
  measure('mfma burn (4 waves/SIMD)', burn, 1024 * 4 * 2000 * 32 * 4096.0, 8, probe_us=3000)
  burn1 = lambda: lib.mfma_burn_launch(256, 8000, sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
  measure('mfma burn (1 wave/SIMD)', burn1, 256 * 4 * 8000 * 32 * 4096.0, 8, probe_us=3000)
  for name, NB, H, C0, C1, Cout in (('m4.conv0', 32, 64, 160, 128, 64), ('m3.conv0', 32, 32, 160, 256, 128),
                                    ('m2.conv0', 32, 16, 160, 512, 256), ('m4.conv1', 32, 64, 64, 0, 64)):
    for nm, fn, fl in conv_loads(name, NB, H, C0, C1, Cout):
      measure(nm, fn, fl, 24)


if __name__ == '__main__':
  main()
