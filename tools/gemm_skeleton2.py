"""Main-loop STRUCTURE sandbox (tools/_src/gemm_skeleton2.hip): the full instruction mix of one K chunk
of conv_fwd_kernel<128,64> arranged in different ways; TFLOP/s at 1-4 co-resident workgroups per CU."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

lib = ctypes.CDLL(os.path.join(ROOT, 'tools', '_bin', 'libgemmskeleton2.so'))
lib.skel2_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_longlong,
                             ctypes.c_void_p]
D = torch.device('cuda', 0)
g = torch.randn(64 * 8192 + 16384, device=D)
sink = torch.zeros(4, device=D)
NAMES = {0: 'sequential phases (current)', 1: 'staggered start', 2: 's_setprio around MFMAs', 3: 'loader math sliced into MFMA block',
         4: 'ping-pong halves (512 thr)', 5: 'sliced math + stores, 1 barrier, 2 LDS images'}
IMG = (128 + 64) * 36 * 4


def run(mode, wg_per_cu, iters=1500):
  per_wg = 2 if mode == 4 else 1                 # a 512-thread workgroup carries two tiles
  n = max(1, wg_per_cu // per_wg)
  blocks = 256 * n
  base = IMG * (2 if mode in (4, 5) else 1)
  extra = max(0, (160 * 1024) // n - base - 1024) if n < 4 else 0
  extra = min(extra, 64 * 1024 - base) if mode not in (4, 5) else min(extra, 150 * 1024 - base)
  st = torch.cuda.current_stream().cuda_stream
  for _ in range(2):
    rc = lib.skel2_launch(mode, g.data_ptr(), sink.data_ptr(), blocks, iters, extra, st)
    assert rc == 0, rc
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(3):
    lib.skel2_launch(mode, g.data_ptr(), sink.data_ptr(), blocks, iters, extra, st)
  e1.record()
  torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / 3
  flops = blocks * per_wg * 4.0 * iters * 32 * 4096.0
  return flops / (ms * 1e-3) / 1e12


def main():
  print('TFLOP/s (fp32 MFMA peak 157.3); columns = co-resident 128x64 TILES per CU (4 waves each)')
  print('%-48s %8s %8s %8s %8s' % ('structure', '1/CU', '2/CU', '3/CU', '4/CU'))
  for mode in range(6):
    vals = []
    for n in (1, 2, 3, 4):
      if mode == 4 and n in (1, 3):
        vals.append(float('nan'))
        continue
      vals.append(run(mode, n))
    print('%-48s %8.1f %8.1f %8.1f %8.1f' % (NAMES[mode], vals[0], vals[1], vals[2], vals[3]), flush=True)


if __name__ == '__main__':
  main()
