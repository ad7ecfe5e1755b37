"""Where does the fp32 matrix pipe go idle in the implicit-GEMM main loop?  (tools/_src/gemm_skeleton.hip)
Runs the loop skeleton at increasing levels of completeness and at 1-4 co-resident workgroups per CU."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

lib = ctypes.CDLL(os.path.join(ROOT, 'tools', '_bin', 'libgemmskeleton.so'))
lib.skeleton_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]
D = torch.device('cuda', 0)
STRIDE = 48
g = torch.randn(STRIDE * 1024 * 256 * 4, device=D)
sink = torch.zeros(4, device=D)
NAMES = ['MFMA only', '+ frag ds_read', '+ 2 barriers', '+ 6 ds_write', '+ 6 global loads', '+ VALU/SALU filler']


def run(level, ksteps, wg_per_cu, iters=1500):
  blocks = 256 * wg_per_cu
  # extra LDS so that exactly wg_per_cu workgroups fit a CU (160 KB)
  base = (128 + 64) * 36 * (ksteps // 16) * 4
  extra = max(0, (160 * 1024) // wg_per_cu - base - 1024) if wg_per_cu < 4 else 0
  it = iters * 16 // ksteps
  st = torch.cuda.current_stream().cuda_stream
  for _ in range(2):
    rc = lib.skeleton_launch(level, ksteps, g.data_ptr(), sink.data_ptr(), blocks, it, STRIDE, extra, st)
    assert rc == 0, rc
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(3):
    lib.skeleton_launch(level, ksteps, g.data_ptr(), sink.data_ptr(), blocks, it, STRIDE, extra, st)
  e1.record()
  torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / 3
  flops = blocks * 4.0 * it * (2 * ksteps) * 4096.0
  return flops / (ms * 1e-3) / 1e12


def main():
  print('TFLOP/s of the loop skeleton (fp32 MFMA peak 157.3); columns = co-resident workgroups per CU')
  for ks in (16, 32):
    print('-- %d MFMAs per barrier interval (BK = %d)' % (2 * ks, 2 * ks))
    print('%-22s %8s %8s %8s %8s' % ('level', '1/CU', '2/CU', '3/CU', '4/CU'))
    for level in range(6):
      vals = []
      for n in (1, 2, 3, 4):
        if ks == 32 and n == 4:
          vals.append(float('nan'))       # 55 KB per workgroup: at most 2-3 fit
          continue
        vals.append(run(level, ks, n))
      print('%-22s %8.1f %8.1f %8.1f %8.1f' % (NAMES[level], vals[0], vals[1], vals[2], vals[3]), flush=True)


if __name__ == '__main__':
  main()
