"""Loader-mix sweep of the GEMM loop sandbox (tools/_src/gemm_skeleton3.hip)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

lib = ctypes.CDLL(os.path.join(ROOT, 'tools', '_bin', 'libgemmskeleton3.so'))
lib.skel3_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                             ctypes.c_longlong, ctypes.c_void_p]
D = torch.device('cuda', 0)
g = torch.randn(64 * 8192 + 16384, device=D)
sink = torch.zeros(4, device=D)
CFGS = [(0, 'nothing (MFMA + frag reads + 2 barriers)'), (5, '80 VALU / chunk'), (1, '160 VALU / chunk'), (6, '320 VALU / chunk'),
        (2, '112 SALU / chunk'), (7, '224 SALU / chunk'), (9, '6 global loads'), (10, '6 ds_write'), (3, '6 loads + 6 ds_write'),
        (8, '80 VALU + 48 SALU + loads + stores'), (4, '160 VALU + 112 SALU + loads + stores')]
IMG = (128 + 64) * 36 * 4


def run(cfg, sliced, n, iters=1500):
  blocks = 256 * n
  extra = max(0, min((160 * 1024) // n - IMG - 1024, 64 * 1024 - IMG)) if n < 4 else 0
  st = torch.cuda.current_stream().cuda_stream
  for _ in range(2):
    rc = lib.skel3_launch(cfg, sliced, g.data_ptr(), sink.data_ptr(), blocks, iters, extra, st)
    assert rc == 0, rc
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(3):
    lib.skel3_launch(cfg, sliced, g.data_ptr(), sink.data_ptr(), blocks, iters, extra, st)
  e1.record()
  torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / 3
  return blocks * 4.0 * iters * 32 * 4096.0 / (ms * 1e-3) / 1e12


def main():
  print('TFLOP/s (peak 157.3) - loader mix per K chunk of 32 MFMAs per wave; 1 / 2 / 4 workgroups per CU; '
        '"sliced" = math placed between the MFMAs')
  print('%-44s | %7s %7s %7s | %7s %7s %7s' % ('loader mix', 'seq 1', 'seq 2', 'seq 4', 'slc 1', 'slc 2', 'slc 4'))
  for cfg, name in CFGS:
    v = [run(cfg, s, n) for s in (0, 1) for n in (1, 2, 4)]
    print('%-44s | %7.1f %7.1f %7.1f | %7.1f %7.1f %7.1f' % ((name,) + tuple(v)), flush=True)


if __name__ == '__main__':
  main()
