"""Build libsg2im_hip.so (gfx950) in-tree with hipcc.  No torch involved: the library is a
plain C-ABI shared object (include/sg2im_hip.h) loaded through ctypes."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libsg2im_hip.so')
SOURCES = ['conv.hip', 'graph.hip', 'norm.hip', 'layout.hip', 'loss.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC]


def _stale(target, deps):
  if not os.path.exists(target):
    return True
  t = os.path.getmtime(target)
  return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
  hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
  os.makedirs(LIBDIR, exist_ok=True)
  headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
  headers.append(os.path.join(ROOT, 'include', 'sg2im_hip.h'))
  objs, procs = [], []
  for src in SOURCES:
    sp = os.path.join(CSRC, src)
    op = os.path.join(LIBDIR, src.replace('.hip', '.o'))
    objs.append(op)
    if force or _stale(op, [sp] + headers):
      cmd = [hipcc] + FLAGS + ['-c', sp, '-o', op]
      if verbose:
        print(' '.join(cmd), flush=True)
      procs.append((src, subprocess.Popen(cmd)))
  for src, pr in procs:
    if pr.wait() != 0:
      raise RuntimeError('hipcc failed on %s' % src)
  if force or procs or _stale(LIB, objs):
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
      print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
  return LIB


if __name__ == '__main__':
  print(build(force='--force' in sys.argv))
