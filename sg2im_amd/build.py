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
SOURCES = ['conv.hip', 'graph.hip', 'gconv.hip', 'gcn_persist.hip', 'heads.hip', 'norm.hip', 'layout.hip', 'loss.hip']
INCLUDES = {}      # (.hip files another translation unit includes: none)
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC]


def _stale(target, deps):
  if not os.path.exists(target):
    return True
  t = os.path.getmtime(target)
  return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, cflags=(), tag=''):
  """tag/cflags: experimental variants, e.g. build(cflags=['-DSG2IM_LDS_STAGES=1'], tag='_lds1')
  -> lib/libsg2im_hip_lds1.so, selected at run time with SG2IM_LIB=<path>."""
  hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
  os.makedirs(LIBDIR, exist_ok=True)
  lib_path = LIB.replace('.so', tag + '.so')
  headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
  headers.append(os.path.join(ROOT, 'include', 'sg2im_hip.h'))
  objs, procs = [], []
  for src in SOURCES:
    sp = os.path.join(CSRC, src)
    op = os.path.join(LIBDIR, src.replace('.hip', tag + '.o'))
    objs.append(op)
    if force or _stale(op, [sp] + headers + [os.path.join(CSRC, f) for f in INCLUDES.get(src, ())]):
      cmd = [hipcc] + FLAGS + list(cflags) + ['-c', sp, '-o', op]
      if verbose:
        print(' '.join(cmd), flush=True)
      procs.append((src, subprocess.Popen(cmd)))
  for src, pr in procs:
    if pr.wait() != 0:
      raise RuntimeError('hipcc failed on %s' % src)
  if force or procs or _stale(lib_path, objs):
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib_path] + objs
    if verbose:
      print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
  return lib_path


if __name__ == '__main__':
  tag = [a[6:] for a in sys.argv if a.startswith('--tag=')]
  print(build(force='--force' in sys.argv, cflags=[a for a in sys.argv[1:] if a.startswith('-D')],
              tag=tag[0] if tag else ''))
