"""CPU: the C-ABI shared library builds for gfx950, loads, and exports every symbol that
include/sg2im_hip.h declares, with the argument counts the ctypes binding expects.  No
compute call is made (there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'sg2im_hip.h')


def declared_functions():
  src = open(HEADER).read()
  src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
  out = {}
  for m in re.finditer(r'\b(?:int|size_t|unsigned long long)\s+(sg2im_\w+)\s*\(([^;]*?)\)\s*;', src, flags=re.S):
    args = m.group(2).strip()
    n = 0 if args in ('', 'void') else len([a for a in args.split(',') if a.strip()])
    out[m.group(1)] = n
  return out


@pytest.fixture(scope='module')
def lib():
  from sg2im_amd import build, _lib
  build.build(verbose=False)
  return _lib.load()


def test_header_declares_the_hot_path():
  fns = declared_functions()
  for name in ('sg2im_conv2d_forward', 'sg2im_conv2d_backward_data', 'sg2im_conv2d_backward_weight',
               'sg2im_csr_build', 'sg2im_segment_sum', 'sg2im_layout_forward', 'sg2im_layout_backward',
               'sg2im_crop_forward', 'sg2im_bn_stats', 'sg2im_adam_step_guarded'):
    assert name in fns


def test_every_declared_symbol_is_exported(lib):
  for name in declared_functions():
    assert hasattr(lib, name), 'libsg2im_hip.so does not export %s' % name


def test_binding_matches_header(lib):
  from sg2im_amd import _lib
  fns = declared_functions()
  assert set(fns) == set(_lib._SIGNATURES), (set(fns) ^ set(_lib._SIGNATURES))
  for name, n in fns.items():
    assert len(_lib._SIGNATURES[name]) == n, (name, n, len(_lib._SIGNATURES[name]))
  assert lib.sg2im_abi_version() >= 1


def test_argument_validation_without_a_gpu(lib):
  """NULL / malformed arguments are rejected with SG2IM_ERR_ARG before any HIP call"""
  from sg2im_amd._lib import ConvDesc, SG2IM_ERR_ARG
  d = ConvDesc()
  d.nsrc = 0
  assert lib.sg2im_conv2d_forward(ctypes.byref(d), None, 8, None, 1.0, None, 8, 0, None, 0, None) == SG2IM_ERR_ARG
  assert lib.sg2im_segment_sum(None, 0, 0, None, 0, None, None, 4, 0, 0, 0, None, 0, None) == SG2IM_ERR_ARG
  assert lib.sg2im_adam_step(None, None, None, None, 0, 1e-4, .9, .999, 1e-8, 1, 1.0, None) == SG2IM_ERR_ARG


def test_product_path_has_no_oracle_import():
  """the oracle is test infrastructure: nothing under sg2im_amd/ may import it"""
  pkg = os.path.join(ROOT, 'sg2im_amd')
  for dirpath, _, files in os.walk(pkg):
    for f in files:
      if f.endswith('.py'):
        text = open(os.path.join(dirpath, f)).read()
        assert not re.search(r'^\s*(from|import)\s+oracle\b', text, flags=re.M), f


def test_plain_cpp_user_of_the_header_compiles_and_links():
  """examples/abi_smoke.cpp - a C++ program with hipMalloc'd buffers and no torch - compiles against include/sg2im_hip.h
  and links against the built library (cross-compiled here; it RUNS on the GPU box: profiles/r5_abi_smoke.log)."""
  import subprocess
  import tempfile
  from sg2im_amd import build
  lib = build.build(verbose=False)
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, 'abi_smoke')
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-std=c++17', '-I' + os.path.join(root, 'include'),
           os.path.join(root, 'examples', 'abi_smoke.cpp'), '-L' + os.path.dirname(lib), '-lsg2im_hip', '-o', out]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    assert os.path.getsize(out) > 0
