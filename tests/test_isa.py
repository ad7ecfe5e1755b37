"""CPU: what the gfx950 code objects inside the built libsg2im_hip.so contain - read from the library the GPU box loads,
not from a fresh compile: the matrix-core instructions the design rests on are there, the implicit-GEMM translation unit
has no scratch (spill) instructions, and the only kernel of the library that spills is the one DESIGN.md section 4.3 says
does (the register-capped low-footprint GraphTripleConv backward)."""
import os
import re
import struct
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def _device_code_objects(lib, tmp):
  """every gfx950 code object of the library's .hip_fatbin section (one offload bundle per translation unit)"""
  fat = os.path.join(tmp, 'fat.bin')
  subprocess.check_call([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, lib])
  blob = open(fat, 'rb').read()
  out = []
  for m in re.finditer(re.escape(MAGIC), blob):
    base = m.start()
    pos = base + len(MAGIC)
    (n,) = struct.unpack_from('<Q', blob, pos)
    pos += 8
    for _ in range(n):
      off, size, idlen = struct.unpack_from('<QQQ', blob, pos)
      pos += 24
      ident = blob[pos:pos + idlen].decode()
      pos += idlen
      if 'gfx950' in ident and size:
        path = os.path.join(tmp, 'dev%d.o' % len(out))
        with open(path, 'wb') as f:
          f.write(blob[base + off:base + off + size])
        out.append(path)
  return out


def _kernels(asm):
  """{function name: its disassembly} of an llvm-objdump -d listing"""
  parts = re.split(r'^[0-9a-f]+ <([^>]+)>:$', asm, flags=re.M)
  return {parts[i]: parts[i + 1] for i in range(1, len(parts) - 1, 2)}


@pytest.fixture(scope='module')
def kernels():
  from sg2im_amd import build
  lib = build.build(verbose=False)
  if not os.path.exists(os.path.join(LLVM, 'llvm-objdump')):
    pytest.skip('no llvm-objdump in this image')
  found = {}
  with tempfile.TemporaryDirectory() as tmp:
    objs = _device_code_objects(lib, tmp)
    assert len(objs) >= 7, objs                      # one code object per translation unit that has kernels (gconv.hip has none)
    for o in objs:
      asm = subprocess.check_output([os.path.join(LLVM, 'llvm-objdump'), '-d', o]).decode()
      found.update(_kernels(asm))
  return found


def _count(kernels, pattern, name_filter=None):
  rx = re.compile(pattern)
  return sum(len(rx.findall(body)) for name, body in kernels.items() if name_filter is None or name_filter(name))


def test_matrix_core_and_transposing_lds_instructions_are_in_the_shipped_library(kernels):
  assert len(kernels) >= 150, len(kernels)
  conv = lambda n: 'conv_' in n or 'splitk' in n
  assert _count(kernels, r'v_mfma_f32_32x32x2_?f32', conv) >= 4000        # fp32 implicit GEMM (conv.hip)
  assert _count(kernels, r'v_mfma_f32_32x32x16_?bf16', conv) >= 400       # bf16 operand forms
  assert _count(kernels, r'ds_read_b64_tr_b16', conv) >= 200              # gfx950 transposing LDS read (section 4.2)
  assert _count(kernels, r'v_mfma_f32_32x32x2_?f32', lambda n: 'gcn_stack' in n) >= 2000      # persistent GraphTripleConv kernels
  # hand-written for gfx950 only: no vendor GEMM kernels linked in
  assert not [n for n in kernels if 'rocblas' in n.lower() or 'Cijk' in n or 'miopen' in n.lower()]


def test_only_the_register_capped_kernel_spills(kernels):
  spilling = sorted(n for n, body in kernels.items() if 'scratch_' in body)
  assert all('gcn_stack_bwd_low_kernel' in n for n in spilling), spilling
  # the hot implicit-GEMM instantiations in particular (round 4 shipped 44 scratch instructions in two of them)
  assert not [n for n in spilling if 'conv_' in n]
