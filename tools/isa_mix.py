"""Instruction mix of kernels in the built library: whole kernel and the innermost loop that holds the MFMAs.
  python tools/isa_mix.py conv_dgrad_kernel      (substring of the demangled name)"""
import collections, os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_isa import _device_code_objects, _kernels, LLVM
from sg2im_amd import build


def klass(op):
  return ('mfma' if op.startswith('v_mfma') else 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else
          'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')) else 'other')


def mix(lines):
  c = collections.Counter()
  for l in lines:
    t = l.split('//')[0].strip()
    if not t or t.endswith(':'):
      continue
    t = re.sub(r'^[0-9a-f]+:\s+', '', t)
    op = t.split()[0] if t.split() else ''
    if op:
      c[klass(op)] += 1
  return c


pat = sys.argv[1] if len(sys.argv) > 1 else 'conv_'
with tempfile.TemporaryDirectory() as tmp:
  for o in _device_code_objects(build.LIB, tmp):
    asm = subprocess.check_output([os.path.join(LLVM, 'llvm-objdump'), '-d', o]).decode()
    for name, body in _kernels(asm).items():
      dn = subprocess.check_output(['c++filt', name]).decode().strip()
      if pat not in dn:
        continue
      lines = [l for l in body.split('\n') if l.strip()]
      ops = []
      for l in lines:
        m = re.match(r'\s*(\S+)\s', l)
        ops.append(m.group(1) if m else '')
      tot = mix(lines)
      # innermost backward branch span containing MFMAs: find "s_cbranch... <name+0xOFF>" with OFF before the branch
      addr = {}
      for i, l in enumerate(lines):
        m = re.search(r'//\s*([0-9A-Fa-f]+):', l)
        if m:
          addr[int(m.group(1), 16)] = i
      base = min(addr) if addr else 0
      best = None
      for i, l in enumerate(lines):
        m = re.search(r's_c?branch\S*\s+\S+\s+//.*<\S+\+0x([0-9a-fA-F]+)>', l)
        if m:
          tgt = base + int(m.group(1), 16)
          j = addr.get(tgt)
          if j is not None and j < i and any('v_mfma' in x for x in lines[j:i]):
            if best is None or (i - j) < (best[1] - best[0]):
              best = (j, i)
      loop = mix(lines[best[0]:best[1] + 1]) if best else collections.Counter()
      print('%-100s' % dn[:100])
      print('   whole: %s' % dict(tot))
      print('   loop : %s   -> outside the loop: valu %d salu %d vmem %d' % (dict(loop), tot['valu'] - loop['valu'], tot['salu'] - loop['salu'], tot['vmem'] - loop['vmem']))
