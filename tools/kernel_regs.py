"""Register / LDS / scratch budget of the kernels in the built library whose (demangled) name contains a pattern:
  python tools/kernel_regs.py conv_halo_kernel"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_isa import _device_code_objects, LLVM
from sg2im_amd import build
pat = sys.argv[1] if len(sys.argv) > 1 else ''
with tempfile.TemporaryDirectory() as tmp:
  for o in _device_code_objects(build.LIB, tmp):
    txt = subprocess.check_output([os.path.join(LLVM, 'llvm-readelf'), '--notes', o]).decode()
    for blk in re.split(r'\n\s+- ', txt):
      m = re.search(r'\.name:\s+(\S+)', blk)
      if not m:
        continue
      name = subprocess.check_output(['c++filt', m.group(1)]).decode().strip()
      if pat not in name:
        continue
      g = lambda k: (re.search(r'\.%s:\s+(\d+)' % k, blk) or [None, '?'])[1]
      print('%-90s vgpr %s agpr %s sgpr %s lds %s scratch %s' % (name[:90], g('vgpr_count'), g('agpr_count'), g('sgpr_count'),
            g('group_segment_fixed_size'), g('private_segment_fixed_size')))
