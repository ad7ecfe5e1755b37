"""Gradient parity of one HIP training iteration against the CPU oracle evaluated in float32 (the
reference's arithmetic) AND in float64 (the exact gradient, up to 1e-16): per parameter tensor
  e_hip64 = |g_hip - g_f64|_max / |g_f64|_max      e_ref = |g_oracle32 - g_f64|_max / |g_f64|_max
Usage (GPU box):  python tools/grad_parity_probe.py [batch] [graph]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from oracle import sg2im_oracle as orc
from sg2im_amd.synthetic import make_vocab, synthetic_batch
from sg2im_amd.trainer import Trainer, GENERATOR_DEFAULTS, D_OBJ_DEFAULTS, D_IMG_DEFAULTS
from tests import hip_harness as hh


def main():
  bs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
  graph = len(sys.argv) > 2 and sys.argv[2] == 'graph'
  dev = hh.dev()
  vocab = make_vocab(184, 7)
  cpu_batch = synthetic_batch(bs, seed=3)
  gcfg, docfg, dicfg = dict(GENERATOR_DEFAULTS, vocab=vocab), dict(D_OBJ_DEFAULTS, vocab=vocab), dict(D_IMG_DEFAULTS)
  PG = orc.init_generator_params(gcfg, 0, randomize_bn=True)
  PDo = orc.init_ac_discriminator_params(docfg, 2, randomize_bn=True)
  PDi = orc.init_patch_discriminator_params(dicfg, 1, randomize_bn=True)
  noise = torch.randn(bs, 32, 64, 64, generator=torch.Generator().manual_seed(5))
  tr = Trainer(vocab, dev, seed=0, use_graphs=graph)
  hh.load_params(tr.model, PG); hh.load_params(tr.d_obj, PDo); hh.load_params(tr.d_img, PDi)
  batch = tuple(t.to(dev) if torch.is_tensor(t) else t for t in cpu_batch)
  with hh.fixed_noise(noise):
    got = Trainer.losses_to_host(tr.step(batch))
  o32 = hh.oracle_trainer(PG, PDo, PDi, gcfg, docfg, dicfg, torch.float32)
  o64 = hh.oracle_trainer(PG, PDo, PDi, gcfg, docfg, dicfg, torch.float64)
  w32 = o32.step(hh.cast_batch(cpu_batch, torch.float32), noise)
  w64 = o64.step(hh.cast_batch(cpu_batch, torch.float64), noise.double())
  for k in w64:
    print('%-18s hip %.9g  o32 %.9g  f64 %.12g' % (k, got[k], w32[k], w64[k]))
  rows = hh.grad_parity_rows3(tr, o32, o64)
  for net in ('G', 'Do', 'Di'):
    sel = [r for r in rows if r[0] == net and r[5] >= 1e-6]
    print('%s: %d tensors; max e_hip64 %.3e, max e_ref %.3e, max e_hip32 %.3e; tensors with e_hip64 > e_ref: %d' % (
      net, len(sel), max(r[2] for r in sel), max(r[3] for r in sel), max(r[4] for r in sel),
      sum(1 for r in sel if r[2] > r[3])))
  rows.sort(key=lambda r: -r[2])
  print('%-58s %10s %10s %10s %10s' % ('tensor', 'e_hip64', 'e_ref', 'e_hip32', 'max|g|'))
  for r in rows[:25]:
    print('%-58s %10.3e %10.3e %10.3e %10.3e' % (r[0] + '.' + r[1], r[2], r[3], r[4], r[5]))
  rows.sort(key=lambda r: -(r[2] / max(r[3], 1e-12)))
  print('largest e_hip64 / e_ref:')
  for r in rows[:15]:
    print('%-58s %10.3e %10.3e %10.3e %10.3e' % (r[0] + '.' + r[1], r[2], r[3], r[4], r[5]))


if __name__ == '__main__':
  main()
