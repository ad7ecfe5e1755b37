import sys, os, torch
sys.path.insert(0, os.getcwd())
from oracle import sg2im_oracle as orc
from sg2im_amd.synthetic import make_vocab, synthetic_batch
from sg2im_amd.trainer import GENERATOR_DEFAULTS
from tests import hip_harness as hh
dev = hh.dev()
vocab = make_vocab(184, 7)
for norm in ('none', 'batch'):
  gcfg = dict(GENERATOR_DEFAULTS, vocab=vocab, normalization=norm)
  PG = orc.init_generator_params(gcfg, 21)
  b = synthetic_batch(8, seed=61)
  noise = torch.randn(8, 32, 64, 64, generator=torch.Generator().manual_seed(62))
  G = hh.build_generator(gcfg, PG).train()
  imgs, objs, boxes, masks, triples, o2i = [hh.to_dev(t) for t in b[:6]]
  with hh.fixed_noise(noise):
    ip, bp, mp, rs = G(objs, triples, o2i, boxes_gt=boxes, masks_gt=masks, num_images=8)
  outs = {}
  for dt in (torch.float32, torch.float64):
    P = {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in PG.items()}
    with hh.fixed_noise(noise.to(dt)):
      o = orc.generator_forward(P, gcfg, b[1], b[4], b[5], boxes_gt=b[2].to(dt), masks_gt=b[3], training=True)
    outs[dt] = o
  i64 = outs[torch.float64][0]
  print(norm, 'img: hip vs f64 %.2e   oracle32 vs f64 %.2e   max %.2e' % (float((ip.cpu().double() - i64).abs().max()), float((outs[torch.float32][0].double() - i64).abs().max()), float(i64.abs().max())))
  print(norm, 'boxes: hip vs f64 %.2e  o32 %.2e' % (float((bp.cpu().double() - outs[torch.float64][1]).abs().max()), float((outs[torch.float32][1].double() - outs[torch.float64][1]).abs().max())))
