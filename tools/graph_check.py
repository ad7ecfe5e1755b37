import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sg2im_amd.synthetic import make_vocab, synthetic_batch
from sg2im_amd.trainer import Trainer
dev = torch.device('cuda', 0)
vocab = make_vocab(184, 7)
batch = tuple(t.to(dev) if torch.is_tensor(t) else t for t in synthetic_batch(4, seed=11))
kw = dict(generator_kwargs={'layout_noise_dim': 0}, seed=7)
a, b, c = Trainer(vocab, dev, use_graphs=False, **kw), Trainer(vocab, dev, use_graphs=True, **kw), Trainer(vocab, dev, use_graphs=False, **kw)
for i in range(7):
  la, lb, lc = [Trainer.losses_to_host(t.step(batch)) for t in (a, b, c)]
  print('step', i + 1, 'graphs' if b._graphs else 'eager ', ' '.join('%s %.2e/%.2e' % (k[:9], abs(la[k] - lb[k]), abs(la[k] - lc[k])) for k in ('total_loss', 'd_obj_gan_loss', 'd_img_gan_loss', 'ac_loss')),
        'param diff g %.2e/%.2e do %.2e/%.2e' % (float((a.flat_g.flat - b.flat_g.flat).abs().max()), float((a.flat_g.flat - c.flat_g.flat).abs().max()),
                                                  float((a.flat_do.flat - b.flat_do.flat).abs().max()), float((a.flat_do.flat - c.flat_do.flat).abs().max())))
