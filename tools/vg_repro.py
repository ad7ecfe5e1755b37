import sys; sys.path.insert(0,'/root/repo')
import torch
from sg2im_amd.synthetic import make_vocab, synthetic_batch
from sg2im_amd.trainer import Trainer
dev=torch.device('cuda',0)
vocab = make_vocab(179, 46)
cpu = synthetic_batch(4, num_objs=179, num_preds=46, style='vg', min_objs=3, max_objs=10, seed=19)
batch = tuple(t.to(dev) if torch.is_tensor(t) else t for t in cpu)
kw = dict(generator_kwargs={'layout_noise_dim': 0}, seed=3, loss_weights=dict(predicate_pred_loss_weight=0.2, mask_loss_weight=0.0))
for use_graphs in (False, False, False, True, True, True, True, True):
  tr = Trainer(vocab, dev, use_graphs=use_graphs, **kw)
  out=[Trainer.losses_to_host(tr.step(batch)) for _ in range(5)]
  print('graphs' if use_graphs else 'eager ', ' '.join('%.6f' % o['bbox_pred'] for o in out), ' | total', ' '.join('%.5f' % o['total_loss'] for o in out))
