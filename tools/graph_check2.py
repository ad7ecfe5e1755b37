import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sg2im_amd import ops
from sg2im_amd.synthetic import make_vocab, synthetic_batch
from sg2im_amd.trainer import Trainer
mode = sys.argv[1]
dev = torch.device('cuda', 0)
vocab = make_vocab(184, 7)
batch = tuple(t.to(dev) if torch.is_tensor(t) else t for t in synthetic_batch(4, seed=11))
b = Trainer(vocab, dev, use_graphs=True, generator_kwargs={'layout_noise_dim': 0}, seed=7)
x = torch.randn(8, 32, 32, 64, device=dev); w = torch.randn(64, 3, 3, 64, device=dev); o = torch.empty(8, 32, 32, 64, device=dev)
d = ops.conv_desc([ops.nhwc_src(x)], 8, 32, 32, 3, 3, 1, 1)
v = torch.randn(1 << 16, device=dev); vo = torch.empty_like(v)
for i in range(7):
  l = Trainer.losses_to_host(b.step(batch))
  torch.cuda.synchronize()
  print('step', i + 1, 'ok', l['total_loss'], flush=True)
  if mode == 'conv_prealloc':
    ops.conv2d_forward(d, w, 64, None, o, 64)
  elif mode == 'sigmoid_prealloc':
    ops.sigmoid_forward(v, vo)
  elif mode == 'small_allocs':
    xs = [torch.randn(1000 + 17 * k, device=dev) for k in range(200)]
    ys = [torch.randint(0, 5, (300 + k,), device=dev) for k in range(200)]
    del xs, ys
  torch.cuda.synchronize()
