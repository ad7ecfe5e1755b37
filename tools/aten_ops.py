"""Which ATen kernels (not ours) does one eager training step launch?  usage: python tools/aten_ops.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from sg2im_amd.synthetic import make_vocab, synthetic_batch
from sg2im_amd.trainer import Trainer

dev = torch.device('cuda', 0)
vocab = make_vocab(184, 7)
batch = tuple(t.to(dev) if torch.is_tensor(t) else t for t in synthetic_batch(32, seed=0))
tr = Trainer(vocab, dev, seed=1, use_graphs=False)
for _ in range(3):
  tr.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
  tr.step(batch)
  torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_stack_n=6) if e.key.startswith('aten::') and e.device_time_total > 0]
rows.sort(key=lambda e: -e.count)
for e in rows[:40]:
  stack = [s for s in e.stack if 'sg2im_amd' in s or 'scripts' in s][:2]
  print('%-28s n=%3d dev_us=%8.1f  %s' % (e.key, e.count, e.device_time_total, ' <- '.join(s.split('/')[-1] for s in stack)))
