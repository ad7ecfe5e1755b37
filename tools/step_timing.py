"""CPU enqueue time vs GPU time of one trainer step (is the step launch-bound?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sg2im_amd.synthetic import make_vocab, synthetic_batch
from sg2im_amd.trainer import Trainer
dev = torch.device('cuda', 0)
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
batch = tuple(t.to(dev) if torch.is_tensor(t) else t for t in synthetic_batch(bs, seed=0))
tr = Trainer(make_vocab(184, 7), dev, seed=1)
for _ in range(5):
  tr.step(batch)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
  tr.step(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('batch %d: cpu enqueue %.2f ms/step, wall %.2f ms/step (gpu drained %.2f ms after the last enqueue)' % (
  bs, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, (t2 - t1) * 1e3))
