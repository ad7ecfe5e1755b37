"""cProfile of the eager trainer step (host-side launch overhead)."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sg2im_amd.synthetic import make_vocab, synthetic_batch
from sg2im_amd.trainer import Trainer
dev = torch.device('cuda', 0)
batch = tuple(t.to(dev) if torch.is_tensor(t) else t for t in synthetic_batch(4, seed=0))   # small batch: CPU-bound
tr = Trainer(make_vocab(184, 7), dev, seed=1)
for _ in range(5):
  tr.step(batch)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
  tr.step(batch)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(45)
st.sort_stats('tottime').print_stats(25)
