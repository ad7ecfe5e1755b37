"""Flag parsers, timing and loss bookkeeping (reference sg2im/utils.py)."""
import time
from contextlib import contextmanager

import torch


def int_tuple(s):
  return tuple(int(i) for i in s.split(','))


def float_tuple(s):
  return tuple(float(i) for i in s.split(','))


def str_tuple(s):
  return tuple(s.split(','))


def bool_flag(s):
  if s == '1':
    return True
  if s == '0':
    return False
  raise ValueError('Invalid value "%s" for bool flag (should be 0 or 1)' % s)


@contextmanager
def timeit(msg, should_time=True):
  """wall-clock a region between device synchronisations (reference sg2im/utils.py:63-73)"""
  if should_time:
    torch.cuda.synchronize()
    t0 = time.time()
  yield
  if should_time:
    torch.cuda.synchronize()
    print('%s: %.2f ms' % (msg, (time.time() - t0) * 1000.0))


class LossManager(object):
  """Sums weighted losses; unlike the reference (sg2im/utils.py:76-91, one ``.item()``
  host sync per loss) the scalars stay on the device until ``items()`` is called."""

  def __init__(self):
    self.total_loss = None
    self.all_losses = {}

  def add_loss(self, loss, name, weight=1.0):
    cur = loss * weight if weight != 1.0 else loss
    self.total_loss = cur if self.total_loss is None else self.total_loss + cur
    self.all_losses[name] = cur.detach()

  def items(self):
    return [(k, float(v)) for k, v in self.all_losses.items()]


IMAGENET_MEAN = [0.485, 0.456, 0.406]
IMAGENET_STD = [0.229, 0.224, 0.225]


def imagenet_deprocess_batch(imgs, rescale=True):
  """(N, 3, H, W) normalised floats -> uint8 images on the host, one image at a time like
  reference sg2im/data/utils.py:50-68 (x * std + mean, optional per-image min/max rescale)."""
  imgs = imgs.detach().float().cpu()
  mean = torch.tensor(IMAGENET_MEAN).view(3, 1, 1)
  std = torch.tensor(IMAGENET_STD).view(3, 1, 1)
  out = []
  for img in imgs:
    img = img * std + mean
    if rescale:
      lo, hi = img.min(), img.max()
      img = (img - lo) / (hi - lo)
    out.append(img.mul(255).clamp(0, 255).byte()[None])
  return torch.cat(out, dim=0)
