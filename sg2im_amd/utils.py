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
