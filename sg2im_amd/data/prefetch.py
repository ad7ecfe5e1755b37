"""GPU-side input pipeline: host batches travel to the device AHEAD of the step that uses them.

The reference moves each batch with blocking ``.cuda()`` calls at the head of the iteration (scripts/train.py:508-519).
Here the DataLoader hands out PINNED host tensors (``pin_memory=True``) and ``CopyAhead`` issues the host-to-device
copies of batch k + 1 on a copy stream of its own while the training stream runs iteration k; the consumer's stream
only waits for the copy's event.  From pageable memory ``.to(device, non_blocking=True)`` is a synchronous copy on the
current stream - what scripts/train.py did until round 4."""
import torch


class CopyAhead(object):
  """Iterate ``batches`` (tuples of CPU tensors / other values) as device batches, one copy ahead.

  ``finish``: optional function applied to the raw host tuple first (e.g. scripts/train.py's 6-tuple -> 7-tuple).
  On a CPU ``device`` the batches pass through (tests).  A device batch stays valid until the batch after the NEXT
  one is requested: its memory is tied to the consumer's stream with ``record_stream``."""

  def __init__(self, batches, device, finish=None, pin=True):
    self.it = iter(batches)
    self.device = torch.device(device)
    self.finish = finish
    self.pin = pin
    self.cuda = self.device.type == 'cuda'
    self.stream = torch.cuda.Stream(device=self.device) if self.cuda else None
    self.ahead = None
    self.copies = 0

  def _host(self, t):
    if self.cuda and self.pin and torch.is_tensor(t) and not t.is_pinned():
      return t.pin_memory()            # (a loader without pin_memory, the synthetic generator)
    return t

  def _issue(self):
    try:
      raw = next(self.it)
    except StopIteration:
      self.ahead = None
      return
    if self.finish is not None:
      raw = self.finish(raw)
    if not self.cuda:
      self.ahead = (tuple(raw), None, None)
      return
    host = tuple(self._host(t) for t in raw)
    with torch.cuda.stream(self.stream):
      dev = tuple(t.to(self.device, non_blocking=True) if torch.is_tensor(t) else t for t in host)
      ev = torch.cuda.Event()
      ev.record(self.stream)
    self.copies += 1
    self.ahead = (dev, ev, host)       # (the pinned host tensors stay referenced until the copy was waited for)

  def __iter__(self):
    return self

  def __next__(self):
    if self.ahead is None:
      self._issue()
    if self.ahead is None:
      raise StopIteration
    dev, ev, _ = self.ahead
    if ev is not None:
      cur = torch.cuda.current_stream(self.device)
      cur.wait_event(ev)
      for t in dev:
        if torch.is_tensor(t):
          t.record_stream(cur)
    self._issue()                      # the next batch starts travelling now, under this iteration
    return dev
