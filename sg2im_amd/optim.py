"""Flat parameter arenas + fused Adam.

All parameters of a network live in ONE contiguous fp32 HBM buffer (and their gradients
in a second one), so the optimiser is a single HIP kernel launch over the arena
(sg2im_adam_step) and a data-parallel gradient exchange is one RCCL all-reduce over one
buffer - instead of the reference's per-tensor torch.optim.Adam loop
(scripts/train.py:426-443, ~150 tensors for the generator).
"""
import weakref

import torch

from . import functional as HF
from . import ops


class FlatParams(object):
  """Re-homes every parameter of ``module`` into one flat buffer (values preserved, conv
  weights keep their channels_last strides) and gives every parameter a ``.grad`` view
  into a flat gradient buffer, which autograd then accumulates into in place."""

  def __init__(self, module):
    params = [p for p in module.parameters()]
    if not params:
      raise ValueError('module has no parameters')
    dev = params[0].device
    offs, total = [], 0
    for p in params:
      offs.append(total)
      total += (p.numel() + 3) // 4 * 4            # keep every tensor 16-byte aligned
    self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
    self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
    self.params, self.offsets = params, offs
    self.sinks = {}                  # parameter data pointer -> dense view of its gradient slot
    me = weakref.ref(self)
    with torch.no_grad():
      for p, off in zip(params, offs):
        view = self._view(self.flat, p, off)
        view.copy_(p.data)
        p.data = view
        p.grad = self._view(self.grad, p, off)
        # backward kernels accumulate straight into the arena (see functional.GRAD_SINKS)
        self.sinks[view.data_ptr()] = self._sink_view(p, off)
        HF.GRAD_SINKS[view.data_ptr()] = me
    self.numel = total
    self.mirror = None               # bfloat16 copy of `flat` (refresh_mirror), same element offsets

  def refresh_mirror(self):
    """bfloat16 mirror of the parameter arena (sg2im_conv_desc.weight_bf16): ONE launch over the arena, on the current
    stream.  The Trainer calls it at the start of every bf16 iteration (inside the captured graph), so the mirror always
    holds RNE(current fp32 weights) whoever changed them - Adam, a checkpoint restore, a broadcast, a test writing into
    the parameters; functional._weight_mirror only hands it out while ops.WEIGHT_MIRROR is set (Trainer.step)."""
    if self.mirror is None:
      if torch.cuda.is_current_stream_capturing():
        raise RuntimeError('the weight mirror must be allocated before the capture (Trainer._prepare_lanes)')
      self.mirror = torch.zeros(self.numel + 16, dtype=torch.bfloat16, device=self.flat.device)    # (+16: the loaders read whole 16-byte pieces)
    ops.cast_f32_to_bf16(self.flat, self.mirror, self.numel)
    return self.mirror

  def close(self):
    """forget the gradient sinks (also done when the object is collected)"""
    for ptr in list(self.sinks):
      ref = HF.GRAD_SINKS.get(ptr)
      if ref is not None and ref() in (self, None):
        HF.GRAD_SINKS.pop(ptr, None)
    self.sinks = {}

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass

  def _sink_view(self, p, off):
    """dense physical view of the parameter's gradient slot: [Cout][KH][KW][Cin] for a
    channels_last conv weight, the parameter's own shape otherwise"""
    if p.dim() == 4:
      co, ci, kh, kw = p.shape
      return self.grad[off:off + p.numel()].view(co, kh, kw, ci)
    return self.grad[off:off + p.numel()].view(p.shape)

  @staticmethod
  def _view(buf, p, off):
    # same sizes/strides as the parameter (dense, possibly permuted) on top of the arena
    return torch.as_strided(buf, p.size(), p.stride(), off)

  def zero_grad(self):
    self.grad.zero_()
    for p, off in zip(self.params, self.offsets):
      if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
        p.grad = self._view(self.grad, p, off)


class FlatAdam(object):
  """torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8) over a FlatParams arena.  A
  parameter that never receives a gradient keeps g = m = v = 0 and is left untouched,
  like a ``grad is None`` parameter that torch.optim.Adam skips."""

  def __init__(self, flat, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
    self.flat, self.lr, self.betas, self.eps = flat, lr, betas, eps
    self.exp_avg = torch.zeros_like(flat.flat)
    self.exp_avg_sq = torch.zeros_like(flat.flat)
    self.t = 0
    self.state = torch.zeros(4, dtype=torch.float32, device=flat.flat.device)   # device-side step counter

  def zero_grad(self):
    self.flat.zero_grad()

  def reset_state(self):
    """a fresh optimiser (the reference re-creates Adam when the generator switches to
    eval mode, scripts/train.py:509-512)"""
    self.exp_avg.zero_()
    self.exp_avg_sq.zero_()
    self.state.zero_()
    self.t = 0

  def step(self, grad_scale=1.0):
    self.t += 1
    ops.adam_step(self.flat.flat, self.flat.grad, self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0],
                  self.betas[1], self.eps, self.t, grad_scale)

  def step_guarded(self, guard, grad_scale=1.0):
    """Apply the update unless the device scalar ``guard`` is non-finite (no host sync)."""
    ops.adam_step_guarded(self.flat.flat, self.flat.grad, self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0],
                          self.betas[1], self.eps, self.state, guard, grad_scale)

  def prepare_guarded(self, guard):
    """first half of step_guarded: the step counter / bias corrections (or the skip flag) of THIS step"""
    ops.adam_prepare_guarded(self.lr, self.betas[0], self.betas[1], self.state, guard)

  def apply_guarded(self, lo, hi, grad_scale=1.0):
    """second half, for the arena slice [lo, hi): may be called for disjoint slices at different times / on
    different streams, each ordered after prepare_guarded and after the slice's gradients are complete"""
    if hi > lo:
      ops.adam_apply_guarded(self.flat.flat[lo:hi], self.flat.grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi],
                             self.betas[0], self.betas[1], self.eps, self.state, grad_scale)

  def _steps_taken(self):
    # the guarded path counts on the device (a skipped non-finite step does not count)
    return max(int(self.t), int(round(float(self.state[0].item()))))

  def state_dict(self):
    """The torch.optim.Adam layout the reference checkpoints hold (scripts/train.py:642-650:
    ``optimizer.state_dict()``): per-parameter ``step / exp_avg / exp_avg_sq`` keyed by the
    index in ``module.parameters()`` order plus one param group, so either side can resume the
    other's checkpoint.  Moments are exported in the parameter's logical (OIHW) layout."""
    step = self._steps_taken()
    state = {}
    if step > 0:
      for i, (p, off) in enumerate(zip(self.flat.params, self.flat.offsets)):
        state[i] = {'step': torch.tensor(float(step)),
                    'exp_avg': FlatParams._view(self.exp_avg, p, off).clone(memory_format=torch.contiguous_format),
                    'exp_avg_sq': FlatParams._view(self.exp_avg_sq, p, off).clone(memory_format=torch.contiguous_format)}
    group = {'lr': self.lr, 'betas': tuple(self.betas), 'eps': self.eps, 'weight_decay': 0, 'amsgrad': False,
             'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
             'params': list(range(len(self.flat.params)))}
    return {'state': state, 'param_groups': [group]}

  def load_state_dict(self, sd):
    """accepts ``torch.optim.Adam.state_dict()`` of the same module (parameters the reference never
    stepped have no entry there and keep zero moments)"""
    self.reset_state()
    step = 0
    for i, ent in sd['state'].items():
      p, off = self.flat.params[int(i)], self.flat.offsets[int(i)]
      FlatParams._view(self.exp_avg, p, off).copy_(ent['exp_avg'])
      FlatParams._view(self.exp_avg_sq, p, off).copy_(ent['exp_avg_sq'])
      step = max(step, int(round(float(ent['step']))))
    groups = sd.get('param_groups') or [{}]
    self.lr = groups[0].get('lr', self.lr)
    self.t = step
    self.state[0] = float(step)
