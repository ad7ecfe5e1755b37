"""Helpers shared by the GPU parity tests: build the HIP modules from an oracle/golden
parameter dict, run them on cuda:0 and pull results back to the CPU."""
import contextlib

import torch

from oracle import sg2im_oracle as orc


def dev():
  return torch.device('cuda', 0)


def to_dev(x):
  if torch.is_tensor(x):
    return x.to(dev())
  if isinstance(x, (list, tuple)):
    return type(x)(to_dev(v) for v in x)
  return x


def load_params(module, P):
  """copy an oracle parameter dict into a HIP module (keeps channels_last conv storage)"""
  sd = module.state_dict()
  missing = [k for k in sd if k not in P]
  extra = [k for k in P if k not in sd]
  assert not missing and not extra, (missing, extra)
  with torch.no_grad():
    for k, v in sd.items():
      v.copy_(P[k].detach())
  return module


def build_generator(gcfg, PG):
  from sg2im_amd.model import Sg2ImModel
  kw = {k: v for k, v in gcfg.items() if k != 'vocab'}
  m = Sg2ImModel(gcfg['vocab'], **kw)
  load_params(m, PG)
  return m.to(dev())


def build_d_obj(docfg, P):
  from sg2im_amd.discriminators import AcCropDiscriminator
  kw = {k: v for k, v in docfg.items() if k != 'vocab'}
  m = AcCropDiscriminator(docfg['vocab'], **kw)
  load_params(m, P)
  return m.to(dev())


def build_d_img(dicfg, P):
  from sg2im_amd.discriminators import PatchDiscriminator
  m = PatchDiscriminator(**dicfg)
  load_params(m, P)
  return m.to(dev())


@contextlib.contextmanager
def fixed_noise(noise):
  """make torch.randn return ``noise`` (model.py:164-168 draws the layout noise with it)"""
  if noise is None:
    yield
    return
  real = torch.randn
  # (staged on the GPU up front: inside a stream capture a host -> device copy of a temporary would be
  # recorded with a dangling host pointer)
  staged = {'cpu': noise}
  if torch.cuda.is_available():
    staged['cuda'] = noise.to(dev())

  def fake(*a, **k):
    d = torch.device(k.get('device', 'cpu'))
    return staged[d.type].clone()
  torch.randn = fake
  try:
    yield
  finally:
    torch.randn = real


def grads_of(module):
  return {k: (None if p.grad is None else p.grad.detach().cpu().contiguous()) for k, p in module.named_parameters()}


def oracle_leafs(P):
  out = {}
  for k, v in P.items():
    t = v.detach().clone()
    if t.is_floating_point() and 'running_' not in k:
      t.requires_grad_(True)
    out[k] = t
  return out


# ---------------------------------------------------------------------------------------
# Backward parity of a whole training iteration: the flat gradient arenas of the
# HIP Trainer, sliced per parameter, against the oracle's ``.grad`` after the SAME step.  (Parameter
# distances after an Adam update cannot serve: Adam moves every element by +-lr whatever the gradient,
# so two runs from the same weights are never more than 2 lr apart.)
# ---------------------------------------------------------------------------------------
GRAD_REL, GRAD_ABS_ZERO = 1e-4, 1e-6


def grad_parity_rows(tr, otr, scale=1.0):
  """rows (net, name, rel-to-max error, cosine, abs error, reference max) for every parameter of
  G / D_obj / D_img; a parameter without a reference gradient must have an all-zero arena slot
  (reported with rel = abs = the slot's max magnitude).  ``scale``: factor applied to the arena first
  (1 / world_size after a SUM all-reduce)."""
  rows = []
  for net, mod, P in (('G', tr.model, otr.PG), ('Do', tr.d_obj, otr.PDo), ('Di', tr.d_img, otr.PDi)):
    if mod is None:
      assert P is None, net
      continue
    for name, p in mod.named_parameters():
      got = p.grad.detach().cpu().double() * scale
      ref = P[name].grad
      if ref is None:
        m = float(got.abs().max())
        rows.append((net, name, m, 1.0, m, 0.0))
        continue
      ref = ref.detach().double()
      assert got.shape == ref.shape, (net, name, tuple(got.shape), tuple(ref.shape))
      err, rmax = float((got - ref).abs().max()), float(ref.abs().max())
      den = float(got.norm() * ref.norm())
      cos = float((got * ref).sum()) / den if den > 0 else 1.0
      rows.append((net, name, err / max(rmax, 1e-30), cos, err, rmax))
  return rows


def assert_grad_parity(tr, otr, label, rel=GRAD_REL, cos_min=None, scale=1.0):
  """every parameter gradient within ``rel`` of its tensor's max magnitude (analytically-zero tensors -
  the reference itself below 1e-6 everywhere, e.g. the bias of a convolution that feeds a batch-statistics
  BatchNorm - within 1e-6 absolute); optionally a cosine bound per tensor (bf16).  Appends the measured
  worst cases to gpurun_out/grad_parity.log and returns (worst rel, worst cosine)."""
  import os
  rows = grad_parity_rows(tr, otr, scale)
  live = [r for r in rows if r[5] >= GRAD_ABS_ZERO]
  worst = max(live, key=lambda r: r[2])
  wcos = min(live, key=lambda r: r[3])
  line = ('%-44s tensors %3d  worst rel-to-max %.3e (%s.%s)  worst cosine %.6f (%s.%s)' %
          (label, len(rows), worst[2], worst[0], worst[1], wcos[3], wcos[0], wcos[1]))
  print(line)
  try:
    os.makedirs('gpurun_out', exist_ok=True)
    with open(os.path.join('gpurun_out', 'grad_parity.log'), 'a') as f:
      f.write(line + '\n')
  except OSError:
    pass
  bad = [r for r in rows if not (r[2] <= rel or (r[4] <= GRAD_ABS_ZERO and r[5] < GRAD_ABS_ZERO))]
  if cos_min is not None:
    bad += [r for r in live if r[3] < cos_min and r not in bad]
  assert not bad, '%s: gradients out of tolerance:\n' % label + '\n'.join(
    '  %s.%s rel %.3e cos %.6f abs %.3e refmax %.3e' % r for r in bad[:20])
  return worst[2], wcos[3]
