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
  torch.randn = lambda *a, **k: noise.clone().to(k.get('device', 'cpu'))
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
