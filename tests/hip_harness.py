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
#
# What the gradients can be held to.  The step is NOT well conditioned: the L1 pixel loss is a sign
# function of the prediction, every LeakyReLU / ReLU has a kink, and ten training-mode BatchNorms subtract
# batch means in their backward - a 1e-7 perturbation of the forward pass flips a few of those decisions.
# The oracle evaluated in float32 (the reference's own arithmetic) therefore deviates from the SAME oracle
# evaluated in float64 by up to 3e-2 of a tensor's max magnitude at batch 4 and 8e-3 at batch 32 (measured:
# profiles/r3_grad_parity_probe_*.log; pinned on the CPU by tests/test_oracle_golden.py::
# test_fp32_gradient_of_the_step_is_only_accurate_to_1e2), although every loss agrees to 1e-7.  A bound
# of 1e-4 against the float32 oracle is thus unattainable even for a bit-exact reimplementation with another
# summation order.  The gradients are compared with the EXACT gradient instead - the oracle in float64 - and
# must be as close to it as the reference's arithmetic is:
#     e_hip64(t) <= max(1e-4, 3 E_ref)   and   cosine(g_hip(t), g_f64(t)) >= 0.999   for every tensor t
#     [or, for a tensor hit by a flipped decision that the float32 oracle happened not to share - single flips move a
#      few elements by a lot on the 2-4 image batches of the small tests -: cosine >= 0.9999 and e_hip64 <= 0.1]
# with e_hip64 / e_ref the max-abs errors of the HIP arena / of the float32 oracle against float64, relative
# to the tensor's max magnitude, and E_ref the largest e_ref over all tensors of the step (which decisions
# flip is a draw per implementation: per tensor the two errors differ by large factors either way - one side
# often has no flip at all where the other has one - while the worst tensor of a step is a stable measure of
# the step's conditioning; measured e_hip64 <= 1.3 E_ref everywhere, profiles/r3_grad_parity.log).
# A genuinely wrong gradient (sign, missing term, missing 1/world) is off by O(1): 10-100x above the bound,
# and every kernel is held to 1e-4 on its own by the op-level sections (tools/gpu_check.py).
# ---------------------------------------------------------------------------------------
GRAD_REL, GRAD_ABS_ZERO, GRAD_COS = 1e-4, 1e-6, 0.999
NO_REFERENCE_GRAD = -1.0     # max|g_f64| of a parameter the reference has no gradient for


def cast_batch(cpu_batch, dtype):
  return tuple((t.to(dtype) if torch.is_tensor(t) and t.is_floating_point() else t) for t in cpu_batch[:6])


def oracle_trainer(PG, PDo, PDi, gcfg, docfg, dicfg, dtype=torch.float32, **kw):
  """an OracleTrainer over copies of the parameter dicts cast to ``dtype`` (float64: the exact gradient)"""
  cv = lambda P: {k: (v.detach().clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in P.items()}
  return orc.OracleTrainer(cv(PG), cv(PDo), cv(PDi), gcfg, docfg, dicfg, **kw)


class OracleRefs(object):
  """The float32 oracle (losses, parameters, running statistics: what the tests always compared with) plus
  the float64 oracle stepped on the same batches for the first ``exact_steps`` iterations (afterwards the two
  have taken different +-lr Adam steps and the float64 run says nothing about the float32 trajectory)."""

  def __init__(self, PG, PDo, PDi, gcfg, docfg, dicfg, exact_steps=1, **kw):
    self.o32 = oracle_trainer(PG, PDo, PDi, gcfg, docfg, dicfg, torch.float32, **kw)
    self.o64 = oracle_trainer(PG, PDo, PDi, gcfg, docfg, dicfg, torch.float64, **kw)
    self.exact_steps, self.steps = exact_steps, 0

  PG = property(lambda self: self.o32.PG)
  PDo = property(lambda self: self.o32.PDo)
  PDi = property(lambda self: self.o32.PDi)

  @property
  def training(self):
    return self.o32.training

  @training.setter
  def training(self, flag):
    self.o32.training = self.o64.training = flag

  def step(self, cpu_batch, noise=None):
    want = self.o32.step(cast_batch(cpu_batch, torch.float32), noise)
    if self.steps < self.exact_steps:
      self.o64.step(cast_batch(cpu_batch, torch.float64), None if noise is None else noise.double())
    self.steps += 1
    return want


def grad_parity_rows3(tr, o32, o64, scale=1.0):
  """rows (net, name, e_hip64, e_ref, e_hip32, max|g_f64|, cosine(hip, f64), numel, ndim, cosine(f32 oracle, f64)): errors relative to the float64
  gradient's max magnitude - of the HIP arena against float64, of the float32 oracle against float64, of HIP
  against the float32 oracle.  A parameter without a reference gradient must have an all-zero arena slot (its row carries
  max|g_f64| = NO_REFERENCE_GRAD and e_hip64 = max|g_hip|; check_grad_rows fails it unless that is exactly 0)."""
  rows = []
  for net, mod, P32, P64 in (('G', tr.model, o32.PG, o64.PG), ('Do', tr.d_obj, o32.PDo, o64.PDo),
                             ('Di', tr.d_img, o32.PDi, o64.PDi)):
    if mod is None:
      assert P64 is None, net
      continue
    for name, p in mod.named_parameters():
      got = p.grad.detach().cpu().double() * scale
      g64 = P64[name].grad
      if g64 is None:
        # the reference's optimiser SKIPS this parameter (no gradient at all): the arena slot must be exactly zero
        # (max|g_f64| = -1 marks the row; check_grad_rows tests e_hip64 == 0 for it)
        m = float(got.abs().max()) if got.numel() else 0.0
        if m != m:
          m = float('inf')
        rows.append((net, name, m, 0.0, m, NO_REFERENCE_GRAD, 1.0, got.numel(), got.dim(), 1.0))
        continue
      g32 = P32[name].grad.detach().double()
      g64 = g64.detach()
      assert got.shape == g64.shape, (net, name, tuple(got.shape), tuple(g64.shape))
      den = max(float(g64.abs().max()), 1e-30)
      nn_ = float(got.norm() * g64.norm())
      nr_ = float(g32.norm() * g64.norm())
      rows.append((net, name, float((got - g64).abs().max()) / den, float((g32 - g64).abs().max()) / den,
                   float((got - g32).abs().max()) / den, float(g64.abs().max()),
                   float((got * g64).sum()) / nn_ if nn_ > 0 else 1.0, got.numel(), got.dim(),
                   float((g32 * g64).sum()) / nr_ if nr_ > 0 else 1.0))
  return rows


def check_grad_rows(rows, rel=None, cos_min=None, vector_bound=None, cos_from_reference=False, strict=False, noise_factor=3.0):
  """-> (bad rows, summary dict).  rel None: the reference-arithmetic bound described above; a number: that
  bound on e_hip64 for every tensor (bf16) - with ``vector_bound`` = (rel, cos) a separate, looser pair for the
  one-dimensional parameters (biases, BatchNorm gamma / beta: sums of cancelling terms over a whole feature map),
  so that ``rel`` / ``cos_min`` can hold the matrices / filters / embedding tables to a bound that says something
  about magnitude.  Analytically-zero tensors (|g_f64| < 1e-6 everywhere: the bias of
  a convolution feeding a batch-statistics BatchNorm, parameters without a gradient) must be below 1e-6
  absolute on the HIP side.  strict: the fixed bound ``rel`` holds for EVERY tensor with a gradient, however few elements
  it has, and no flipped-decision escape (the tight form for steps without normalisation layers, VERDICT r5 item 5: with
  rel None the bound is max(1e-4, noise_factor x E_ref) - noise_factor 1: the kernels must be AT LEAST as close to the exact
  gradient as the reference's own fp32 arithmetic is)."""
  bad, summ = [], {}
  E_all = max([r[3] for r in rows if r[5] >= GRAD_ABS_ZERO] or [0.0])
  flip_ok = rel is None and not strict
  if rel is None:
    rel, cos_min = max(GRAD_REL, noise_factor * E_all), (GRAD_COS if cos_min is None else cos_min)
    if cos_from_reference:
      # The cosine bound from the reference arithmetic too (the bf16-operand emulation: its fp32 run is itself only
      # cos ~0.995 from its float64 run at small batches - rounding to bfloat16 turns fp32 noise into 0.4 % operand
      # changes): 3x the distance means 9x (1 - cos), since 1 - cos ~ angle^2 / 2.
      C_all = min([r[9] for r in rows if r[5] >= GRAD_ABS_ZERO] or [1.0])
      cos_min = min(cos_min, 1.0 - 9.0 * (1.0 - C_all))
  for net in ('G', 'Do', 'Di'):
    sel = [r for r in rows if r[0] == net]
    live = [r for r in sel if r[5] >= GRAD_ABS_ZERO]
    if not sel:
      continue
    for r in sel:
      if r[5] == NO_REFERENCE_GRAD:
        if r[2] != 0.0:              # anything but an all-zero slot would move a parameter the reference leaves alone
          bad.append(r)
        continue
      if r[5] < GRAD_ABS_ZERO:
        if r[2] * max(r[5], 1e-30) > GRAD_ABS_ZERO and r[4] * max(r[5], 1e-30) > GRAD_ABS_ZERO:
          bad.append(r)
        continue
      if r[7] < 16 and not flip_ok and not strict:
        # (fixed-bound mode, i.e. bf16: a scalar / few-element gradient - the bias of mask_net's 1-channel output
        # conv - is one sum of cancelling terms: its relative error is unbounded and its cosine says nothing;
        # it is held to 1 % of the largest gradient magnitude of its network instead)
        gmax = max(q[5] for q in sel)
        if r[2] * r[5] > 1e-2 * gmax:
          bad.append(r)
        continue
      r_rel, r_cos = (vector_bound if (vector_bound is not None and r[8] <= 1) else (rel, cos_min))
      # strict: a tensor behind a LeakyReLU whose pre-activation changed sign under the forward pass' 1e-6 differences
    # may exceed the bound - by at most 1e-3, and only with a cosine of six nines
    if (r[2] > r_rel or (r_cos is not None and r[6] < r_cos)) and not (flip_ok and r[6] >= 0.9999 and r[2] <= 0.1) and \
       not (strict and r[6] >= 0.999999 and r[2] <= 1e-3):
        bad.append(r)
    if live:
      w = max(live, key=lambda r: r[2])
      summ[net] = (w[2], w[1], max(r[3] for r in live), max(r[4] for r in live), min(r[6] for r in live))
      mats = [r for r in live if r[8] >= 2 and r[7] >= 16]
      if mats:                      # worst matrix / filter tensor on its own (the bf16 bound is stated per class)
        wm = max(mats, key=lambda r: r[2])
        summ[net + ':matrices'] = (wm[2], wm[1], min(r[6] for r in mats))
  return bad, summ


def assert_grad_parity(tr, refs, label, rel=None, cos_min=None, scale=1.0, vector_bound=None, cos_from_reference=False,
                       strict=False, noise_factor=3.0):
  """every parameter gradient of G / D_obj / D_img against the float64 oracle under the bound above (``rel`` /
  ``cos_min``: fixed bounds instead, for bf16).  Appends the measured worst cases to
  gpurun_out/grad_parity.log and returns (worst e_hip64, worst cosine)."""
  import os
  rows = grad_parity_rows3(tr, refs.o32, refs.o64, scale)
  bad, summ = check_grad_rows(rows, rel, cos_min, vector_bound, cos_from_reference, strict, noise_factor)
  line = '%-40s' % label + '  '.join(
    '%s: e_hip64 %.2e (%s) E_ref %.2e e_hip32 %.2e cos %.6f' % ((n,) + summ[n]) for n in ('G', 'Do', 'Di') if n in summ)
  if vector_bound is not None:
    line += '  | matrices: ' + '  '.join('%s %.2e (%s) cos %.6f' % ((n,) + summ[n + ':matrices'])
                                         for n in ('G', 'Do', 'Di') if n + ':matrices' in summ)
  print(line)
  try:
    os.makedirs('gpurun_out', exist_ok=True)
    with open(os.path.join('gpurun_out', 'grad_parity.log'), 'a') as f:
      f.write(line + '\n')
  except OSError:
    pass
  assert not bad, '%s: gradients out of tolerance:\n' % label + '\n'.join(
    '  %s.%s e_hip64 %.3e e_ref %.3e e_hip32 %.3e max|g| %.3e cos %.6f' % r[:7] for r in bad[:20])
  nets = [v for k, v in summ.items() if ':' not in k]
  return max(v[0] for v in nets), min(v[4] for v in nets)


def gain_reducer(like, gain=2.0, force=True):
  """Test instrumentation for the data-parallel exchange on ONE GPU (VERDICT r3 weak #1d; moved out of the product
  class in round 5): a 1-rank SUM is the identity, so a gradient slice that is reduced twice, never, or BEFORE its
  last writer ran would go unnoticed on a one-GPU box.  This reducer multiplies every tensor by ``gain`` right after
  its collective, on the stream that carries the collective (GradReducer._reduced), and reports grad_scale =
  1 / (world x gain): the arena x grad_scale equals the plain gradient exactly (gain a power of two) if and only if
  every element went through exactly one reduction after it was complete."""
  from sg2im_amd.distributed import GradReducer

  class GainReducer(GradReducer):
    @property
    def grad_scale(self):
      return 1.0 / (self.world_size * gain)

    def _reduced(self, tensor):
      tensor.mul_(gain)
  r = GainReducer(like.world_size, like.group, force=force, payload=like.payload)
  return r
