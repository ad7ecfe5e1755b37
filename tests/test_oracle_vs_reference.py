"""CPU, build container only (skipped where /root/reference is absent): the oracle against
the LIVE imported reference modules at the reference's default architecture."""
import contextlib
import io
import os
import sys

import pytest
import torch

from oracle import sg2im_oracle as orc
from sg2im_amd.synthetic import make_vocab, synthetic_batch
from tests.util import assert_close

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree not available')


def _ref():
  sys.dont_write_bytecode = True
  if REF not in sys.path:
    sys.path.insert(0, REF)
  from sg2im.model import Sg2ImModel
  from sg2im.discriminators import AcCropDiscriminator, PatchDiscriminator
  return Sg2ImModel, AcCropDiscriminator, PatchDiscriminator


def test_default_architecture_forward_and_input_gradients():
  Sg2ImModel, AcCrop, Patch = _ref()
  from sg2im_amd.trainer import GENERATOR_DEFAULTS, D_OBJ_DEFAULTS, D_IMG_DEFAULTS
  vocab = make_vocab(184, 7)
  torch.manual_seed(0)
  G = Sg2ImModel(vocab, **GENERATOR_DEFAULTS).train()
  with contextlib.redirect_stdout(io.StringIO()):
    Do = AcCrop(vocab, **D_OBJ_DEFAULTS).train()
    Di = Patch(**D_IMG_DEFAULTS).train()
  imgs, objs, boxes, masks, triples, o2i, _ = synthetic_batch(2, seed=5)
  noise = torch.randn(2, 32, 64, 64)
  real = torch.randn
  torch.randn = lambda *a, **k: noise.clone()
  try:
    want = G(objs, triples, o2i, boxes_gt=boxes, masks_gt=masks)
  finally:
    torch.randn = real
  P = {k: v.clone() for k, v in G.state_dict().items()}
  # state_dict was cloned AFTER the forward: rewind the BN statistics the forward advanced
  for k in P:
    if k.endswith('running_mean'):
      P[k].zero_()
    elif k.endswith('running_var'):
      P[k].fill_(1.0)
  got = orc.generator_forward(P, dict(GENERATOR_DEFAULTS, vocab=vocab), objs, triples, o2i, boxes, masks, noise, True)
  for a, b, name in zip(got, want, ('img', 'boxes', 'masks', 'rel')):
    assert_close(a, b, 2e-5, 2e-5, name)
  ip = want[0].detach()
  sr, ac = Do(ip, objs, boxes, o2i)
  PD = {k: v.clone() for k, v in Do.state_dict().items()}
  for k in PD:
    if k.endswith('running_mean'):
      PD[k].zero_()
    elif k.endswith('running_var'):
      PD[k].fill_(1.0)
  sr2, ac2 = orc.ac_crop_discriminator(PD, dict(D_OBJ_DEFAULTS, vocab=vocab), ip, objs, boxes, o2i)
  assert_close(sr2, sr, 2e-5, 2e-5, 'd_obj scores')
  assert abs(float(ac2) - float(ac)) < 1e-4
  PI = {k: v.clone() for k, v in Di.state_dict().items()}
  assert_close(orc.patch_discriminator(PI, dict(D_IMG_DEFAULTS), ip), Di(ip), 2e-5, 2e-5, 'd_img scores')


def test_build_cnn_state_dict_matches_reference_for_random_arch_strings():
  """drop-in check for arbitrary --d_obj_arch / --d_img_arch strings: the module list (and so the
  checkpoint keys and shapes) of sg2im_amd.layers.build_cnn and of the oracle's parameter walker equal
  the reference's build_cnn (layers.py:129-213) for random token sequences and every normalization"""
  import random
  _ref()
  from sg2im.layers import build_cnn as ref_build_cnn
  from sg2im_amd.layers import build_cnn
  rng = random.Random(7)
  for trial in range(40):
    toks, flat = [], False
    if rng.random() < 0.5:
      toks.append('I%d' % rng.choice([1, 3, 4]))
    cin = int(toks[0][1:]) if toks else 3
    for _ in range(rng.randint(1, 6)):
      kind = rng.choice(['C', 'C', 'C', 'R', 'U', 'P'])
      if kind == 'C':
        k, c = rng.choice([1, 3, 5]), rng.choice([4, 8, 12])
        toks.append('C%d-%d' % (k, c) if rng.random() < 0.5 else 'C%d-%d-%d' % (k, c, rng.choice([1, 2])))
      elif kind == 'R':
        toks.append('R')
      else:
        toks.append('%s%d' % (kind, rng.choice([2, 3])))
    for _ in range(rng.randint(0, 2)):
      toks.append('FC-%d-%d' % (rng.choice([16, 32]), rng.choice([8, 10])))
    arch = ','.join(toks)
    norm = rng.choice(['batch', 'instance', 'none'])
    pool = rng.choice(['max', 'avg'])
    with contextlib.redirect_stdout(io.StringIO()):
      ref, ref_c = ref_build_cnn(arch, normalization=norm, activation='leakyrelu-0.2', padding='same', pooling=pool)
    mine, my_c = build_cnn(arch, normalization=norm, activation='leakyrelu-0.2', padding='same', pooling=pool)
    assert my_c == ref_c, arch
    want = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    got = {k: tuple(v.shape) for k, v in mine.state_dict().items()}
    assert got == want, (arch, norm)
    assert [type(m).__name__ for m in mine] == [type(m).__name__ for m in ref], (arch, norm)
    P = {}
    body = arch if arch.startswith('I') else 'I%d,%s' % (cin, arch)
    assert orc._init_disc_cnn(P, 'cnn', body, cin, torch.Generator().manual_seed(0), False, norm) == ref_c
    assert {('cnn.' + k): s for k, s in want.items()} == {k: tuple(v.shape) for k, v in P.items()}, (arch, norm)


def test_oracle_cnn_forward_matches_reference_for_arch_tokens():
  """numerical check of the oracle's build_cnn restatement (R / U / P / FC, all normalizations, including
  the doubled BatchNorm running-statistics update inside ResidualBlock) against the live reference"""
  _ref()
  from sg2im.layers import build_cnn as ref_build_cnn
  cases = [('I3,C3-8,R,P2,C3-12-2,U2,R', 'batch', 'avg'), ('I3,R,C3-8-2,P2,FC-128-16,FC-16-4', 'batch', 'max'),
           ('I4,C5-8,P3,R,U3,C1-6', 'instance', 'max'), ('I3,C3-8,R,R,P2', 'none', 'avg')]
  for arch, norm, pool in cases:
    torch.manual_seed(11)
    with contextlib.redirect_stdout(io.StringIO()):
      ref, _ = ref_build_cnn(arch, normalization=norm, activation='leakyrelu-0.2', padding='same', pooling=pool)
    ref.train()
    P = {'cnn.' + k: v.clone() for k, v in ref.state_dict().items()}
    x = torch.randn(4, int(arch.split(',')[0][1:]), 16, 16)
    want = ref(x)
    got = orc.disc_cnn(P, 'cnn', x, arch, 0.2, 'same', True, norm, pool)
    assert_close(got, want, 2e-5, 2e-5, arch)
    for k, v in ref.state_dict().items():          # running statistics after ONE forward (moved twice in R)
      assert_close(P['cnn.' + k].float(), v.float(), 2e-5, 2e-5, arch + ' ' + k)


def test_oracle_align_corners_true_matches_reference_under_torch04_sampling():
  """The authors trained under torch 0.4, whose F.grid_sample had today's align_corners=True
  semantics (SURVEY.md section 8c caveat i).  The reference calls F.grid_sample without the argument
  (sg2im/layout.py:53,88, sg2im/bilinear.py:132); forcing the old semantics into those calls gives the
  pin for the oracle's ``align_corners=True`` mode: layouts (masks, boxes-only), crops, and their
  gradients."""
  _ref()
  import torch.nn.functional as F
  import sg2im.layout as ref_layout
  import sg2im.bilinear as ref_bilinear
  imgs, objs, boxes, masks, triples, o2i, _ = synthetic_batch(3, seed=12)
  g = torch.Generator().manual_seed(2)
  vecs = torch.randn(objs.numel(), 24, generator=g)
  soft = torch.rand(objs.numel(), 16, 16, generator=g)
  real = F.grid_sample
  patched = lambda inp, grid, *a, **k: real(inp, grid, *a, **dict(k, align_corners=True))
  for mode in (True, False):
    F.grid_sample = patched if mode else real
    try:
      vr, br, mr = vecs.clone().requires_grad_(True), boxes.clone().requires_grad_(True), soft.clone().requires_grad_(True)
      want_m = ref_layout.masks_to_layout(vr, br, mr, o2i, 32)
      want_b = ref_layout.boxes_to_layout(vr, br, o2i, 32)
      ir = imgs.clone().requires_grad_(True)
      want_c = ref_bilinear.crop_bbox_batch(ir, boxes, o2i, 16)
      (want_m.sum() * 0.5 + want_b.square().sum() + want_c.square().sum()).backward()
    finally:
      F.grid_sample = real
    vo, bo, mo = vecs.clone().requires_grad_(True), boxes.clone().requires_grad_(True), soft.clone().requires_grad_(True)
    io_ = imgs.clone().requires_grad_(True)
    got_m = orc.masks_to_layout(vo, bo, mo, o2i, 32, align_corners=mode)
    got_b = orc.boxes_to_layout(vo, bo, o2i, 32, align_corners=mode)
    got_c = orc.crop_bbox_batch(io_, boxes, o2i, 16, align_corners=mode)
    (got_m.sum() * 0.5 + got_b.square().sum() + got_c.square().sum()).backward()
    for a, b, name in ((got_m, want_m, 'masks_to_layout'), (got_b, want_b, 'boxes_to_layout'), (got_c, want_c, 'crops'),
                       (vo.grad, vr.grad, 'd_vecs'), (bo.grad, br.grad, 'd_boxes'), (mo.grad, mr.grad, 'd_masks'),
                       (io_.grad, ir.grad, 'd_imgs')):
      assert_close(a, b, 2e-5, 2e-5, '%s (align_corners=%s)' % (name, mode))
  # the two conventions really differ (an identity-box crop is the identity only under the old one)
  eye = torch.tensor([[0., 0., 1., 1.]])
  one = torch.zeros(1, dtype=torch.long)
  x = torch.randn(1, 3, 8, 8, generator=g)
  assert torch.allclose(orc.crop_bbox_batch(x, eye, one, 8, align_corners=True), x, atol=1e-6)
  assert not torch.allclose(orc.crop_bbox_batch(x, eye, one, 8, align_corners=False), x, atol=1e-3)
