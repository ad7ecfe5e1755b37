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
