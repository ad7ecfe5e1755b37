"""Shared helpers for the test-suite (fixture loading, tolerant comparison)."""
import os

import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
GOLDEN_NAMES = ('tiny_coco', 'tiny_vg')
GOLDEN_TRAIN_NAMES = GOLDEN_NAMES + ('tiny_coco_nonorm', 'tiny_coco_mlpbn', 'tiny_coco_instnorm', 'tiny_coco_archtokens')     # (no eval-mode fixtures for these)


def load_golden(name):
  return torch.load(os.path.join(GOLDEN_DIR, name + '.pt'), weights_only=False)


def clone_params(sd):
  return {k: v.clone() for k, v in sd.items()}


def max_rel_err(a, b):
  """max |a-b| / max(|b|max, tiny): scale-aware error for whole-tensor comparison."""
  a, b = a.double(), b.double()
  return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def assert_close(a, b, rtol, atol, what=''):
  a, b = a.detach().cpu(), b.detach().cpu()
  assert a.shape == b.shape, '%s: shape %s vs %s' % (what, tuple(a.shape), tuple(b.shape))
  diff = (a.double() - b.double()).abs()
  tol = atol + rtol * b.double().abs()
  bad = diff > tol
  if bool(bad.any()):
    i = int(torch.argmax(diff - tol))
    raise AssertionError('%s: %d/%d elements out of tolerance (rtol=%g atol=%g); worst |diff|=%.3e at flat %d (got %.6e want %.6e); max|want|=%.3e'
                         % (what, int(bad.sum()), bad.numel(), rtol, atol, float(diff.flatten()[i]), i,
                            float(a.flatten()[i]), float(b.flatten()[i]), float(b.abs().max())))
