"""GAN / reconstruction losses (reference sg2im/losses.py, scripts/train.py:387-412) as
fused loss+gradient HIP kernels.  Means are layout independent, so scores / images may
be NCHW or NHWC as long as prediction and target agree."""
from . import functional as HF


def get_gan_losses(gan_type):
  """reference sg2im/losses.py:21-36"""
  if gan_type == 'gan':
    return gan_g_loss, gan_d_loss
  if gan_type in ('wgan', 'lsgan'):
    raise NotImplementedError('"%s" GAN losses are not on the HIP path yet (SURVEY.md 8f rank 3)' % gan_type)
  raise ValueError('Unrecognized GAN type "%s"' % gan_type)


def bce_loss(input, target):
  """Numerically stable BCE-with-logits against a constant target (reference
  sg2im/losses.py:39-57): mean(max(x,0) - x*t + log(1+exp(-|x|)))."""
  return HF.BceLogitsLoss.apply(input, float(target), 1.0)


def gan_g_loss(scores_fake):
  return bce_loss(scores_fake, 1.0)


def gan_d_loss(scores_real, scores_fake):
  assert scores_real.size() == scores_fake.size()
  return bce_loss(scores_real, 1.0) + bce_loss(scores_fake, 0.0)


def l1_loss(pred, target, weight=1.0):
  return HF.L1Loss.apply(pred, target, float(weight))


def mse_loss(pred, target, weight=1.0):
  return HF.MseLoss.apply(pred, target, float(weight))


def cross_entropy(scores, labels, weight=1.0):
  return HF.CrossEntropyLoss.apply(scores, labels, float(weight))
