"""GAN / reconstruction losses (reference sg2im/losses.py, scripts/train.py:387-412) as
fused loss+gradient HIP kernels.  Means are layout independent, so scores / images may
be NCHW or NHWC as long as prediction and target agree."""
from . import functional as HF


def get_gan_losses(gan_type):
  """reference sg2im/losses.py:21-36"""
  if gan_type == 'gan':
    return gan_g_loss, gan_d_loss
  if gan_type == 'wgan':
    return wgan_g_loss, wgan_d_loss
  if gan_type == 'lsgan':
    return lsgan_g_loss, lsgan_d_loss
  raise ValueError('Unrecognized GAN type "%s"' % gan_type)


def bce_loss(input, target, weight=1.0, count=None):
  """Numerically stable BCE-with-logits against a constant target (reference
  sg2im/losses.py:39-57): mean(max(x,0) - x*t + log(1+exp(-|x|))).  ``weight`` (all losses here):
  the loss weight of scripts/train.py folded into the kernel instead of a separate multiply."""
  return HF.BceLogitsLoss.apply(input, float(target), float(weight), count)


def _sum(*terms):
  return HF.SumScalars.apply(*terms)


# ``count`` (every GAN loss below): None or (int32 device scalar, unit) - the scores of a padded object
# batch (sg2im_amd/bucketing.py): the mean runs over the first count * unit scores only.

def gan_g_loss(scores_fake, weight=1.0, count=None):
  return bce_loss(scores_fake, 1.0, weight, count)


def gan_d_loss(scores_real, scores_fake, count=None):
  return _sum(*gan_d_loss.terms(scores_real, scores_fake, count))


def _gan_d_terms(scores_real, scores_fake, count=None):
  """the addends of the discriminator loss, for a caller that sums them with further terms itself"""
  assert scores_real.size() == scores_fake.size()
  return bce_loss(scores_real, 1.0, 1.0, count), bce_loss(scores_fake, 0.0, 1.0, count)


gan_d_loss.terms = _gan_d_terms


def l1_loss(pred, target, weight=1.0):
  return HF.L1Loss.apply(pred, target, float(weight))


def mse_loss(pred, target, weight=1.0, count=None):
  return HF.MseLoss.apply(pred, target, float(weight), count)


def cross_entropy(scores, labels, weight=1.0, count=None):
  return HF.CrossEntropyLoss.apply(scores, labels, float(weight), count)


def wgan_g_loss(scores_fake, weight=1.0, count=None):
  """reference sg2im/losses.py:106-114: -mean(scores_fake)"""
  return HF.GanScoreLoss.apply(scores_fake, 1, -1.0, float(weight), count)


def wgan_d_loss(scores_real, scores_fake, count=None):
  """reference sg2im/losses.py:117-124: mean(fake) - mean(real)"""
  return _sum(*wgan_d_loss.terms(scores_real, scores_fake, count))


wgan_d_loss.terms = lambda scores_real, scores_fake, count=None: (
  HF.GanScoreLoss.apply(scores_fake, 1, 1.0, 1.0, count), HF.GanScoreLoss.apply(scores_real, 1, -1.0, 1.0, count))


def lsgan_g_loss(scores_fake, weight=1.0, count=None):
  """reference sg2im/losses.py:127-131: mse(sigmoid(fake), 1)"""
  return HF.GanScoreLoss.apply(scores_fake, 2, 1.0, float(weight), count)


def lsgan_d_loss(scores_real, scores_fake, count=None):
  """reference sg2im/losses.py:134-145"""
  assert scores_real.size() == scores_fake.size()
  return _sum(*lsgan_d_loss.terms(scores_real, scores_fake, count))


lsgan_d_loss.terms = lambda scores_real, scores_fake, count=None: (
  HF.GanScoreLoss.apply(scores_real, 2, 1.0, 1.0, count), HF.GanScoreLoss.apply(scores_fake, 2, 0.0, 1.0, count))


def binary_cross_entropy(prob, target, weight=1.0, count=None):
  """F.binary_cross_entropy on probabilities (mask loss of scripts/train.py:407-410)"""
  return HF.BceProbLoss.apply(prob, target, float(weight), count)
