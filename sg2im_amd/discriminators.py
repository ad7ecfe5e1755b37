"""Image / object PatchGAN discriminators (reference sg2im/discriminators.py) on HIP."""
import os

import torch
import torch.nn as nn

from . import functional as HF
from . import ops
from .bilinear import crop_bbox_batch_nhwc
from .layers import DiscCnn, GlobalAvgPool, build_cnn
from .layout import ALIGN_CORNERS

TWO_HEADS = os.environ.get('SG2IM_TWO_HEADS', '1') != '0'      # (A/B knob)


class PatchDiscriminator(nn.Module):
  def __init__(self, arch, normalization='batch', activation='leakyrelu-0.2', padding='same', pooling='avg',
               input_size=(128, 128), layout_dim=0):
    super(PatchDiscriminator, self).__init__()
    input_dim = 3 + layout_dim
    self.cnn, output_dim = build_cnn(arch='I%d,%s' % (input_dim, arch), normalization=normalization,
                                     activation=activation, pooling=pooling, padding=padding)
    # present in the state_dict but never applied (reference sg2im/discriminators.py:40-45)
    self.classifier = nn.Conv2d(output_dim, 1, kernel_size=1, stride=1)

  def forward_nhwc(self, x_nhwc, share=None):
    """share: None or a functional.SharedPass (only the 'C'-token architectures can share a pass)"""
    if share is not None and isinstance(self.cnn, DiscCnn):
      return self.cnn(x_nhwc, share=share)       # (a recorded pass is adopted; x is then only checked against the recording)
    return self.cnn(x_nhwc)

  def forward(self, x, layout=None):
    if layout is not None:
      x = torch.cat([x, layout], dim=1)
    return HF.NhwcToNchw.apply(self.cnn(HF.NchwToNhwc.apply(x)))


class AcDiscriminator(nn.Module):
  def __init__(self, vocab, arch, normalization='none', activation='relu', padding='same', pooling='avg'):
    super(AcDiscriminator, self).__init__()
    self.vocab = vocab
    cnn, D = build_cnn(arch=arch, normalization=normalization, activation=activation, pooling=pooling,
                       padding=padding)
    self.cnn = nn.Sequential(cnn, GlobalAvgPool(), nn.Linear(D, 1024))
    num_objects = len(vocab['object_idx_to_name'])
    self.real_classifier = nn.Linear(1024, 1)
    self.obj_classifier = nn.Linear(1024, num_objects)

  def scores_nhwc(self, x_nhwc, count=None, share=None):
    if share is not None and isinstance(self.cnn[0], DiscCnn):
      feats = self.cnn[0](x_nhwc, count, share)
    else:
      feats = self.cnn[0](x_nhwc, count) if count is not None else self.cnn[0](x_nhwc)
    vecs = self.cnn[1](feats)
    fc = self.cnn[2]
    vecs = HF.LinearAct.apply(vecs, fc.weight, fc.bias, 1.0)
    rc, oc = self.real_classifier, self.obj_classifier
    if TWO_HEADS and ops.two_heads_supported(vecs.size(1), rc.weight.size(0), oc.weight.size(0)):
      return HF.TwoHeads.apply(vecs, rc.weight, rc.bias, oc.weight, oc.bias)     # (both heads, one launch)
    real = HF.LinearAct.apply(vecs, rc.weight, rc.bias, 1.0)
    cls = HF.LinearAct.apply(vecs, oc.weight, oc.bias, 1.0)
    return real, cls

  def forward_nhwc(self, x_nhwc, y, ac_weight=1.0, count=None, share=None):
    real, cls = self.scores_nhwc(x_nhwc, count, share)
    return real, HF.CrossEntropyLoss.apply(cls, y, float(ac_weight), count)   # reference sg2im/discriminators.py:74

  def forward(self, x, y):
    if x.dim() == 3:
      x = x[:, None]
    return self.forward_nhwc(HF.NchwToNhwc.apply(x), y)


class AcCropDiscriminator(nn.Module):
  def __init__(self, vocab, arch, normalization='none', activation='relu', object_size=64, padding='same',
               pooling='avg', align_corners=ALIGN_CORNERS):
    """align_corners (not a reference argument): sampling convention of the crops
    (reference sg2im/bilinear.py:132), see Sg2ImModel"""
    super(AcCropDiscriminator, self).__init__()
    self.vocab = vocab
    self.discriminator = AcDiscriminator(vocab, arch, normalization, activation, padding, pooling)
    self.object_size = object_size
    self.align_corners = bool(align_corners)

  def forward_nhwc(self, imgs_nhwc, objs, boxes, obj_to_img, ac_weight=1.0, obj_count=None, share=None):
    """ac_weight: loss weight folded into the classification loss (the Trainer's ac_loss_weight);
    obj_count: (int32 device scalar, 1) when the object axis is padded (sg2im_amd/bucketing.py);
    share: None or a functional.SharedPass - the crops and the CNN over them are computed by the first call that
    gets it and adopted by the second (same images, boxes and weights: scripts/train.py:544 and :566-568)"""
    if share is not None and not isinstance(self.discriminator.cnn[0], DiscCnn):
      share = None                     # (architectures with R / U / P / FC tokens run layer by layer)
    if share is not None and share.recorded:
      crops = None
    else:
      crops = crop_bbox_batch_nhwc(imgs_nhwc, boxes, obj_to_img, self.object_size,
                                   align_corners=self.align_corners)
    return self.discriminator.forward_nhwc(crops, objs, ac_weight, obj_count, share)

  def forward(self, imgs, objs, boxes, obj_to_img):
    """imgs (N,3,H,W) -> (real_scores (O,1), ac_loss scalar)  (reference :87-90)"""
    return self.forward_nhwc(HF.NchwToNhwc.apply(imgs), objs, boxes, obj_to_img)
