"""Differentiable bilinear object crops (reference sg2im/bilinear.py) on the HIP path."""
from . import functional as HF
from .layout import ALIGN_CORNERS


def crop_bbox_batch_nhwc(feats_nhwc, bbox, bbox_to_feats, HH, WW=None, align_corners=ALIGN_CORNERS):
  if WW is not None and WW != HH:
    raise NotImplementedError('only square crops are on the HIP path')
  return HF.CropFn.apply(feats_nhwc, bbox, bbox_to_feats, HH, int(align_corners))


def crop_bbox_batch(feats, bbox, bbox_to_feats, HH, WW=None, backend='cudnn', align_corners=ALIGN_CORNERS):
  """reference sg2im/bilinear.py:28-62: feats (N,C,H,W) -> crops (B,C,HH,WW).  One kernel
  reads image ``bbox_to_feats[b]`` directly instead of the per-image Python loop with an
  expand().contiguous() copy per object (bilinear.py:76-87)."""
  if backend != 'cudnn':
    raise NotImplementedError('only the grid_sample ("cudnn") crop backend is on the HIP path')
  crops = crop_bbox_batch_nhwc(HF.NchwToNhwc.apply(feats), bbox, bbox_to_feats, HH, WW, align_corners)
  return HF.NhwcToNchw.apply(crops)
