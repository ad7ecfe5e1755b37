"""Scene layouts from object vectors, boxes and masks (reference sg2im/layout.py) on the
HIP path.  The functions keep the reference's signatures and NCHW result; the model uses
``layout_nhwc`` directly so no conversion happens inside a training step."""
from . import functional as HF

ALIGN_CORNERS = False    # what F.grid_sample does under torch >= 1.3 (SURVEY.md 8c caveat i)


def _num_images(obj_to_img, n_images):
  if n_images is not None:
    return int(n_images)
  return int(obj_to_img.max().item()) + 1        # reference sg2im/layout.py:143 (host sync)


def layout_nhwc(vecs, boxes, masks, obj_to_img, H, W=None, noise=None, n_images=None,
                align_corners=ALIGN_CORNERS, img_csr=None, pyramid_levels=0, link=None):
  """(N, H, W, D [+ noise channels]) NHWC layout; masks=None gives boxes_to_layout.
  link / pyramid_levels: see functional.LayoutFn (the hand-over to a refinement network that is the only consumer)."""
  W = H if W is None else W
  if masks is not None:
    O, M = masks.size(0), masks.size(1)
    assert masks.size() == (O, M, M)              # reference sg2im/layout.py:81
  return HF.LayoutFn.apply(vecs, boxes, masks, obj_to_img, noise, _num_images(obj_to_img, n_images), H, W,
                           int(align_corners), img_csr, int(pyramid_levels), link)


def boxes_to_layout(vecs, boxes, obj_to_img, H, W=None, pooling='sum', n_images=None,
                    align_corners=ALIGN_CORNERS):
  """reference sg2im/layout.py:30-63 -> (N, D, H, W)"""
  if pooling != 'sum':
    raise ValueError('Invalid pooling "%s"' % pooling)
  return HF.NhwcToNchw.apply(layout_nhwc(vecs, boxes, None, obj_to_img, H, W, n_images=n_images,
                                         align_corners=align_corners))


def masks_to_layout(vecs, boxes, masks, obj_to_img, H, W=None, pooling='sum', n_images=None,
                    align_corners=ALIGN_CORNERS):
  """reference sg2im/layout.py:66-91 -> (N, D, H, W)"""
  if pooling != 'sum':
    raise ValueError('Invalid pooling "%s"' % pooling)
  return HF.NhwcToNchw.apply(layout_nhwc(vecs, boxes, masks, obj_to_img, H, W, n_images=n_images,
                                         align_corners=align_corners))
