"""Shape buckets for the collated scene-graph batch.

Every batch the reference's loaders emit has its own object / triple count (3-8 objects per
COCO image, reference sg2im/data/coco.py:286-359; scripts/train.py:508-514 just feeds whatever
``coco_collate_fn`` concatenated), while a captured hipGraph bakes tensor shapes and addresses
in.  The Trainer therefore pads the object axis O and the triple axis T up to a bucket size
and replays one graph per bucket; the true sizes travel in device memory (``counts``), which
a replay re-reads.

Padding is constructed so that it is *exactly* neutral:

  * dummy objects: category 0, box (2, 2, 3, 3) - outside the unit square, so every bilinear
    sample of their layout footprint and of their crop is an out-of-range zero
    (sg2im_layout_forward / sg2im_crop_forward add exactly +0 for them and read no pixel);
    they belong to the last image, which keeps ``obj_to_img`` sorted;
  * dummy triples: (O_pad-1, 0, O_pad-1) - they only touch the last dummy object, so the
    pooled vectors (and avg-pool counts) of real objects are bit-identical;
  * every cross-row reduction takes the true row count from ``counts``: BatchNorm statistics
    and their backward in D_obj / mask_net, the means of the box / score / classification /
    predicate / mask losses (sg2im_bn_stats, sg2im_bn_act_backward and the loss kernels'
    ``count`` argument).  Padding rows receive a zero loss gradient there, hence zero
    gradients everywhere upstream (row-wise GEMMs map zero rows to zero rows, and a weight
    gradient sums x^T dy over rows with dy = 0).

Only the path with ground-truth boxes laid out (what scripts/train.py:526 does) is supported:
predicted boxes of dummy objects would not stay outside the image.
"""
import torch


def round_up(n, m):
  return (int(n) + m - 1) // m * m


class Bucketer(object):
  """(O, T) -> (O_pad, T_pad).  At least one dummy object always exists (dummy triples need it)."""

  def __init__(self, obj_multiple=32, triple_multiple=64):
    self.obj_multiple, self.triple_multiple = int(obj_multiple), int(triple_multiple)

  def bucket(self, num_objs, num_triples):
    return (round_up(num_objs + 1, self.obj_multiple), round_up(max(num_triples, 1), self.triple_multiple))


FAR_BOX = (2.0, 2.0, 3.0, 3.0)


def pad_batch(batch, o_pad, t_pad):
  """Functional form (host logic + tests): returns the padded 6-tuple and an int32 ``counts``
  tensor [O, T] on the batch's device."""
  imgs, objs, boxes, masks, triples, obj_to_img = batch[:6]
  O, T, N = objs.numel(), triples.size(0), imgs.size(0)
  if o_pad <= O or t_pad < T:
    raise ValueError('bucket (%d, %d) too small for O=%d, T=%d' % (o_pad, t_pad, O, T))
  dev = objs.device
  po, pt = o_pad - O, t_pad - T
  objs_p = torch.cat([objs, torch.zeros(po, dtype=objs.dtype, device=dev)])
  boxes_p = torch.cat([boxes, torch.tensor(FAR_BOX, dtype=boxes.dtype, device=dev).expand(po, 4)])
  masks_p = None
  if masks is not None:
    masks_p = torch.cat([masks, torch.zeros((po,) + tuple(masks.shape[1:]), dtype=masks.dtype, device=dev)])
  dummy = torch.tensor([o_pad - 1, 0, o_pad - 1], dtype=triples.dtype, device=dev).expand(pt, 3)
  triples_p = torch.cat([triples.reshape(T, 3), dummy])
  o2i_p = torch.cat([obj_to_img, torch.full((po,), N - 1, dtype=obj_to_img.dtype, device=dev)])
  counts = torch.tensor([O, T], dtype=torch.int32, device=dev)
  return (imgs, objs_p, boxes_p, masks_p, triples_p, o2i_p), counts


class StaticBatch(object):
  """The static input buffers one captured graph reads: allocated once per bucket, refilled in
  place from every batch of that bucket (a handful of tiny copy / fill launches)."""

  def __init__(self, batch, o_pad, t_pad):
    (imgs, objs, boxes, masks, triples, o2i), counts = pad_batch(batch, o_pad, t_pad)
    self.o_pad, self.t_pad, self.n_images = o_pad, t_pad, imgs.size(0)
    self.imgs, self.objs, self.boxes, self.masks = imgs.clone(), objs, boxes, masks
    self.triples, self.obj_to_img = triples.contiguous(), o2i
    self.counts = counts
    self.obj_count = (self.counts[0:1], 1)
    self.triple_count = (self.counts[1:2], 1)
    dev = objs.device
    # Constant padding rows (whole-bucket tensors: load() copies their [O:] / [T:] tails) and lookup
    # tables for the two counts - so that a refill is three multi-tensor copies, one per dtype
    # (torch._foreach_copy_), instead of ~15 copy / fill launches in front of every replay.  [These are
    # torch's kernels on purpose: an eager launch of THIS library between two replays would invalidate
    # the captured graphs, see Trainer._graph_step.]
    self._pad_objs = torch.zeros_like(self.objs)
    self._pad_boxes = torch.tensor(FAR_BOX, dtype=boxes.dtype, device=dev).expand(o_pad, 4).contiguous()
    self._pad_masks = None if masks is None else torch.zeros_like(self.masks)
    self._pad_triples = torch.tensor([o_pad - 1, 0, o_pad - 1], dtype=triples.dtype, device=dev).expand(t_pad, 3).contiguous()
    self._pad_o2i = torch.full_like(self.obj_to_img, self.n_images - 1)
    self._tab = torch.arange(max(o_pad, t_pad) + 1, dtype=torch.int32, device=dev)

  def tensors(self):
    return (self.imgs, self.objs, self.boxes, self.masks, self.triples, self.obj_to_img)

  def load(self, batch):
    imgs, objs, boxes, masks, triples, o2i = batch[:6]
    O, T = objs.numel(), triples.size(0)
    if O >= self.o_pad or T > self.t_pad or imgs.shape != self.imgs.shape or (masks is None) != (self.masks is None):
      raise ValueError('batch does not fit this bucket')
    groups = {}
    def put(dst, src):
      if dst.numel() > 0:
        d, s_ = groups.setdefault(dst.dtype, ([], []))
        d.append(dst); s_.append(src)
    put(self.imgs, imgs)
    put(self.objs[:O], objs); put(self.objs[O:], self._pad_objs[O:])
    put(self.boxes[:O], boxes); put(self.boxes[O:], self._pad_boxes[O:])
    if masks is not None:
      put(self.masks[:O], masks); put(self.masks[O:], self._pad_masks[O:])
    put(self.triples[:T], triples.reshape(T, 3)); put(self.triples[T:], self._pad_triples[T:])
    put(self.obj_to_img[:O], o2i); put(self.obj_to_img[O:], self._pad_o2i[O:])
    put(self.counts[0:1], self._tab[O:O + 1]); put(self.counts[1:2], self._tab[T:T + 1])
    if _stage_with_library(groups):
      return
    for dst, src in groups.values():
      torch._foreach_copy_(dst, src)


def _stage_with_library(groups):
  """all copies of a refill as ONE launch of the library (sg2im_stage_batch) instead of three multi-tensor
  copies (~13 torch launches, ~70 us at the head of every replayed step).  The launch is issued straight
  through the binding, NOT through ``_lib.call``: it is part of every training step, so it must not count as
  "eager use" that re-captures the graphs (Trainer._graph_step)."""
  import ctypes
  import os
  if os.environ.get('SG2IM_STAGE', '1') == '0':         # (A/B knob)
    return False
  from . import _lib
  pairs = [(d, s) for dst, src in groups.values() for d, s in zip(dst, src)]
  n = len(pairs)
  if n == 0 or n > 16:
    return n == 0
  for d, s in pairs:
    if not (d.is_cuda and s.is_cuda and d.is_contiguous() and s.is_contiguous() and d.dtype == s.dtype and
            d.numel() == s.numel() and (d.numel() * d.element_size()) % 4 == 0):
      return False
  if not _lib._inited:
    _lib.init()
  P = ctypes.c_void_p * n
  Z = ctypes.c_size_t * n
  rc = _lib.load().sg2im_stage_batch(n, P(*[d.data_ptr() for d, _ in pairs]), P(*[s.data_ptr() for _, s in pairs]),
                                     Z(*[d.numel() * d.element_size() for d, _ in pairs]),
                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
  if rc != 0:
    raise _lib.Sg2imHipError('sg2im_stage_batch failed (%d)' % rc)
  return True
