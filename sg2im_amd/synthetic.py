"""Seeded synthetic scene-graph batches in the reference's collate layout.

The flat batch contract is the one ``coco_collate_fn`` / ``vg_collate_fn`` emit
(reference sg2im/data/coco.py:376-419, sg2im/data/vg.py:144-186; per-image graph
shape coco.py:286-359, vg.py:96-141): objects of all images concatenated, the last
object of every image is ``__image__`` (category 0, box [0,0,1,1], all-ones mask),
triples are (s, p, o) with s/o already offset into the flat object axis, each image
contributes its relationship triples first and then one ``__in_image__`` (p = 0)
triple per real object, ``obj_to_img`` / ``triple_to_img`` are non-decreasing.

Shapes follow SURVEY.md section 8d.  All tensors are created on the CPU with a
``torch.Generator`` seeded by ``seed`` so the CPU oracle and every GPU rank
(``seed + rank``) can build bit-identical inputs.
"""
import torch


def make_vocab(num_objs, num_preds):
  """Vocab dict with the keys Sg2ImModel reads (reference sg2im/model.py:47-48):
  object 0 is ``__image__`` and predicate 0 is ``__in_image__``
  (reference scripts/preprocess_vg.py:231,335; sg2im/data/coco.py:181-205)."""
  objs = ['__image__'] + ['obj%d' % i for i in range(1, num_objs)]
  preds = ['__in_image__'] + ['pred%d' % i for i in range(1, num_preds)]
  return {
    'object_idx_to_name': objs,
    'object_name_to_idx': {n: i for i, n in enumerate(objs)},
    'pred_idx_to_name': preds,
    'pred_name_to_idx': {n: i for i, n in enumerate(preds)},
  }


def synthetic_batch(batch_size, image_size=(64, 64), num_objs=184, num_preds=7,
                    min_objs=3, max_objs=8, mask_size=16, style='coco',
                    extra_rels=6, seed=0):
  """Returns ``(imgs, objs, boxes, masks, triples, obj_to_img, triple_to_img)``;
  ``masks`` is None for ``style='vg'`` (the 6-tuple of vg_collate_fn, reference
  scripts/train.py:516-519).

  coco: k ~ U{min_objs..max_objs} real objects per image, k random relationship
        triples (one per real object, partner != self, p ~ U{1..P-1}) followed by
        k ``__in_image__`` triples; Bernoulli(.5) int64 masks as GT.
  vg:   k real objects, r ~ U{1..k+extra_rels} random relationships, then k
        ``__in_image__`` triples; no masks.
  """
  g = torch.Generator().manual_seed(seed)
  H, W = image_size
  imgs = torch.randn(batch_size, 3, H, W, generator=g)
  all_objs, all_boxes, all_masks, all_triples = [], [], [], []
  obj_to_img, triple_to_img = [], []
  off = 0
  for n in range(batch_size):
    k = int(torch.randint(min_objs, max_objs + 1, (1,), generator=g))
    cats = torch.randint(1, num_objs, (k,), generator=g)
    objs = torch.cat([cats, torch.zeros(1, dtype=torch.long)])
    xy0 = torch.rand(k, 2, generator=g) * 0.6
    wh = 0.05 + torch.rand(k, 2, generator=g) * 0.35
    xy1 = (xy0 + wh).clamp(max=1.0)
    boxes = torch.cat([torch.cat([xy0, xy1], 1), torch.tensor([[0., 0., 1., 1.]])])
    O = k + 1
    trip = []
    n_rel = k if style == 'coco' else int(torch.randint(1, k + extra_rels + 1, (1,), generator=g))
    for r in range(n_rel):
      s = r % k if style == 'coco' else int(torch.randint(0, k, (1,), generator=g))
      o = int(torch.randint(0, k - 1, (1,), generator=g))
      o = o + 1 if o >= s else o            # partner != self
      if style == 'coco' and float(torch.rand(1, generator=g)) > 0.5:
        s, o = o, s
      p = int(torch.randint(1, num_preds, (1,), generator=g))
      trip.append([s + off, p, o + off])
    for i in range(k):
      trip.append([i + off, 0, O - 1 + off])
    trip = torch.tensor(trip, dtype=torch.long)
    if style == 'coco':
      m = (torch.rand(k, mask_size, mask_size, generator=g) > 0.5).long()
      m = torch.cat([m, torch.ones(1, mask_size, mask_size, dtype=torch.long)])
      all_masks.append(m)
    all_objs.append(objs)
    all_boxes.append(boxes)
    all_triples.append(trip)
    obj_to_img.append(torch.full((O,), n, dtype=torch.long))
    triple_to_img.append(torch.full((trip.size(0),), n, dtype=torch.long))
    off += O
  masks = torch.cat(all_masks) if style == 'coco' else None
  return (imgs, torch.cat(all_objs), torch.cat(all_boxes), masks, torch.cat(all_triples),
          torch.cat(obj_to_img), torch.cat(triple_to_img))


def shard_batch(batch, rank, world_size):
  """Data-parallel shard of a collated batch: rank r takes a
  contiguous block of N / world_size whole images with their objects / triples
  re-based to a local flat index, i.e. exactly what the collate function would
  have produced for those images alone."""
  imgs, objs, boxes, masks, triples, obj_to_img, triple_to_img = batch
  N = imgs.size(0)
  assert N % world_size == 0, 'batch must split evenly across ranks'
  per = N // world_size
  lo, hi = rank * per, (rank + 1) * per
  osel = (obj_to_img >= lo) & (obj_to_img < hi)
  tsel = (triple_to_img >= lo) & (triple_to_img < hi)
  obase = int(osel.nonzero()[0]) if bool(osel.any()) else 0
  tri = triples[tsel].clone()
  tri[:, 0] -= obase
  tri[:, 2] -= obase
  return (imgs[lo:hi].contiguous(), objs[osel], boxes[osel],
          None if masks is None else masks[osel], tri,
          obj_to_img[osel] - lo, triple_to_img[tsel] - lo)
