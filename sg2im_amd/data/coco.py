"""COCO / COCO-Stuff scene-graph dataset (reference sg2im/data/coco.py:32-419): object annotations become a scene
graph on the fly - one `__image__` object per image, one random spatial relationship per real object, an
`__in_image__` triple per object.  Same constructor arguments, vocabulary layout, item tuple and collate contract
as the reference, so `scripts/train.py --dataset coco` consumes either; see data/__init__.py for what is and is not
pinned."""
import json
import math
import os
import random
from collections import defaultdict

import numpy as np
import torch
from torch.utils.data import Dataset

from .masks import resize_mask, seg_to_mask
from .utils import ImageTransform, load_image

PREDICATES = ('__in_image__', 'left of', 'right of', 'above', 'below', 'inside', 'surrounding')   # coco.py:201-209


def _read_json(path):
  with open(path, 'r') as f:
    return json.load(f)


class CocoSceneGraphDataset(Dataset):
  """item: (image (3, H, W) float, objs (O,) long, boxes (O, 4) float in [0, 1] as (x0, y0, x1, y1),
  masks (O, M, M) long, triples (T, 3) long [subject index, predicate, object index]); the last object is
  `__image__` with the unit box and an all-ones mask (reference coco.py:248-360)."""

  def __init__(self, image_dir, instances_json, stuff_json=None, stuff_only=True, image_size=(64, 64), mask_size=16,
               normalize_images=True, max_samples=None, include_relationships=True, min_object_size=0.02,
               min_objects_per_image=3, max_objects_per_image=8, include_other=False, instance_whitelist=None,
               stuff_whitelist=None, seed=None):
    super(CocoSceneGraphDataset, self).__init__()
    if stuff_only and not stuff_json:
      print('WARNING: Got stuff_only=True but stuff_json=None.')
      print('Falling back to stuff_only=False.')
    self.image_dir, self.mask_size, self.max_samples = image_dir, int(mask_size), max_samples
    self.normalize_images, self.include_relationships = normalize_images, include_relationships
    self._rng = None if seed is None else random.Random(seed)     # (the reference draws from the global module)
    self.set_image_size(image_size)

    instances = _read_json(instances_json)
    stuff = _read_json(stuff_json) if stuff_json else None

    self.image_ids = [im['id'] for im in instances['images']]
    self.image_id_to_filename = {im['id']: im['file_name'] for im in instances['images']}
    self.image_id_to_size = {im['id']: (im['width'], im['height']) for im in instances['images']}

    # categories: COCO ids are kept as the object indices (they start at 1; 0 becomes __image__)
    id_to_name, names = {}, {'instance': [], 'stuff': []}
    for kind, data in (('instance', instances), ('stuff', stuff)):
      for cat in (data['categories'] if data else ()):
        id_to_name[cat['id']] = cat['name']
        names[kind].append(cat['name'])
    allowed = set(names['instance'] if instance_whitelist is None else instance_whitelist)
    allowed |= set(names['stuff'] if stuff_whitelist is None else stuff_whitelist)

    def keep(ann):
      w_img, h_img = self.image_id_to_size[ann['image_id']]
      _, _, w, h = ann['bbox']
      name = id_to_name[ann['category_id']]
      return (w * h) / (w_img * h_img) > min_object_size and name in allowed and (name != 'other' or include_other)

    self.image_id_to_objects = defaultdict(list)
    for ann in instances['annotations']:
      if keep(ann):
        self.image_id_to_objects[ann['image_id']].append(ann)
    if stuff:
      with_stuff = set()
      for ann in stuff['annotations']:
        with_stuff.add(ann['image_id'])
        if keep(ann):
          self.image_id_to_objects[ann['image_id']].append(ann)
      if stuff_only:                  # only the images COCO-Stuff annotates
        self.image_ids = [i for i in self.image_ids if i in with_stuff]
        for i in set(self.image_id_to_filename) - with_stuff:
          self.image_id_to_filename.pop(i, None)
          self.image_id_to_size.pop(i, None)
          self.image_id_to_objects.pop(i, None)

    name_to_idx = {name: cid for cid, name in id_to_name.items()}
    name_to_idx['__image__'] = 0
    if len(set(name_to_idx.values())) != len(name_to_idx):
      raise ValueError('category ids are not unique')
    idx_to_name = ['NONE'] * (1 + max(name_to_idx.values()))
    for name, idx in name_to_idx.items():
      idx_to_name[idx] = name
    self.vocab = {'object_name_to_idx': name_to_idx, 'object_idx_to_name': idx_to_name,
                  'pred_idx_to_name': list(PREDICATES), 'pred_name_to_idx': {p: i for i, p in enumerate(PREDICATES)}}

    # images with too few / too many (kept) objects are dropped
    self.image_ids = [i for i in self.image_ids
                      if min_objects_per_image <= len(self.image_id_to_objects[i]) <= max_objects_per_image]

  def set_image_size(self, image_size):
    self.transform = ImageTransform(image_size, self.normalize_images)
    self.image_size = image_size

  def rng(self):
    """the global `random` module when no seed was given (what the reference draws from) - resolved at use, so that the
    dataset object stays picklable for DataLoader workers under the spawn / forkserver start methods"""
    return self._rng if self._rng is not None else random

  def __len__(self):
    n = len(self.image_ids)
    return n if self.max_samples is None else min(n, self.max_samples)

  def total_objects(self):
    return sum(len(self.image_id_to_objects[i]) for i in self.image_ids[:len(self)])

  # -- one item ----------------------------------------------------------------------------------------------------
  def _object_mask(self, ann, ww, hh):
    """the object's mask cropped to its box (at least one pixel) and resized to M x M, thresholded at 128 / 255"""
    x, y, w, h = ann['bbox']
    full = seg_to_mask(ann['segmentation'], ww, hh)
    x0, y0 = int(round(x)), int(round(y))
    x1, y1 = max(x0 + 1, int(round(x + w))), max(y0 + 1, int(round(y + h)))
    crop = full[y0:y1, x0:x1]
    if crop.size == 0:
      return np.zeros((self.mask_size, self.mask_size), dtype=np.int64)
    return (resize_mask(255.0 * crop, self.mask_size) > 128).astype(np.int64)

  @staticmethod
  def _centers(boxes, masks):
    """per object the mean (x, y) of its mask cells, each cell at its linspace position inside the box; the box
    centre when the mask is empty"""
    m = masks.shape[1]
    lin = np.linspace(0.0, 1.0, m, dtype=np.float32)
    out = np.empty((len(boxes), 2), dtype=np.float32)
    for i, (x0, y0, x1, y1) in enumerate(boxes):
      on = masks[i] == 1
      n = int(on.sum())
      if n == 0:
        out[i] = (0.5 * (x0 + x1), 0.5 * (y0 + y1))
      else:
        xs = (x0 + (x1 - x0) * lin)[None, :].repeat(m, 0)
        ys = (y0 + (y1 - y0) * lin)[:, None].repeat(m, 1)
        out[i] = (xs[on].mean(), ys[on].mean())
    return out

  @staticmethod
  def _predicate(sbox, obox, d):
    """spatial relationship of subject to object: containment first, else the quadrant of the centre offset d"""
    if sbox[0] < obox[0] and sbox[2] > obox[2] and sbox[1] < obox[1] and sbox[3] > obox[3]:
      return 'surrounding'
    if sbox[0] > obox[0] and sbox[2] < obox[2] and sbox[1] > obox[1] and sbox[3] < obox[3]:
      return 'inside'
    theta = math.atan2(float(d[1]), float(d[0]))
    q = math.pi / 4
    if theta >= 3 * q or theta <= -3 * q:
      return 'left of'
    if theta < -q:
      return 'above'
    if theta < q:
      return 'right of'
    return 'below'

  def __getitem__(self, index):
    image_id = self.image_ids[index]
    image, ww, hh = load_image(os.path.join(self.image_dir, self.image_id_to_filename[image_id]), self.transform)
    anns = self.image_id_to_objects[image_id]
    n = len(anns)
    m = self.mask_size
    objs = np.zeros(n + 1, dtype=np.int64)                       # (last: __image__ = 0)
    boxes = np.zeros((n + 1, 4), dtype=np.float32)
    masks = np.ones((n + 1, m, m), dtype=np.int64)
    boxes[n] = (0, 0, 1, 1)
    for i, ann in enumerate(anns):
      x, y, w, h = ann['bbox']
      objs[i] = ann['category_id']
      boxes[i] = (x / ww, y / hh, (x + w) / ww, (y + h) / hh)
      masks[i] = self._object_mask(ann, ww, hh)

    triples = []
    if self.include_relationships and n > 1:
      centers = self._centers(boxes, masks)
      pred_idx = self.vocab['pred_name_to_idx']
      for cur in range(n):
        other = self.rng().choice([j for j in range(n) if j != cur])
        s, o = (cur, other) if self.rng().random() > 0.5 else (other, cur)
        triples.append((s, pred_idx[self._predicate(boxes[s], boxes[o], centers[s] - centers[o])], o))
    triples += [(i, 0, n) for i in range(n)]                     # __in_image__
    return (image, torch.from_numpy(objs), torch.from_numpy(boxes), torch.from_numpy(masks),
            torch.tensor(triples, dtype=torch.int64).view(-1, 3))


def coco_collate_fn(batch):
  """list of items -> (imgs (N, 3, H, W), objs (O,), boxes (O, 4), masks (O, M, M), triples (T, 3) with GLOBAL object
  indices, obj_to_img (O,), triple_to_img (T,)) - reference coco.py:378-419"""
  imgs, objs, boxes, masks, triples, o2i, t2i = [], [], [], [], [], [], []
  first = 0
  for i, (img, ob, bx, mk, tr) in enumerate(batch):
    imgs.append(img[None])
    if ob.dim() == 0 or tr.dim() == 0:
      continue
    tr = tr.clone()
    tr[:, 0] += first
    tr[:, 2] += first
    objs.append(ob); boxes.append(bx); masks.append(mk); triples.append(tr)
    o2i.append(torch.full((ob.size(0),), i, dtype=torch.int64))
    t2i.append(torch.full((tr.size(0),), i, dtype=torch.int64))
    first += ob.size(0)
  return (torch.cat(imgs), torch.cat(objs), torch.cat(boxes), torch.cat(masks), torch.cat(triples),
          torch.cat(o2i), torch.cat(t2i))
