"""Visual Genome scene-graph dataset (reference sg2im/data/vg.py:32-208) over the arrays the reference's
preprocess_vg.py writes: `image_paths`, `object_names`, `object_boxes`, `objects_per_image`,
`relationships_per_image`, `relationship_subjects / _predicates / _objects`.  The container is the reference's HDF5
file when h5py is importable, or an `.npz` with the same keys (what this build's tests use: h5py is not installed
here - a missing h5py with an .h5 path is an error, not a fallback)."""
import os
import random

import numpy as np
import torch
from torch.utils.data import Dataset

from .utils import ImageTransform, load_image


def _load_arrays(path):
  if path.endswith('.npz'):
    with np.load(path, allow_pickle=False) as z:
      return {k: z[k] for k in z.files}
  try:
    import h5py
  except ImportError:
    raise ImportError('reading %s needs h5py (not installed); an .npz with the same keys is accepted too' % path)
  with h5py.File(path, 'r') as f:
    return {k: (list(v) if k == 'image_paths' else np.asarray(v)) for k, v in f.items()}


class VgSceneGraphDataset(Dataset):
  """item: (image (3, H, W), objs (O,) long, boxes (O, 4) float in [0, 1], triples (T, 3) long); objects that take
  part in relationships come first (at most max_objects - 1 of them... see __getitem__), `__image__` is last."""

  def __init__(self, vocab, h5_path, image_dir, image_size=(256, 256), normalize_images=True, max_objects=10,
               max_samples=None, include_relationships=True, use_orphaned_objects=True, seed=None):
    super(VgSceneGraphDataset, self).__init__()
    self.image_dir, self.image_size, self.vocab = image_dir, image_size, vocab
    self.num_objects = len(vocab['object_idx_to_name'])
    self.use_orphaned_objects, self.max_objects = use_orphaned_objects, max_objects
    self.max_samples, self.include_relationships = max_samples, include_relationships
    self._rng = None if seed is None else random.Random(seed)
    self.transform = ImageTransform(image_size, normalize_images)
    arrays = _load_arrays(h5_path)
    paths = arrays.pop('image_paths')
    self.image_paths = [p.decode('utf-8') if isinstance(p, bytes) else str(p) for p in paths]
    self.data = {k: np.asarray(v).astype(np.int64) for k, v in arrays.items()}

  def rng(self):
    """the global `random` module when no seed was given (what the reference draws from) - resolved at use, so that the
    dataset object stays picklable for DataLoader workers under the spawn / forkserver start methods"""
    return self._rng if self._rng is not None else random

  def __len__(self):
    n = self.data['object_names'].shape[0]
    return n if self.max_samples is None else min(self.max_samples, n)

  def __getitem__(self, index):
    d = self.data
    image, ww, hh = load_image(os.path.join(self.image_dir, self.image_paths[index]), self.transform)
    n_rel = int(d['relationships_per_image'][index])
    subj = d['relationship_subjects'][index, :n_rel]
    obj = d['relationship_objects'][index, :n_rel]
    pred = d['relationship_predicates'][index, :n_rel]
    related = set(subj.tolist()) | set(obj.tolist())
    orphans = [i for i in range(int(d['objects_per_image'][index])) if i not in related]
    chosen = list(related)
    # reference vg.py:95-100: more related objects than fit -> a random max_objects of them (one more than the
    # max_objects - 1 it otherwise aims for; kept as is); fewer -> filled up with objects outside any relationship
    if len(chosen) > self.max_objects - 1:
      chosen = self.rng().sample(chosen, self.max_objects)
    if len(chosen) < self.max_objects - 1 and self.use_orphaned_objects:
      chosen += self.rng().sample(orphans, min(self.max_objects - 1 - len(chosen), len(orphans)))
    n = len(chosen)
    objs = np.empty(n + 1, dtype=np.int64)
    boxes = np.empty((n + 1, 4), dtype=np.float32)
    objs[n] = self.vocab['object_name_to_idx']['__image__']
    boxes[n] = (0, 0, 1, 1)
    local = {}
    for i, k in enumerate(chosen):
      x, y, w, h = (float(v) for v in d['object_boxes'][index, k])
      objs[i] = d['object_names'][index, k]
      boxes[i] = (x / ww, y / hh, (x + w) / ww, (y + h) / hh)
      local[k] = i
    triples = []
    if self.include_relationships:
      triples = [(local[int(s)], int(p), local[int(o)]) for s, p, o in zip(subj, pred, obj)
                 if int(s) in local and int(o) in local]
    in_image = self.vocab['pred_name_to_idx']['__in_image__']
    triples += [(i, in_image, n) for i in range(n)]
    return (image, torch.from_numpy(objs), torch.from_numpy(boxes),
            torch.tensor(triples, dtype=torch.int64).view(-1, 3))


def vg_collate_fn(batch):
  """list of items -> (imgs, objs, boxes, triples with GLOBAL object indices, obj_to_img, triple_to_img) -
  reference vg.py:141-176"""
  imgs, objs, boxes, triples, o2i, t2i = [], [], [], [], [], []
  first = 0
  for i, (img, ob, bx, tr) in enumerate(batch):
    tr = tr.clone()
    tr[:, 0] += first
    tr[:, 2] += first
    imgs.append(img[None]); objs.append(ob); boxes.append(bx); triples.append(tr)
    o2i.append(torch.full((ob.size(0),), i, dtype=torch.int64))
    t2i.append(torch.full((tr.size(0),), i, dtype=torch.int64))
    first += ob.size(0)
  return (torch.cat(imgs), torch.cat(objs), torch.cat(boxes), torch.cat(triples), torch.cat(o2i), torch.cat(t2i))


def vg_uncollate_fn(batch):
  """inverse of vg_collate_fn: a list of (img, objs, boxes, triples) with local object indices"""
  imgs, objs, boxes, triples, obj_to_img, triple_to_img = batch
  out, first = [], 0
  for i in range(imgs.size(0)):
    rows = (obj_to_img == i).nonzero().view(-1)
    tr = triples[(triple_to_img == i).nonzero().view(-1)].clone()
    tr[:, 0] -= first
    tr[:, 2] -= first
    first += rows.numel()
    out.append((imgs[i], objs[rows], boxes[rows], tr))
  return out
