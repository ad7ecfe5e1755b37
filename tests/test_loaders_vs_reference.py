"""CPU, build container only (skipped where /root/reference is absent): sg2im_amd/data against the LIVE reference
loaders (sg2im/data/coco.py, vg.py, utils.py) on the tiny on-disk datasets of tests/data_fixtures.py.

The reference modules import four third-party packages this image does not have.  They are replaced - for the
duration of this module - by the minimal stand-ins below, so that everything the REFERENCE'S OWN code does runs as
written: annotation filtering and pruning, the vocabulary, image ids and their order, box normalisation, the mask crop
and threshold, object centres, the randomly drawn spatial relationships, the __image__ object and its __in_image__
triples, VG's object selection (including `random.sample(obj_idxs, self.max_objects)`), the HDF5 reading loop of
vg.py:53-59, and the three collate functions.  What the stand-ins cover is third-party behaviour, not reference code:
  * torchvision.transforms: Compose, ToTensor (uint8 HWC -> float CHW / 255), Normalize ((x - mean) / std);
  * pycocotools.mask frPyObjects / merge / decode, skimage.transform.resize: on sg2im_amd.data.masks (the polygon
    rasteriser and the bilinear resize this repo ships: tests/test_data_loaders.py checks those on their own);
  * h5py.File: the dataset dictionary of an .npz file.
Both sides draw from the global `random` module, seeded identically before every item: results must be EQUAL."""
import contextlib
import io
import os
import random
import sys
import types

import numpy as np
import pytest
import torch

from tests import data_fixtures as fx

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree not available')


def _stand_ins():
  from sg2im_amd.data import masks as M

  tv, T = types.ModuleType('torchvision'), types.ModuleType('torchvision.transforms')

  class Compose(object):
    def __init__(self, transforms):
      self.transforms = transforms

    def __call__(self, x):
      for t in self.transforms:
        x = t(x)
      return x

  class ToTensor(object):
    def __call__(self, img):
      a = np.asarray(img, dtype=np.uint8)
      return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).float().div(255)

  class Normalize(object):
    def __init__(self, mean, std):
      self.mean, self.std = mean, std

    def __call__(self, t):
      t = t.clone()
      mean = torch.as_tensor(self.mean, dtype=t.dtype).view(-1, 1, 1)
      std = torch.as_tensor(self.std, dtype=t.dtype).view(-1, 1, 1)
      return t.sub_(mean).div_(std)
  T.Compose, T.ToTensor, T.Normalize = Compose, ToTensor, Normalize
  tv.transforms = T

  coco, mask = types.ModuleType('pycocotools'), types.ModuleType('pycocotools.mask')

  def frPyObjects(seg, h, w):
    if isinstance(seg, list):
      return [{'_mask': M.polygons_to_mask([p], h, w)} for p in seg]
    return {'_mask': M.rle_counts_to_mask(seg['counts'], *seg['size'])}

  def merge(rles):
    out = np.zeros_like(rles[0]['_mask'])
    for r in rles:
      out |= r['_mask']
    return {'_mask': out}

  def decode(rle):
    if '_mask' in rle:
      return rle['_mask']
    h, w = rle['size']
    return M.seg_to_mask(rle, w, h)                 # (a compressed RLE is handed through as it is, coco.py:372-374)
  mask.frPyObjects, mask.merge, mask.decode = frPyObjects, merge, decode
  coco.mask = mask

  sk, skt = types.ModuleType('skimage'), types.ModuleType('skimage.transform')

  def resize(image, output_shape, mode='constant'):
    assert mode == 'constant' and output_shape[0] == output_shape[1]
    return M.resize_mask(image, output_shape[0])
  skt.resize = resize
  sk.transform = skt

  h5 = types.ModuleType('h5py')

  class File(object):
    def __init__(self, path, mode='r'):
      assert mode == 'r'
      self._arrays = np.load(path)

    def __enter__(self):
      return self

    def __exit__(self, *exc):
      self._arrays.close()

    def items(self):
      for k in self._arrays.files:
        yield k, self._arrays[k]
  h5.File = File
  return {'torchvision': tv, 'torchvision.transforms': T, 'pycocotools': coco, 'pycocotools.mask': mask,
          'skimage': sk, 'skimage.transform': skt, 'h5py': h5}


@pytest.fixture(scope='module')
def ref():
  """the reference's data modules, imported over the stand-ins (which are removed from sys.modules afterwards)"""
  sys.dont_write_bytecode = True
  if REF not in sys.path:
    sys.path.insert(0, REF)
  fake = _stand_ins()
  missing = [k for k in fake if k not in sys.modules]
  for k in missing:
    sys.modules[k] = fake[k]
  try:
    import sg2im.data.coco as rcoco
    import sg2im.data.vg as rvg
    yield types.SimpleNamespace(coco=rcoco, vg=rvg)
  finally:
    for k in missing:
      sys.modules.pop(k, None)


def _quiet(fn, *a, **k):
  with contextlib.redirect_stdout(io.StringIO()):
    return fn(*a, **k)


def _same(a, b, what):
  assert type(a) == type(b) or (torch.is_tensor(a) and torch.is_tensor(b)), what
  if torch.is_tensor(a):
    assert a.dtype == b.dtype and a.shape == b.shape, (what, a.dtype, b.dtype, tuple(a.shape), tuple(b.shape))
    assert torch.equal(a, b), what
  else:
    assert a == b, what


COCO_VARIANTS = [
  dict(),
  dict(stuff_only=False),
  dict(include_other=True, min_object_size=0.0, max_objects_per_image=20),
  dict(include_relationships=False, normalize_images=False),
  dict(instance_whitelist=['thing0', 'thing2', 'thing3'], stuff_whitelist=['sky'], min_objects_per_image=1),
  dict(image_size=(32, 48), mask_size=8, max_samples=3),
  dict(stuff_json=None, stuff_only=False, min_objects_per_image=1),
]


@pytest.mark.parametrize('kw', COCO_VARIANTS, ids=lambda kw: ','.join(sorted(kw)) or 'defaults')
def test_coco_dataset_equals_the_reference(ref, tmp_path, kw):
  from sg2im_amd.data import coco as mine
  paths = fx.make_coco(str(tmp_path), n_images=7, seed=3)
  args = dict(image_dir=paths['image_dir'], instances_json=paths['instances_json'], stuff_json=paths['stuff_json'])
  args.update(kw)
  a = _quiet(ref.coco.CocoSceneGraphDataset, **args)
  b = _quiet(mine.CocoSceneGraphDataset, **args)
  assert a.vocab == b.vocab
  assert list(a.image_ids) == list(b.image_ids) and len(a) == len(b) and len(a) > 0
  assert a.total_objects() == b.total_objects()
  items_a, items_b = [], []
  for i in range(len(a)):
    random.seed(1000 + i)
    ia = a[i]
    ra = random.random()
    random.seed(1000 + i)
    ib = b[i]
    rb = random.random()
    assert ra == rb, 'item %d consumed a different number of random draws' % i
    assert len(ia) == len(ib) == 5
    for name, x, y in zip(('image', 'objs', 'boxes', 'masks', 'triples'), ia, ib):
      _same(x, y, 'item %d %s' % (i, name))
    items_a.append(ia)
    items_b.append(ib)
  ca, cb = ref.coco.coco_collate_fn(items_a), mine.coco_collate_fn(items_b)
  assert len(ca) == len(cb) == 7
  for k, (x, y) in enumerate(zip(ca, cb)):
    _same(x, y, 'collated field %d' % k)
  if kw.get('image_size') is None:         # set_image_size (the reference's progressive-growing hook, coco.py:222-227)
    _quiet(a.set_image_size, (40, 40))
    _quiet(b.set_image_size, (40, 40))
    random.seed(5)
    ia = a[0]
    random.seed(5)
    ib = b[0]
    _same(ia[0], ib[0], 'image after set_image_size')


VG_VARIANTS = [
  dict(),
  dict(max_objects=5),
  dict(use_orphaned_objects=False, normalize_images=False),
  dict(include_relationships=False, image_size=(32, 48)),
  dict(max_samples=2, max_objects=30),
]


@pytest.mark.parametrize('kw', VG_VARIANTS, ids=lambda kw: ','.join(sorted(kw)) or 'defaults')
def test_vg_dataset_equals_the_reference(ref, tmp_path, kw):
  """h5_path is the .npz of the fixture: the reference reads it through vg.py:53-59's loop over `f.items()`"""
  from sg2im_amd.data import vg as mine
  vocab, path, img_dir = fx.make_vg(str(tmp_path), n_images=6, seed=4, max_objs=14)
  args = dict(image_size=(64, 64))
  args.update(kw)
  a = ref.vg.VgSceneGraphDataset(vocab, path, img_dir, **args)
  b = mine.VgSceneGraphDataset(vocab, path, img_dir, **args)
  assert len(a) == len(b) and len(a) > 0
  items_a, items_b = [], []
  for i in range(len(a)):
    random.seed(77 + i)
    ia = a[i]
    ra = random.random()
    random.seed(77 + i)
    ib = b[i]
    rb = random.random()
    assert ra == rb, 'item %d consumed a different number of random draws' % i
    assert len(ia) == len(ib) == 4
    for name, x, y in zip(('image', 'objs', 'boxes', 'triples'), ia, ib):
      _same(x, y, 'item %d %s' % (i, name))
    items_a.append(ia)
    items_b.append(ib)
  ca, cb = ref.vg.vg_collate_fn(items_a), mine.vg_collate_fn(items_b)
  assert len(ca) == len(cb) == 6
  for k, (x, y) in enumerate(zip(ca, cb)):
    _same(x, y, 'collated field %d' % k)
  ua, ub = ref.vg.vg_uncollate_fn(ca), mine.vg_uncollate_fn(cb)
  assert len(ua) == len(ub) == len(a)
  for i, (ta, tb) in enumerate(zip(ua, ub)):
    for x, y in zip(ta, tb):
      _same(x, y, 'uncollated item %d' % i)


def test_image_transforms_equal_the_reference(ref):
  import PIL.Image
  import sg2im.data.utils as rutils
  from sg2im_amd.data import utils as mine
  rng = np.random.RandomState(0)
  img = PIL.Image.fromarray(rng.randint(0, 255, (37, 53, 3)).astype(np.uint8))
  for size in ((64, 64), (24, 40), 32):
    _same(np.asarray(rutils.Resize(size)(img)).tolist(), np.asarray(mine.Resize(size)(img)).tolist(), 'Resize %r' % (size,))
  x = torch.rand(3, 8, 8)
  _same(rutils.imagenet_preprocess()(x), mine.imagenet_preprocess()(x), 'imagenet_preprocess')
  batch = torch.randn(2, 3, 8, 8)
  for rescale in (True, False):
    _same(rutils.imagenet_deprocess_batch(batch, rescale=rescale), mine.imagenet_deprocess_batch(batch, rescale=rescale),
          'imagenet_deprocess_batch rescale=%s' % rescale)
