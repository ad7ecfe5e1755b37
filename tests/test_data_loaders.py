"""The data front end (sg2im_amd/data, reference sg2im/data/*.py) against its documented contract on tiny on-disk
datasets (tests/data_fixtures.py).  These are contract tests and independent re-derivations; the comparison with the
LIVE reference loaders (over stand-ins for the four absent third-party packages) is tests/test_loaders_vs_reference.py."""
import json
import math
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import data_fixtures as fx  # noqa: E402

from sg2im_amd.data import (CocoSceneGraphDataset, VgSceneGraphDataset, coco_collate_fn, split_graph_batch,  # noqa: E402
                            vg_collate_fn, vg_uncollate_fn)
from sg2im_amd.data import masks as M  # noqa: E402
from sg2im_amd.data.utils import ImageTransform  # noqa: E402


def test_rle_and_polygon_decoding():
  rng = np.random.RandomState(3)
  for h, w in ((7, 5), (16, 16), (33, 20)):
    m = (rng.rand(h, w) > 0.6).astype(np.uint8)
    s, counts = fx.rle_encode_string(m)
    assert M.rle_string_to_counts(s) == counts
    assert np.array_equal(M.seg_to_mask({'counts': s, 'size': [h, w]}, w, h), m)
    assert np.array_equal(M.seg_to_mask({'counts': counts, 'size': [h, w]}, w, h), m)
  # known answer: a 2 x 3 mask [[0 1 1], [1 1 0]] is, column-major, 0 1 1 1 1 0 -> runs 1, 4, 1
  assert np.array_equal(M.rle_counts_to_mask([1, 4, 1], 2, 3), np.array([[0, 1, 1], [1, 1, 0]], np.uint8))
  # a long run needs more than one character and a shrinking one a negative difference
  big = np.zeros((40, 50), np.uint8); big[:, 10:45] = 1; big[5:9, 20:22] = 0
  s, counts = fx.rle_encode_string(big)
  assert max(counts) > 31 and M.rle_string_to_counts(s) == counts
  # polygon: an axis-aligned rectangle covers exactly its pixels (inclusive outline)
  sq = M.seg_to_mask([[2.0, 3.0, 9.0, 3.0, 9.0, 7.0, 2.0, 7.0]], 12, 10)
  assert sq.shape == (10, 12) and sq[3:8, 2:10].all() and sq.sum() == 5 * 8


def test_polygon_rasterisation_differs_from_the_pixel_centre_rule_only_on_the_boundary():
  """the documented deviation from pycocotools: against an independent even-odd test of every pixel centre, the PIL
  rasterisation may only differ within one pixel of the polygon's boundary"""
  rng = np.random.RandomState(5)
  H, W = 48, 64
  for trial in range(6):
    n = rng.randint(3, 9)
    ang = np.sort(rng.rand(n) * 2 * np.pi)
    rad = 8 + rng.rand(n) * 14
    px, py = 32 + rad * np.cos(ang), 24 + rad * np.sin(ang)
    poly = [float(v) for xy in zip(px, py) for v in xy]
    got = M.seg_to_mask([poly], W, H).astype(bool)
    ys, xs = np.mgrid[0:H, 0:W]
    cx, cy = xs + 0.5, ys + 0.5                     # pixel centres
    inside = np.zeros((H, W), bool)
    dist = np.full((H, W), 1e9)
    for i in range(n):
      x0, y0, x1, y1 = px[i], py[i], px[(i + 1) % n], py[(i + 1) % n]
      crosses = ((y0 > cy) != (y1 > cy)) & (cx < (x1 - x0) * (cy - y0) / (y1 - y0 + 1e-12) + x0)
      inside ^= crosses
      t = np.clip(((cx - x0) * (x1 - x0) + (cy - y0) * (y1 - y0)) / ((x1 - x0) ** 2 + (y1 - y0) ** 2 + 1e-12), 0, 1)
      dist = np.minimum(dist, np.hypot(cx - (x0 + t * (x1 - x0)), cy - (y0 + t * (y1 - y0))))
    diff = got != inside
    assert got.sum() > 0
    assert (dist[diff] <= 1.5).all(), float(dist[diff].max())


def test_mask_resize_is_bilinear_at_pixel_centres():
  m = np.zeros((4, 4)); m[1:3, 1:3] = 1
  assert np.allclose(M.resize_mask(m, 4), m)                       # identity at equal size
  up = M.resize_mask(np.ones((2, 2)), 4)
  assert np.allclose(up[1:3, 1:3], 1.0) and np.allclose(up[0, 0], 0.75 * 0.75)     # zeros outside the image
  down = M.resize_mask(np.arange(16.0).reshape(4, 4), 2)           # centres fall between four pixels: their mean
  assert np.allclose(down, [[2.5, 4.5], [10.5, 12.5]])


def _rederive_predicate(boxes, masks, s, o):
  """independent restatement of reference coco.py:296-345 on the RETURNED tensors"""
  sx0, sy0, sx1, sy1 = boxes[s].tolist(); ox0, oy0, ox1, oy1 = boxes[o].tolist()
  if sx0 < ox0 and sx1 > ox1 and sy0 < oy0 and sy1 > oy1:
    return 'surrounding'
  if sx0 > ox0 and sx1 < ox1 and sy0 > oy0 and sy1 < oy1:
    return 'inside'
  c = []
  for i in (s, o):
    x0, y0, x1, y1 = boxes[i]
    mk = masks[i] == 1
    mm = mk.size(0)
    xs = torch.linspace(float(x0), float(x1), mm).view(1, mm).expand(mm, mm)
    ys = torch.linspace(float(y0), float(y1), mm).view(mm, 1).expand(mm, mm)
    c.append((float(xs[mk].mean()), float(ys[mk].mean())) if mk.any() else (0.5 * float(x0 + x1), 0.5 * float(y0 + y1)))
  th = math.atan2(c[0][1] - c[1][1], c[0][0] - c[1][0])
  if th >= 3 * math.pi / 4 or th <= -3 * math.pi / 4:
    return 'left of'
  if -3 * math.pi / 4 <= th < -math.pi / 4:
    return 'above'
  if -math.pi / 4 <= th < math.pi / 4:
    return 'right of'
  return 'below'


def test_coco_dataset_contract(tmp_path):
  p = fx.make_coco(str(tmp_path), n_images=6)
  ds = CocoSceneGraphDataset(p['image_dir'], p['instances_json'], p['stuff_json'], image_size=(32, 48), mask_size=16, seed=1)
  v = ds.vocab
  assert v['object_idx_to_name'][0] == '__image__' and v['object_name_to_idx']['__image__'] == 0
  assert v['object_name_to_idx']['thing0'] == 1 and v['object_name_to_idx']['sky'] == 92       # COCO ids are the indices
  assert v['object_idx_to_name'][6] == 'NONE' and len(v['object_idx_to_name']) == 95
  assert v['pred_idx_to_name'] == ['__in_image__', 'left of', 'right of', 'above', 'below', 'inside', 'surrounding']
  assert len(ds) == 5                                   # the image without stuff annotations is gone (stuff_only)
  total = 0
  for i in range(len(ds)):
    img, objs, boxes, masks, triples = ds[i]
    O = objs.numel()
    total += O - 1
    assert img.shape == (3, 32, 48) and img.dtype == torch.float32
    assert objs.dtype == torch.int64 and objs[-1] == 0 and (objs[:-1] > 0).all()
    assert 94 not in objs.tolist()                      # 'other' is excluded by default
    assert boxes.shape == (O, 4) and boxes.dtype == torch.float32 and boxes[-1].tolist() == [0, 0, 1, 1]
    assert (boxes >= 0).all() and (boxes <= 1).all() and (boxes[:, 2] > boxes[:, 0]).all()
    assert ((boxes[:-1, 2] - boxes[:-1, 0]) * (boxes[:-1, 3] - boxes[:-1, 1]) > 0.02).all()     # min_object_size
    assert masks.shape == (O, 16, 16) and masks.dtype == torch.int64 and set(masks.unique().tolist()) <= {0, 1}
    assert masks[-1].all()
    for j in range(O - 1):
      assert masks[j].float().mean() > 0.5              # rectangles / half planes fill their own boxes
    n = O - 1
    assert triples.shape == (2 * n, 3)                  # one spatial relationship per real object + __in_image__
    assert triples[n:].tolist() == [[j, 0, n] for j in range(n)]
    for cur, (s, pr, o) in enumerate(triples[:n].tolist()):
      assert cur in (s, o) and s != o and s < n and o < n
      assert v['pred_idx_to_name'][pr] == _rederive_predicate(boxes, masks, s, o)
  assert ds.total_objects() == total
  # filters
  assert len(CocoSceneGraphDataset(p['image_dir'], p['instances_json'], p['stuff_json'], stuff_only=False)) == 6
  with_other = CocoSceneGraphDataset(p['image_dir'], p['instances_json'], p['stuff_json'], include_other=True)
  assert any(94 in with_other[i][1].tolist() for i in range(len(with_other)))
  only3 = CocoSceneGraphDataset(p['image_dir'], p['instances_json'], None, stuff_only=False, min_objects_per_image=5,
                                max_objects_per_image=5)
  assert all(only3[i][1].numel() == 6 for i in range(len(only3))) and 0 < len(only3) < 6
  wl = CocoSceneGraphDataset(p['image_dir'], p['instances_json'], p['stuff_json'], instance_whitelist=('thing0', 'thing1'),
                             min_objects_per_image=1)
  assert all(set(wl[i][1].tolist()) <= {0, 1, 2, 92, 93} for i in range(len(wl)))
  norel = CocoSceneGraphDataset(p['image_dir'], p['instances_json'], p['stuff_json'], include_relationships=False)
  assert (norel[0][4][:, 1] == 0).all()
  assert len(CocoSceneGraphDataset(p['image_dir'], p['instances_json'], p['stuff_json'], max_samples=2)) == 2
  # image transform: resize + /255 + ImageNet normalisation
  import PIL.Image
  raw = PIL.Image.open(os.path.join(p['image_dir'], 'img000.png')).convert('RGB').resize((48, 32), PIL.Image.BILINEAR)
  want = (torch.from_numpy(np.asarray(raw)).permute(2, 0, 1).float() / 255 - torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)) \
      / torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
  assert torch.allclose(ds[0][0], want, atol=1e-6)
  assert float(ImageTransform((8, 8), normalize=False)(raw).max()) <= 1.0


def test_coco_collate_and_loader(tmp_path):
  from torch.utils.data import DataLoader
  p = fx.make_coco(str(tmp_path), n_images=6)
  ds = CocoSceneGraphDataset(p['image_dir'], p['instances_json'], p['stuff_json'], image_size=(32, 32), seed=0)
  items = [ds[i] for i in range(3)]
  imgs, objs, boxes, masks, triples, o2i, t2i = coco_collate_fn(items)
  O = sum(it[1].numel() for it in items)
  assert imgs.shape == (3, 3, 32, 32) and objs.shape == (O,) and boxes.shape == (O, 4) and masks.shape == (O, 16, 16)
  assert o2i.tolist() == sum([[i] * it[1].numel() for i, it in enumerate(items)], [])
  assert (o2i[triples[:, 0]] == t2i).all() and (o2i[triples[:, 2]] == t2i).all()          # triples stay inside their image
  tr_back, (objs_back, boxes_back) = split_graph_batch(triples, [objs, boxes], o2i, t2i)
  for i, it in enumerate(items):
    assert torch.equal(tr_back[i], it[4]) and torch.equal(objs_back[i], it[1]) and torch.equal(boxes_back[i], it[2])
  for workers in (0, 2):
    n = 0
    for batch in DataLoader(ds, batch_size=2, shuffle=True, num_workers=workers, collate_fn=coco_collate_fn):
      assert len(batch) == 7 and batch[0].size(0) <= 2
      n += batch[0].size(0)
    assert n == len(ds)


def test_vg_dataset_contract(tmp_path):
  vocab, npz, img_dir = fx.make_vg(str(tmp_path))
  raw = np.load(npz)
  ds = VgSceneGraphDataset(vocab, npz, img_dir, image_size=(40, 40), max_objects=6, seed=2)
  assert len(ds) == 5
  for k in range(len(ds)):
    img, objs, boxes, triples = ds[k]
    O = objs.numel()
    assert img.shape == (3, 40, 40) and objs[-1] == 0 and (objs[:-1] > 0).all() and boxes[-1].tolist() == [0, 0, 1, 1]
    assert O - 1 <= 6                                        # (the reference samples max_objects when too many are related)
    n_rel = int(raw['relationships_per_image'][k])
    related = set(raw['relationship_subjects'][k, :n_rel].tolist()) | set(raw['relationship_objects'][k, :n_rel].tolist())
    if len(related) <= 5:
      assert O - 1 == min(5, int(raw['objects_per_image'][k]))   # filled up with orphaned objects
    n = O - 1
    assert triples[-n:].tolist() == [[j, 0, n] for j in range(n)]
    rel = triples[:-n]
    assert (rel[:, 1] > 0).all() and (rel[:, 0] < n).all() and (rel[:, 2] < n).all()
    if len(related) <= 5:
      assert rel.size(0) == n_rel                            # every relationship survives when its objects do
    # boxes are the raw (x, y, w, h) over the ORIGINAL image size
    cats = raw['object_names'][k]; rb = raw['object_boxes'][k]
    for j in range(n):
      hits = [(x / 120.0, y / 90.0, (x + w) / 120.0, (y + h) / 90.0) for (x, y, w, h), c in zip(rb.tolist(), cats.tolist()) if c == int(objs[j])]
      assert any(np.allclose(boxes[j].tolist(), h_, atol=1e-6) for h_ in hits)
  no_orph = VgSceneGraphDataset(vocab, npz, img_dir, image_size=(40, 40), max_objects=10, use_orphaned_objects=False, seed=2)
  img, objs, boxes, triples = no_orph[0]
  n_rel = int(raw['relationships_per_image'][0])
  related = set(raw['relationship_subjects'][0, :n_rel].tolist()) | set(raw['relationship_objects'][0, :n_rel].tolist())
  assert objs.numel() - 1 == len(related)
  assert (VgSceneGraphDataset(vocab, npz, img_dir, include_relationships=False)[1][3][:, 1] == 0).all()
  batch = vg_collate_fn([ds[0], ds[1], ds[2]])
  assert len(batch) == 6
  for (img, objs, boxes, triples), want in zip(vg_uncollate_fn(batch), [ds[0], ds[1], ds[2]]):
    assert objs.numel() == want[1].numel() and triples.size(0) == want[3].size(0) and int(triples[:, [0, 2]].max()) < objs.numel()
  with pytest.raises(ImportError):
    VgSceneGraphDataset(vocab, str(tmp_path / 'train.h5'), img_dir)          # an .h5 needs h5py, loudly


@pytest.mark.gpu
def test_train_script_on_real_coco_loaders(tmp_path):
  """scripts/train.py end to end on the COCO loaders (no synthetic banner): 6 iterations in graph mode incl. a
  checkpoint with check_model on train and val"""
  import subprocess
  p = fx.make_coco(str(tmp_path), n_images=12, size=(96, 96))
  out = str(tmp_path / 'out'); os.makedirs(out)
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  cmd = [sys.executable, os.path.join(root, 'scripts', 'train.py'), '--dataset', 'coco', '--batch_size', '4',
         '--num_iterations', '6', '--print_every', '2', '--checkpoint_every', '6', '--output_dir', out, '--loader_num_workers', '0',
         '--num_val_samples', '4', '--coco_stuff_only', '0', '--min_objects_per_image', '2',
         '--coco_train_image_dir', p['image_dir'], '--coco_val_image_dir', p['image_dir'],
         '--coco_train_instances_json', p['instances_json'], '--coco_val_instances_json', p['instances_json'],
         '--coco_train_stuff_json', p['stuff_json'], '--coco_val_stuff_json', p['stuff_json']]
  r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
  assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
  assert 'SYNTHETIC' not in r.stdout and 'Training dataset has' in r.stdout and 'Starting epoch 2' in r.stdout
  ck = torch.load(os.path.join(out, 'checkpoint_with_model.pt'), map_location='cpu', weights_only=False)
  assert ck['vocab']['object_name_to_idx']['sky'] == 92 and ck['counters']['t'] == 6 and ck['counters']['epoch'] >= 2
  assert len(ck['val_losses']['bbox_pred']) == 1 and ck['model_state'] is not None
  # scripts/sample_images.py (reference scripts/sample_images.py) on that checkpoint and the same files: predicted
  # boxes / masks, then ground-truth boxes + masks with the GT images saved
  samples = str(tmp_path / 'samples')
  base = [sys.executable, os.path.join(root, 'scripts', 'sample_images.py'), '--checkpoint', os.path.join(out, 'checkpoint_with_model.pt'),
          '--dataset', 'coco', '--batch_size', '4', '--num_samples', '6', '--loader_num_workers', '0',
          '--coco_image_dir', p['image_dir'], '--instances_json', p['instances_json'], '--stuff_json', p['stuff_json']]
  for extra, sub in (([], 'pred'), (['--use_gt_boxes', '1', '--use_gt_masks', '1', '--save_gt_imgs', '1'], 'gt')):
    r = subprocess.run(base + ['--output_dir', os.path.join(samples, sub)] + extra, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    data = torch.load(os.path.join(samples, sub, 'data.pt'), map_location='cpu', weights_only=False)
    n = len(data['filenames'])
    assert n == 6 and data['filenames'][0] == '0000.png' and data['vocab']['object_name_to_idx']['sky'] == 92
    assert all(len(data[k]) == n for k in ('objs', 'boxes_pred', 'boxes_gt', 'masks_pred', 'masks_gt'))
    for i in range(n):
      k = data['objs'][i].numel()
      assert data['boxes_pred'][i].shape == (k, 4) and data['boxes_gt'][i].shape == (k, 4)
      assert data['masks_pred'][i].shape == (k, 16, 16) and data['masks_gt'][i].shape == (k, 16, 16)
      assert int(data['objs'][i][-1]) == 0                  # (the __image__ object closes every image)
      assert os.path.isfile(os.path.join(samples, sub, 'images', data['filenames'][i]))
    assert os.path.isdir(os.path.join(samples, sub, 'images_gt')) == (sub == 'gt')
    import PIL.Image
    assert PIL.Image.open(os.path.join(samples, sub, 'images', '0003.png')).size == (64, 64)


def test_copy_ahead_keeps_order_and_applies_finish():
  """sg2im_amd/data/prefetch.py::CopyAhead (the GPU-side input pipeline of scripts/train.py; on a CPU device the
  batches pass through): order, the 6-tuple -> 7-tuple `finish`, one batch requested ahead, clean exhaustion."""
  import torch
  from sg2im_amd.data.prefetch import CopyAhead
  pulled = []

  def gen():
    for i in range(4):
      pulled.append(i)
      yield (torch.full((2,), float(i)), torch.tensor([i]), 'tag%d' % i)
  it = CopyAhead(gen(), 'cpu', finish=lambda b: b + (None,))
  first = next(it)
  assert pulled == [0, 1]                         # batch 1 was already requested while batch 0 is in use
  assert float(first[0][0]) == 0.0 and first[2] == 'tag0' and first[3] is None
  rest = list(it)
  assert [int(b[1]) for b in rest] == [1, 2, 3]
  with pytest.raises(StopIteration):
    next(it)


def test_sample_images_checkpoint_list_conventions(tmp_path):
  """scripts/sample_images.py::sampling_jobs - the reference's output-directory conventions for --checkpoint /
  --checkpoint_list (reference scripts/sample_images.py:243-286), without a GPU: file on line k -> result%03d % k,
  snapshot directory on line k -> result%03d_<tag> % (k - 1) for its *snapshot* files in sorted order, missing
  paths skipped, exactly one of the two flags."""
  import argparse
  import importlib.util
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  spec = importlib.util.spec_from_file_location('sample_images_script', os.path.join(root, 'scripts', 'sample_images.py'))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  ck = tmp_path / 'a.pt'; ck.write_bytes(b'x')
  snaps = tmp_path / 'run'; snaps.mkdir()
  for fn in ('snapshot_00200K.pt', 'snapshot_00100K.pt', 'notes.txt'):
    (snaps / fn).write_bytes(b'x')
  lst = tmp_path / 'list.txt'
  lst.write_text('%s\n%s\n%s\n' % (ck, tmp_path / 'missing.pt', snaps))
  out = str(tmp_path / 'out')
  jobs = mod.sampling_jobs(argparse.Namespace(checkpoint=None, checkpoint_list=str(lst), output_dir=out))
  assert jobs == [(str(ck), os.path.join(out, 'result001')),
                  (str(snaps / 'snapshot_00100K.pt'), os.path.join(out, 'result002_00100K')),
                  (str(snaps / 'snapshot_00200K.pt'), os.path.join(out, 'result002_00200K'))]
  assert mod.sampling_jobs(argparse.Namespace(checkpoint=str(ck), checkpoint_list=None, output_dir=out)) == [(str(ck), out)]
  for bad in (dict(checkpoint=None, checkpoint_list=None), dict(checkpoint=str(ck), checkpoint_list=str(lst))):
    with pytest.raises(ValueError):
      mod.sampling_jobs(argparse.Namespace(output_dir=out, **bad))
  made = mod.output_dirs(out, {'images': True, 'images_gt': False})
  assert os.path.isdir(made['images']) and made['images_gt'] is None and not os.path.exists(os.path.join(out, 'images_gt'))
