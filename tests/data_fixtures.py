"""Tiny on-disk COCO / COCO-Stuff and Visual Genome datasets for the loader tests (and for scripts/train.py end to
end on real loaders): PNG images drawn with PIL, annotation JSON in the COCO format with all three segmentation
encodings, VG arrays as an .npz with the keys the reference's preprocess_vg.py writes."""
import json
import os

import numpy as np
import PIL.Image
import PIL.ImageDraw


def rle_encode_string(mask):
  """binary (h, w) array -> the compressed-RLE string of the COCO format (the inverse of
  sg2im_amd.data.masks.rle_string_to_counts, written from the format description: column-major runs starting with
  zeros, every count from the third on as a difference to the one two before, 5 data bits + continuation bit per
  character, offset 48)"""
  flat = np.asarray(mask, dtype=np.uint8).T.reshape(-1)
  counts, cur, run = [], 0, 0
  for v in flat:
    if v == cur:
      run += 1
    else:
      counts.append(run)
      cur, run = v, 1
  counts.append(run)
  out = []
  for i, c in enumerate(counts):
    x = c - counts[i - 2] if i > 2 else c
    more = True
    while more:
      ch = x & 0x1f
      x >>= 5
      more = not ((x == 0 and not (ch & 0x10)) or (x == -1 and (ch & 0x10)))
      if more:
        ch |= 0x20
      out.append(chr(ch + 48))
  return ''.join(out), counts


def make_coco(root, n_images=6, seed=0, size=(96, 80)):
  """-> dict of the paths scripts/train.py's --coco_* flags take.  Every image gets 3-5 instance annotations
  (polygons; one uncompressed and one compressed RLE), 1-2 stuff annotations and one object below min_object_size."""
  rng = np.random.RandomState(seed)
  W, H = size
  img_dir = os.path.join(root, 'images')
  os.makedirs(img_dir, exist_ok=True)
  inst_cats = [{'id': i + 1, 'name': 'thing%d' % i} for i in range(5)]
  stuff_cats = [{'id': 92 + i, 'name': n} for i, n in enumerate(['sky', 'grass', 'other'])]
  images, inst_anns, stuff_anns = [], [], []
  aid = 0
  for k in range(n_images):
    name = 'img%03d.png' % k
    pil = PIL.Image.fromarray(rng.randint(0, 255, (H, W, 3)).astype(np.uint8))
    pil.save(os.path.join(img_dir, name))
    images.append({'id': 100 + k, 'file_name': name, 'width': W, 'height': H})
    for j in range(3 + k % 3):
      x, y = rng.randint(0, W - 40), rng.randint(0, H - 40)
      w, h = rng.randint(20, 40), rng.randint(20, 40)
      poly = [x, y, x + w, y, x + w, y + h, x, y + h]
      seg = [[float(v) for v in poly]]
      if j == 1:          # uncompressed RLE of the same rectangle
        m = np.zeros((H, W), np.uint8); m[y:y + h, x:x + w] = 1
        seg = {'counts': rle_encode_string(m)[1], 'size': [H, W]}
      elif j == 2:        # compressed RLE
        m = np.zeros((H, W), np.uint8); m[y:y + h, x:x + w] = 1
        seg = {'counts': rle_encode_string(m)[0], 'size': [H, W]}
      aid += 1
      inst_anns.append({'id': aid, 'image_id': 100 + k, 'category_id': 1 + (k + j) % 5, 'bbox': [float(x), float(y), float(w), float(h)],
                        'segmentation': seg, 'area': float(w * h), 'iscrowd': 0})
    aid += 1                # a tiny object: filtered by min_object_size
    inst_anns.append({'id': aid, 'image_id': 100 + k, 'category_id': 1, 'bbox': [1.0, 1.0, 4.0, 4.0],
                      'segmentation': [[1.0, 1.0, 5.0, 1.0, 5.0, 5.0, 1.0, 5.0]], 'area': 16.0, 'iscrowd': 0})
    if k != n_images - 1:   # the last image has no stuff annotation (dropped under stuff_only)
      for j in range(1 + k % 2):
        m = np.zeros((H, W), np.uint8); m[: H // 2 + 5 * j, :] = 1
        aid += 1
        stuff_anns.append({'id': aid, 'image_id': 100 + k, 'category_id': 92 + j, 'bbox': [0.0, 0.0, float(W), float(H // 2 + 5 * j)],
                           'segmentation': {'counts': rle_encode_string(m)[0], 'size': [H, W]}, 'area': float(m.sum()), 'iscrowd': 0})
      aid += 1              # category 'other': dropped unless include_other
      stuff_anns.append({'id': aid, 'image_id': 100 + k, 'category_id': 94, 'bbox': [0.0, float(H // 2), float(W), float(H // 2)],
                         'segmentation': [[0.0, float(H // 2), float(W), float(H // 2), float(W), float(H), 0.0, float(H)]],
                         'area': float(W * H // 2), 'iscrowd': 0})
  inst_json, stuff_json = os.path.join(root, 'instances.json'), os.path.join(root, 'stuff.json')
  with open(inst_json, 'w') as f:
    json.dump({'images': images, 'annotations': inst_anns, 'categories': inst_cats}, f)
  with open(stuff_json, 'w') as f:
    json.dump({'images': images, 'annotations': stuff_anns, 'categories': stuff_cats}, f)
  return {'image_dir': img_dir, 'instances_json': inst_json, 'stuff_json': stuff_json, 'size': size}


def make_vg(root, n_images=5, seed=0, size=(120, 90), max_objs=12, max_rels=8):
  """-> (vocab, npz path, image dir)"""
  rng = np.random.RandomState(seed)
  W, H = size
  img_dir = os.path.join(root, 'vg_images')
  os.makedirs(img_dir, exist_ok=True)
  objs = ['__image__'] + ['o%d' % i for i in range(1, 9)]
  preds = ['__in_image__'] + ['p%d' % i for i in range(1, 5)]
  vocab = {'object_idx_to_name': objs, 'object_name_to_idx': {n: i for i, n in enumerate(objs)},
           'pred_idx_to_name': preds, 'pred_name_to_idx': {n: i for i, n in enumerate(preds)}}
  names = np.full((n_images, max_objs), -1, np.int32)
  boxes = np.full((n_images, max_objs, 4), -1, np.int32)
  opi = np.zeros(n_images, np.int32); rpi = np.zeros(n_images, np.int32)
  rs = np.full((n_images, max_rels), -1, np.int32); rp = rs.copy(); ro = rs.copy()
  paths = []
  for k in range(n_images):
    name = 'v%03d.png' % k
    PIL.Image.fromarray(rng.randint(0, 255, (H, W, 3)).astype(np.uint8)).save(os.path.join(img_dir, name))
    paths.append(name)
    n = 4 + 2 * k               # image 4 has 12 objects: more than max_objects
    opi[k] = n
    for j in range(n):
      x, y = rng.randint(0, W - 30), rng.randint(0, H - 30)
      names[k, j] = 1 + rng.randint(0, 8)
      boxes[k, j] = (x, y, rng.randint(5, 30), rng.randint(5, 30))
    r = min(max_rels, 2 + k)
    rpi[k] = r
    for j in range(r):
      s = rng.randint(0, n); o = (s + 1 + rng.randint(0, n - 1)) % n
      rs[k, j], rp[k, j], ro[k, j] = s, 1 + rng.randint(0, 4), o
  path = os.path.join(root, 'vg_train.npz')
  np.savez(path, image_paths=np.array(paths), object_names=names, object_boxes=boxes, objects_per_image=opi,
           relationships_per_image=rpi, relationship_subjects=rs, relationship_predicates=rp, relationship_objects=ro)
  return vocab, path, img_dir
