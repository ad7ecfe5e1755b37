#!/usr/bin/env python
"""Sample many images from a trained model for evaluation (reference scripts/sample_images.py:1-290): iterate a
COCO / Visual Genome validation loader, run ``Sg2ImModel.forward`` on the MI355X (predicted - or, with
``--use_gt_boxes`` / ``--use_gt_masks``, ground-truth - boxes and masks), write ``images/%04d.png`` (and
``images_gt/``) and a ``data.pt`` with the per-image objects / predicted and ground-truth boxes and masks / file names.
Same flags, defaults, output layout and checkpoint conventions (``--checkpoint`` or ``--checkpoint_list`` of files /
snapshot directories) as the reference.  Differences: images are written with PIL (scipy.misc.imsave no longer
exists), there is no CPU path, and ``--save_graphs`` needs graphviz' ``dot`` through sg2im/vis.py, which is outside
the hot-path scope (NotImplementedError, like scripts/run_model.py --draw_scene_graphs)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
from PIL import Image
from torch.utils.data import DataLoader

from sg2im_amd.data import (CocoSceneGraphDataset, VgSceneGraphDataset, coco_collate_fn, split_graph_batch,
                            vg_collate_fn)
from sg2im_amd.model import Sg2ImModel
from sg2im_amd.utils import bool_flag, imagenet_deprocess_batch, int_tuple

parser = argparse.ArgumentParser()
parser.add_argument('--checkpoint', default='sg2im-models/vg64.pt')
parser.add_argument('--checkpoint_list', default=None)
parser.add_argument('--model_mode', default='eval', choices=['train', 'eval'])

# Shared dataset options
parser.add_argument('--dataset', default='vg', choices=['coco', 'vg'])
parser.add_argument('--image_size', default=(64, 64), type=int_tuple)
parser.add_argument('--batch_size', default=24, type=int)
parser.add_argument('--shuffle', default=False, type=bool_flag)
parser.add_argument('--loader_num_workers', default=4, type=int)
parser.add_argument('--num_samples', default=10000, type=int)
parser.add_argument('--save_gt_imgs', default=False, type=bool_flag)
parser.add_argument('--save_graphs', default=False, type=bool_flag)
parser.add_argument('--use_gt_boxes', default=False, type=bool_flag)
parser.add_argument('--use_gt_masks', default=False, type=bool_flag)
parser.add_argument('--save_layout', default=True, type=bool_flag)

parser.add_argument('--output_dir', default='output')

# For VG
VG_DIR = os.path.expanduser('datasets/vg')
parser.add_argument('--vg_h5', default=os.path.join(VG_DIR, 'val.h5'))
parser.add_argument('--vg_image_dir', default=os.path.join(VG_DIR, 'images'))

# For COCO
COCO_DIR = os.path.expanduser('~/datasets/coco/2017')
parser.add_argument('--coco_image_dir', default=os.path.join(COCO_DIR, 'images/val2017'))
parser.add_argument('--instances_json', default=os.path.join(COCO_DIR, 'annotations/instances_val2017.json'))
parser.add_argument('--stuff_json', default=os.path.join(COCO_DIR, 'annotations/stuff_val2017.json'))


def build_coco_dset(args, checkpoint):
  """sample_images.py:77-95: the dataset options the model was trained with come from the checkpoint"""
  ca = checkpoint['args']
  print('include other: ', ca.get('coco_include_other'))
  return CocoSceneGraphDataset(
    image_dir=args.coco_image_dir, instances_json=args.instances_json, stuff_json=args.stuff_json,
    stuff_only=ca['coco_stuff_only'], image_size=args.image_size, mask_size=ca['mask_size'], max_samples=args.num_samples,
    min_object_size=ca['min_object_size'], min_objects_per_image=ca['min_objects_per_image'],
    instance_whitelist=ca['instance_whitelist'], stuff_whitelist=ca['stuff_whitelist'],
    include_other=ca.get('coco_include_other', True))


def build_vg_dset(args, checkpoint):
  """sample_images.py:98-110"""
  return VgSceneGraphDataset(
    vocab=checkpoint['model_kwargs']['vocab'], h5_path=args.vg_h5, image_dir=args.vg_image_dir, image_size=args.image_size,
    max_samples=args.num_samples, max_objects=checkpoint['args']['max_objects_per_image'],
    use_orphaned_objects=checkpoint['args']['vg_use_orphaned_objects'])


def build_loader(args, checkpoint):
  if args.dataset == 'coco':
    dset, collate_fn = build_coco_dset(args, checkpoint), coco_collate_fn
  else:
    dset, collate_fn = build_vg_dset(args, checkpoint), vg_collate_fn
  return DataLoader(dset, batch_size=args.batch_size, num_workers=args.loader_num_workers, shuffle=args.shuffle,
                    collate_fn=collate_fn)


def build_model(args, checkpoint, device):
  model = Sg2ImModel(**checkpoint['model_kwargs'])
  model.load_state_dict(checkpoint['model_state'])
  if args.model_mode == 'eval':
    model.eval()
  else:
    model.train()
  model.image_size = args.image_size          # (read at call time: a 64 x 64 model can sample at another size)
  return model.to(device)


def output_dirs(root, wanted):
  """create root/<name> for every wanted name -> {name: path or None}; the reference's output layout
  (images/, images_gt/ only with --save_gt_imgs)"""
  made = {}
  for name, want in wanted.items():
    made[name] = os.path.join(root, name) if want else None
    if want:
      os.makedirs(made[name], exist_ok=True)
  return made


def imsave(path, chw_uint8):
  Image.fromarray(chw_uint8.numpy().transpose(1, 2, 0)).save(path)


def run_model(args, checkpoint, output_dir, loader=None):
  """sample_images.py:151-243"""
  if args.save_graphs:
    raise NotImplementedError('--save_graphs draws with graphviz (sg2im/vis.py), which is out of scope')
  device = torch.device('cuda:0')
  vocab = checkpoint['model_kwargs']['vocab']
  model = build_model(args, checkpoint, device)
  if loader is None:
    loader = build_loader(args, checkpoint)
  dirs = output_dirs(output_dir, {'images': True, 'images_gt': args.save_gt_imgs})
  img_dir, gt_img_dir = dirs['images'], dirs['images_gt']
  data_path = os.path.join(output_dir, 'data.pt')
  data = {'vocab': vocab, 'objs': [], 'masks_pred': [], 'boxes_pred': [], 'masks_gt': [], 'boxes_gt': [], 'filenames': []}
  img_idx = 0
  for batch in loader:
    masks = None
    if len(batch) == 6:
      imgs, objs, boxes, triples, obj_to_img, triple_to_img = [x.to(device) for x in batch]
    else:
      imgs, objs, boxes, masks, triples, obj_to_img, triple_to_img = [x.to(device) for x in batch]
    imgs_gt = imagenet_deprocess_batch(imgs)
    boxes_gt = boxes if args.use_gt_boxes else None
    masks_gt = masks if args.use_gt_masks else None
    with torch.no_grad():
      imgs_pred, boxes_pred, masks_pred, _ = model(objs, triples, obj_to_img, boxes_gt=boxes_gt, masks_gt=masks_gt)
    imgs_pred = imagenet_deprocess_batch(imgs_pred)
    if masks_pred is None:                      # (a model without mask_net: the reference would fail on the split)
      masks_pred = torch.zeros(objs.size(0), 0, 0, device=device)
    _, (objs_l, boxes_pred_l, masks_pred_l) = split_graph_batch(triples, [objs, boxes_pred, masks_pred], obj_to_img,
                                                                triple_to_img)
    obj_data_gt = [boxes] + ([masks] if masks is not None else [])
    _, obj_data_gt = split_graph_batch(triples, obj_data_gt, obj_to_img, triple_to_img)
    boxes_gt_l, masks_gt_l = obj_data_gt[0], (obj_data_gt[1] if masks is not None else None)
    for i in range(imgs_pred.size(0)):
      img_filename = '%04d.png' % img_idx
      if args.save_gt_imgs:
        imsave(os.path.join(gt_img_dir, img_filename), imgs_gt[i])
      imsave(os.path.join(img_dir, img_filename), imgs_pred[i])
      data['objs'].append(objs_l[i].cpu().clone())
      data['masks_pred'].append(masks_pred_l[i].cpu().clone())
      data['boxes_pred'].append(boxes_pred_l[i].cpu().clone())
      data['boxes_gt'].append(boxes_gt_l[i].cpu().clone())
      data['filenames'].append(img_filename)
      data['masks_gt'].append(masks_gt_l[i].cpu().clone() if masks_gt_l is not None else None)
      img_idx += 1
    torch.save(data, data_path)
    print('Saved %d images' % img_idx)


def sampling_jobs(args):
  """[(checkpoint file, output directory)] in the order they are sampled - the reference's conventions
  (sample_images.py:243-286): ``--checkpoint`` writes into ``--output_dir`` itself; line k (1-based) of
  ``--checkpoint_list`` writes into ``result%03d`` % k when it names a file, and when it names a directory every
  ``*snapshot*`` file in it, in sorted order, writes into ``result%03d_<tag>`` % (k - 1) with <tag> the second
  underscore field of the file name ("snapshot_00100K.pt" -> "00100K").  Lines naming nothing that exists are skipped."""
  if (args.checkpoint is None) == (args.checkpoint_list is None):
    raise ValueError('Must specify exactly one of --checkpoint and --checkpoint_list')
  if args.checkpoint is not None:
    return [(args.checkpoint, args.output_dir)]
  jobs = []
  with open(args.checkpoint_list) as f:
    entries = [line.strip() for line in f]
  for k, entry in enumerate(entries, 1):
    if os.path.isfile(entry):
      jobs.append((entry, os.path.join(args.output_dir, 'result%03d' % k)))
    elif os.path.isdir(entry):
      snaps = sorted(fn for fn in os.listdir(entry) if 'snapshot' in fn)
      jobs += [(os.path.join(entry, fn), os.path.join(args.output_dir, 'result%03d_%s' % (k - 1, os.path.splitext(fn)[0].split('_')[1])))
               for fn in snaps]
  return jobs


def main(args):
  jobs = sampling_jobs(args)
  if not torch.cuda.is_available():
    raise RuntimeError('sg2im_amd runs on an MI355X only: no CPU path')
  loader = None                               # (one loader serves every checkpoint of a list)
  for path, out_dir in jobs:
    print('Loading model from ', path)
    checkpoint = torch.load(path, map_location='cpu', weights_only=False)
    if loader is None and len(jobs) > 1:
      loader = build_loader(args, checkpoint)
    run_model(args, checkpoint, out_dir, loader)
  return 0


if __name__ == '__main__':
  # (--checkpoint has a default, as in the reference: naming a list means giving up that default explicitly)
  a = parser.parse_args()
  if a.checkpoint_list is not None and '--checkpoint' not in ' '.join(sys.argv[1:]).replace('--checkpoint_list', ''):
    a.checkpoint = None
  sys.exit(main(a))
