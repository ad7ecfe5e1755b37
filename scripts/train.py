#!/usr/bin/env python
"""Training entry point with the reference's option surface (reference scripts/train.py:47-140)
driving sg2im_amd.trainer.Trainer on MI355X; one process per GPU under
``python -m torch.distributed.run --nproc-per-node N scripts/train.py ...``.

Data: when the dataset files the flags name exist, the reference's loaders are used (sg2im_amd/data: COCO /
COCO-Stuff annotations + images; Visual Genome arrays as the reference's .h5 when h5py is importable, or an .npz
with the same keys) - ``build_loaders`` below, reference train.py:230-306.  When they do not exist (this build
container: no datasets), batches come from the seeded synthetic generator that reproduces the collate layout
(``--dataset coco|vg`` selects the graph style), and the script says so loudly.  Every reference flag is
accepted; flags whose feature is not on the HIP path yet fail loudly.
"""
import argparse
import collections
import json
import os
import sys
import time
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
import torch.distributed as dist

from sg2im_amd import losses as L
from sg2im_amd.metrics import jaccard
from sg2im_amd.synthetic import make_vocab, synthetic_batch
from sg2im_amd.trainer import Trainer
from sg2im_amd.utils import bool_flag, imagenet_deprocess_batch, int_tuple, str_tuple, timeit

VG_DIR = os.path.expanduser('datasets/vg')
COCO_DIR = os.path.expanduser('datasets/coco')

parser = argparse.ArgumentParser()
parser.add_argument('--dataset', default='coco', choices=['vg', 'coco'])
# Optimization hyperparameters
parser.add_argument('--batch_size', default=32, type=int)
parser.add_argument('--num_iterations', default=1000000, type=int)
parser.add_argument('--learning_rate', default=1e-4, type=float)
parser.add_argument('--eval_mode_after', default=100000, type=int)
# Dataset options (accepted for compatibility; the synthetic generator uses image_size only)
parser.add_argument('--image_size', default='64,64', type=int_tuple)
parser.add_argument('--num_train_samples', default=None, type=int)
parser.add_argument('--num_val_samples', default=1024, type=int)
parser.add_argument('--shuffle_val', default=True, type=bool_flag)
parser.add_argument('--loader_num_workers', default=4, type=int)
parser.add_argument('--include_relationships', default=True, type=bool_flag)
parser.add_argument('--vg_image_dir', default=os.path.join(VG_DIR, 'images'))
parser.add_argument('--train_h5', default=os.path.join(VG_DIR, 'train.h5'))
parser.add_argument('--val_h5', default=os.path.join(VG_DIR, 'val.h5'))
parser.add_argument('--vocab_json', default=os.path.join(VG_DIR, 'vocab.json'))
parser.add_argument('--max_objects_per_image', default=10, type=int)
parser.add_argument('--vg_use_orphaned_objects', default=True, type=bool_flag)
parser.add_argument('--coco_train_image_dir', default=os.path.join(COCO_DIR, 'images/train2017'))
parser.add_argument('--coco_val_image_dir', default=os.path.join(COCO_DIR, 'images/val2017'))
parser.add_argument('--coco_train_instances_json', default=os.path.join(COCO_DIR, 'annotations/instances_train2017.json'))
parser.add_argument('--coco_train_stuff_json', default=os.path.join(COCO_DIR, 'annotations/stuff_train2017.json'))
parser.add_argument('--coco_val_instances_json', default=os.path.join(COCO_DIR, 'annotations/instances_val2017.json'))
parser.add_argument('--coco_val_stuff_json', default=os.path.join(COCO_DIR, 'annotations/stuff_val2017.json'))
parser.add_argument('--instance_whitelist', default=None, type=str_tuple)
parser.add_argument('--stuff_whitelist', default=None, type=str_tuple)
parser.add_argument('--coco_include_other', default=False, type=bool_flag)
parser.add_argument('--min_object_size', default=0.02, type=float)
parser.add_argument('--min_objects_per_image', default=3, type=int)
parser.add_argument('--coco_stuff_only', default=True, type=bool_flag)
# Generator options
parser.add_argument('--mask_size', default=16, type=int)
parser.add_argument('--embedding_dim', default=128, type=int)
parser.add_argument('--gconv_dim', default=128, type=int)
parser.add_argument('--gconv_hidden_dim', default=512, type=int)
parser.add_argument('--gconv_num_layers', default=5, type=int)
parser.add_argument('--mlp_normalization', default='none', type=str)
parser.add_argument('--refinement_network_dims', default='1024,512,256,128,64', type=int_tuple)
parser.add_argument('--normalization', default='batch')
parser.add_argument('--activation', default='leakyrelu-0.2')
parser.add_argument('--layout_noise_dim', default=32, type=int)
parser.add_argument('--use_boxes_pred_after', default=-1, type=int)
# Generator losses
parser.add_argument('--mask_loss_weight', default=0, type=float)
parser.add_argument('--l1_pixel_loss_weight', default=1.0, type=float)
parser.add_argument('--bbox_pred_loss_weight', default=10, type=float)
parser.add_argument('--predicate_pred_loss_weight', default=0, type=float)
# Generic discriminator options
parser.add_argument('--discriminator_loss_weight', default=0.01, type=float)
parser.add_argument('--gan_loss_type', default='gan')
parser.add_argument('--d_clip', default=None, type=float)
parser.add_argument('--d_normalization', default='batch')
parser.add_argument('--d_padding', default='valid')
parser.add_argument('--d_activation', default='leakyrelu-0.2')
# Object / image discriminators
parser.add_argument('--d_obj_arch', default='C4-64-2,C4-128-2,C4-256-2')
parser.add_argument('--crop_size', default=32, type=int)
parser.add_argument('--d_obj_weight', default=1.0, type=float)
parser.add_argument('--ac_loss_weight', default=0.1, type=float)
parser.add_argument('--d_img_arch', default='C4-64-2,C4-128-2,C4-256-2')
parser.add_argument('--d_img_weight', default=1.0, type=float)
# Output options
parser.add_argument('--print_every', default=10, type=int)
parser.add_argument('--timing', default=False, type=bool_flag)
parser.add_argument('--checkpoint_every', default=10000, type=int)
parser.add_argument('--output_dir', default=os.getcwd())
parser.add_argument('--checkpoint_name', default='checkpoint')
parser.add_argument('--checkpoint_start_from', default=None)
parser.add_argument('--restore_from_checkpoint', default=False, type=bool_flag)
# launcher-level additions
parser.add_argument('--seed', default=0, type=int)
# one captured hipGraph per batch-shape bucket (object / triple axes padded to these multiples with
# exactly neutral rows, sg2im_amd/bucketing.py); --use_graphs 0 launches every kernel from Python
parser.add_argument('--use_graphs', default=True, type=bool_flag)
parser.add_argument('--bucket_objects', default=32, type=int)
parser.add_argument('--bucket_triples', default=64, type=int)
parser.add_argument('--align_corners', default=False, type=bool_flag,
                    help='bilinear sampling convention of layout / crops: False = F.grid_sample of torch >= 1.3, '
                         'True = torch 0.4 (what the reference authors trained with)')


_DATASET_FLAGS = ('vg_image_dir', 'train_h5', 'val_h5', 'vocab_json', 'coco_train_image_dir', 'coco_val_image_dir',
                  'coco_train_instances_json', 'coco_train_stuff_json', 'coco_val_instances_json',
                  'coco_val_stuff_json', 'num_train_samples', 'instance_whitelist', 'stuff_whitelist')


def dataset_files(args):
  """the files / directories the selected dataset needs, and which of them are missing"""
  if args.dataset == 'coco':
    need = [args.coco_train_image_dir, args.coco_train_instances_json, args.coco_val_image_dir, args.coco_val_instances_json]
    need += [p for p in (args.coco_train_stuff_json, args.coco_val_stuff_json) if p]
  else:
    need = [args.vg_image_dir, args.train_h5, args.val_h5, args.vocab_json]
  return need, [p for p in need if not os.path.exists(p)]


def build_coco_dsets(args):
  """reference train.py:230-262"""
  from sg2im_amd.data import CocoSceneGraphDataset
  kw = dict(image_dir=args.coco_train_image_dir, instances_json=args.coco_train_instances_json,
            stuff_json=args.coco_train_stuff_json, stuff_only=args.coco_stuff_only, image_size=args.image_size,
            mask_size=args.mask_size, max_samples=args.num_train_samples, min_object_size=args.min_object_size,
            min_objects_per_image=args.min_objects_per_image, instance_whitelist=args.instance_whitelist,
            stuff_whitelist=args.stuff_whitelist, include_other=args.coco_include_other,
            include_relationships=args.include_relationships)
  train = CocoSceneGraphDataset(**kw)
  print('Training dataset has %d images and %d objects' % (len(train), train.total_objects()))
  print('(%.2f objects per image)' % (float(train.total_objects()) / max(len(train), 1)))
  kw.update(image_dir=args.coco_val_image_dir, instances_json=args.coco_val_instances_json,
            stuff_json=args.coco_val_stuff_json, max_samples=args.num_val_samples)
  val = CocoSceneGraphDataset(**kw)
  assert train.vocab == val.vocab
  return json.loads(json.dumps(train.vocab)), train, val


def build_vg_dsets(args):
  """reference train.py:265-285"""
  from sg2im_amd.data import VgSceneGraphDataset
  with open(args.vocab_json, 'r') as f:
    vocab = json.load(f)
  kw = dict(vocab=vocab, h5_path=args.train_h5, image_dir=args.vg_image_dir, image_size=args.image_size,
            max_samples=args.num_train_samples, max_objects=args.max_objects_per_image,
            use_orphaned_objects=args.vg_use_orphaned_objects, include_relationships=args.include_relationships)
  train = VgSceneGraphDataset(**kw)
  print('There are %d iterations per epoch' % (len(train) // args.batch_size))
  kw.update(h5_path=args.val_h5)
  del kw['max_samples']
  return vocab, train, VgSceneGraphDataset(**kw)


def build_loaders(args, rank=0):
  """reference train.py:288-306 -> (vocab, train_loader, val_loader).  Data parallel: every rank draws its own
  shuffle of the whole training set (generator seeded with seed + rank)."""
  from torch.utils.data import DataLoader
  from sg2im_amd.data import coco_collate_fn, vg_collate_fn
  if args.dataset == 'vg':
    (vocab, train, val), collate = build_vg_dsets(args), vg_collate_fn
  else:
    (vocab, train, val), collate = build_coco_dsets(args), coco_collate_fn
  # pinned host batches: CopyAhead (sg2im_amd/data/prefetch.py) copies batch k + 1 on its own stream under iteration k
  kw = dict(batch_size=args.batch_size, num_workers=args.loader_num_workers, collate_fn=collate,
            pin_memory=torch.cuda.is_available())
  gen = torch.Generator().manual_seed(args.seed + rank)
  return vocab, DataLoader(train, shuffle=True, generator=gen, **kw), DataLoader(val, shuffle=args.shuffle_val, **kw)


def as_step_tuple(batch):
  """a collated batch -> the 7-tuple (imgs, objs, boxes, masks | None, triples, obj_to_img, triple_to_img)
  (reference train.py:514-519: the VG collate has no masks)"""
  if len(batch) == 6:
    imgs, objs, boxes, triples, obj_to_img, triple_to_img = batch
    batch = (imgs, objs, boxes, None, triples, obj_to_img, triple_to_img)
  return tuple(batch)


def as_step_batch(batch, device):
  """as_step_tuple on the device (one-off batches: validation samples; the training stream goes through CopyAhead)"""
  return tuple(x.to(device, non_blocking=True) if torch.is_tensor(x) else x for x in as_step_tuple(batch))


def warn_synthetic_data(args, missing=()):
  """No dataset files here: batches are seeded synthetic scene graphs.  Say so loudly - above all when the caller
  pointed the dataset flags somewhere, which would otherwise be ignored silently."""
  given = [f for f in _DATASET_FLAGS if getattr(args, f) != parser.get_default(f)]
  print('=' * 100)
  print('WARNING: training on SYNTHETIC scene graphs (--dataset %s shape) with a fabricated vocabulary;' % args.dataset)
  print('         dataset files not found: ' + ', '.join(missing))
  if given:
    print('         IGNORED dataset flags: ' + ', '.join('--' + f for f in given))
  print('=' * 100)
  return given


def check_args(args):
  H, W = args.image_size
  for _ in args.refinement_network_dims[1:]:
    H = H // 2
  if H == 0:
    raise ValueError('Too many layers in refinement network')      # reference train.py:153-158


def calculate_model_losses(args, skip_pixel_loss, img, img_pred, bbox, bbox_pred, masks, masks_pred,
                           predicates, predicate_scores):
  """reference train.py:387-412 on the fused loss kernels; values come back as host floats"""
  total, losses = None, {}
  def add(name, v):
    nonlocal total
    total = v if total is None else total + v
    losses[name] = v
  if not skip_pixel_loss:
    add('L1_pixel_loss', L.l1_loss(img_pred, img, args.l1_pixel_loss_weight))
  add('bbox_pred', L.mse_loss(bbox_pred, bbox, args.bbox_pred_loss_weight))
  if args.predicate_pred_loss_weight > 0:
    add('predicate_pred', L.cross_entropy(predicate_scores, predicates, args.predicate_pred_loss_weight))
  if args.mask_loss_weight > 0 and masks is not None and masks_pred is not None:
    add('mask_loss', L.binary_cross_entropy(masks_pred, masks.float(), args.mask_loss_weight))
  return total, {k: float(v) for k, v in losses.items()}


def check_model(args, t, loader, model):
  """reference train.py:309-384: mean losses and box IoU over ``num_val_samples`` images plus
  sample images of the last batch under the three box / mask supervision modes.  The model is
  run as it is during training (so training-mode BatchNorm keeps updating its running stats)."""
  num_samples, total_iou, total_boxes = 0, 0.0, 0
  all_losses = defaultdict(list)
  with torch.no_grad():
    for batch in loader:
      imgs, objs, boxes, masks, triples, obj_to_img, triple_to_img = batch
      predicates = triples[:, 1].contiguous()
      N = imgs.size(0)
      imgs_pred, boxes_pred, masks_pred, predicate_scores = model(objs, triples, obj_to_img, boxes_gt=boxes,
                                                                  masks_gt=masks, num_images=N)
      _, losses = calculate_model_losses(args, False, imgs, imgs_pred, boxes, boxes_pred, masks, masks_pred,
                                         predicates, predicate_scores)
      total_iou += float(jaccard(boxes_pred, boxes))
      total_boxes += boxes_pred.size(0)
      for name, val in losses.items():
        all_losses[name].append(val)
      num_samples += N
      if num_samples >= args.num_val_samples:
        break
    samples = {'gt_img': imgs,
               'gt_box_gt_mask': model(objs, triples, obj_to_img, boxes_gt=boxes, masks_gt=masks, num_images=N)[0],
               'gt_box_pred_mask': model(objs, triples, obj_to_img, boxes_gt=boxes, num_images=N)[0],
               'pred_box_pred_mask': model(objs, triples, obj_to_img, num_images=N)[0]}
    samples = {k: imagenet_deprocess_batch(v) for k, v in samples.items()}
  mean_losses = {k: sum(v) / len(v) for k, v in all_losses.items()}
  host = lambda x: None if x is None else x.detach().cpu().clone()
  batch_data = {'objs': host(objs), 'boxes_gt': host(boxes), 'masks_gt': host(masks), 'triples': host(triples),
                'obj_to_img': host(obj_to_img), 'triple_to_img': host(triple_to_img),
                'boxes_pred': host(boxes_pred), 'masks_pred': host(masks_pred)}
  return mean_losses, samples, batch_data, total_iou / total_boxes


def main(args):
  check_args(args)
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  torch.cuda.set_device(local_rank)
  device = torch.device('cuda', local_rank)
  if world > 1:
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
  need, missing = dataset_files(args)
  real_data = not missing
  if rank == 0:
    print(args)
    if not real_data:
      warn_synthetic_data(args, missing)
  num_objs, num_preds = (184, 7) if args.dataset == 'coco' else (179, 46)
  if real_data:
    vocab, train_dl, val_dl = build_loaders(args, rank)
    if len(train_dl) == 0 or len(val_dl) == 0:
      raise ValueError('the %s dataset is empty after filtering (train: %d batches, val: %d)' % (args.dataset, len(train_dl), len(val_dl)))
  else:
    vocab, train_dl, val_dl = make_vocab(num_objs, num_preds), None, None
  gk = dict(image_size=args.image_size, embedding_dim=args.embedding_dim, gconv_dim=args.gconv_dim,
            gconv_hidden_dim=args.gconv_hidden_dim, gconv_num_layers=args.gconv_num_layers,
            mlp_normalization=args.mlp_normalization, refinement_dims=args.refinement_network_dims,
            normalization=args.normalization, activation=args.activation, mask_size=args.mask_size,
            layout_noise_dim=args.layout_noise_dim)
  dk = dict(normalization=args.d_normalization, activation=args.d_activation, padding=args.d_padding)
  lw = {k: getattr(args, k) for k in ('l1_pixel_loss_weight', 'bbox_pred_loss_weight', 'predicate_pred_loss_weight',
                                      'mask_loss_weight', 'discriminator_loss_weight', 'd_obj_weight',
                                      'd_img_weight', 'ac_loss_weight')}
  trainer = Trainer(vocab, device, generator_kwargs=gk,
                    d_obj_kwargs=dict(dk, arch=args.d_obj_arch, object_size=args.crop_size),
                    d_img_kwargs=dict(dk, arch=args.d_img_arch), loss_weights=lw,
                    learning_rate=args.learning_rate, world_size=world, seed=args.seed, rank=rank,
                    gan_loss_type=args.gan_loss_type, use_graphs=args.use_graphs,
                    bucket=(args.bucket_objects, args.bucket_triples) if args.use_graphs else None,
                    align_corners=args.align_corners)
  if args.checkpoint_start_from is not None:                        # reference train.py:162-172
    ck = torch.load(args.checkpoint_start_from, map_location='cpu', weights_only=False)
    sd = {(k[7:] if k.startswith('module.') else k): v for k, v in ck['model_state'].items()}
    trainer.model.load_state_dict(sd)
    trainer.broadcast_state()

  epoch_box = [0]
  # the epoch every training batch belongs to, in the order the batches were PRODUCED: CopyAhead pulls host batch k + 1
  # while iteration k runs, so the counter (what a checkpoint stores, reference train.py:506-514) moves when a batch is
  # CONSUMED by the loop below, not when the loader hands it out
  pending_epochs = collections.deque()

  def batches(split, start, training=False):
    """the train / val DataLoader, cycled over epochs (reference train.py:506-514; ``training``: the iterator that
    feeds the optimisation loop counts the epochs) - or its endless seeded synthetic stand-in"""
    def host_batches():
      produced = epoch_box[0]
      while real_data:
        produced += 1
        for cpu_batch in (train_dl if split == 'train' else val_dl):
          if training:
            pending_epochs.append(produced)
          yield cpu_batch
      i = start
      while True:
        i += 1
        seed = args.seed + 1000003 * i + rank + (0 if split == 'train' else 500000009)
        yield synthetic_batch(args.batch_size, image_size=args.image_size, num_objs=num_objs,
                              num_preds=num_preds, mask_size=max(args.mask_size, 1), style=args.dataset, seed=seed)
    # GPU-side input pipeline: pinned host batches, batch k + 1 copied on a stream of its own under iteration k
    from sg2im_amd.data.prefetch import CopyAhead
    return CopyAhead(host_batches(), device, finish=as_step_tuple)

  sd_of = lambda m: None if m is None else m.state_dict()           # reference train.py:634-641
  restore_path = None
  if args.restore_from_checkpoint:                                  # reference train.py:445-467
    restore_path = os.path.join(args.output_dir, '%s_with_model.pt' % args.checkpoint_name)
  if restore_path is not None and os.path.isfile(restore_path):
    print('Restoring from checkpoint:')
    print(restore_path)
    checkpoint = torch.load(restore_path, map_location='cpu', weights_only=False)
    trainer.model.load_state_dict(checkpoint['model_state'])
    trainer.opt_g.load_state_dict(checkpoint['optim_state'])
    for d, opt, key in ((trainer.d_obj, trainer.opt_do, 'd_obj'), (trainer.d_img, trainer.opt_di, 'd_img')):
      if d is not None:
        d.load_state_dict(checkpoint[key + '_state'])
        opt.load_state_dict(checkpoint[key + '_optim_state'])
    t = checkpoint['counters']['t']
    if 0 <= args.eval_mode_after <= t:
      trainer.model.eval()
    else:
      trainer.model.train()
    epoch_box[0] = checkpoint['counters']['epoch'] or 0
    trainer.broadcast_state()
  else:
    t = 0
    checkpoint = {'args': args.__dict__, 'vocab': vocab, 'model_kwargs': trainer.model_kwargs,
                  'd_obj_kwargs': trainer.d_obj_kwargs, 'd_img_kwargs': trainer.d_img_kwargs, 'losses_ts': [],
                  'losses': defaultdict(list), 'd_losses': defaultdict(list), 'checkpoint_ts': [],
                  'train_batch_data': [], 'train_samples': [], 'train_iou': [], 'val_batch_data': [],
                  'val_samples': [], 'val_losses': defaultdict(list), 'val_iou': [], 'norm_d': [], 'norm_g': [],
                  'counters': {'t': None, 'epoch': None},
                  'model_state': None, 'model_best_state': None, 'optim_state': None,
                  'd_obj_state': None, 'd_obj_best_state': None, 'd_obj_optim_state': None,
                  'd_img_state': None, 'd_img_best_state': None, 'd_img_optim_state': None, 'best_t': []}
  t0 = time.time()
  train_loader = batches('train', t, training=True)
  while t < args.num_iterations:
    if t == args.eval_mode_after:                                   # reference train.py:509-512
      if rank == 0:
        print('switching to eval mode')
      trainer.set_generator_eval()
    t += 1
    batch = next(train_loader)
    if pending_epochs:
      ep = pending_epochs.popleft()
      if ep != epoch_box[0]:
        epoch_box[0] = ep
        if rank == 0:
          print('Starting epoch %d' % ep)
    with timeit('step', args.timing):
      losses = trainer.step(batch)
    if t % args.print_every == 0:
      vals = Trainer.losses_to_host(losses)                         # the only host sync
      if world > 1:
        tv = torch.tensor([vals[k] for k in sorted(vals)], device=device)
        dist.all_reduce(tv)
        vals = dict(zip(sorted(vals), (tv / world).tolist()))
      if rank == 0:
        ips = args.batch_size * world * args.print_every / (time.time() - t0)
        print('t = %d / %d  (%.1f images/sec)' % (t, args.num_iterations, ips))
        for name, val in vals.items():
          is_d = name.startswith('d_')
          print(' %s [%s]: %.4f' % ('D' if is_d else 'G', name, val))
          checkpoint['d_losses' if is_d else 'losses'][name].append(val)
        checkpoint['losses_ts'].append(t)
      t0 = time.time()
    if t % args.checkpoint_every == 0 and rank == 0:                # reference train.py:611-661
      print('checking on train')
      t_losses, t_samples, t_batch_data, t_avg_iou = check_model(args, t, batches('train', t), trainer.model)
      checkpoint['train_batch_data'].append(t_batch_data)
      checkpoint['train_samples'].append(t_samples)
      checkpoint['checkpoint_ts'].append(t)
      checkpoint['train_iou'].append(t_avg_iou)
      print('checking on val')
      val_losses, val_samples, val_batch_data, val_avg_iou = check_model(args, t, batches('val', 0), trainer.model)
      checkpoint['val_samples'].append(val_samples)
      checkpoint['val_batch_data'].append(val_batch_data)
      checkpoint['val_iou'].append(val_avg_iou)
      print('train iou: ', t_avg_iou)
      print('val iou: ', val_avg_iou)
      for k, v in val_losses.items():
        checkpoint['val_losses'][k].append(v)
      checkpoint.update(model_state=trainer.model.state_dict(), d_obj_state=sd_of(trainer.d_obj),
                        d_img_state=sd_of(trainer.d_img), optim_state=trainer.opt_g.state_dict(),
                        d_obj_optim_state=sd_of(trainer.opt_do), d_img_optim_state=sd_of(trainer.opt_di))
      checkpoint['counters']['t'] = t
      checkpoint['counters']['epoch'] = epoch_box[0]
      path = os.path.join(args.output_dir, '%s_with_model.pt' % args.checkpoint_name)
      print('Saving checkpoint to ', path)
      torch.save(checkpoint, path)
      # a second checkpoint without any model or optimiser state (reference train.py:651-661)
      small = {k: v for k, v in checkpoint.items()
               if not (k.endswith('_state') or k.endswith('_best_state'))}
      torch.save(small, os.path.join(args.output_dir, '%s_no_model.pt' % args.checkpoint_name))
    if t % args.checkpoint_every == 0 and world > 1:
      # rank 0's check_model ran the networks in training mode (like the reference's): its BatchNorm
      # running statistics moved; bring every replica back in line
      trainer.broadcast_state()
  if rank == 0 and args.use_graphs:
    print('hipGraph statistics:', trainer.graph_stats)
  if world > 1:
    dist.destroy_process_group()


if __name__ == '__main__':
  main(parser.parse_args())
