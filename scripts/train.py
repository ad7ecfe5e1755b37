#!/usr/bin/env python
"""Training entry point with the reference's option surface (reference scripts/train.py:47-140)
driving sg2im_amd.trainer.Trainer on MI355X; one process per GPU under
``python -m torch.distributed.run --nproc-per-node N scripts/train.py ...``.

Datasets are outside the hot-path scope (COCO/VG loaders need
torchvision/h5py/pycocotools and the data); batches come from the seeded synthetic generator
that reproduces the collate layout (``--dataset coco|vg`` selects the graph style).  Every
reference flag is accepted; flags whose feature is not on the HIP path yet fail loudly.
"""
import argparse
import math
import os
import sys
import time
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
import torch.distributed as dist

from sg2im_amd.synthetic import make_vocab, synthetic_batch
from sg2im_amd.trainer import Trainer
from sg2im_amd.utils import bool_flag, int_tuple, str_tuple, timeit

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


def check_args(args):
  H, W = args.image_size
  for _ in args.refinement_network_dims[1:]:
    H = H // 2
  if H == 0:
    raise ValueError('Too many layers in refinement network')      # reference train.py:153-158


def main(args):
  check_args(args)
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  torch.cuda.set_device(local_rank)
  device = torch.device('cuda', local_rank)
  if world > 1:
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
  if rank == 0:
    print(args)
  num_objs, num_preds = (184, 7) if args.dataset == 'coco' else (179, 46)
  vocab = make_vocab(num_objs, num_preds)
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
                    learning_rate=args.learning_rate, world_size=world, seed=args.seed,
                    gan_loss_type=args.gan_loss_type)
  if args.checkpoint_start_from is not None:                        # reference train.py:162-172
    ck = torch.load(args.checkpoint_start_from, map_location='cpu', weights_only=False)
    sd = {(k[7:] if k.startswith('module.') else k): v for k, v in ck['model_state'].items()}
    trainer.model.load_state_dict(sd)
  checkpoint = {'args': args.__dict__, 'vocab': vocab, 'model_kwargs': trainer.model_kwargs,
                'd_obj_kwargs': trainer.d_obj_kwargs, 'd_img_kwargs': trainer.d_img_kwargs, 'losses_ts': [],
                'losses': defaultdict(list), 'checkpoint_ts': [], 'counters': {'t': None, 'epoch': None}}
  t, t0 = 0, time.time()
  style = args.dataset
  while t < args.num_iterations:
    if t == args.eval_mode_after:                                   # reference train.py:509-512
      if rank == 0:
        print('switching to eval mode')
      trainer.set_generator_eval()
    t += 1
    cpu_batch = synthetic_batch(args.batch_size, image_size=args.image_size, num_objs=num_objs, num_preds=num_preds,
                                mask_size=max(args.mask_size, 1), style=style, seed=args.seed + 1000003 * t + rank)
    batch = tuple(x.to(device, non_blocking=True) if torch.is_tensor(x) else x for x in cpu_batch)
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
          tag = 'G' if not name.startswith('d_') else 'D'
          print(' %s [%s]: %.4f' % (tag, name, val))
          checkpoint['losses'][name].append(val)
        checkpoint['losses_ts'].append(t)
      t0 = time.time()
    if t % args.checkpoint_every == 0 and rank == 0:                # reference train.py:611-661
      sd_of = lambda m: None if m is None else m.state_dict()     # reference train.py:634-641
      checkpoint.update(model_state=trainer.model.state_dict(), d_obj_state=sd_of(trainer.d_obj),
                        d_img_state=sd_of(trainer.d_img), optim_state=trainer.opt_g.state_dict(),
                        d_obj_optim_state=sd_of(trainer.opt_do), d_img_optim_state=sd_of(trainer.opt_di))
      checkpoint['counters']['t'] = t
      checkpoint['checkpoint_ts'].append(t)
      path = os.path.join(args.output_dir, '%s_with_model.pt' % args.checkpoint_name)
      torch.save(checkpoint, path)
      print('Saved checkpoint to', path)
  if world > 1:
    dist.destroy_process_group()


if __name__ == '__main__':
  main(parser.parse_args())
