"""Generate golden vectors from the *imported reference* (runs only in the build
container, where /root/reference exists).

  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

For each small configuration it instantiates the reference's own Sg2ImModel /
AcCropDiscriminator / PatchDiscriminator (sg2im/model.py, sg2im/discriminators.py),
runs the loop body of scripts/train.py:524-592 once (losses + backward, no optimiser
step) on a seeded synthetic batch, and stores: the state_dicts *before* the step,
the batch, the injected layout noise, every forward output, every loss, and the
gradient of every parameter.  The fixtures pin oracle/sg2im_oracle.py (CPU tests)
which in turn checks the HIP path (GPU tests).
"""
import contextlib
import io
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')

import torch
import torch.nn.functional as F

from sg2im.model import Sg2ImModel                       # noqa: E402  (reference)
from sg2im.discriminators import PatchDiscriminator, AcCropDiscriminator   # noqa: E402
from sg2im.losses import get_gan_losses                  # noqa: E402
from sg2im.layout import boxes_to_layout, masks_to_layout   # noqa: E402
from sg2im.bilinear import crop_bbox_batch               # noqa: E402
from sg2im.graph import GraphTripleConv                  # noqa: E402

from sg2im_amd.synthetic import make_vocab, synthetic_batch   # noqa: E402

CONFIGS = {
  # COCO-style: GT masks feed the layout, mask_net gets no gradient.
  'tiny_coco': dict(
    batch=dict(batch_size=3, image_size=(16, 16), num_objs=12, num_preds=5, min_objs=2,
               max_objs=4, mask_size=4, style='coco', seed=11),
    g=dict(image_size=(16, 16), embedding_dim=16, gconv_dim=16, gconv_hidden_dim=32,
           gconv_num_layers=3, refinement_dims=(32, 16), normalization='batch',
           activation='leakyrelu-0.2', mask_size=4, layout_noise_dim=4),
    d_obj=dict(arch='C4-8-2,C4-16-2', normalization='batch', activation='leakyrelu-0.2',
               padding='valid', object_size=16),
    d_img=dict(arch='C4-8-2,C4-16-2', normalization='batch', activation='leakyrelu-0.2',
               padding='valid'),
  ),
  # generator without normalisation layers in the refinement network (--normalization none)
  'tiny_coco_nonorm': dict(
    batch=dict(batch_size=2, image_size=(16, 16), num_objs=10, num_preds=4, min_objs=2,
               max_objs=3, mask_size=4, style='coco', seed=37),
    g=dict(image_size=(16, 16), embedding_dim=16, gconv_dim=16, gconv_hidden_dim=32,
           gconv_num_layers=2, refinement_dims=(24, 16, 8), normalization='none',
           activation='leakyrelu-0.2', mask_size=4, layout_noise_dim=4),
    d_obj=dict(arch='C4-8-2,C4-16-2', normalization='none', activation='leakyrelu-0.2',
               padding='valid', object_size=16),
    d_img=dict(arch='C3-8-2,C3-16-2,C3-16', normalization='none', activation='leakyrelu-0.2',
               padding='same'),
  ),
  # BatchNorm1d inside every build_mlp (--mlp_normalization batch, sg2im/layers.py:224-225)
  'tiny_coco_mlpbn': dict(
    batch=dict(batch_size=4, image_size=(16, 16), num_objs=10, num_preds=4, min_objs=3,
               max_objs=5, mask_size=4, style='coco', seed=53),
    g=dict(image_size=(16, 16), embedding_dim=16, gconv_dim=16, gconv_hidden_dim=32,
           gconv_num_layers=2, mlp_normalization='batch', refinement_dims=(24, 16),
           normalization='batch', activation='leakyrelu-0.2', mask_size=4, layout_noise_dim=4),
    d_obj=dict(arch='C4-8-2,C4-16-2', normalization='batch', activation='leakyrelu-0.2',
               padding='valid', object_size=16),
    d_img=dict(arch='C4-8-2,C4-16-2', normalization='batch', activation='leakyrelu-0.2',
               padding='valid'),
  ),
  # InstanceNorm2d in the refinement network and both discriminators (--normalization instance,
  # --d_normalization instance; sg2im/layers.py:27-28)
  'tiny_coco_instnorm': dict(
    batch=dict(batch_size=3, image_size=(16, 16), num_objs=10, num_preds=4, min_objs=2,
               max_objs=4, mask_size=4, style='coco', seed=71),
    g=dict(image_size=(16, 16), embedding_dim=16, gconv_dim=16, gconv_hidden_dim=32,
           gconv_num_layers=2, refinement_dims=(24, 16, 8), normalization='instance',
           activation='leakyrelu-0.2', mask_size=4, layout_noise_dim=4),
    d_obj=dict(arch='C4-8-2,C4-16-2', normalization='instance', activation='leakyrelu-0.2',
               padding='valid', object_size=16),
    d_img=dict(arch='C4-8-2,C4-16-2', normalization='instance', activation='leakyrelu-0.2',
               padding='valid'),
  ),
  # discriminators built from the other build_cnn tokens (R, P, U, FC; sg2im/layers.py:183-206)
  'tiny_coco_archtokens': dict(
    batch=dict(batch_size=3, image_size=(16, 16), num_objs=10, num_preds=4, min_objs=2,
               max_objs=4, mask_size=4, style='coco', seed=83),
    g=dict(image_size=(16, 16), embedding_dim=16, gconv_dim=16, gconv_hidden_dim=32,
           gconv_num_layers=2, refinement_dims=(24, 16), normalization='batch',
           activation='leakyrelu-0.2', mask_size=4, layout_noise_dim=4),
    d_obj=dict(arch='R,C3-8-2,P2,FC-128-32', normalization='batch', activation='leakyrelu-0.2',
               padding='same', object_size=16, pooling='max'),
    d_img=dict(arch='C3-8,R,P2,C3-16-2,U2,R', normalization='batch', activation='leakyrelu-0.2',
               padding='same', pooling='avg'),
  ),
  # VG-style: no GT masks -> masks_pred feeds the layout and mask_net trains.
  'tiny_vg': dict(
    batch=dict(batch_size=2, image_size=(32, 32), num_objs=9, num_preds=6, min_objs=3,
               max_objs=5, mask_size=8, style='vg', seed=23),
    g=dict(image_size=(32, 32), embedding_dim=32, gconv_dim=32, gconv_hidden_dim=64,
           gconv_num_layers=2, refinement_dims=(32, 16, 8), normalization='batch',
           activation='leakyrelu-0.2', mask_size=8, layout_noise_dim=0),
    d_obj=dict(arch='C4-8-2,C4-16-2', normalization='batch', activation='leakyrelu-0.2',
               padding='valid', object_size=16),
    d_img=dict(arch='C4-8-2,C4-16-2,C4-32-2', normalization='batch',
               activation='leakyrelu-0.2', padding='valid'),
  ),
}

W = dict(l1=1.0, bbox=10.0, d=0.01, d_obj=1.0, d_img=1.0, ac=0.1)   # scripts/train.py:108-131


def quiet(fn, *a, **k):
  with contextlib.redirect_stdout(io.StringIO()):   # build_cnn prints every layer (layers.py:211-212)
    return fn(*a, **k)


def clone_sd(m, only_buffers=False):
  return {k: v.detach().clone() for k, v in m.state_dict().items()
          if not only_buffers or 'running_' in k or 'num_batches' in k}


def grads_of(m):
  return {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in m.named_parameters()}


def run_config(name, cfg):
  bc = cfg['batch']
  vocab = make_vocab(bc['num_objs'], bc['num_preds'])
  batch = synthetic_batch(**bc)
  imgs, objs, boxes, masks, triples, obj_to_img, _ = batch
  torch.manual_seed(1234)
  G = Sg2ImModel(vocab, **cfg['g'])
  Do = quiet(AcCropDiscriminator, vocab, **cfg['d_obj'])
  Di = quiet(PatchDiscriminator, **cfg['d_img'])
  # non-trivial BN affine parameters so gamma/beta paths are exercised
  g = torch.Generator().manual_seed(99)
  for m in list(G.modules()) + list(Do.modules()) + list(Di.modules()):
    if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
      m.weight.data = 0.5 + torch.rand(m.weight.shape, generator=g)
      m.bias.data = 0.2 * torch.randn(m.bias.shape, generator=g)
  for m in (G, Do, Di):
    m.train()
  sd0 = dict(G=clone_sd(G), Do=clone_sd(Do), Di=clone_sd(Di))

  nd = cfg['g']['layout_noise_dim']
  H, Wd = cfg['g']['image_size']
  noise = torch.randn(imgs.size(0), nd, H, Wd, generator=g) if nd > 0 else None
  real_randn = torch.randn
  if nd > 0:
    torch.randn = lambda *a, **k: noise.clone()       # model.py:167
  try:
    out = G(objs, triples, obj_to_img, boxes_gt=boxes, masks_gt=masks)
  finally:
    torch.randn = real_randn
  imgs_pred, boxes_pred, masks_pred, rel_scores = out
  gan_g, gan_d = get_gan_losses('gan')

  # --- generator loss, scripts/train.py:533-550
  l1 = F.l1_loss(imgs_pred, imgs) * W['l1']
  lb = F.mse_loss(boxes_pred, boxes) * W['bbox']
  sf_obj, ac = Do(imgs_pred, objs, boxes, obj_to_img)
  l_ac = ac * W['ac']
  l_go = gan_g(sf_obj) * (W['d'] * W['d_obj'])
  sf_img = Di(imgs_pred)
  l_gi = gan_g(sf_img) * (W['d'] * W['d_img'])
  total = l1 + lb + l_ac + l_go + l_gi
  for m in (G, Do, Di):
    m.zero_grad()
  total.backward()
  g_grads = grads_of(G)
  sd_after_g = dict(G=clone_sd(G, True), Do=clone_sd(Do, True), Di=clone_sd(Di, True))

  # --- D_obj loss, scripts/train.py:566-575
  fake = imgs_pred.detach()
  Do.zero_grad()
  sf, ac_f = Do(fake, objs, boxes, obj_to_img)
  sr, ac_r = Do(imgs, objs, boxes, obj_to_img)
  ld_obj = gan_d(sr, sf) + ac_r + ac_f
  ld_obj.backward()
  do_grads = grads_of(Do)
  # --- D_img loss, scripts/train.py:581-588
  Di.zero_grad()
  sfi = Di(fake)
  sri = Di(imgs)
  ld_img = gan_d(sri, sfi)
  ld_img.backward()
  di_grads = grads_of(Di)

  # --- stand-alone op vectors
  torch.manual_seed(5)
  D = cfg['g']['gconv_dim']
  O = objs.size(0)
  vecs = torch.randn(O, D)
  soft_masks = torch.rand(O, bc['mask_size'], bc['mask_size'])
  ops = dict(
    vecs=vecs, soft_masks=soft_masks,
    boxes_to_layout=boxes_to_layout(vecs, boxes, obj_to_img, H, Wd),
    masks_to_layout_soft=masks_to_layout(vecs, boxes, soft_masks, obj_to_img, H, Wd),
    crops=crop_bbox_batch(imgs, boxes, obj_to_img, cfg['d_obj']['object_size']),
  )
  if masks is not None:
    ops['masks_to_layout_gt'] = masks_to_layout(vecs, boxes, masks, obj_to_img, H, Wd)
  gc = GraphTripleConv(D, output_dim=D, hidden_dim=2 * D, pooling='sum')
  pv = torch.randn(triples.size(0), D)
  edges = torch.stack([triples[:, 0], triples[:, 2]], dim=1)
  no, np_ = gc(vecs, pv, edges)
  ops.update(gconv_sum_sd=clone_sd(gc), gconv_sum_pred_in=pv,
             gconv_sum_obj_out=no.detach(), gconv_sum_pred_out=np_.detach())

  fix = dict(
    name=name, config=cfg, vocab=vocab, batch=batch, noise=noise, weights=W,
    state_before=sd0, state_after_g_forward=sd_after_g,
    outputs=dict(imgs_pred=imgs_pred.detach(), boxes_pred=boxes_pred.detach(),
                 masks_pred=None if masks_pred is None else masks_pred.detach(),
                 rel_scores=rel_scores.detach(), d_obj_scores_fake=sf_obj.detach(),
                 d_img_scores_fake=sf_img.detach()),
    losses=dict(l1=l1.item(), bbox=lb.item(), ac=l_ac.item(), g_gan_obj=l_go.item(),
                g_gan_img=l_gi.item(), total=total.item(), d_obj=ld_obj.item(),
                d_img=ld_img.item(), d_ac_real=ac_r.item(), d_ac_fake=ac_f.item()),
    grads=dict(G=g_grads, Do=do_grads, Di=di_grads),
    ops={k: (v.detach() if torch.is_tensor(v) else v) for k, v in ops.items()},
    torch_version=torch.__version__,
  )
  path = os.path.join(HERE, name + '.pt')
  torch.save(fix, path)
  print('wrote %s (%.1f KB)' % (path, os.path.getsize(path) / 1024.0))


def run_eval_config(name, cfg):
  """Generator in eval() mode (scripts/train.py:509-512 switches to it after
  `eval_mode_after` iterations): BatchNorm uses its running statistics, the discriminators
  stay in train mode.  Stored as <name>_eval.pt: state, outputs, generator losses and grads."""
  bc = cfg['batch']
  vocab = make_vocab(bc['num_objs'], bc['num_preds'])
  batch = synthetic_batch(**bc)
  imgs, objs, boxes, masks, triples, obj_to_img, _ = batch
  torch.manual_seed(4321)
  G = Sg2ImModel(vocab, **cfg['g'])
  Do = quiet(AcCropDiscriminator, vocab, **cfg['d_obj'])
  Di = quiet(PatchDiscriminator, **cfg['d_img'])
  g = torch.Generator().manual_seed(77)
  for m in list(G.modules()) + list(Do.modules()) + list(Di.modules()):
    if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
      m.weight.data = 0.5 + torch.rand(m.weight.shape, generator=g)
      m.bias.data = 0.2 * torch.randn(m.bias.shape, generator=g)
      # running statistics as after some training: not the (0, 1) initial values
      m.running_mean.data = 0.3 * torch.randn(m.running_mean.shape, generator=g)
      m.running_var.data = 0.5 + torch.rand(m.running_var.shape, generator=g)
  G.eval(); Do.train(); Di.train()
  sd0 = dict(G=clone_sd(G), Do=clone_sd(Do), Di=clone_sd(Di))
  nd = cfg['g']['layout_noise_dim']
  H, Wd = cfg['g']['image_size']
  noise = torch.randn(imgs.size(0), nd, H, Wd, generator=g) if nd > 0 else None
  real_randn = torch.randn
  if nd > 0:
    torch.randn = lambda *a, **k: noise.clone()
  try:
    imgs_pred, boxes_pred, masks_pred, rel_scores = G(objs, triples, obj_to_img, boxes_gt=boxes, masks_gt=masks)
  finally:
    torch.randn = real_randn
  gan_g, _ = get_gan_losses('gan')
  l1 = F.l1_loss(imgs_pred, imgs) * W['l1']
  lb = F.mse_loss(boxes_pred, boxes) * W['bbox']
  sf_obj, ac = Do(imgs_pred, objs, boxes, obj_to_img)
  sf_img = Di(imgs_pred)
  total = l1 + lb + ac * W['ac'] + gan_g(sf_obj) * (W['d'] * W['d_obj']) + gan_g(sf_img) * (W['d'] * W['d_img'])
  for m in (G, Do, Di):
    m.zero_grad()
  total.backward()
  fix = dict(
    name=name + '_eval', config=cfg, vocab=vocab, batch=batch, noise=noise, weights=W, state_before=sd0,
    state_after_g_forward=dict(G=clone_sd(G, True)),        # must be unchanged in eval mode
    outputs=dict(imgs_pred=imgs_pred.detach(), boxes_pred=boxes_pred.detach(),
                 masks_pred=None if masks_pred is None else masks_pred.detach(),
                 rel_scores=rel_scores.detach(), d_obj_scores_fake=sf_obj.detach(),
                 d_img_scores_fake=sf_img.detach()),
    losses=dict(l1=l1.item(), bbox=lb.item(), ac=(ac * W['ac']).item(), total=total.item()),
    grads=dict(G=grads_of(G)), torch_version=torch.__version__,
  )
  path = os.path.join(HERE, name + '_eval.pt')
  torch.save(fix, path)
  print('wrote %s (%.1f KB)' % (path, os.path.getsize(path) / 1024.0))


if __name__ == '__main__':
  which = sys.argv[1:] or ['train', 'eval']
  only = [w for w in which if w in CONFIGS]
  for n, c in CONFIGS.items():
    if only and n not in only:
      continue
    if 'train' in which or only:
      run_config(n, c)
    if 'eval' in which and n in ('tiny_coco', 'tiny_vg'):
      run_eval_config(n, c)
