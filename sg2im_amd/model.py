"""Sg2ImModel with the reference's constructor / forward / state_dict surface
(reference sg2im/model.py) running on hand-written HIP kernels for gfx950."""
import torch
import torch.nn as nn

from . import functional as HF
from . import ops
from .crn import RefinementNetwork
from .graph import GraphTripleConv, GraphTripleConvNet
from .layers import build_mlp, to_channels_last
from .layout import ALIGN_CORNERS, layout_nhwc

import os as _os
EMB_CSR_AHEAD = _os.environ.get('SG2IM_EMB_CSR_AHEAD', '1') != '0'      # (A/B knob)
TRIPLES_CSR = _os.environ.get('SG2IM_TRIPLES_CSR', '1') != '0'          # (A/B knob: 0 = three strided copies + sg2im_csr_build)


class Sg2ImModel(nn.Module):
  def __init__(self, vocab, image_size=(64, 64), embedding_dim=64, gconv_dim=128, gconv_hidden_dim=512,
               gconv_pooling='avg', gconv_num_layers=5, refinement_dims=(1024, 512, 256, 128, 64),
               normalization='batch', activation='leakyrelu-0.2', mask_size=None,
               mlp_normalization='none', layout_noise_dim=0, align_corners=ALIGN_CORNERS, **kwargs):
    """align_corners (not a reference argument): the bilinear sampling convention of the layout
    (reference sg2im/layout.py:53,88 call F.grid_sample without it): False = torch >= 1.3, what the
    reference computes under a current torch; True = torch 0.4, what the authors trained their
    released checkpoints with (SURVEY.md section 8c caveat i)."""
    super(Sg2ImModel, self).__init__()
    if len(kwargs) > 0:      # reference sg2im/model.py:41-42
      print('WARNING: Model got unexpected kwargs ', kwargs)
    self.vocab = vocab
    self.image_size = image_size
    self.layout_noise_dim = layout_noise_dim
    self.align_corners = bool(align_corners)

    num_objs = len(vocab['object_idx_to_name'])
    num_preds = len(vocab['pred_idx_to_name'])
    self.obj_embeddings = nn.Embedding(num_objs + 1, embedding_dim)
    self.pred_embeddings = nn.Embedding(num_preds, embedding_dim)

    if gconv_num_layers == 0:
      self.gconv = nn.Linear(embedding_dim, gconv_dim)
    elif gconv_num_layers > 0:
      self.gconv = GraphTripleConv(input_dim=embedding_dim, output_dim=gconv_dim, hidden_dim=gconv_hidden_dim,
                                   pooling=gconv_pooling, mlp_normalization=mlp_normalization)
    self.gconv_net = None
    if gconv_num_layers > 1:
      self.gconv_net = GraphTripleConvNet(input_dim=gconv_dim, hidden_dim=gconv_hidden_dim,
                                          pooling=gconv_pooling, num_layers=gconv_num_layers - 1,
                                          mlp_normalization=mlp_normalization)

    self.box_net = build_mlp([gconv_dim, gconv_hidden_dim, 4], batch_norm=mlp_normalization)
    self.mask_net = None
    if mask_size is not None and mask_size > 0:
      self.mask_net = self._build_mask_net(num_objs, gconv_dim, mask_size)
    self.rel_aux_net = build_mlp([2 * embedding_dim + 8, gconv_hidden_dim, num_preds],
                                 batch_norm=mlp_normalization)
    self.refinement_net = RefinementNetwork(dims=(gconv_dim + layout_noise_dim,) + tuple(refinement_dims),
                                            normalization=normalization, activation=activation)

  def _build_mask_net(self, num_objs, dim, mask_size):
    """Container with the Sequential indices of reference sg2im/model.py:94-106."""
    layers, cur = [], 1
    while cur < mask_size:
      layers += [nn.Upsample(scale_factor=2, mode='nearest'), nn.BatchNorm2d(dim),
                 nn.Conv2d(dim, dim, kernel_size=3, padding=1), nn.ReLU()]
      cur *= 2
    if cur != mask_size:
      raise ValueError('Mask size must be a power of 2')
    layers.append(nn.Conv2d(dim, 1, kernel_size=1))
    return to_channels_last(nn.Sequential(*layers))

  def _run_mask_net(self, obj_vecs, obj_count=None):
    bns = [m for m in self.mask_net if isinstance(m, nn.BatchNorm2d)]
    convs = [m for m in self.mask_net if isinstance(m, nn.Conv2d)]
    params = []
    for bn, cv in zip(bns, convs[:-1]):
      params += [bn.weight, bn.bias, cv.weight, cv.bias]
    params += [convs[-1].weight, convs[-1].bias]
    return HF.MaskNetFn.apply(obj_vecs, bns, self.training, obj_count, *params)

  def forward_nhwc(self, objs, triples, obj_to_img=None, boxes_gt=None, masks_gt=None, num_images=None,
                   obj_count=None, triple_count=None, aux_stream=None, detach_masks=False, detach_rel=False,
                   aux_work=None):
    """Same computation as ``forward`` (reference sg2im/model.py:108-171) but the image
    is returned NHWC, the internal layout of the kernels.  ``num_images`` avoids the host
    sync of reference sg2im/layout.py:143 (N = obj_to_img.max()+1).  ``obj_count``: (int32 device
    scalar, 1) when the object / triple axes are padded to a bucket size (sg2im_amd/bucketing.py):
    the batch statistics of mask_net - and of the MLPs' BatchNorm1d layers under mlp_normalization='batch' -
    then only see the real objects / triples.

    ``aux_stream`` (the Trainer passes its side stream while it captures the iteration): everything that is NOT
    on the path embeddings -> graph convolutions -> layout -> refinement network runs there, next to that path
    instead of in front of it: the layout noise and the per-image object lists (need only the inputs), and -
    when the caller does not back-propagate through them (``detach_masks``: ground-truth masks are laid out and
    the mask loss is off, as in the reference's COCO configuration; ``detach_rel``: the predicate loss is off) -
    mask_net and rel_aux_net, whose outputs are then returned detached.  [mask_net is 0.37 ms of the 0.9 ms the
    refinement network used to wait for at the head of every step.]  ``aux_work``: a callable of the caller's that
    the refinement network depends on (the Trainer: the bf16 weight mirror refresh, 28 us over 180 MB) - issued on
    ``aux_stream`` BEHIND the small launches at the head of the step (issued first, its 8192 workgroups kept the
    one-workgroup CSR build waiting for a free CU: +25 us on the critical path) and joined behind the graph convolutions."""
    O = objs.size(0)
    if obj_to_img is None:
      obj_to_img = torch.zeros(O, dtype=objs.dtype, device=objs.device)
      num_images = 1
    H, W = self.image_size
    main = torch.cuda.current_stream() if aux_stream is not None else None
    noise = img_csr = ev_pre = ev_work = None
    emb_csr = (None, None)            # the CSRs the embeddings' backward sums over, built ahead of time
    if aux_stream is not None and num_images is not None:
      aux_stream.wait_stream(main)
    # (s, p, o = triples.chunk(3, dim=1) of model.py:116-118 and the pooling CSR of the graph convolutions: one launch;
    # padded batch: the padding triples stay out of the CSR - no long tail row on the dummy object)
    if TRIPLES_CSR:
      s, p, o, pool_csr = ops.triples_csr(triples, O, live=triple_count)
    else:
      s, p, o = (triples[:, c].contiguous() for c in range(3))
      pool_csr = ops.Csr(s, o, O, live=triple_count)
    if aux_stream is not None and num_images is not None:
      with torch.cuda.stream(aux_stream):
        if self.layout_noise_dim > 0:                           # reference sg2im/model.py:164-168
          noise = torch.randn((num_images, self.layout_noise_dim, H, W), dtype=torch.float32, device=objs.device)
        img_csr = ops.Csr(obj_to_img, None, num_images)
        ev_pre = torch.cuda.Event()
        ev_pre.record(aux_stream)
        # (not waited for by the layout: joined with everything else of this stream at the end of the forward pass)
        if torch.is_grad_enabled() and self.obj_embeddings.weight.requires_grad and EMB_CSR_AHEAD:
          aux_stream.wait_stream(main)                          # (p)
          emb_csr = (ops.Csr(objs, None, self.obj_embeddings.weight.size(0)),
                     ops.Csr(p, None, self.pred_embeddings.weight.size(0)))
        if aux_work is not None:
          aux_stream.wait_stream(main)                          # (behind the pooling CSR)
          aux_work()
          ev_work = torch.cuda.Event()
          ev_work.record(aux_stream)
    elif aux_work is not None:
      aux_work()
    edges = (s, o, pool_csr)

    ops.mark('csr_done')
    obj_vecs = HF.Embedding.apply(self.obj_embeddings.weight, objs, emb_csr[0])
    obj_vecs_orig = obj_vecs
    pred_vecs = HF.Embedding.apply(self.pred_embeddings.weight, p, emb_csr[1])
    stack = self._gconv_stack()
    if stack is not None:
      # every GraphTripleConv layer in ONE persistent launch (csrc/gcn_persist.hip)
      params = [t for g in stack for lin in (g.net1.linears() + g.net2.linears()) for t in (lin.weight, lin.bias)]
      obj_vecs, pred_vecs = HF.GraphTripleConvStackFn.apply(obj_vecs, pred_vecs, edges[0], edges[1], edges[2],
                                                           stack[0].pooling == 'avg', *params)
    else:
      if isinstance(self.gconv, nn.Linear):
        obj_vecs = HF.LinearAct.apply(obj_vecs, self.gconv.weight, self.gconv.bias, 1.0)
      else:
        obj_vecs, pred_vecs = self.gconv(obj_vecs, pred_vecs, edges, (obj_count, triple_count))
      if self.gconv_net is not None:
        obj_vecs, pred_vecs = self.gconv_net(obj_vecs, pred_vecs, edges, (obj_count, triple_count))

    ops.mark('gcn_layers_done')
    if ev_work is not None:
      main.wait_event(ev_work)                                  # (mask_net's convolutions are the first to read the mirror)
    masks_pred = rel_scores = None
    on_aux = aux_stream is not None and ((self.mask_net is not None and detach_masks) or detach_rel)
    if on_aux:
      aux_stream.wait_stream(main)
    if on_aux and self.mask_net is not None and detach_masks:
      with torch.cuda.stream(aux_stream), torch.no_grad():
        masks_pred = self._run_mask_net(obj_vecs.detach(), obj_count)
    boxes_pred = self.box_net(obj_vecs, obj_count)
    if self.mask_net is not None and masks_pred is None:
      masks_pred = self._run_mask_net(obj_vecs, obj_count)

    r1, r2 = self.rel_aux_net.linears()
    if on_aux and detach_rel and not self.rel_aux_net.norms():
      aux_stream.wait_stream(main)               # (boxes_pred)
      with torch.cuda.stream(aux_stream), torch.no_grad():
        rel_scores = HF.RelAux.apply(boxes_pred.detach(), obj_vecs_orig.detach(), s, o, edges[2], r1.weight, r1.bias,
                                     r2.weight, r2.bias)
    elif self.rel_aux_net.norms():       # mlp_normalization='batch'
      h = self.rel_aux_net.tail(HF.RelAuxLinear.apply(boxes_pred, obj_vecs_orig, s, o, edges[2], r1.weight,
                                                      r1.bias, self.training), 0, triple_count)
      rel_scores = self.rel_aux_net.tail(HF.LinearAct.apply(h, r2.weight, r2.bias, 1.0, self.training), 1,
                                         triple_count)
    else:
      rel_scores = HF.RelAux.apply(boxes_pred, obj_vecs_orig, s, o, edges[2], r1.weight, r1.bias, r2.weight, r2.bias)

    layout_boxes = boxes_pred if boxes_gt is None else boxes_gt
    layout_masks = None
    if masks_pred is not None:
      layout_masks = masks_pred if masks_gt is None else masks_gt
    if num_images is None:
      num_images = int(obj_to_img.max().item()) + 1          # reference sg2im/layout.py:143
    if ev_pre is not None:
      main.wait_event(ev_pre)                                 # (noise + per-image object lists of the aux stream)
    elif self.layout_noise_dim > 0:                           # reference sg2im/model.py:164-168
      noise = torch.randn((num_images, self.layout_noise_dim, H, W), dtype=obj_vecs.dtype,
                          device=obj_vecs.device)
    ops.mark('gcn_done')
    # (the layout has exactly one consumer, the refinement network: pyramid forward / per-level gradients backward
    # are handed over through an explicit per-forward link, functional.LayoutLink)
    link = HF.LayoutLink()
    layout = layout_nhwc(obj_vecs, layout_boxes, layout_masks, obj_to_img, H, W, noise=noise,
                         n_images=num_images, align_corners=self.align_corners, img_csr=img_csr,
                         pyramid_levels=len(self.refinement_net.refinement_modules) - 1, link=link)
    # the appended noise channels need no gradient: only the first D layout channels do
    img = self.refinement_net.forward_nhwc(layout, layout_grad_channels=obj_vecs.size(1), link=link)
    if aux_stream is not None:
      main.wait_stream(aux_stream)                            # (the detached outputs; long finished by now)
    return img, boxes_pred, masks_pred, rel_scores

  def _gconv_stack(self):
    """the GraphTripleConv layers as a list when the persistent stack kernels can run them (no MLP normalisation,
    one pooling mode, feature sizes that are multiples of 32), else None (layer-by-layer launches)"""
    from .graph import GraphTripleConv
    if not isinstance(self.gconv, GraphTripleConv):
      return None
    layers = [self.gconv] + (list(self.gconv_net.gconvs) if self.gconv_net is not None else [])
    if any(g.net1.norms() or g.net2.norms() or g.pooling != layers[0].pooling for g in layers):
      return None
    if not ops.gconv_stack_supported([(g.input_dim, g.hidden_dim, g.output_dim) for g in layers]):
      return None
    return layers

  def forward(self, objs, triples, obj_to_img=None, boxes_gt=None, masks_gt=None, num_images=None):
    """Required: objs (O,) int64 categories, triples (T,3) int64 [s,p,o].  Optional:
    obj_to_img (O,), boxes_gt (O,4), masks_gt (O,M,M).  Returns (img (N,3,H,W),
    boxes_pred (O,4), masks_pred (O,M,M) | None, rel_scores (T,P))."""
    img, boxes_pred, masks_pred, rel_scores = self.forward_nhwc(objs, triples, obj_to_img, boxes_gt,
                                                                masks_gt, num_images)
    return HF.NhwcToNchw.apply(img), boxes_pred, masks_pred, rel_scores

  def encode_scene_graphs(self, scene_graphs):
    """Scene-graph dict(s) {'objects': [...], 'relationships': [[s, pred, o], ...]} ->
    (objs, triples, obj_to_img) LongTensors on the model's device.  Like the reference
    (sg2im/model.py:173-227) this appends '__image__' / '__in_image__' to the *caller's*
    dicts and raises ValueError on out-of-vocabulary names."""
    if isinstance(scene_graphs, dict):
      scene_graphs = [scene_graphs]
    name_to_obj = self.vocab['object_name_to_idx']
    name_to_pred = self.vocab['pred_name_to_idx']
    objs, triples, obj_to_img = [], [], []
    offset = 0
    for img_idx, sg in enumerate(scene_graphs):
      sg['objects'].append('__image__')
      image_node = len(sg['objects']) - 1
      sg['relationships'].extend([j, '__in_image__', image_node] for j in range(image_node))
      for name in sg['objects']:
        if name not in name_to_obj:
          raise ValueError('Object "%s" not in vocab' % name)
        objs.append(name_to_obj[name])
        obj_to_img.append(img_idx)
      for s, pred, o in sg['relationships']:
        if pred not in name_to_pred:
          raise ValueError('Relationship "%s" not in vocab' % pred)
        triples.append([s + offset, name_to_pred[pred], o + offset])
      offset += len(sg['objects'])
    device = next(self.parameters()).device
    as_long = lambda v: torch.tensor(v, dtype=torch.int64, device=device)
    return as_long(objs), as_long(triples), as_long(obj_to_img)

  def forward_json(self, scene_graphs):
    """encode_scene_graphs + forward (reference sg2im/model.py:229-232)."""
    objs, triples, obj_to_img = self.encode_scene_graphs(scene_graphs)
    return self.forward(objs, triples, obj_to_img)
