"""CPU oracle for the sg2im training hot path -- TEST INFRASTRUCTURE ONLY.

A functional restatement (plain torch CPU ops over a ``{state_dict name: tensor}``
mapping, no nn.Module) of the algorithm in the reference tree.  Every function
cites the reference file:line it follows (paths relative to /root/reference).

Pinning: the reference ships no tests / golden vectors, so the
oracle is pinned by (a) ``tests/golden/*.pt`` -- outputs of the *imported
reference modules* produced by ``tests/golden/make_golden.py`` in the build
container, and (b) ``tests/test_oracle_vs_reference.py`` which compares against
the live reference whenever /root/reference exists.

This file must never be imported from ``sg2im_amd/`` (the product path).
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-5        # torch.nn.BatchNorm2d default (sg2im/layers.py:26)
BN_MOMENTUM = 0.1

# ----------------------------------------------------------------------------
# Emulation of the HIP library's bf16 OPERAND mode (sg2im_conv_desc.compute_dtype = 1, BASELINE configs[2..4]) -
# not reference behaviour (the reference is fp32 only, scripts/train.py:418,423): with OPERAND_ROUND = 'bf16' every
# SPATIAL convolution multiplies operands rounded to bfloat16 (round-to-nearest-even) and accumulates exactly as
# before, in all three passes - forward conv(round(x), round(W)), data gradient conv_T(round(dy), round(W)), weight
# gradient corr(round(x), round(dy)); bias and everything else untouched.  Which passes round follows the library's
# dispatch (csrc/conv.hip: only the float4 loaders have a bf16 form): forward iff the input channels are a multiple
# of 4; data / weight gradient iff, in addition, the output channels are (and, for the data gradient, there are more
# than 4 input channels - fewer take the few-channel gather kernel).  Linear layers never round.  The float64 run of
# this emulation is the exact result of the arithmetic the bf16 mode is SPECIFIED to perform, which is what its
# parity tests compare against (tests/test_gpu_parity.py::test_bf16_*).
# ----------------------------------------------------------------------------
OPERAND_ROUND = None

# bfloat16 STORAGE of the refinement network's chain (round 6; only with OPERAND_ROUND = 'bf16'): on the refinement
# modules whose maps qualify (_storage_level: the shape rule of the library's planner - a map that 128-pixel patches
# tile and that fills the GPU without split-K) the pre-normalisation convolution outputs y, the gradients w.r.t. the
# activated outputs and the BatchNorm-backward results are ROUNDED to bfloat16 where the library stores them:
#   forward   y_s = round(conv + b);  batch statistics from the UNROUNDED values;  a = leaky(scale * y_s + shift)
#   backward  g = d loss / d a (unrounded accumulators):  dgamma, dbeta and the two sums from du = g * leaky'(.) with
#             x_hat from y_s;  the stored g_s = round(g);  dy = gamma * invstd * (g_s * leaky'(.) [2x2-summed behind an
#             upsampling] - mean(du) - x_hat * mean(du * x_hat)),  stored as round(dy)
# (_BnLeakyStored).  The layout, the output convolutions' z / dz, statistics and parameter gradients stay unrounded.
STORAGE_ROUND = True


def _round_bf16(t):
  return t.to(torch.bfloat16).to(t.dtype)


def _storage_level(N, h, w, C):
  """the planner's rule (csrc/conv.hip halo_plan, 256 CUs): 8 x 16, 4 x 32 or 2 x 64 patches tile the map and the
  launch has at least 1.5 workgroups per CU without split-K - (N h w / 128) * ceil(min(C, 128) / 64) >= 384"""
  if not ((w % 16 == 0 and h % 8 == 0) or (w % 32 == 0 and h % 4 == 0) or (w % 64 == 0 and h % 2 == 0)):
    return False
  return (N * h * w // 128) * ((min(C, 128) + 63) // 64) >= 384


class _BnLeakyStored(torch.autograd.Function):
  """leaky(BatchNorm2d(y)) [+ nearest upsampling x2] with the bfloat16 storage points of the HIP path (see above)"""

  @staticmethod
  def forward(ctx, y, gamma, beta, rm, rv, nbt, training, slope, up2):
    if training:
      mean = y.mean((0, 2, 3))
      var = y.var((0, 2, 3), unbiased=False)
      n = y.numel() // y.size(1)
      with torch.no_grad():
        if rm is not None:
          rm.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean)
          rv.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var * n / max(n - 1, 1))
        if nbt is not None:
          nbt += 1
    else:
      mean, var = rm, rv
    invstd = (var + BN_EPS).rsqrt()
    scale = gamma * invstd
    shift = beta - mean * scale
    ys = _round_bf16(y)
    u = ys * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    ctx.save_for_backward(ys, gamma, mean, invstd, u)
    ctx.cfg = (training, slope, up2)
    a = F.leaky_relu(u, slope)
    return F.interpolate(a, scale_factor=2, mode='nearest') if up2 else a

  @staticmethod
  def backward(ctx, g):
    ys, gamma, mean, invstd, u = ctx.saved_tensors
    training, slope, up2 = ctx.cfg
    mask = torch.where(u > 0, torch.ones((), dtype=g.dtype), torch.full((), slope, dtype=g.dtype))
    xhat = (ys - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
    up = (lambda t: F.interpolate(t, scale_factor=2, mode='nearest')) if up2 else (lambda t: t)
    du = g * up(mask)
    s0, s1 = du.sum((0, 2, 3)), (du * up(xhat)).sum((0, 2, 3))
    dus = _round_bf16(g) * up(mask)                       # from the STORED gradient
    if up2:
      dus = F.avg_pool2d(dus, 2) * 4.0
    M = ys.numel() // ys.size(1)
    a = (gamma * invstd).view(1, -1, 1, 1)
    if training:
      dy = a * (dus - (s0 / M).view(1, -1, 1, 1) - xhat * (s1 / M).view(1, -1, 1, 1))
    else:
      dy = a * dus
    return _round_bf16(dy), s1, s0, None, None, None, None, None, None


class _RoundedConv2d(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x, w, b, stride, padding, rf, rd, rw):
    ctx.save_for_backward(x, w)
    ctx.cfg = (stride, padding, rd, rw, b is not None)
    return F.conv2d(_round_bf16(x) if rf else x, _round_bf16(w) if rf else w, b, stride=stride, padding=padding)

  @staticmethod
  def backward(ctx, g):
    x, w = ctx.saved_tensors
    stride, padding, rd, rw, has_b = ctx.cfg
    dx = dw = db = None
    if ctx.needs_input_grad[0]:
      dx = torch.nn.grad.conv2d_input(x.shape, _round_bf16(w) if rd else w, _round_bf16(g) if rd else g, stride=stride,
                                      padding=padding)
    if ctx.needs_input_grad[1]:
      dw = torch.nn.grad.conv2d_weight(_round_bf16(x) if rw else x, w.shape, _round_bf16(g) if rw else g, stride=stride,
                                       padding=padding)
    if has_b and ctx.needs_input_grad[2]:
      db = g.sum((0, 2, 3))
    return dx, dw, db, None, None, None, None, None


def conv2d(x, w, b=None, stride=1, padding=0, cin_eff=None):
  """F.conv2d, or its bf16-operand emulation (OPERAND_ROUND).  cin_eff: the input channels the HIP path actually
  runs over when it drops a constant all-zero channel (the first refinement module, crn.py:105)"""
  if OPERAND_ROUND is None:
    return F.conv2d(x, w, b, stride=stride, padding=padding)
  if OPERAND_ROUND != 'bf16':
    raise ValueError('OPERAND_ROUND must be None or "bf16"')
  cout, cin = w.size(0), (w.size(1) if cin_eff is None else cin_eff)
  rf = cin % 4 == 0
  rd = rf and cout % 4 == 0 and cin > 4
  rw = rf and cout % 4 == 0
  return _RoundedConv2d.apply(x, w, b, stride, padding, rf, rd, rw)


# ----------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------

def make_vocab(num_objs, num_preds):
  """Synthetic vocab with the conventions of scripts/preprocess_vg.py:231,335 and
  sg2im/data/coco.py:181-205: object 0 is ``__image__``, predicate 0 is
  ``__in_image__``."""
  objs = ['__image__'] + ['obj%d' % i for i in range(1, num_objs)]
  preds = ['__in_image__'] + ['pred%d' % i for i in range(1, num_preds)]
  return {
    'object_idx_to_name': objs,
    'object_name_to_idx': {n: i for i, n in enumerate(objs)},
    'pred_idx_to_name': preds,
    'pred_name_to_idx': {n: i for i, n in enumerate(preds)},
  }


def activation_slope(name):
  """sg2im/layers.py:33-46.  The reference overwrites ``name = 'leakyrelu'``
  unconditionally (line 39), so every activation string yields a LeakyReLU; only a
  ``leakyrelu-<slope>`` string changes the slope from torch's default 0.01."""
  slope = 0.01
  if name.lower().startswith('leakyrelu') and '-' in name:
    slope = float(name.split('-')[1])
  return slope


def mlp(P, prefix, x, n_linear=2, training=True):
  """sg2im/layers.py:216-232 with activation='relu', final_nonlinearity=True:
  Linear [-BatchNorm1d] -ReLU per layer.  Without BatchNorm1d (batch_norm='none', the
  default) the Sequential indices of the Linear layers are 0, 2, 4, ...; with it
  (batch_norm='batch', layers.py:224-225) they are 0, 3, 6, ... and the norm of layer i sits
  at 3i+1.  Which one applies is read off the parameter names."""
  with_bn = ('%s.1.weight' % prefix) in P
  for i in range(n_linear):
    k = 3 * i if with_bn else 2 * i
    x = F.linear(x, P['%s.%d.weight' % (prefix, k)], P['%s.%d.bias' % (prefix, k)])
    if with_bn:
      x = batch_norm(P, '%s.%d' % (prefix, k + 1), x, training)
    x = F.relu(x)
  return x


def batch_norm(P, prefix, x, training):
  """nn.BatchNorm2d / nn.BatchNorm1d forward (sg2im/layers.py:26, 225).  In training mode the running
  statistics in ``P`` are updated in place, like the module does."""
  rm, rv = P.get(prefix + '.running_mean'), P.get(prefix + '.running_var')
  y = F.batch_norm(x, rm, rv, P[prefix + '.weight'], P[prefix + '.bias'],
                   training=training, momentum=BN_MOMENTUM, eps=BN_EPS)
  nbt = P.get(prefix + '.num_batches_tracked')
  if training and nbt is not None:
    nbt += 1
  return y


# ----------------------------------------------------------------------------
# graph convolution (sg2im/graph.py)
# ----------------------------------------------------------------------------

def gconv_pool(new_t, s_idx, o_idx, O, H, Dout, pooling='avg'):
  """sg2im/graph.py:87-114.  Column split of the net1 output into (s, p, o) parts,
  then two scatter_add calls (all subject edges, then all object edges) and the
  'avg' division by clamp(count, min=1).  Returns (pooled (O,H), new_p (T,Dout))."""
  T = new_t.size(0)
  new_s = new_t[:, :H]
  new_p = new_t[:, H:H + Dout]
  new_o = new_t[:, H + Dout:2 * H + Dout]
  pooled = torch.zeros(O, H, dtype=new_t.dtype)
  pooled = pooled.scatter_add(0, s_idx.view(-1, 1).expand_as(new_s), new_s)
  pooled = pooled.scatter_add(0, o_idx.view(-1, 1).expand_as(new_o), new_o)
  if pooling == 'avg':
    counts = torch.zeros(O, dtype=new_t.dtype)
    ones = torch.ones(T, dtype=new_t.dtype)
    counts = counts.scatter_add(0, s_idx, ones)
    counts = counts.scatter_add(0, o_idx, ones)
    pooled = pooled / counts.clamp(min=1).view(-1, 1)
  elif pooling != 'sum':
    raise AssertionError('Invalid pooling "%s"' % pooling)   # graph.py:45
  return pooled, new_p


def gconv_pool_sequential(new_t, s_idx, o_idx, O, H, Dout, pooling='avg'):
  """Order rule of the reference's CPU scatter_add, spelled out (SURVEY.md section 7,
  probe-verified): per object, starting from +0.0, add every subject-role row in
  increasing t, then every object-role row in increasing t, in fp32; then divide by
  max(1, count).  Pure-Python loops: small cases only.  The HIP pool kernel must be
  *bit-identical* to this."""
  T = new_t.size(0)
  pooled = torch.zeros(O, H, dtype=torch.float32)
  counts = [0] * O
  for t in range(T):
    j = int(s_idx[t])
    pooled[j] = pooled[j] + new_t[t, :H]
    counts[j] += 1
  for t in range(T):
    j = int(o_idx[t])
    pooled[j] = pooled[j] + new_t[t, H + Dout:2 * H + Dout]
    counts[j] += 1
  if pooling == 'avg':
    div = torch.tensor([max(1, c) for c in counts], dtype=torch.float32)
    pooled = pooled / div.view(-1, 1)
  return pooled


def graph_triple_conv(P, prefix, obj_vecs, pred_vecs, edges, hidden_dim, out_dim,
                      pooling='avg', training=True):
  """sg2im/graph.py:56-120 (one GraphTripleConv layer)."""
  O = obj_vecs.size(0)
  s_idx = edges[:, 0].contiguous()
  o_idx = edges[:, 1].contiguous()
  triple_in = torch.cat([obj_vecs[s_idx], pred_vecs, obj_vecs[o_idx]], dim=1)
  new_t = mlp(P, prefix + '.net1', triple_in, training=training)
  pooled, new_p = gconv_pool(new_t, s_idx, o_idx, O, hidden_dim, out_dim, pooling)
  new_obj = mlp(P, prefix + '.net2', pooled, training=training)
  return new_obj, new_p


# ----------------------------------------------------------------------------
# layout (sg2im/layout.py) and crops (sg2im/bilinear.py)
# ----------------------------------------------------------------------------

def boxes_to_grid(boxes, H, W):
  """sg2im/layout.py:94-128."""
  O = boxes.size(0)
  b = boxes.view(O, 4, 1, 1)
  x0, y0, x1, y1 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
  X = torch.linspace(0, 1, steps=W).view(1, 1, W).to(boxes)
  Y = torch.linspace(0, 1, steps=H).view(1, H, 1).to(boxes)
  X = ((X - x0) / (x1 - x0)).expand(O, H, W)
  Y = ((Y - y0) / (y1 - y0)).expand(O, H, W)
  return torch.stack([X, Y], dim=3).mul(2).sub(1)


def pool_samples(samples, obj_to_img, n_images=None):
  """sg2im/layout.py:131-162 ('sum' pooling, the only mode model.py uses).  The
  reference derives N from obj_to_img.max()+1 (line 143); ``n_images`` lets a caller
  pin it explicitly."""
  O, D, H, W = samples.size()
  N = int(obj_to_img.max().item()) + 1 if n_images is None else n_images
  out = torch.zeros(N, D, H, W, dtype=samples.dtype)
  idx = obj_to_img.view(O, 1, 1, 1).expand(O, D, H, W)
  return out.scatter_add(0, idx, samples)


def masks_to_layout(vecs, boxes, masks, obj_to_img, H, W=None, align_corners=False,
                    n_images=None):
  """sg2im/layout.py:66-91.  ``align_corners=False`` is what F.grid_sample does when
  the reference is imported under torch >= 1.3 (SURVEY.md section 8c caveat i)."""
  O, D = vecs.size()
  M = masks.size(1)
  assert masks.size() == (O, M, M)
  W = H if W is None else W
  grid = boxes_to_grid(boxes, H, W)
  img_in = vecs.view(O, D, 1, 1) * masks.to(vecs.dtype).view(O, 1, M, M)      # (.float() in the reference; the tests also run this oracle in float64)
  sampled = F.grid_sample(img_in, grid, mode='bilinear', padding_mode='zeros',
                          align_corners=align_corners)
  return pool_samples(sampled, obj_to_img, n_images)


def boxes_to_layout(vecs, boxes, obj_to_img, H, W=None, align_corners=False,
                    n_images=None):
  """sg2im/layout.py:30-63 (constant 8x8 input per object)."""
  O, D = vecs.size()
  W = H if W is None else W
  grid = boxes_to_grid(boxes, H, W)
  img_in = vecs.view(O, D, 1, 1).expand(O, D, 8, 8)
  sampled = F.grid_sample(img_in, grid, mode='bilinear', padding_mode='zeros',
                          align_corners=align_corners)
  return pool_samples(sampled, obj_to_img, n_images)


def tensor_linspace(start, end, steps):
  """sg2im/bilinear.py:249-278: start*linspace(1,0) + end*linspace(0,1)."""
  w0 = torch.linspace(1, 0, steps=steps).to(start)
  w1 = torch.linspace(0, 1, steps=steps).to(start)
  return start.unsqueeze(-1) * w0 + end.unsqueeze(-1) * w1


def crop_bbox_batch(feats, bbox, bbox_to_feats, HH, WW=None, align_corners=False):
  """sg2im/bilinear.py:28-132, 'cudnn' backend.  The reference loops over images and
  copies each image once per box (bilinear.py:76-87); gathering ``feats[bbox_to_feats]``
  is the same computation for every box ordering because the final inverse
  permutation (bilinear.py:96-100) restores box order."""
  WW = HH if WW is None else WW
  B = bbox.size(0)
  per_box = feats[bbox_to_feats]
  bb = 2 * bbox - 1
  X = tensor_linspace(bb[:, 0], bb[:, 2], WW).view(B, 1, WW).expand(B, HH, WW)
  Y = tensor_linspace(bb[:, 1], bb[:, 3], HH).view(B, HH, 1).expand(B, HH, WW)
  grid = torch.stack([X, Y], dim=3)
  return F.grid_sample(per_box, grid, mode='bilinear', padding_mode='zeros',
                       align_corners=align_corners)


# ----------------------------------------------------------------------------
# generator (sg2im/model.py, sg2im/crn.py)
# ----------------------------------------------------------------------------

def refinement_network(P, prefix, layout, n_modules, slope, training, normalization='batch'):
  """sg2im/crn.py:88-111 (+ RefinementModule.forward crn.py:53-65); normalization 'batch' or
  'none' (crn.py:41-47 then drops the norm layers, so the second conv is net.2 instead of net.3)."""
  N, _, H, W = layout.size()
  h, w = H, W
  for _ in range(n_modules):
    h //= 2
    w //= 2
  assert h != 0 and w != 0
  feats = torch.zeros(N, 1, h, w).to(layout)
  pre_upsampled = False              # (_BnLeakyStored hands the features over already upsampled)
  for i in range(n_modules):
    if not pre_upsampled:
      feats = F.interpolate(feats, scale_factor=2, mode='nearest')
    pre_upsampled = False
    hh = feats.size(2)
    lay = layout
    if H > hh:
      factor = H // hh
      lay = F.avg_pool2d(layout, kernel_size=factor, stride=factor)
    x = torch.cat([lay, feats], dim=1)
    p = '%s.refinement_modules.%d.net' % (prefix, i)
    x = conv2d(x, P[p + '.0.weight'], P[p + '.0.bias'], padding=1,
               cin_eff=x.size(1) - 1 if (i == 0 and normalization == 'batch') else None)
    if normalization == 'none':
      x = F.leaky_relu(x, slope)
      feats = F.leaky_relu(conv2d(x, P[p + '.2.weight'], P[p + '.2.bias'], padding=1), slope)
      continue
    if normalization == 'instance':     # nn.InstanceNorm2d(C): no affine, no running stats (layers.py:27-28)
      x = F.leaky_relu(F.instance_norm(x, eps=BN_EPS), slope)
      x = conv2d(x, P[p + '.3.weight'], P[p + '.3.bias'], padding=1)
      feats = F.leaky_relu(F.instance_norm(x, eps=BN_EPS), slope)
      continue
    if OPERAND_ROUND == 'bf16' and STORAGE_ROUND and _storage_level(N, hh, hh * W // H, x.size(1)):
      bnl = lambda q, t, up2: _BnLeakyStored.apply(t, P[q + '.weight'], P[q + '.bias'], P.get(q + '.running_mean'),
                                                   P.get(q + '.running_var'), P.get(q + '.num_batches_tracked'),
                                                   training, slope, up2)
      x = conv2d(bnl(p + '.1', x, False), P[p + '.3.weight'], P[p + '.3.bias'], padding=1)
      feats = bnl(p + '.4', x, i + 1 < n_modules)
      pre_upsampled = i + 1 < n_modules
      continue
    x = F.leaky_relu(batch_norm(P, p + '.1', x, training), slope)
    x = conv2d(x, P[p + '.3.weight'], P[p + '.3.bias'], padding=1)
    feats = F.leaky_relu(batch_norm(P, p + '.4', x, training), slope)
  o = prefix + '.output_conv'
  x = F.leaky_relu(conv2d(feats, P[o + '.0.weight'], P[o + '.0.bias'], padding=1), slope)
  return conv2d(x, P[o + '.2.weight'], P[o + '.2.bias'])


def mask_net(P, prefix, obj_vecs, mask_size, training):
  """sg2im/model.py:94-106,146-147: [up2, BN, conv3x3, ReLU] x log2(mask_size), then
  conv1x1 -> sigmoid.  Sequential indices: block b has BN at 4b+1, conv at 4b+2."""
  O = obj_vecs.size(0)
  x = obj_vecs.view(O, -1, 1, 1)
  size, b = 1, 0
  while size < mask_size:
    x = F.interpolate(x, scale_factor=2, mode='nearest')
    x = batch_norm(P, '%s.%d' % (prefix, 4 * b + 1), x, training)
    x = F.relu(conv2d(x, P['%s.%d.weight' % (prefix, 4 * b + 2)],
                      P['%s.%d.bias' % (prefix, 4 * b + 2)], padding=1))
    size *= 2
    b += 1
  if size != mask_size:
    raise ValueError('Mask size must be a power of 2')   # model.py:104
  k = 4 * b
  scores = conv2d(x, P['%s.%d.weight' % (prefix, k)], P['%s.%d.bias' % (prefix, k)])
  return scores.squeeze(1).sigmoid()


def generator_forward(P, cfg, objs, triples, obj_to_img=None, boxes_gt=None,
                      masks_gt=None, noise=None, training=True, align_corners=False):
  """sg2im/model.py:108-171.  ``cfg`` holds the Sg2ImModel constructor kwargs.
  ``noise`` (N, layout_noise_dim, H, W) replaces the torch.randn draw of
  model.py:164-168 so both sides of a parity test see the same noise."""
  O = objs.size(0)
  s, p, o = triples[:, 0], triples[:, 1], triples[:, 2]
  edges = torch.stack([s, o], dim=1)
  if obj_to_img is None:
    obj_to_img = torch.zeros(O, dtype=objs.dtype)
  obj_vecs = P['obj_embeddings.weight'][objs]
  obj_vecs_orig = obj_vecs
  pred_vecs = P['pred_embeddings.weight'][p]
  L = cfg.get('gconv_num_layers', 5)
  Hd, Dg = cfg.get('gconv_hidden_dim', 512), cfg.get('gconv_dim', 128)
  pooling = cfg.get('gconv_pooling', 'avg')
  if L == 0:
    obj_vecs = F.linear(obj_vecs, P['gconv.weight'], P['gconv.bias'])   # model.py:53-54
  else:
    obj_vecs, pred_vecs = graph_triple_conv(P, 'gconv', obj_vecs, pred_vecs, edges, Hd, Dg, pooling, training)
  for i in range(L - 1):
    obj_vecs, pred_vecs = graph_triple_conv(P, 'gconv_net.gconvs.%d' % i, obj_vecs,
                                            pred_vecs, edges, Hd, Dg, pooling, training)
  boxes_pred = mlp(P, 'box_net', obj_vecs, training=training)

  masks_pred = None
  mask_size = cfg.get('mask_size', None)
  if mask_size is not None and mask_size > 0:
    masks_pred = mask_net(P, 'mask_net', obj_vecs, mask_size, training)

  rel_in = torch.cat([boxes_pred[s], boxes_pred[o], obj_vecs_orig[s], obj_vecs_orig[o]], dim=1)
  rel_scores = mlp(P, 'rel_aux_net', rel_in, training=training)

  H, W = cfg.get('image_size', (64, 64))
  layout_boxes = boxes_pred if boxes_gt is None else boxes_gt
  if masks_pred is None:
    layout = boxes_to_layout(obj_vecs, layout_boxes, obj_to_img, H, W, align_corners)
  else:
    layout_masks = masks_pred if masks_gt is None else masks_gt
    layout = masks_to_layout(obj_vecs, layout_boxes, layout_masks, obj_to_img, H, W, align_corners)

  nd = cfg.get('layout_noise_dim', 0)
  if nd > 0:
    if noise is None:
      noise = torch.randn(layout.size(0), nd, H, W, dtype=layout.dtype)
    layout = torch.cat([layout, noise], dim=1)
  n_modules = len(cfg.get('refinement_dims', (1024, 512, 256, 128, 64)))
  slope = activation_slope(cfg.get('activation', 'leakyrelu-0.2'))
  norm = cfg.get('normalization', 'batch')
  if norm not in ('batch', 'none', 'instance'):
    raise ValueError('Unrecognized normalization type "%s"' % norm)
  img = refinement_network(P, 'refinement_net', layout, n_modules, slope, training, norm)
  return img, boxes_pred, masks_pred, rel_scores


# ----------------------------------------------------------------------------
# discriminators (sg2im/discriminators.py, sg2im/layers.py:129-213)
# ----------------------------------------------------------------------------

def _norm2d(P, name, x, normalization, training):
  """get_normalization_2d (layers.py:22-31) applied: 'batch' reads P[name.*], 'instance' is parameter free"""
  if normalization == 'batch':
    return batch_norm(P, name, x, training)
  if normalization == 'instance':
    return F.instance_norm(x, eps=BN_EPS)
  raise ValueError('Unrecognized normalization type "%s"' % normalization)


def cnn_layers(arch, normalization):
  """The module list build_cnn (layers.py:129-213) produces for ``arch``, as (kind, Sequential index,
  args) tuples - the index is what the state_dict keys carry.  A 'none' normalization adds no
  module (layers.py:208 drops the None entries), 'batch' and 'instance' each take a slot."""
  toks = arch.split(',') if isinstance(arch, str) else list(arch)
  cur = 3
  if toks and toks[0][0] == 'I':
    cur = int(toks[0][1:])
    toks = toks[1:]
  out, idx, first_conv, flat = [], 0, True, False
  def push(kind, *args):
    nonlocal idx
    out.append((kind, idx) + args)
    idx += 1
  for i, t in enumerate(toks):
    if t[0] == 'C':
      vals = [int(v) for v in t[1:].split('-')]
      k, c = vals[0], vals[1]
      stride = vals[2] if len(vals) == 3 else 1
      if not first_conv:
        if normalization != 'none':
          push('norm', cur)
        push('act')
      first_conv = False
      push('conv', cur, c, k, stride)
      cur = c
    elif t[0] == 'R':
      push('res', cur, 'none' if first_conv else normalization)
      first_conv = False
    elif t[0] == 'U':
      push('up', int(t[1:]))
    elif t[0] == 'P':
      push('pool', int(t[1:]))
    elif t[:2] == 'FC':
      _, din, dout = t.split('-')
      if not flat:
        push('flatten')
      flat = True
      push('fc', int(din), int(dout))
      if i + 1 < len(toks):
        push('act')
      cur = int(dout)
    else:
      raise ValueError('Invalid layer "%s"' % t)
  return out, cur


def residual_block(P, prefix, x, C, normalization, slope, padding, training):
  """ResidualBlock (layers.py:88-117), kernel 3: net = [norm, act, conv, norm, act, conv] minus absent
  norms.  forward computes ``self.net(x)`` TWICE (:116-117) and uses the second value - the first
  call only matters for the BatchNorm running statistics, which move twice."""
  pad = 0 if padding == 'valid' else 1
  def net(t):
    j = 0
    for _ in range(2):
      if normalization != 'none':
        t = _norm2d(P, '%s.net.%d' % (prefix, j), t, normalization, training)
        j += 1
      t = F.leaky_relu(t, slope)
      j += 1
      t = conv2d(t, P['%s.net.%d.weight' % (prefix, j)], P['%s.net.%d.bias' % (prefix, j)], padding=pad)
      j += 1
    return t
  shortcut = x
  if pad == 0:
    shortcut = x[:, :, pad:-pad, pad:-pad]        # (:113-114) empty for P = 0: the reference fails in the add
  net(x)
  return shortcut + net(x)


def disc_cnn(P, prefix, x, arch, slope, padding, training, normalization='batch', pooling='max'):
  """Forward of the Sequential build_cnn makes (layers.py:129-213): every conv except the first is
  preceded by [norm +] activation, nothing follows the last (:166-169); R / U / P / FC as documented
  there.  With C tokens only and normalization='batch' conv i sits at index 3i, its BN at 3i-2."""
  layers, _ = cnn_layers(arch, normalization)
  for lay in layers:
    kind, idx = lay[0], lay[1]
    name = '%s.%d' % (prefix, idx)
    if kind == 'norm':
      x = _norm2d(P, name, x, normalization, training)
    elif kind == 'act':
      x = F.leaky_relu(x, slope)
    elif kind == 'conv':
      _cin, _cout, k, stride = lay[2:]
      pad = 0 if padding == 'valid' else (k - 1) // 2
      x = conv2d(x, P[name + '.weight'], P[name + '.bias'], stride=stride, padding=pad)
    elif kind == 'res':
      x = residual_block(P, name, x, lay[2], lay[3], slope, padding, training)
    elif kind == 'up':
      x = F.interpolate(x, scale_factor=lay[2], mode='nearest')
    elif kind == 'pool':
      x = (F.max_pool2d if pooling == 'max' else F.avg_pool2d)(x, kernel_size=lay[2], stride=lay[2])
    elif kind == 'flatten':
      x = x.reshape(x.size(0), -1)
    elif kind == 'fc':
      x = F.linear(x, P[name + '.weight'], P[name + '.bias'])
  return x


def patch_discriminator(P, dcfg, x, training=True):
  """sg2im/discriminators.py:42-45: returns the raw CNN features; ``classifier``
  (line 40) is never applied."""
  slope = activation_slope(dcfg.get('activation', 'leakyrelu-0.2'))
  return disc_cnn(P, 'cnn', x, dcfg['arch'], slope, dcfg.get('padding', 'same'), training,
                  dcfg.get('normalization', 'batch'), dcfg.get('pooling', 'avg'))     # discriminators.py:27


def ac_crop_discriminator(P, dcfg, imgs, objs, boxes, obj_to_img, training=True,
                          align_corners=False):
  """sg2im/discriminators.py:87-90 + AcDiscriminator.forward :68-75."""
  crops = crop_bbox_batch(imgs, boxes, obj_to_img, dcfg.get('object_size', 64),
                          align_corners=align_corners)
  slope = activation_slope(dcfg.get('activation', 'relu'))
  feats = disc_cnn(P, 'discriminator.cnn.0', crops, dcfg['arch'], slope,
                   dcfg.get('padding', 'same'), training, dcfg.get('normalization', 'none'),
                   dcfg.get('pooling', 'avg'))                                        # discriminators.py:50
  vecs = feats.view(feats.size(0), feats.size(1), -1).mean(dim=2)     # GlobalAvgPool layers.py:83-86
  vecs = F.linear(vecs, P['discriminator.cnn.2.weight'], P['discriminator.cnn.2.bias'])
  real = F.linear(vecs, P['discriminator.real_classifier.weight'], P['discriminator.real_classifier.bias'])
  cls = F.linear(vecs, P['discriminator.obj_classifier.weight'], P['discriminator.obj_classifier.bias'])
  return real, F.cross_entropy(cls, objs)


# ----------------------------------------------------------------------------
# losses (sg2im/losses.py, scripts/train.py:387-412)
# ----------------------------------------------------------------------------

def bce_loss(x, target):
  """sg2im/losses.py:39-57."""
  return (x.clamp(min=0) - x * target + (1 + (-x.abs()).exp()).log()).mean()


def gan_g_loss(scores_fake):
  """sg2im/losses.py:72-84."""
  s = scores_fake.reshape(-1)
  return bce_loss(s, torch.ones_like(s))


def gan_d_loss(scores_real, scores_fake):
  """sg2im/losses.py:87-103."""
  assert scores_real.size() == scores_fake.size()
  r, f = scores_real.reshape(-1), scores_fake.reshape(-1)
  return bce_loss(r, torch.ones_like(r)) + bce_loss(f, torch.zeros_like(f))


DEFAULT_LOSS_WEIGHTS = dict(            # scripts/train.py:108-131
  l1_pixel_loss_weight=1.0, bbox_pred_loss_weight=10.0, predicate_pred_loss_weight=0.0,
  mask_loss_weight=0.0, discriminator_loss_weight=0.01, d_obj_weight=1.0,
  d_img_weight=1.0, ac_loss_weight=0.1)


def get_gan_losses(gan_type):
  """sg2im/losses.py:21-36, 106-145"""
  if gan_type == 'gan':
    return gan_g_loss, gan_d_loss
  if gan_type == 'wgan':
    return (lambda f: -f.mean()), (lambda r, f: f.mean() - r.mean())
  if gan_type == 'lsgan':
    def g(f):
      f = f.reshape(-1)
      return F.mse_loss(f.sigmoid(), torch.full_like(f, 1))
    def d(r, f):
      r, f = r.reshape(-1), f.reshape(-1)
      return F.mse_loss(r.sigmoid(), torch.full_like(r, 1)) + F.mse_loss(f.sigmoid(), torch.full_like(f, 0))
    return g, d
  raise ValueError('Unrecognized GAN type "%s"' % gan_type)


def generator_losses(w, imgs, imgs_pred, boxes, boxes_pred, masks, masks_pred,
                     predicates, rel_scores):
  """scripts/train.py:387-412 (calculate_model_losses)."""
  losses = {}
  total = torch.zeros(1).to(imgs)
  losses['L1_pixel_loss'] = F.l1_loss(imgs_pred, imgs) * w['l1_pixel_loss_weight']
  total = total + losses['L1_pixel_loss']
  losses['bbox_pred'] = F.mse_loss(boxes_pred, boxes) * w['bbox_pred_loss_weight']
  total = total + losses['bbox_pred']
  if w['predicate_pred_loss_weight'] > 0:
    losses['predicate_pred'] = F.cross_entropy(rel_scores, predicates) * w['predicate_pred_loss_weight']
    total = total + losses['predicate_pred']
  if w['mask_loss_weight'] > 0 and masks is not None and masks_pred is not None:
    losses['mask_loss'] = F.binary_cross_entropy(masks_pred, masks.to(masks_pred.dtype)) * w['mask_loss_weight']
    total = total + losses['mask_loss']
  return total, losses


# ----------------------------------------------------------------------------
# one training iteration (scripts/train.py:524-592)
# ----------------------------------------------------------------------------

class OracleTrainer(object):
  """The loop body of scripts/train.py:524-592 on CPU: generator forward + losses +
  Adam, then the object and image discriminator updates.  Parameters are leaf tensors
  in three dicts (generator / D_obj / D_img) keyed by state_dict names; buffers (BN
  running stats) live in the same dicts without grad."""

  def __init__(self, PG, PDo, PDi, gcfg, docfg, dicfg, weights=None, lr=1e-4,
               align_corners=False, gan_loss_type='gan'):
    self.gan_g, self.gan_d = get_gan_losses(gan_loss_type)       # train.py:467
    self.gcfg, self.docfg, self.dicfg = gcfg, docfg, dicfg
    self.w = dict(DEFAULT_LOSS_WEIGHTS)
    if weights:
      self.w.update(weights)
    # build_obj_discriminator / build_img_discriminator return None when the weight is zero
    # (train.py:198-200, 221-223): that discriminator and its loss terms do not exist
    if self.w['discriminator_loss_weight'] == 0 or self.w['d_obj_weight'] == 0:
      PDo = None
    if self.w['discriminator_loss_weight'] == 0 or self.w['d_img_weight'] == 0:
      PDi = None
    self.PG, self.PDo, self.PDi = PG, PDo, PDi
    self.align_corners = align_corners
    self.training = True
    for P in (PG, PDo, PDi):
      for k, v in (P or {}).items():
        if v.is_floating_point() and not ('running_' in k):
          v.requires_grad_(True)
    # torch.optim.Adam(model.parameters(), lr) -- train.py:426,436,443
    self.opt_g = torch.optim.Adam([v for v in PG.values() if v.requires_grad], lr=lr)
    self.opt_do = PDo and torch.optim.Adam([v for v in PDo.values() if v.requires_grad], lr=lr)
    self.opt_di = PDi and torch.optim.Adam([v for v in PDi.values() if v.requires_grad], lr=lr)

  def g_forward_loss(self, batch, noise=None):
    imgs, objs, boxes, masks, triples, obj_to_img = batch
    w = self.w
    out = generator_forward(self.PG, self.gcfg, objs, triples, obj_to_img, boxes_gt=boxes,
                            masks_gt=masks, noise=noise, training=self.training,
                            align_corners=self.align_corners)
    imgs_pred, boxes_pred, masks_pred, rel_scores = out
    total, losses = generator_losses(w, imgs, imgs_pred, boxes, boxes_pred, masks, masks_pred,
                                     triples[:, 1], rel_scores)
    # train.py:538-550
    if self.PDo is not None:
      scores_fake, ac_loss = ac_crop_discriminator(self.PDo, self.docfg, imgs_pred, objs, boxes,
                                                   obj_to_img, True, self.align_corners)
      losses['ac_loss'] = ac_loss * w['ac_loss_weight']
      total = total + losses['ac_loss']
      losses['g_gan_obj_loss'] = self.gan_g(scores_fake) * (w['discriminator_loss_weight'] * w['d_obj_weight'])
      total = total + losses['g_gan_obj_loss']
    if self.PDi is not None:
      scores_fake = patch_discriminator(self.PDi, self.dicfg, imgs_pred, True)
      losses['g_gan_img_loss'] = self.gan_g(scores_fake) * (w['discriminator_loss_weight'] * w['d_img_weight'])
      total = total + losses['g_gan_img_loss']
    losses['total_loss'] = total
    return total, losses, out

  def d_obj_loss(self, batch, imgs_fake):
    imgs, objs, boxes, masks, triples, obj_to_img = batch
    # train.py:566-575
    sf, ac_f = ac_crop_discriminator(self.PDo, self.docfg, imgs_fake, objs, boxes, obj_to_img,
                                     True, self.align_corners)
    sr, ac_r = ac_crop_discriminator(self.PDo, self.docfg, imgs, objs, boxes, obj_to_img,
                                     True, self.align_corners)
    parts = {'d_obj_gan_loss': self.gan_d(sr, sf), 'd_ac_loss_real': ac_r, 'd_ac_loss_fake': ac_f}
    return parts['d_obj_gan_loss'] + ac_r + ac_f, parts

  def d_img_loss(self, batch, imgs_fake):
    imgs = batch[0]
    # train.py:581-588
    sf = patch_discriminator(self.PDi, self.dicfg, imgs_fake, True)
    sr = patch_discriminator(self.PDi, self.dicfg, imgs, True)
    loss = self.gan_d(sr, sf)
    return loss, {'d_img_gan_loss': loss}

  def step(self, batch, noise=None):
    total, losses, out = self.g_forward_loss(batch, noise)
    if not math.isfinite(float(total.detach())):          # train.py:553-555
      return None
    self.opt_g.zero_grad()
    # the reference also deposits (discarded) grads in the D params here (train.py:559)
    total.backward()
    self.opt_g.step()
    imgs_fake = out[0].detach()
    parts_o, parts_i = {}, {}
    if self.PDo is not None:
      ld, parts_o = self.d_obj_loss(batch, imgs_fake)
      self.opt_do.zero_grad()
      ld.backward()
      self.opt_do.step()
    if self.PDi is not None:
      li, parts_i = self.d_img_loss(batch, imgs_fake)
      self.opt_di.zero_grad()
      li.backward()
      self.opt_di.step()
    res = {k: float(v) for k, v in losses.items()}
    res.update({k: float(v) for k, v in parts_o.items()})
    res.update({k: float(v) for k, v in parts_i.items()})
    return res


# ----------------------------------------------------------------------------
# parameter initialisation with the reference's shapes / state_dict names
# ----------------------------------------------------------------------------

def _lin(P, name, dout, din, gen, kaiming=False):
  if kaiming:      # graph.py:26-29 -> kaiming_normal_ on Linear weights
    P[name + '.weight'] = torch.randn(dout, din, generator=gen) * math.sqrt(2.0 / din)
  else:            # nn.Linear default init: U(-1/sqrt(din), 1/sqrt(din)) for W and b
    P[name + '.weight'] = (torch.rand(dout, din, generator=gen) * 2 - 1) / math.sqrt(din)
  P[name + '.bias'] = (torch.rand(dout, generator=gen) * 2 - 1) / math.sqrt(din)


def _conv(P, name, cout, cin, k, gen, kaiming=False):
  fan_in = cin * k * k
  if kaiming:      # crn.py:49-51,84-85
    P[name + '.weight'] = torch.randn(cout, cin, k, k, generator=gen) * math.sqrt(2.0 / fan_in)
  else:
    P[name + '.weight'] = (torch.rand(cout, cin, k, k, generator=gen) * 2 - 1) / math.sqrt(fan_in)
  P[name + '.bias'] = (torch.rand(cout, generator=gen) * 2 - 1) / math.sqrt(fan_in)


def _bn(P, name, c, gen, randomize):
  P[name + '.weight'] = torch.ones(c) if not randomize else 0.5 + torch.rand(c, generator=gen)
  P[name + '.bias'] = torch.zeros(c) if not randomize else 0.2 * torch.randn(c, generator=gen)
  P[name + '.running_mean'] = torch.zeros(c)
  P[name + '.running_var'] = torch.ones(c)
  P[name + '.num_batches_tracked'] = torch.tensor(0, dtype=torch.long)


def init_generator_params(cfg, seed=0, randomize_bn=False):
  """Random parameters with exactly the names/shapes of Sg2ImModel.state_dict()
  (SURVEY.md section 8b).  The distributions follow the reference's initialisers but the
  draws are this function's own (seeded) -- fixtures carry explicit weights when the
  reference's own draws matter."""
  g = torch.Generator().manual_seed(seed)
  vocab = cfg['vocab']
  C, Pn = len(vocab['object_idx_to_name']), len(vocab['pred_idx_to_name'])
  E, Dg, Hd = cfg.get('embedding_dim', 64), cfg.get('gconv_dim', 128), cfg.get('gconv_hidden_dim', 512)
  L = cfg.get('gconv_num_layers', 5)
  P = {}
  P['obj_embeddings.weight'] = torch.randn(C + 1, E, generator=g)      # model.py:50
  P['pred_embeddings.weight'] = torch.randn(Pn, E, generator=g)
  mbn = cfg.get('mlp_normalization', 'none') == 'batch'
  def mlp2(prefix, d0, d1, d2, kaiming=False):
    """build_mlp([d0, d1, d2], batch_norm=mlp_normalization) parameter names"""
    step = 3 if mbn else 2
    for i, (din, dout) in enumerate(((d0, d1), (d1, d2))):
      _lin(P, '%s.%d' % (prefix, step * i), dout, din, g, kaiming)
      if mbn:
        _bn(P, '%s.%d' % (prefix, step * i + 1), dout, g, randomize_bn)
  def gconv(prefix, din, dout):
    mlp2(prefix + '.net1', 3 * din, Hd, 2 * Hd + dout, True)
    mlp2(prefix + '.net2', Hd, Hd, dout, True)
  if L == 0:
    _lin(P, 'gconv', Dg, E, g)
  else:
    gconv('gconv', E, Dg)
  for i in range(L - 1):
    gconv('gconv_net.gconvs.%d' % i, Dg, Dg)
  mlp2('box_net', Dg, Hd, 4)
  ms = cfg.get('mask_size', None)
  if ms is not None and ms > 0:
    size, b = 1, 0
    while size < ms:
      _bn(P, 'mask_net.%d' % (4 * b + 1), Dg, g, randomize_bn)
      _conv(P, 'mask_net.%d' % (4 * b + 2), Dg, Dg, 3, g)
      size *= 2
      b += 1
    _conv(P, 'mask_net.%d' % (4 * b), 1, Dg, 1, g)
  mlp2('rel_aux_net', 2 * E + 8, Hd, Pn)
  dims = (Dg + cfg.get('layout_noise_dim', 0),) + tuple(cfg.get('refinement_dims', (1024, 512, 256, 128, 64)))
  for i in range(1, len(dims)):
    cin = 1 if i == 1 else dims[i - 1]
    p = 'refinement_net.refinement_modules.%d.net' % (i - 1)
    _conv(P, p + '.0', dims[i], dims[0] + cin, 3, g, True)
    if cfg.get('normalization', 'batch') == 'none':
      _conv(P, p + '.2', dims[i], dims[i], 3, g, True)
      continue
    if cfg.get('normalization', 'batch') == 'instance':
      _conv(P, p + '.3', dims[i], dims[i], 3, g, True)
      continue
    _bn(P, p + '.1', dims[i], g, randomize_bn)
    _conv(P, p + '.3', dims[i], dims[i], 3, g, True)
    _bn(P, p + '.4', dims[i], g, randomize_bn)
  _conv(P, 'refinement_net.output_conv.0', dims[-1], dims[-1], 3, g, True)
  _conv(P, 'refinement_net.output_conv.2', 3, dims[-1], 1, g, True)
  return P


def _init_disc_cnn(P, prefix, arch, cin, gen, randomize_bn, normalization='batch'):
  """parameters of build_cnn's Sequential in module order (the draw order for C-only strings is
  BN then conv, as before)"""
  if not arch.startswith('I'):
    arch = 'I%d,%s' % (cin, arch)
  layers, cout = cnn_layers(arch, normalization)
  for lay in layers:
    kind, name = lay[0], '%s.%d' % (prefix, lay[1])
    if kind == 'norm' and normalization == 'batch':
      _bn(P, name, lay[2], gen, randomize_bn)
    elif kind == 'conv':
      _conv(P, name, lay[3], lay[2], lay[4], gen)
    elif kind == 'fc':
      _lin(P, name, lay[3], lay[2], gen)
    elif kind == 'res':
      C, norm = lay[2], lay[3]
      j = 0
      for _ in range(2):
        if norm != 'none':
          if norm == 'batch':
            _bn(P, '%s.net.%d' % (name, j), C, gen, randomize_bn)
          j += 1
        j += 1
        _conv(P, '%s.net.%d' % (name, j), C, C, 3, gen)
        j += 1
  return cout


def init_patch_discriminator_params(dcfg, seed=1, randomize_bn=False):
  g = torch.Generator().manual_seed(seed)
  P = {}
  c = _init_disc_cnn(P, 'cnn', dcfg['arch'], 3, g, randomize_bn, dcfg.get('normalization', 'batch'))
  _conv(P, 'classifier', 1, c, 1, g)         # discriminators.py:40 (present, unused)
  return P


def init_ac_discriminator_params(dcfg, seed=2, randomize_bn=False):
  g = torch.Generator().manual_seed(seed)
  P = {}
  c = _init_disc_cnn(P, 'discriminator.cnn.0', dcfg['arch'], 3, g, randomize_bn, dcfg.get('normalization', 'none'))
  _lin(P, 'discriminator.cnn.2', 1024, c, g)
  _lin(P, 'discriminator.real_classifier', 1, 1024, g)
  _lin(P, 'discriminator.obj_classifier', len(dcfg['vocab']['object_idx_to_name']), 1024, g)
  return P
