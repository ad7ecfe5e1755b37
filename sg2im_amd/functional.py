"""autograd.Function wrappers: each one is a whole block of the reference model whose
forward AND backward are explicit sequences of HIP kernels (sg2im_amd.ops).  Internally
image-like tensors are NHWC and BatchNorm+LeakyReLU stay *pending* (folded into the next
convolution's operand loader), so no torch compute kernel runs on the hot path.
"""
import os
from ctypes import c_void_p

import torch
from torch.autograd import Function

from . import ops
from .ops import conv_desc, nhwc_src, rows_src

BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def _fptr(t, offset_floats=0):
  return c_void_p(t.data_ptr() + 4 * offset_floats)


def _new(like, *shape):
  return torch.empty(*shape, dtype=torch.float32, device=like.device)


def _new_s(like, bf, *shape):
  """a tensor of the refinement network's chain in its STORAGE type: bfloat16 where the layer qualifies (see
  _storage_levels), float32 otherwise"""
  if not bf:
    return torch.empty(*shape, dtype=torch.float32, device=like.device)
  n = 1
  for d in shape:
    n *= int(d)
  # (8 elements of slack: the mixed-storage weight gradient loads 16 bytes where it needs 8, include/sg2im_hip.h)
  return torch.empty(n + 8, dtype=torch.bfloat16, device=like.device)[:n].view(*shape)


# A/B knob: 0 = every tensor of the bf16 mode stays float32 in memory (rounds 2-5: operand rounding only)
BF16_STORAGE = os.environ.get('SG2IM_BF16_STORAGE', '1') != '0'


def _storage_levels(N, H, W, L, dims):
  """Which refinement modules keep their chain tensors - the convolution outputs y0 / y1 (pre-normalisation), the
  gradients w.r.t. the activated outputs and the BatchNorm-backward results dy0 / dy1 - as bfloat16 in HBM: bf16 compute
  mode (Trainer(compute_dtype='bf16')) and a map that the halo'd 3x3 kernels run WITHOUT split-K for the module's channel
  count (the launches that carry bfloat16 storage have no split-K form: sg2im_conv_halo_unsplit - at the COCO-64 batch
  the 32 x 32 and 64 x 64 levels, 85 % of the chain's bytes).  BatchNorm statistics, the layout, the image and every
  parameter gradient stay float32; the reductions inside the GEMM epilogues see the unrounded accumulators."""
  if not (BF16_STORAGE and ops.CONV_COMPUTE == 1):
    return [False] * L
  return [ops.halo_unsplit(N, H >> (L - 1 - i), W >> (L - 1 - i), min(int(dims[i]), 128)) for i in range(L)]


def _cl_weight(w):
  """physical [Cout][KH][KW][Cin] view of a conv weight parameter (channels_last storage)"""
  p = w.permute(0, 2, 3, 1)
  if not p.is_contiguous():
    raise RuntimeError('conv weights must be stored channels_last; call sg2im_amd.layers.to_channels_last(module)')
  return p


# ---------------------------------------------------------------------------------------
# gradient sinks: sg2im_amd.optim.FlatParams registers, for every parameter, a view into its
# flat gradient arena.  Backward kernels then accumulate straight into the arena (beta = 1)
# and report ``None`` to autograd, which removes ~150 per-parameter "grad += g" launches per
# step.  Without a registered sink the gradient is returned through autograd as usual.
#
# The registry maps a parameter's data pointer to a WEAK reference of the owning FlatParams,
# which holds the views: an entry is only honoured while its owner (hence the arena the
# address lies in) is alive, so a freed arena can neither be kept alive by the registry nor be
# mistaken for the home of a later tensor that the allocator placed at the same address.
# ---------------------------------------------------------------------------------------
GRAD_SINKS = {}


def _sink(param):
  if param is None:
    return None
  ptr = param.data_ptr()
  ref = GRAD_SINKS.get(ptr)
  if ref is None:
    return None
  owner = ref()
  if owner is None:                       # the arena is gone: a stale address
    GRAD_SINKS.pop(ptr, None)
    return None
  view = owner.sinks.get(ptr)
  if view is None or view.numel() != param.numel():
    return None
  return view


def _weight_mirror(w):
  """device address of the bfloat16 mirror of conv weight ``w`` (FlatParams.refresh_mirror), or None: only while the
  Trainer has refreshed it for this iteration (ops.WEIGHT_MIRROR) and ``w`` lives in an arena that has one"""
  if not ops.WEIGHT_MIRROR or w is None:
    return None
  ref = GRAD_SINKS.get(w.data_ptr())
  owner = ref() if ref is not None else None
  if owner is None or owner.mirror is None:
    return None
  off = (w.data_ptr() - owner.flat.data_ptr()) // 4
  if off < 0 or off + w.numel() > owner.numel:
    return None
  return owner.mirror.data_ptr() + 2 * off


ops.WEIGHT_MIRROR_LOOKUP = _weight_mirror


def _cl_grad(dw_phys):
  """[Cout][KH][KW][Cin] gradient -> (Cout,Cin,KH,KW)-shaped channels_last tensor"""
  return dw_phys.permute(0, 3, 1, 2)


# ----------------------------------------------------------------------------
# layout conversions at the API boundary
# ----------------------------------------------------------------------------

class NchwToNhwc(Function):
  @staticmethod
  def forward(ctx, x):
    N, C, H, W = x.shape
    return ops.nchw_to_nhwc(x.contiguous(), _new(x, N, H, W, C))

  @staticmethod
  def backward(ctx, g):
    N, H, W, C = g.shape
    return ops.nhwc_to_nchw(g.contiguous(), _new(g, N, C, H, W))


class NhwcToNchw(Function):
  @staticmethod
  def forward(ctx, x):
    N, H, W, C = x.shape
    return ops.nhwc_to_nchw(x.contiguous(), _new(x, N, C, H, W))

  @staticmethod
  def backward(ctx, g):
    N, C, H, W = g.shape
    return ops.nchw_to_nhwc(g.contiguous(), _new(g, N, H, W, C))


# ----------------------------------------------------------------------------
# linear layers
# ----------------------------------------------------------------------------

def _linear_bwd(desc, W, dpre, need_dx, need_dw, need_db, K, b=None, group=None, act=None):
  """dpre: dense (M,N) gradient of the pre-activation.  Returns (dx (M,K), dW, db); a
  parameter gradient that went straight into its registered sink is reported as None.
  group: a list - parameter gradients that go into sinks are only QUEUED on it, for one grouped
  launch by the caller (_flush_wgrad_group).  act = (z, slope): the layer's input x is z = leaky_slope(pre) of a
  previous layer and the caller wants the gradient w.r.t. that pre-activation: dx *= leaky'(z), applied by the data
  gradient's own launches (ops.conv2d_backward_data_act)."""
  M, N = dpre.shape
  dx = dw = db = None
  if need_dx:
    dx = _new(dpre, M, K)
    if act is not None and act[1] != 1.0 and M > 0:
      ops.conv2d_backward_data_act(desc, W, N, dpre, N, 0, K, dx, K, act[0], act[0].stride(0), act[1])
    else:
      ops.conv2d_backward_data(desc, W, N, dpre, N, 0, K, dx, K)
  if group is not None and need_dw and M > 0:
    sk, skb = _sink(W), (_sink(b) if need_db else None)
    if sk is not None and (not need_db or skb is not None):
      group.append((desc, dpre, N, sk, skb))
      return dx, None, None
  if need_dw:
    sk = _sink(W)
    skb = _sink(b) if need_db else None
    if sk is not None:
      # (with both sinks registered the bias gradient rides along in the same pass over dpre)
      ops.conv2d_backward_weight(desc, dpre, N, N, sk, accumulate=True, dbias=skb)
      need_db = need_db and skb is None
    else:
      dw = _new(dpre, N, K)
      if need_db and skb is None:
        db = _new(dpre, N)
        need_db = False
      ops.conv2d_backward_weight(desc, dpre, N, N, dw, dbias=db)
  if need_db:
    sk = _sink(b)
    if sk is not None:
      ops.column_sum(_fptr(dpre), M, N, N, sk, accumulate=True)
    else:
      db = _new(dpre, N)
      ops.column_sum(_fptr(dpre), M, N, N, db)
  return dx, dw, db


def _flush_wgrad_group(group):
  """the queued weight gradients of _linear_bwd: one grouped launch, or one by one when a problem does not
  qualify for it"""
  if not group:
    return
  # (bench.py's instrumented pass times every launch on its own: no grouping while a KernelTimer is active)
  if not (len(group) > 1 and ops.TIMER is None and ops.conv2d_backward_weight_group(group)):
    for desc, dpre, N, sk, skb in group:
      ops.conv2d_backward_weight(desc, dpre, N, N, sk, accumulate=True, dbias=skb)
  del group[:]


def _act_bwd_rows(g, y, slope):
  """dpre = g * leaky'(y) for dense row matrices (y is the activated output)"""
  M, N = y.shape
  g = g.contiguous()
  if slope == 1.0:
    return g
  return ops.act_backward(_fptr(g), N, 0, M, 1, 1, y, N, N, slope, _new(y, M, N))


class LinearAct(Function):
  """y = leaky_slope(x W^T + b)   (nn.Linear [+ ReLU], reference sg2im/layers.py:216-232)"""

  @staticmethod
  def forward(ctx, x, W, b, slope, shadowed=False):
    """shadowed: a training-mode BatchNorm consumes y, so db is exactly zero (_shadowed_bias_grad)"""
    M, K = x.shape
    N = W.size(0)
    desc = conv_desc([rows_src(x)], M, 1, 1)
    y = ops.conv2d_forward(desc, W, N, b, _new(x, M, N), N, slope)
    ctx.save_for_backward(x, W, y, b)
    ctx.slope, ctx.shadowed = slope, shadowed
    return y

  @staticmethod
  def backward(ctx, g):
    x, W, y, b = ctx.saved_tensors
    ni = ctx.needs_input_grad
    dpre = _act_bwd_rows(g, y, ctx.slope)
    desc = conv_desc([rows_src(x)], x.size(0), 1, 1)
    dx, dw, db = _linear_bwd(desc, W, dpre, ni[0], ni[1], ni[2] and not ctx.shadowed, x.size(1), b)
    if ctx.shadowed:
      db = _shadowed_bias_grad(b, ni[2])
    return dx, dw, db, None, None


class TwoHeads(Function):
  """(x W1^T + b1, x W2^T + b2): two nn.Linear layers on the same input - AcDiscriminator's real_classifier and
  obj_classifier (reference sg2im/discriminators.py:66-75) - one launch forward, one for the summed input gradient
  (csrc/heads.hip); their weight / bias gradients as one grouped launch of the implicit-GEMM family."""

  @staticmethod
  def forward(ctx, x, W1, b1, W2, b2):
    if not (x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0):    # (column views are fine)
      x = x.contiguous()
    M = x.size(0)
    y1, y2 = ops.two_heads_forward(x, W1, b1, W2, b2, _new(x, M, W1.size(0)), _new(x, M, W2.size(0)))
    ctx.save_for_backward(x, W1, b1, W2, b2)
    return y1, y2

  @staticmethod
  def backward(ctx, g1, g2):
    x, W1, b1, W2, b2 = ctx.saved_tensors
    ni = ctx.needs_input_grad
    M, K = x.shape
    g1 = torch.zeros(M, W1.size(0), dtype=x.dtype, device=x.device) if g1 is None else g1.contiguous()
    g2 = torch.zeros(M, W2.size(0), dtype=x.dtype, device=x.device) if g2 is None else g2.contiguous()
    dx = ops.two_heads_backward_data(g1, g2, W1, W2, _new(x, M, K)) if ni[0] else None
    desc = conv_desc([rows_src(x)], M, 1, 1)
    group = []
    _, dw1, db1 = _linear_bwd(desc, W1, g1, False, ni[1], ni[2], K, b1, group)
    _, dw2, db2 = _linear_bwd(desc, W2, g2, False, ni[3], ni[4], K, b2, group)
    _flush_wgrad_group(group)
    return dx, dw1, db1, dw2, db2


class Mlp2(Function):
  """ReLU(W2 ReLU(W1 x + b1) + b2): build_mlp([D, H, out]) (reference sg2im/model.py:76-78)"""

  @staticmethod
  def forward(ctx, x, W1, b1, W2, b2):
    M = x.size(0)
    d1 = conv_desc([rows_src(x)], M, 1, 1)
    h = ops.conv2d_forward(d1, W1, W1.size(0), b1, _new(x, M, W1.size(0)), W1.size(0), 0.0)
    d2 = conv_desc([rows_src(h)], M, 1, 1)
    y = ops.conv2d_forward(d2, W2, W2.size(0), b2, _new(x, M, W2.size(0)), W2.size(0), 0.0)
    ctx.save_for_backward(x, W1, W2, h, y, b1, b2)
    return y

  @staticmethod
  def backward(ctx, g):
    x, W1, W2, h, y, b1, b2 = ctx.saved_tensors
    M = x.size(0)
    ni = ctx.needs_input_grad
    dp2 = _act_bwd_rows(g, y, 0.0)
    d2 = conv_desc([rows_src(h)], M, 1, 1)
    # (dp1 = (dp2 W2) * relu'(h): the mask rides in the data gradient's launches)
    dp1, dW2, db2 = _linear_bwd(d2, W2, dp2, True, ni[3], ni[4], h.size(1), b2, act=(h, 0.0))
    d1 = conv_desc([rows_src(x)], M, 1, 1)
    dx, dW1, db1 = _linear_bwd(d1, W1, dp1, ni[0], ni[1], ni[2], x.size(1), b1)
    return dx, dW1, db1, dW2, db2


# ---- composable pieces for MLPs with BatchNorm1d (build_mlp(batch_norm='batch'), reference
# sg2im/layers.py:216-232).  The default configuration has no norm and runs through the fused
# Mlp2 / GraphTripleConvFn / RelAux above; with a BatchNorm between every Linear and its ReLU the
# layers are chained from these instead: LinearAct(slope=1) -> BnActRows -> ...

class BnActRows(Function):
  """z = leaky_slope(batch_norm(y)) materialised; y is any dense tensor with the channels last:
  (rows, C) for BatchNorm1d + ReLU, NHWC for a BatchNorm2d + LeakyReLU that cannot ride in a
  neighbouring conv (the layer-by-layer build_cnn path)"""

  @staticmethod
  def forward(ctx, y, bn, training, gamma, beta, slope=0.0, count=None):
    """count: None or (int32 device scalar, unit) - only the first count * unit rows of a padded batch are
    real (sg2im_amd/bucketing.py): the statistics run over those, the padding rows get a zero gradient"""
    y = y.contiguous()
    C = y.size(-1)
    rows = y.numel() // C
    st = ops.bn_stats(y, rows, C, C, bn, training, BN_EPS, BN_MOMENTUM, count=count)
    z = ops.affine_act_forward(y.view(rows, C), st, slope, _new(y, rows, C)).view(y.shape)
    ctx.save_for_backward(y, gamma, beta)
    ctx.st, ctx.training, ctx.slope, ctx.count = st, training, slope, count
    return z

  @staticmethod
  def backward(ctx, g):
    y, gamma, beta = ctx.saved_tensors
    C = y.size(-1)
    rows = y.numel() // C
    g = g.contiguous()
    ni = ctx.needs_input_grad
    dgam, dbet, acc, ggam, gbet = _bn_grad_bufs(g, C, gamma, beta, ni[3], ni[4])
    dy = ops.bn_act_backward(_fptr(g), C, 0, rows, 1, 1, y, C, C, gamma, ctx.st, ctx.slope, ctx.training,
                             _new(g, *y.shape), dgam, dbet, acc, count=ctx.count)
    return dy, None, None, ggam, gbet, None, None


class TripleLinear(Function):
  """W [obj[s], pred, obj[o]] + b for every triple (the first Linear of GraphTripleConv.net1,
  reference sg2im/graph.py:73-83), no activation; the gather + concat is the GEMM's loader."""

  @staticmethod
  def forward(ctx, obj_vecs, pred_vecs, s_idx, o_idx, csr, W, b, shadowed=False):
    T = pred_vecs.size(0)
    obj_vecs, pred_vecs = obj_vecs.contiguous(), pred_vecs.contiguous()
    d = conv_desc([rows_src(obj_vecs, s_idx), rows_src(pred_vecs), rows_src(obj_vecs, o_idx)], T, 1, 1)
    y = ops.conv2d_forward(d, W, W.size(0), b, _new(obj_vecs, T, W.size(0)), W.size(0))
    ctx.save_for_backward(obj_vecs, pred_vecs, s_idx, o_idx, W, b)
    ctx.csr, ctx.shadowed = csr, shadowed
    return y

  @staticmethod
  def backward(ctx, g):
    obj_vecs, pred_vecs, s_idx, o_idx, W, b = ctx.saved_tensors
    T, Din, O = pred_vecs.size(0), obj_vecs.size(1), obj_vecs.size(0)
    ni = ctx.needs_input_grad
    d = conv_desc([rows_src(obj_vecs, s_idx), rows_src(pred_vecs), rows_src(obj_vecs, o_idx)], T, 1, 1)
    dX, dW, db = _linear_bwd(d, W, g.contiguous(), ni[0] or ni[1], ni[5], ni[6] and not ctx.shadowed, 3 * Din, b)
    if ctx.shadowed:
      db = _shadowed_bias_grad(b, ni[6])
    d_obj = d_pred = None
    if ni[0]:
      d_obj = ops.segment_sum(dX[:, :Din], dX[:, 2 * Din:], ctx.csr, Din, False, _new(obj_vecs, O, Din))
    if ni[1]:
      d_pred = dX[:, Din:2 * Din]
    return d_obj, d_pred, None, None, None, dW, db, None


class TriplePool(Function):
  """pooled[j] = sum / avg of new_t[t, :H] over s_t = j and new_t[t, H+Dout:] over o_t = j
  (reference sg2im/graph.py:87-114), plus the predicate slice new_t[:, H:H+Dout]."""

  @staticmethod
  def forward(ctx, new_t, s_idx, o_idx, csr, avg, H, Dout, O):
    pooled = ops.segment_sum(new_t[:, :H], new_t[:, H + Dout:], csr, H, avg, _new(new_t, O, H))
    ctx.save_for_backward(s_idx, o_idx)
    ctx.misc = (csr, avg, H, Dout, tuple(new_t.shape))
    return pooled, new_t[:, H:H + Dout].contiguous()

  @staticmethod
  def backward(ctx, g_pooled, g_pred):
    s_idx, o_idx = ctx.saved_tensors
    csr, avg, H, Dout, shape = ctx.misc
    d = torch.zeros(shape, dtype=torch.float32, device=s_idx.device)
    if g_pooled is not None:
      g_pooled = g_pooled.contiguous()
      cavg = csr if avg else None
      ops.gather_rows(g_pooled, s_idx, d[:, :H], cavg)
      ops.gather_rows(g_pooled, o_idx, d[:, H + Dout:], cavg)
    if g_pred is not None:
      ops.copy_2d(g_pred.contiguous(), d[:, H:H + Dout])
    return d, None, None, None, None, None, None, None


class RelAuxLinear(Function):
  """W cat[boxes[s], boxes[o], vecs[s], vecs[o]] + b, no activation: the first Linear of
  rel_aux_net (reference sg2im/model.py:149-152) when a BatchNorm1d follows it."""

  @staticmethod
  def _desc(boxes, vecs, s_idx, o_idx):
    return conv_desc([rows_src(boxes, s_idx), rows_src(boxes, o_idx), rows_src(vecs, s_idx),
                      rows_src(vecs, o_idx)], s_idx.numel(), 1, 1)

  @staticmethod
  def forward(ctx, boxes, vecs, s_idx, o_idx, csr, W, b, shadowed=False):
    boxes, vecs = boxes.contiguous(), vecs.contiguous()
    y = ops.conv2d_forward(RelAuxLinear._desc(boxes, vecs, s_idx, o_idx), W, W.size(0), b,
                           _new(vecs, s_idx.numel(), W.size(0)), W.size(0))
    ctx.save_for_backward(boxes, vecs, s_idx, o_idx, W, b)
    ctx.csr, ctx.shadowed = csr, shadowed
    return y

  @staticmethod
  def backward(ctx, g):
    boxes, vecs, s_idx, o_idx, W, b = ctx.saved_tensors
    E = vecs.size(1)
    ni = ctx.needs_input_grad
    dX, dW, db = _linear_bwd(RelAuxLinear._desc(boxes, vecs, s_idx, o_idx), W, g.contiguous(), ni[0] or ni[1],
                             ni[5], ni[6] and not ctx.shadowed, 8 + 2 * E, b)
    if ctx.shadowed:
      db = _shadowed_bias_grad(b, ni[6])
    d_boxes = d_vecs = None
    if ni[0]:
      d_boxes = ops.segment_sum(dX[:, 0:4], dX[:, 4:8], ctx.csr, 4, False, _new(vecs, boxes.size(0), 4))
    if ni[1]:
      d_vecs = ops.segment_sum(dX[:, 8:8 + E], dX[:, 8 + E:], ctx.csr, E, False, _new(vecs, vecs.size(0), E))
    return d_boxes, d_vecs, None, None, None, dW, db, None


class Embedding(Function):
  """weight[idx]  (nn.Embedding, reference sg2im/model.py:131,133); bit-exact row copy.
  Backward is a deterministic CSR segment-sum instead of atomics."""

  @staticmethod
  def forward(ctx, weight, idx, csr=None):
    """csr: None or ops.Csr(idx, None, weight.size(0)) built ahead of time (Sg2ImModel.forward_nhwc builds it on its
    aux stream at the head of the step instead of at the very end of the backward pass)"""
    out = ops.gather_rows(weight, idx, _new(weight, idx.numel(), weight.size(1)))
    ctx.save_for_backward(idx, weight)
    ctx.rows, ctx.csr = weight.size(0), csr
    return out

  @staticmethod
  def backward(ctx, g):
    idx, weight = ctx.saved_tensors
    g = g.contiguous()
    csr, ctx.csr = ctx.csr, None
    if csr is None:
      csr = ops.Csr(idx, None, ctx.rows)
    sk = _sink(weight)
    if sk is not None:
      ops.segment_sum(g, None, csr, g.size(1), False, sk, accumulate=True)
      return None, None, None
    dw = ops.segment_sum(g, None, csr, g.size(1), False, _new(g, ctx.rows, g.size(1)))
    return dw, None, None


# ----------------------------------------------------------------------------
# graph convolution
# ----------------------------------------------------------------------------

class GraphTripleConvFn(Function):
  """One GraphTripleConv layer (reference sg2im/graph.py:56-120): gather+concat folded into
  the net1 GEMM, deterministic CSR pooling, net2 - one call into the library per direction
  (sg2im_gconv_layer_forward / _backward), which runs the layer's launch sequence."""

  @staticmethod
  def forward(ctx, obj_vecs, pred_vecs, s_idx, o_idx, csr, avg, W1a, b1a, W1b, b1b, W2a, b2a, W2b, b2b):
    T, O = pred_vecs.size(0), obj_vecs.size(0)
    H, Dout = W2a.size(0), W2b.size(0)
    NT = W1b.size(0)                                  # 2H + Dout
    if obj_vecs.stride(1) != 1:
      obj_vecs = obj_vecs.contiguous()
    if pred_vecs.size(0) > 0 and pred_vecs.stride(1) != 1:
      pred_vecs = pred_vecs.contiguous()
    L = ops._gconv_layer_struct(obj_vecs, pred_vecs, s_idx, o_idx, csr, avg, (W1a, b1a, W1b, b1b, W2a, b2a, W2b, b2b))
    h1, new_t = _new(obj_vecs, T, H), _new(obj_vecs, T, NT)
    pooled, h2, new_obj = _new(obj_vecs, O, H), _new(obj_vecs, O, H), _new(obj_vecs, O, Dout)
    ops.gconv_layer_forward(L, h1, new_t, pooled, h2, new_obj)
    ctx.save_for_backward(obj_vecs, pred_vecs, s_idx, o_idx, W1a, W1b, W2a, W2b, h1, new_t, pooled, h2, new_obj,
                          b1a, b1b, b2a, b2b)
    ctx.csr, ctx.avg = csr, avg
    return new_obj, new_t[:, H:H + Dout]

  @staticmethod
  def backward(ctx, g_obj, g_pred):
    (obj_vecs, pred_vecs, s_idx, o_idx, W1a, W1b, W2a, W2b, h1, new_t, pooled, h2, new_obj,
     b1a, b1b, b2a, b2b) = ctx.saved_tensors
    csr, avg = ctx.csr, ctx.avg
    T, O, Din = pred_vecs.size(0), obj_vecs.size(0), obj_vecs.size(1)
    ni = ctx.needs_input_grad
    params = (W1a, b1a, W1b, b1b, W2a, b2a, W2b, b2b)
    need = ni[6:14]
    # parameter gradients: straight into the registered gradient sinks (accumulate), or into fresh tensors
    sinks = [_sink(p) if n else None for p, n in zip(params, need)]
    use_sinks = all((not n) or (sk is not None) for n, sk in zip(need, sinks))
    if use_sinks:
      bufs, ret = sinks, [None] * 8
    else:
      bufs = [torch.empty_like(p) if n else None for p, n in zip(params, need)]
      ret = bufs
    L = ops._gconv_layer_struct(obj_vecs, pred_vecs, s_idx, o_idx, csr, avg, params)
    d_triple = _new(obj_vecs, T, 3 * Din)
    d_obj = _new(obj_vecs, O, Din) if ni[0] else None
    ops.gconv_layer_backward(L, h1, new_t, pooled, h2, new_obj, None if g_obj is None else g_obj.contiguous(),
                             None if g_pred is None else (g_pred if g_pred.stride(1) == 1 else g_pred.contiguous()),
                             d_triple, d_obj, bufs, use_sinks)
    d_pred = d_triple[:, Din:2 * Din] if ni[1] else None
    return (d_obj, d_pred, None, None, None, None) + tuple(ret)


class GraphTripleConvStackFn(Function):
  """ALL GraphTripleConv layers of the model (reference sg2im/model.py:136-140: `gconv` then `gconv_net`, each layer
  sg2im/graph.py:56-120) as one persistent launch per direction (csrc/gcn_persist.hip): the stages of the layers
  are separated by grid barriers inside the kernel instead of by ~9 dependent launches per layer."""

  @staticmethod
  def forward(ctx, obj_vecs, pred_vecs, s_idx, o_idx, csr, avg, *params):
    nl = len(params) // 8
    T, O = pred_vecs.size(0), obj_vecs.size(0)
    if obj_vecs.stride(1) != 1:
      obj_vecs = obj_vecs.contiguous()
    if T > 0 and pred_vecs.stride(1) != 1:
      pred_vecs = pred_vecs.contiguous()
    weights = [tuple(params[8 * l:8 * l + 8]) for l in range(nl)]
    acts = []
    for w in weights:
      H, Dout, NT = w[4].size(0), w[6].size(0), w[2].size(0)
      acts.append((_new(obj_vecs, T, H), _new(obj_vecs, T, NT), _new(obj_vecs, O, H), _new(obj_vecs, O, H),
                   _new(obj_vecs, O, Dout)))
    S = ops.gconv_stack_struct(obj_vecs, pred_vecs, s_idx, o_idx, csr, avg, weights, acts)
    ops.gconv_stack_forward(S, obj_vecs.device)
    ctx.save_for_backward(obj_vecs, pred_vecs, s_idx, o_idx, *params, *[t for a in acts for t in a])
    ctx.csr, ctx.avg, ctx.nl = csr, avg, nl
    H, Dout = weights[-1][4].size(0), weights[-1][6].size(0)
    return acts[-1][4], acts[-1][1][:, H:H + Dout]

  @staticmethod
  def backward(ctx, g_obj, g_pred):
    nl = ctx.nl
    sv = ctx.saved_tensors
    obj_vecs, pred_vecs, s_idx, o_idx = sv[:4]
    params, flat_acts = sv[4:4 + 8 * nl], sv[4 + 8 * nl:]
    csr, avg = ctx.csr, ctx.avg
    T, O = pred_vecs.size(0), obj_vecs.size(0)
    need = ctx.needs_input_grad[6:]
    sinks = [_sink(p) if n else None for p, n in zip(params, need)]
    use_sinks = all((not n) or (sk is not None) for n, sk in zip(need, sinks))
    if use_sinks:
      bufs, ret = sinks, [None] * (8 * nl)
    else:
      bufs = [torch.empty_like(p) if n else None for p, n in zip(params, need)]
      ret = bufs
    g_obj = None if g_obj is None else g_obj.contiguous()
    if g_pred is not None and g_pred.stride(1) != 1:
      g_pred = g_pred.contiguous()
    if T > 0 and ops.gconv_stack_backward_in_one_launch():
      # ONE persistent launch for all layers (sg2im_gconv_stack_backward)
      weights = [tuple(params[8 * l:8 * l + 8]) for l in range(nl)]
      acts = [tuple(flat_acts[5 * l:5 * l + 5]) for l in range(nl)]
      S = ops.gconv_stack_struct(obj_vecs, pred_vecs, s_idx, o_idx, csr, avg, weights, acts)
      Din0 = obj_vecs.size(1)
      d_triple = _new(obj_vecs, T, 3 * Din0)
      d_obj = _new(obj_vecs, O, Din0) if ctx.needs_input_grad[0] else None
      ops.gconv_stack_backward(S, g_obj, g_pred, d_triple, d_obj, [bufs[8 * l:8 * l + 8] for l in range(nl)], use_sinks,
                               obj_vecs.device)
      return (d_obj, d_triple[:, Din0:2 * Din0] if ctx.needs_input_grad[1] else None, None, None, None, None) + tuple(ret)
    d_obj = d_pred = None
    # layer by layer, last first (each: sg2im_gconv_layer_backward)
    for l in range(nl - 1, -1, -1):
      w = params[8 * l:8 * l + 8]
      h1, new_t, pooled, h2, new_obj = flat_acts[5 * l:5 * l + 5]
      if l == 0:
        xin, pin = obj_vecs, pred_vecs
      else:
        pa = flat_acts[5 * (l - 1):5 * (l - 1) + 5]
        Hp, Dp = params[8 * (l - 1) + 4].size(0), params[8 * (l - 1) + 6].size(0)
        xin, pin = pa[4], pa[1][:, Hp:Hp + Dp]
      Din = xin.size(1)
      L = ops._gconv_layer_struct(xin, pin, s_idx, o_idx, csr, avg, w)
      d_triple = _new(obj_vecs, T, 3 * Din)
      want_obj = l > 0 or ctx.needs_input_grad[0]
      d_obj = _new(obj_vecs, O, Din) if want_obj else None
      ops.gconv_layer_backward(L, h1, new_t, pooled, h2, new_obj, g_obj, g_pred, d_triple, d_obj, bufs[8 * l:8 * l + 8], use_sinks)
      g_obj, g_pred = d_obj, d_triple[:, Din:2 * Din]
      d_pred = g_pred
    return (d_obj if ctx.needs_input_grad[0] else None, d_pred if ctx.needs_input_grad[1] else None,
            None, None, None, None) + tuple(ret)


class RelAux(Function):
  """rel_aux_net on cat[boxes[s], boxes[o], vecs[s], vecs[o]] (reference sg2im/model.py:149-152)"""

  @staticmethod
  def forward(ctx, boxes, vecs, s_idx, o_idx, csr, W1, b1, W2, b2):
    T = s_idx.numel()
    boxes, vecs = boxes.contiguous(), vecs.contiguous()
    d1 = conv_desc([rows_src(boxes, s_idx), rows_src(boxes, o_idx), rows_src(vecs, s_idx), rows_src(vecs, o_idx)],
                   T, 1, 1)
    h = ops.conv2d_forward(d1, W1, W1.size(0), b1, _new(vecs, T, W1.size(0)), W1.size(0), 0.0)
    y = ops.conv2d_forward(conv_desc([rows_src(h)], T, 1, 1), W2, W2.size(0), b2, _new(vecs, T, W2.size(0)),
                           W2.size(0), 0.0)
    ctx.save_for_backward(boxes, vecs, s_idx, o_idx, W1, W2, h, y, b1, b2)
    ctx.csr = csr
    return y

  @staticmethod
  def backward(ctx, g):
    boxes, vecs, s_idx, o_idx, W1, W2, h, y, b1, b2 = ctx.saved_tensors
    T, E = s_idx.numel(), vecs.size(1)
    ni = ctx.needs_input_grad
    dp2 = _act_bwd_rows(g, y, 0.0)
    dp1, dW2, db2 = _linear_bwd(conv_desc([rows_src(h)], T, 1, 1), W2, dp2, True, ni[7], ni[8], h.size(1), b2,
                                act=(h, 0.0))
    d1 = conv_desc([rows_src(boxes, s_idx), rows_src(boxes, o_idx), rows_src(vecs, s_idx), rows_src(vecs, o_idx)],
                   T, 1, 1)
    dX, dW1, db1 = _linear_bwd(d1, W1, dp1, ni[0] or ni[1], ni[5], ni[6], 8 + 2 * E, b1)
    d_boxes = d_vecs = None
    if ni[0]:
      d_boxes = ops.segment_sum(dX[:, 0:4], dX[:, 4:8], ctx.csr, 4, False, _new(vecs, boxes.size(0), 4))
    if ni[1]:
      d_vecs = ops.segment_sum(dX[:, 8:8 + E], dX[:, 8 + E:], ctx.csr, E, False, _new(vecs, vecs.size(0), E))
    return d_boxes, d_vecs, None, None, None, dW1, db1, dW2, db2


# ----------------------------------------------------------------------------
# layout
# ----------------------------------------------------------------------------

class LayoutFn(Function):
  """masks_to_layout / boxes_to_layout (+ the layout-noise concat of model.py:164-169):
  returns the NHWC tensor (N, H, W, D + noise_dim) the refinement network consumes."""

  @staticmethod
  def forward(ctx, vecs, boxes, masks, obj_to_img, noise, n_images, H, W, align_corners, img_csr=None,
              pyramid_levels=0, link=None):
    """img_csr: the per-image object lists (ops.Csr(obj_to_img, None, n_images)) when the caller built them
    already (Sg2ImModel does, off the critical path).  link (a LayoutLink) + pyramid_levels > 0: the caller feeds
    the result to a refinement network of pyramid_levels + 1 modules AND TO NOTHING ELSE; that network's 2 x 2
    average-pool pyramid (crn.py:58-62) is then produced by the same kernel and handed over through the link, and
    the backward pass takes the per-level gradients back through it (see LayoutLink)."""
    D = vecs.size(1)
    nd = noise.size(1) if noise is not None else 0
    if vecs.stride(1) != 1:
      vecs = vecs.contiguous()
    boxes = boxes.contiguous()
    if img_csr is None:
      img_csr = ops.Csr(obj_to_img, None, n_images)
    out = _new(vecs, n_images, H, W, D + nd)
    nlev = min(int(pyramid_levels), 4) if link is not None else 0
    if ops.LAYOUT_PYRAMID and D % 32 == 0 and nd % 32 == 0 and H % 16 == 0 and W % 16 == 0 and n_images > 0:
      levels = [out] + [_new(vecs, n_images, H >> l, W >> l, D + nd) for l in range(1, nlev + 1)]
      ops.layout_pyramid_forward(vecs, boxes, masks, img_csr, n_images, H, W, align_corners,
                                 noise.contiguous() if nd > 0 else None, levels)
      if nlev > 0:
        link.offer(out, levels)
    else:
      ops.layout_forward(vecs, boxes, masks, img_csr, n_images, H, W, align_corners, out)
      if nd > 0:
        ops.nchw_to_nhwc(noise.contiguous(), out, D)
    ctx.save_for_backward(vecs, boxes, masks, obj_to_img)
    ctx.img_csr, ctx.geom = img_csr, (n_images, H, W, align_corners)
    ctx.link = link
    return out

  @staticmethod
  def backward(ctx, g):
    vecs, boxes, masks, obj_to_img = ctx.saved_tensors
    n_images, H, W, ac = ctx.geom
    ni = ctx.needs_input_grad
    # (the per-level gradients of the refinement network, if it left them in the link: `g` is then the UNWRITTEN
    # tensor it returned - taken BEFORE anything touches g)
    lazy = ctx.link.take_grad(g) if ctx.link is not None else None
    g = g.contiguous()
    d_vecs = _new(vecs, vecs.size(0), vecs.size(1)) if ni[0] else None
    d_boxes = _new(vecs, boxes.size(0), 4) if ni[1] else None      # predicted boxes laid out (boxes_gt=None)
    d_masks = None
    if ni[2] and masks is not None and masks.is_floating_point():
      d_masks = _new(vecs, *masks.shape)
    todo_vecs, todo_maps = d_vecs is not None, d_masks is not None or d_boxes is not None
    if lazy is not None:
      levels, factors, Cg = lazy
      D = vecs.size(1)
      if (D % 4 == 0 and D <= Cg and (g.size(1), g.size(2)) == (H, W) and all(t.size(3) % 4 == 0 for t in levels) and
          LAYOUT_GRAD_FROM_LEVELS):
        # both halves straight from the per-level gradients: the full-resolution sum is never written (COCO style: only
        # d_vecs; VG style / predicted boxes: d_masks, d_boxes too - mask_net trains through the layout)
        ops.mark('layout_bwd_start')
        if todo_vecs:
          ops.layout_backward_vecs_levels(levels, factors, vecs, boxes, masks, ctx.img_csr, n_images, H, W, ac, d_vecs)
          todo_vecs = False
        if todo_maps:
          todo_maps = not ops.layout_backward_maps_levels(levels, factors, vecs, boxes, masks, ctx.img_csr, n_images, H, W,
                                                          ac, d_masks, d_boxes)
        ops.mark('layout_bwd_done')
      if todo_vecs or todo_maps:                         # the summed gradient is needed as a tensor after all
        if Cg < g.size(3):
          g[..., Cg:].zero_()
        ops.pyramid_backward(levels, factors, [Cg] * len(levels), g.size(0), H, W, Cg, g)
    if todo_vecs or todo_maps:
      ops.layout_backward(g, vecs, boxes, masks, obj_to_img, ctx.img_csr, n_images, H, W, ac,
                          d_vecs if todo_vecs else None, d_masks if todo_maps else None, d_boxes if todo_maps else None)
    return d_vecs, d_boxes, d_masks, None, None, None, None, None, None, None, None, None


class LayoutLink(object):
  """The explicit hand-over between ONE LayoutFn call and the ONE RefinementFn call that consumes its result
  (Sg2ImModel.forward_nhwc creates a link per forward pass and passes it to both; nothing else may read that layout).

  forward:  LayoutFn leaves the refinement network's average-pool pyramid (crn.py:58-62), which its kernel produced
            next to the full-resolution layout, in ``levels``; RefinementFn takes it over instead of pooling.
  backward: a refinement network that took the pyramid does not sum its per-level layout gradients into a
            full-resolution tensor - it returns an UNWRITTEN tensor of that shape and leaves the levels here;
            LayoutFn.backward computes d_vecs from the levels directly (ops.layout_backward_vecs_levels) or, when
            mask / box gradients are wanted too, materialises the sum first.  [-(67 MB written + 67 MB re-read) per step]

  The unwritten tensor is only ever interpreted through the link: LayoutFn.backward checks that the gradient it was
  handed IS that tensor (same storage, same version, nothing accumulated into or copied from it) and raises
  otherwise - a second consumer of the layout, a tensor hook or retain_grad on it would make autograd sum or copy
  uninitialised memory, which must not go unnoticed.  Callers that cannot promise a single consumer pass
  no link and get the plain, materialised gradient."""

  def __init__(self):
    self.layout_ptr = None
    self.levels = None
    self.taken = False
    self.grad = None            # (unwritten tensor, its _version at hand-over, levels, factors, Cg)

  def offer(self, layout, levels):
    self.layout_ptr, self.levels, self.taken = layout.data_ptr(), levels, False

  def take_pyramid(self, layout):
    """the levels LayoutFn produced for exactly this tensor, else None"""
    if self.levels is None or self.layout_ptr != layout.data_ptr():
      return None
    levels, self.levels, self.taken = self.levels, None, True
    return levels

  def leave_grad(self, dlayout, levels, factors, Cg):
    if self.grad is not None:
      raise RuntimeError('LayoutLink: a pending layout gradient was never consumed (two backward passes through one link?)')
    self.grad = (dlayout, dlayout._version, levels, factors, Cg)

  def take_grad(self, g):
    if self.grad is None:
      return None
    dlayout, version, levels, factors, Cg = self.grad
    self.grad = None
    if g is not dlayout and (g.data_ptr() != dlayout.data_ptr() or g.shape != dlayout.shape or g.stride() != dlayout.stride()):
      raise RuntimeError('LayoutLink: the layout gradient reaching LayoutFn.backward is not the tensor the refinement '
                         'network handed over - the layout has a second consumer, a hook or retain_grad; build the model '
                         'with SG2IM_LAZY_LAYOUT_GRAD=0 (materialised layout gradient) for such graphs')
    if dlayout._version != version:
      raise RuntimeError('LayoutLink: the handed-over (unwritten) layout gradient was modified in place before '
                         'LayoutFn.backward ran')
    return levels, factors, Cg


import os as _os
WGRAD_FLUSH_AT = int(_os.environ.get('SG2IM_WGRAD_FLUSH_AT', '-1'))          # (probe knob: early release at module i)
LAZY_LAYOUT_GRAD = _os.environ.get('SG2IM_LAZY_LAYOUT_GRAD', '1') != '0'     # (A/B knob)
LAYOUT_GRAD_FROM_LEVELS = _os.environ.get('SG2IM_LAYOUT_GRAD_LEVELS', '1') != '0'      # (A/B knob: 0 = materialise the sum)


def _hand_over_layout_grad(like, dlevels, N, H, W, Cg, Cl, link):
  """the full-resolution d layout as LayoutFn.backward's input: lazily through the link (see LayoutLink) or summed now"""
  lazy = link is not None and LAZY_LAYOUT_GRAD and len(dlevels) <= 6 and all(f & (f - 1) == 0 for _, f in dlevels)
  if lazy:
    dlayout = _new(like, N, H, W, Cl)                  # never read as a tensor: LayoutFn.backward takes the levels
    link.leave_grad(dlayout, [t for t, _ in dlevels], [f for _, f in dlevels], Cg)
    return dlayout
  dlayout = _new(like, N, H, W, Cl) if Cg == Cl else torch.zeros(N, H, W, Cl, dtype=torch.float32, device=like.device)
  ops.pyramid_backward([t for t, _ in dlevels], [f for _, f in dlevels], [Cg] * len(dlevels), N, H, W, Cg, dlayout)
  return dlayout


def _layout_pyramid(layout, L, link=None):
  """[coarsest, ..., full resolution]: the L levels a refinement network of L modules reads (crn.py:58-62 pools
  the full-resolution layout once per module; each level here is the 2 x 2 mean of the next finer one - the same
  value up to fp32 summation order).  Levels LayoutFn already produced are taken over (LayoutLink), the rest pooled.
  Second result: the pyramid came through the link (LayoutFn.backward then accepts the per-level gradients)."""
  N, H, W, Cl = layout.shape
  handed = link.take_pyramid(layout) if link is not None else None
  pyr = (handed or [layout])[:L]
  for i in range(len(pyr), L):
    pyr.append(ops.avgpool_forward(pyr[-1], 2, _new(layout, N, H >> i, W >> i, Cl)))
  return pyr[::-1], handed is not None


class CropFn(Function):
  """crop_bbox_batch (reference sg2im/bilinear.py:28-132) on NHWC images -> NHWC crops"""

  @staticmethod
  def forward(ctx, imgs, boxes, obj_to_img, size, align_corners):
    imgs, boxes = imgs.contiguous(), boxes.contiguous()
    out = ops.crop_forward(imgs, boxes, obj_to_img, size, align_corners,
                           _new(imgs, boxes.size(0), size, size, imgs.size(3)))
    ctx.save_for_backward(boxes, obj_to_img)
    ctx.geom = (tuple(imgs.shape), size, align_corners)
    return out

  @staticmethod
  def backward(ctx, g):
    boxes, obj_to_img = ctx.saved_tensors
    shape, size, ac = ctx.geom
    d = torch.empty(shape, dtype=torch.float32, device=g.device)     # every pixel is written
    ops.crop_backward(g.contiguous(), boxes, obj_to_img, size, ac, d)
    return d, None, None, None, None


# ----------------------------------------------------------------------------
# conv blocks
# ----------------------------------------------------------------------------

def _conv_param_grads(desc, dy, cout, w_phys_shape, need_w, need_b, W=None, b=None):
  """dy: dense NHWC gradient of the conv output.  W / b: the parameters (sink lookup)."""
  dw = db = None
  rows = dy.numel() // cout
  if need_w:
    sk = _sink(W)
    skb = _sink(b) if need_b else None
    if sk is not None:
      ops.conv2d_backward_weight(desc, dy, cout, cout, sk, accumulate=True, dbias=skb)
      need_b = need_b and skb is None
    else:
      # (weight rows wider than the sources: the columns the kernel does not write are exact zeros)
      dw = torch.zeros(*w_phys_shape, dtype=torch.float32, device=dy.device) if desc.weight_channels else _new(dy, *w_phys_shape)
      if need_b and skb is None:
        db = _new(dy, cout)
        need_b = False
      ops.conv2d_backward_weight(desc, dy, cout, cout, dw, dbias=db)
      dw = _cl_grad(dw)
  if need_b:
    sk = _sink(b)
    if sk is not None:
      ops.column_sum(_fptr(dy), rows, cout, cout, sk, accumulate=True)
    else:
      db = _new(dy, cout)
      ops.column_sum(_fptr(dy), rows, cout, cout, db)
  return dw, db


def _shadowed_bias_grad(b, need):
  """Gradient of a conv bias that feeds a *training-mode* BatchNorm: the batch-mean
  subtraction cancels the bias, so the gradient is exactly zero.  The reference computes
  it numerically (sum of dy, ~1e-8 rounding noise); here it is not computed at all."""
  if not need or _sink(b) is not None:
    return None                      # the zeroed gradient arena already holds the answer
  return torch.zeros_like(b)


def _bn_grad_bufs(like, C, gamma, beta, need_g, need_b):
  """(dgamma buffer, dbeta buffer, accumulate, return_dgamma, return_dbeta)"""
  sg, sb = _sink(gamma) if need_g else None, _sink(beta) if need_b else None
  if need_g and need_b and sg is not None and sb is not None:
    return sg, sb, True, None, None
  dg = _new(like, C) if need_g else None
  db = _new(like, C) if need_b else None
  return dg, db, False, dg, db


def _crn_conv0_desc(lay, feat_src, N, h, w, W0p):
  """descriptor of a refinement module's first convolution: [layout level, previous features]; the first
  module has no previous features (the reference's all-zero channel, see RefinementFn._forward)"""
  if feat_src is None:
    return conv_desc([nhwc_src(lay)], N, h, w, 3, 3, 1, 1, weight_channels=W0p.size(1))
  return conv_desc([nhwc_src(lay), feat_src], N, h, w, 3, 3, 1, 1)


class RefinementFn(Function):
  """RefinementNetwork.forward (reference sg2im/crn.py:88-111) on an NHWC layout.

  params (flat): per module [W0, b0, W1, b1] ..., then [Wo0, bo0, Wo2, bo2];
  bns: list of (bn0, bn1) module pairs (parameter containers);  BN gamma/beta are passed
  as tensors as well so autograd sees them: per module [g0, be0, g1, be1] after the conv
  params of ALL modules and the output convs.
  """

  @staticmethod
  def forward(ctx, layout, bns, slope, training, grad_channels, link, *params):
    ops.TIMER_TAG = 'crn'
    ops.mark('crn_fwd_start')
    try:
      return RefinementFn._forward(ctx, layout, bns, slope, training, grad_channels, link, *params)
    finally:
      ops.TIMER_TAG = None

  @staticmethod
  def _forward(ctx, layout, bns, slope, training, grad_channels, link, *params):
    L = len(bns)
    N, H, W, Cl = layout.shape
    convp = params[:4 * L]
    Wo0, bo0, Wo2, bo2 = params[4 * L:4 * L + 4]
    bnp = params[4 * L + 4:]
    h0, w0 = H >> L, W >> L
    if h0 == 0 or w0 == 0:
      raise AssertionError('too many refinement modules for this image size')     # crn.py:103-104
    layout = layout.contiguous()
    # crn.py:105 feeds the first module a constant all-zero 1-channel feature map: it adds nothing to the
    # convolution and its weight-gradient column is exactly zero, so the first conv0 runs over the layout
    # channels only, on weight rows that are one channel wider (sg2im_conv_desc.weight_channels) - with the
    # zero channel its 161 input channels would fall off the vector loaders
    feat_src = None
    saved = []
    pyr, from_link = _layout_pyramid(layout, L, link)
    ctx.link = link if from_link else None
    sbm = _storage_levels(N, H, W, L, [convp[4 * i].size(0) for i in range(L)])

    def activated(y, st, up):
      """the source the next convolution reads: leaky(bn(y)), pending in its loader"""
      return nhwc_src(y, up, st.scale, st.shift, slope)
    for i in range(L):
      h, w = H >> (L - 1 - i), W >> (L - 1 - i)
      lay = pyr[i]
      W0p, b0, W1p, b1 = convp[4 * i:4 * i + 4]
      C = W0p.size(0)
      bn0, bn1 = bns[i]
      d0 = _crn_conv0_desc(lay, feat_src, N, h, w, W0p)
      # (the BatchNorm statistics of a conv output come out of the conv's own launches: epilogue or split-K finish)
      y0 = _new_s(layout, sbm[i], N, h, w, C)
      st0 = ops.conv2d_forward_bn(d0, _cl_weight(W0p), C, b0, y0, C, bn0, training, BN_EPS, BN_MOMENTUM)
      src0 = activated(y0, st0, 0)
      d1 = conv_desc([src0], N, h, w, 3, 3, 1, 1)
      y1 = _new_s(layout, sbm[i], N, h, w, C)
      st1 = ops.conv2d_forward_bn(d1, _cl_weight(W1p), C, b1, y1, C, bn1, training, BN_EPS, BN_MOMENTUM)
      saved.append((lay, feat_src, y0, st0, y1, st1, h, w, C, src0))
      feat_src = activated(y1, st1, 1)
    last = saved[-1]
    Cf = last[8]
    # (the last module's activated output at full resolution: the same tensor `feat_src` holds, not upsampled)
    do0 = conv_desc([ops.SrcSpec(feat_src.t, feat_src.channels, feat_src.ld, 0, None, feat_src.scale, feat_src.shift,
                                 feat_src.slope)], N, H, W, 3, 3, 1, 1)
    z = ops.conv2d_forward(do0, _cl_weight(Wo0), Wo0.size(0), bo0, _new(layout, N, H, W, Wo0.size(0)),
                           Wo0.size(0), slope)
    do2 = conv_desc([nhwc_src(z)], N, H, W, 1, 1, 1, 0)
    img = ops.conv2d_forward(do2, _cl_weight(Wo2), Wo2.size(0), bo2, _new(layout, N, H, W, Wo2.size(0)),
                             Wo2.size(0))
    ctx.saved = saved
    ctx.misc = (L, slope, training, z, do0, do2, Cl, Cf, grad_channels)
    ctx.sbm = sbm
    ctx.save_for_backward(*params)
    ctx.shape = (N, H, W, Cl)
    return img

  @staticmethod
  def backward(ctx, g):
    ops.mark('crn_bwd_start')
    ops.TIMER_TAG = 'crn'
    try:
      out = RefinementFn._backward(ctx, g)
      ops.mark('crn_bwd_done')
      return out
    finally:
      ops.TIMER_TAG = None

  @staticmethod
  def _backward(ctx, g):
    params = ctx.saved_tensors
    L, slope, training, z, do0, do2, Cl, Cf, grad_channels = ctx.misc
    N, H, W, _ = ctx.shape
    saved = ctx.saved
    convp = params[:4 * L]
    Wo0, bo0, Wo2, bo2 = params[4 * L:4 * L + 4]
    bnp = params[4 * L + 4:]
    ni = ctx.needs_input_grad[6:]
    grads = [None] * len(params)
    g = g.contiguous()
    Co = Wo0.size(0)
    # weight gradients go to a second stream when they accumulate straight into registered sinks
    # (otherwise their results would be allocated on that stream and handed to autograd from it)
    side = ops.SideLane(g.device)
    side.on = side.on and all(_sink(p) is not None for p in params[:4 * L + 4])
    deferred = side.on and ops.DEFERRED is not None       # (Trainer: released at the end of the dgrad chain)
    def wgrad(desc, dy, cout, shape, need_w, need_b, Wp, bp):
      if deferred:
        def run(background):
          if background:
            desc.launch_hints |= ops.HINT_BACKGROUND
          _conv_param_grads(desc, dy, cout, shape, need_w, need_b, Wp, bp)
        side.defer(run, dy, desc, completes=(Wp, bp))
        return None, None
      return side.run(lambda: _conv_param_grads(desc, dy, cout, shape, need_w, need_b, Wp, bp), dy, desc)
    # Order per layer: data gradient (big, alone on the GPU), then its weight gradient on the side
    # stream underneath the small kernels that lead to the next data gradient, which waits for it.
    # output 1x1 conv
    dz = _new(g, N, H, W, Co)
    # (d loss / d pre-activation of output_conv[0]: the LeakyReLU mask of z rides in the data gradient's epilogue)
    ops.conv2d_backward_data_act(do2, _cl_weight(Wo2), Wo2.size(0), g, Wo2.size(0), 0, Co, dz, Co, z, Co, slope)
    grads[4 * L + 2], grads[4 * L + 3] = wgrad(do2, g, Wo2.size(0), (Wo2.size(0), 1, 1, Co),
                                               ni[4 * L + 2], ni[4 * L + 3], Wo2, bo2)
    sbm = ctx.sbm
    gz = _new_s(g, sbm[L - 1], N, H, W, Cf)        # grad w.r.t. activated feats of the last module
    side.barrier()
    # Every data gradient below that produces the gradient of a BatchNorm'd layer's activated output also
    # produces that BatchNorm's backward reductions (epilogue / split-K finish) and coefficients: per layer
    # only the elementwise `apply` pass is left between two data gradients of the chain.
    def bn_bufs(i, second):
      k_ = 4 * L + 4 + 4 * i + (2 if second else 0)
      gam, bet = bnp[4 * i + (2 if second else 0)], bnp[4 * i + (3 if second else 1)]
      dgm, dbt, acc_, grads[k_], grads[k_ + 1] = _bn_grad_bufs(g, gam.numel(), gam, bet, ni[k_], ni[k_ + 1])
      return gam, dgm, dbt, acc_
    lastm = saved[L - 1]
    g1, dg1, db1n, acc1 = bn_bufs(L - 1, True)
    coef1 = ops.conv2d_backward_data_bn(do0, _cl_weight(Wo0), Co, dz, Co, 0, Cf, gz, Cf, lastm[4], Cf, 0, g1, lastm[5],
                                        slope, training, dg1, db1n, acc1)
    grads[4 * L], grads[4 * L + 1] = wgrad(do0, dz, Co, (Co, 3, 3, Cf), ni[4 * L], ni[4 * L + 1], Wo0, bo0)
    pool2 = 0
    need_layout = ctx.needs_input_grad[0]
    # layout channels that need gradients (the noise channels appended by the model do not)
    Cg = Cl if grad_channels is None else min(int(grad_channels), Cl)
    dlevels = []
    for i in range(L - 1, -1, -1):
      lay, feat_src, y0, st0, y1, st1, h, w, C, src0 = saved[i]
      W0p, b0, W1p, b1 = convp[4 * i:4 * i + 4]
      if deferred and i == WGRAD_FLUSH_AT:
        side.flush(final=False)
      dy1 = ops.bn_backward_apply(_fptr(gz), gz.size(3), pool2, N, h, w, y1, C, C, st1, slope, coef1,
                                  _new_s(g, sbm[i], N, h, w, C), g_dtype=ops._dt(gz))
      d1 = conv_desc([src0], N, h, w, 3, 3, 1, 1)
      gz0 = _new_s(g, sbm[i], N, h, w, C)
      side.barrier()
      g0, dg0, db0n, acc0 = bn_bufs(i, False)
      coef0 = ops.conv2d_backward_data_bn(d1, _cl_weight(W1p), C, dy1, C, 0, C, gz0, C, y0, C, 0, g0, st0, slope,
                                          training, dg0, db0n, acc0)
      grads[4 * i + 2], grads[4 * i + 3] = wgrad(d1, dy1, C, (C, 3, 3, C), ni[4 * i + 2],
                                                  ni[4 * i + 3] and not training, W1p, b1)
      if training:
        grads[4 * i + 3] = _shadowed_bias_grad(b1, ni[4 * i + 3])
      # (dy1's buffer is recycled for dy0 unless a side-stream weight gradient may still be reading it)
      dy0 = ops.bn_backward_apply(_fptr(gz0), C, 0, N, h, w, y0, C, C, st0, slope, coef0,
                                  _new_s(g, sbm[i], N, h, w, C) if side.on else dy1, g_dtype=ops._dt(gz0))
      Cprev = feat_src.channels if feat_src is not None else W0p.size(1) - Cl
      d0 = _crn_conv0_desc(lay, feat_src, N, h, w, W0p)
      side.barrier()
      if need_layout:
        dl = _new(g, N, h, w, Cg)
        ops.conv2d_backward_data(d0, _cl_weight(W0p), C, dy0, C, 0, Cg, dl, Cg)
        dlevels.append((dl, H // h))
      if i > 0:
        gz = _new_s(g, sbm[i - 1], N, h, w, Cprev)  # at this (upsampled) resolution; summed 2x2 next
        prevm = saved[i - 1]
        g1, dg1, db1n, acc1 = bn_bufs(i - 1, True)
        coef1 = ops.conv2d_backward_data_bn(d0, _cl_weight(W0p), C, dy0, C, Cl, Cprev, gz, Cprev, prevm[4], Cprev, 1, g1,
                                            prevm[5], slope, training, dg1, db1n, acc1)
        pool2 = 1
      grads[4 * i], grads[4 * i + 1] = wgrad(d0, dy0, C, (C, 3, 3, Cl + Cprev), ni[4 * i],
                                              ni[4 * i + 1] and not training, W0p, b0)
      if training:
        grads[4 * i + 1] = _shadowed_bias_grad(b0, ni[4 * i + 1])
    if deferred:
      side.flush()
      ops.DEFERRED.append(side)                    # (joined by the Trainer before the optimiser step)
    dlayout = None
    if need_layout:
      dlayout = _hand_over_layout_grad(g, dlevels, N, H, W, Cg, Cl, getattr(ctx, 'link', None))
    if not deferred:
      side.join()
    ctx.saved = None
    return (dlayout, None, None, None, None, None) + tuple(grads)


class RefinementNoNormFn(Function):
  """RefinementNetwork with normalization='none' (reference sg2im/crn.py:41-47 drops the norm
  layers): conv3x3 + LeakyReLU fused in the conv epilogue, twice per module.
  params (flat): per module [W0, b0, W1, b1] ..., then [Wo0, bo0, Wo2, bo2]."""

  @staticmethod
  def forward(ctx, layout, n_modules, slope, grad_channels, inorm, link, *params):
    ops.TIMER_TAG = 'crn'
    try:
      L = n_modules
      N, H, W, Cl = layout.shape
      h0, w0 = H >> L, W >> L
      if h0 == 0 or w0 == 0:
        raise AssertionError('too many refinement modules for this image size')     # crn.py:103-104
      layout = layout.contiguous()
      feats = torch.zeros(N, h0, w0, 1, dtype=torch.float32, device=layout.device)   # crn.py:105
      feat_src = nhwc_src(feats, up=1)
      pyr, from_link = _layout_pyramid(layout, L, link)
      ctx.link = link if from_link else None
      saved = []
      for i in range(L):
        h, w = H >> (L - 1 - i), W >> (L - 1 - i)
        W0p, b0, W1p, b1 = params[4 * i:4 * i + 4]
        C = W0p.size(0)
        d0 = conv_desc([nhwc_src(pyr[i]), feat_src], N, h, w, 3, 3, 1, 1)
        # 'none': the activation rides in the conv epilogue; 'instance': the conv output is kept
        # (y), normalised per image and activated into a separate tensor (a)
        y0 = st0 = y1 = st1 = None
        a0 = ops.conv2d_forward(d0, _cl_weight(W0p), C, b0, _new(layout, N, h, w, C), C, 1.0 if inorm else slope)
        if inorm:
          y0, st0 = a0, ops.instnorm_stats(a0, BN_EPS)
          a0 = ops.instnorm_act_forward(y0, st0, slope, _new(layout, N, h, w, C))
        d1 = conv_desc([nhwc_src(a0)], N, h, w, 3, 3, 1, 1)
        a1 = ops.conv2d_forward(d1, _cl_weight(W1p), C, b1, _new(layout, N, h, w, C), C, 1.0 if inorm else slope)
        if inorm:
          y1, st1 = a1, ops.instnorm_stats(a1, BN_EPS)
          a1 = ops.instnorm_act_forward(y1, st1, slope, _new(layout, N, h, w, C))
        saved.append((pyr[i], feat_src, a0, a1, h, w, C, y0, st0, y1, st1))
        feat_src = nhwc_src(a1, 1)
      Wo0, bo0, Wo2, bo2 = params[4 * L:4 * L + 4]
      Cf = saved[-1][6]
      do0 = conv_desc([nhwc_src(saved[-1][3])], N, H, W, 3, 3, 1, 1)
      z = ops.conv2d_forward(do0, _cl_weight(Wo0), Wo0.size(0), bo0, _new(layout, N, H, W, Wo0.size(0)),
                             Wo0.size(0), slope)
      do2 = conv_desc([nhwc_src(z)], N, H, W, 1, 1, 1, 0)
      img = ops.conv2d_forward(do2, _cl_weight(Wo2), Wo2.size(0), bo2, _new(layout, N, H, W, Wo2.size(0)),
                               Wo2.size(0))
      ctx.saved = saved
      ctx.misc = (L, slope, z, do0, do2, Cl, Cf, grad_channels, (N, H, W), inorm)
      ctx.save_for_backward(*params)
      return img
    finally:
      ops.TIMER_TAG = None

  @staticmethod
  def backward(ctx, g):
    ops.TIMER_TAG = 'crn'
    try:
      params = ctx.saved_tensors
      L, slope, z, do0, do2, Cl, Cf, grad_channels, (N, H, W), inorm = ctx.misc
      saved = ctx.saved
      ni = ctx.needs_input_grad[6:]
      grads = [None] * len(params)
      Wo0, bo0, Wo2, bo2 = params[4 * L:4 * L + 4]
      g = g.contiguous()
      Co = Wo0.size(0)
      grads[4 * L + 2], grads[4 * L + 3] = _conv_param_grads(do2, g, Wo2.size(0), (Wo2.size(0), 1, 1, Co),
                                                             ni[4 * L + 2], ni[4 * L + 3], Wo2, bo2)
      dz = _new(g, N, H, W, Co)
      ops.conv2d_backward_data_act(do2, _cl_weight(Wo2), Wo2.size(0), g, Wo2.size(0), 0, Co, dz, Co, z, Co, slope)
      grads[4 * L], grads[4 * L + 1] = _conv_param_grads(do0, dz, Co, (Co, 3, 3, Cf), ni[4 * L], ni[4 * L + 1], Wo0, bo0)
      gz = _new(g, N, H, W, Cf)
      ops.conv2d_backward_data(do0, _cl_weight(Wo0), Co, dz, Co, 0, Cf, gz, Cf)
      pool2 = 0
      need_layout = ctx.needs_input_grad[0]
      Cg = Cl if grad_channels is None else min(int(grad_channels), Cl)
      dlevels = []
      for i in range(L - 1, -1, -1):
        lay, feat_src, a0, a1, h, w, C, y0, st0, y1, st1 = saved[i]
        W0p, b0, W1p, b1 = params[4 * i:4 * i + 4]
        # through the second activation (with the 2x2 sum of the nearest-upsample backward)
        dy1 = ops.act_backward(_fptr(gz), gz.size(3), pool2, N, h, w, a1, C, C, slope, _new(g, N, h, w, C))
        if inorm:
          ops.instnorm_backward(dy1, y1, st1, dy1)
        d1 = conv_desc([nhwc_src(a0)], N, h, w, 3, 3, 1, 1)
        # (a bias in front of an instance norm is cancelled by the mean subtraction: zero gradient)
        grads[4 * i + 2], grads[4 * i + 3] = _conv_param_grads(d1, dy1, C, (C, 3, 3, C), ni[4 * i + 2],
                                                                ni[4 * i + 3] and not inorm, W1p, b1)
        if inorm:
          grads[4 * i + 3] = _shadowed_bias_grad(b1, ni[4 * i + 3])
        gz0 = _new(g, N, h, w, C)
        ops.conv2d_backward_data(d1, _cl_weight(W1p), C, dy1, C, 0, C, gz0, C)
        dy0 = ops.act_backward(_fptr(gz0), C, 0, N, h, w, a0, C, C, slope, gz0)
        if inorm:
          ops.instnorm_backward(dy0, y0, st0, dy0)
        Cprev = feat_src.channels
        d0 = conv_desc([nhwc_src(lay), feat_src], N, h, w, 3, 3, 1, 1)
        grads[4 * i], grads[4 * i + 1] = _conv_param_grads(d0, dy0, C, (C, 3, 3, Cl + Cprev), ni[4 * i],
                                                            ni[4 * i + 1] and not inorm, W0p, b0)
        if inorm:
          grads[4 * i + 1] = _shadowed_bias_grad(b0, ni[4 * i + 1])
        if need_layout:
          dl = _new(g, N, h, w, Cg)
          ops.conv2d_backward_data(d0, _cl_weight(W0p), C, dy0, C, 0, Cg, dl, Cg)
          dlevels.append((dl, H // h))
        if i > 0:
          gz = _new(g, N, h, w, Cprev)
          ops.conv2d_backward_data(d0, _cl_weight(W0p), C, dy0, C, Cl, Cprev, gz, Cprev)
          pool2 = 1
      dlayout = None
      if need_layout:
        dlayout = _hand_over_layout_grad(g, dlevels, N, H, W, Cg, Cl, getattr(ctx, 'link', None))
      ctx.saved = None
      return (dlayout, None, None, None, None, None) + tuple(grads)
    finally:
      ops.TIMER_TAG = None


class MaskNetFn(Function):
  """mask_net (reference sg2im/model.py:94-106,146-147): [up2, BN, conv3x3, ReLU] x k, conv1x1,
  sigmoid.  params: per block [gamma, beta, W, b] ..., then [Wf, bf]."""

  @staticmethod
  def forward(ctx, obj_vecs, bns, training, count, *params):
    """count: None or (int32 device scalar, 1) - the number of real objects of a padded batch"""
    nb = len(bns)
    O, D = obj_vecs.shape
    cnt = lambda unit: None if count is None else (count[0], count[1] * unit)
    x = obj_vecs.contiguous().view(O, 1, 1, D)
    saved = []
    s = 1
    # statistics of the upsampled tensor == statistics of x (each value repeated 4x); only the unbiased
    # running_var factor sees the repeated count.  The first block normalises obj_vecs (standalone pass), every
    # later one the previous block's conv + ReLU output, whose statistics ride in that conv's launches.
    st = ops.bn_stats(x, O, D, D, bns[0], training, BN_EPS, BN_MOMENTUM, unbiased_rows=4 * O, count=cnt(1))
    for b in range(nb):
      gam, bet, Wp, bias = params[4 * b:4 * b + 4]
      d = conv_desc([nhwc_src(x, 1, st.scale, st.shift, 1.0)], O, 2 * s, 2 * s, 3, 3, 1, 1)
      y = _new(x, O, 2 * s, 2 * s, D)
      s2 = 2 * s
      if b + 1 < nb:
        st_next = ops.conv2d_forward_bn(d, _cl_weight(Wp), D, bias, y, D, bns[b + 1], training, BN_EPS, BN_MOMENTUM,
                                        out_slope=0.0, unbiased_rows=4 * O * s2 * s2, count=cnt(s2 * s2))
      else:
        st_next = None
        ops.conv2d_forward(d, _cl_weight(Wp), D, bias, y, D, 0.0)
      saved.append((x, st, y, s))
      x, st = y, st_next
      s = s2
    Wf, bf = params[4 * nb:4 * nb + 2]
    df = conv_desc([nhwc_src(x)], O, s, s, 1, 1, 1, 0)
    scores = ops.conv2d_forward(df, _cl_weight(Wf), 1, bf, _new(x, O, s, s, 1), 1)
    masks = ops.sigmoid_forward(scores, _new(x, O, s, s))
    ctx.saved, ctx.misc = saved, (nb, training, x, df, s, count)
    ctx.save_for_backward(masks, *params)
    return masks

  @staticmethod
  def backward(ctx, g):
    masks = ctx.saved_tensors[0]
    params = ctx.saved_tensors[1:]
    nb, training, xl, df, s, count = ctx.misc
    saved = ctx.saved
    O, D = masks.size(0), xl.size(3)
    ni = ctx.needs_input_grad[4:]
    grads = [None] * len(params)
    Wf, bf = params[4 * nb:4 * nb + 2]
    # (Round 5 measured these five weight gradients OFF this chain - it is part of the tail that ends a VG-style step - in
    # two placements: on the weight-gradient lane behind the refinement network's, and on a lane that is idle during the
    # tail.  Both made the step 4-6 % SLOWER: the tail itself got 0.3 ms longer with ten launches fewer on it - every extra
    # fork / join edge there changes how clr maps the graph's branches onto hardware queues, as round 4 saw with an Adam
    # slice on a stream of its own.  profiles/r5_mask_net_wgrad_lane_ab.txt)
    wgrad = _conv_param_grads
    ds = ops.sigmoid_backward(masks, g.contiguous(), _new(g, O, s, s, 1))
    grads[4 * nb], grads[4 * nb + 1] = wgrad(df, ds, 1, (1, 1, 1, D), ni[4 * nb], ni[4 * nb + 1], Wf, bf)
    gz = _new(g, O, s, s, D)
    # (the last block's ReLU mask rides in the 1x1 convolution's data gradient)
    ops.conv2d_backward_data_act(df, _cl_weight(Wf), 1, ds, 1, 0, D, gz, D, saved[nb - 1][2], D, 0.0)
    for b in range(nb - 1, -1, -1):
      x, st, y, sb = saved[b]
      gam, bet, Wp, bias = params[4 * b:4 * b + 4]
      s2 = 2 * sb
      dpre = gz if b == nb - 1 else ops.act_backward(_fptr(gz), D, 0, O, s2, s2, y, D, D, 0.0, gz)
      d = conv_desc([nhwc_src(x, 1, st.scale, st.shift, 1.0)], O, s2, s2, 3, 3, 1, 1)
      grads[4 * b + 2], grads[4 * b + 3] = wgrad(d, dpre, D, (D, 3, 3, D), ni[4 * b + 2], ni[4 * b + 3], Wp, bias)
      gup = _new(g, O, s2, s2, D)
      dgam, dbet, accb, grads[4 * b], grads[4 * b + 1] = _bn_grad_bufs(g, D, gam, bet, ni[4 * b], ni[4 * b + 1])
      bcnt = None if count is None else (count[0], count[1] * sb * sb)
      coef = ops.conv2d_backward_data_bn(d, _cl_weight(Wp), D, dpre, D, 0, D, gup, D, x, D, 1, gam, st, 1.0, training,
                                         dgam, dbet, accb, count=bcnt)
      gz = ops.bn_backward_apply(_fptr(gup), D, 1, O, sb, sb, x, D, D, st, 1.0, coef, _new(g, O, sb, sb, D), count=bcnt)
    ctx.saved = None
    d_obj = gz.view(O, D) if ctx.needs_input_grad[0] else None
    return (d_obj, None, None, None) + tuple(grads)


class SharedPass(object):
  """One forward pass of a discriminator CNN that two autograd graphs use.

  scripts/train.py runs every discriminator over the generated images TWICE per iteration - :544-548 inside the
  generator loss and :566-568 / :581-583 (on ``imgs_pred.detach()``) inside the discriminator's own step - with
  the same weights, the same inputs and (training-mode BatchNorm) the same batch statistics: the discriminator's
  optimiser only steps afterwards.  The Trainer computes the pass once: the first call records the activations
  here and lets every BatchNorm move its running statistics twice (sg2im_bn_fwd.training = 2, bit-identical to a
  second pass); the second call builds its autograd node - on the stream of the discriminator step - around the
  recorded activations without launching anything (it checks network, mode and - when it is handed the input - that
  the input is the recorded tensor).  NOTE for callers outside the Trainer: the second running-statistics update is
  made by the RECORDING pass, so a recorded pass that is never adopted has still moved the running statistics
  twice; every Trainer path adopts it (trainer._seg_d_obj_forward / _seg_d_img)."""

  def __init__(self):
    self.saved = self.misc = self.input = None

  @property
  def recorded(self):
    return self.saved is not None


class DiscCnnFn(Function):
  """build_cnn with 'CK-X-S' tokens, batch norm, valid/same padding (reference
  sg2im/layers.py:129-213): conv, then [BN, LeakyReLU, conv] ... on an NHWC input.
  specs: list of (k, cout, stride, pad); params: [W0, b0], then per later conv
  [gamma, beta, W, b] - or, with normalization='none' (bns is None), just [W, b] per conv:
  the LeakyReLU in front of conv i+1 is then fused into conv i's epilogue.
  share: None or a SharedPass (x may be None once it is recorded)."""

  @staticmethod
  def forward(ctx, x, bns, specs, slope, training, count, share, *params):
    """count: None or (int32 device scalar, 1) - the number of real batch entries (objects) of a
    padded batch: BatchNorm statistics and their backward only see those"""
    if share is not None and share.recorded:
      # the second pass over the same input with the same weights: nothing to compute (SharedPass)
      if (specs, slope, bool(training)) != (share.misc[0], share.misc[1], bool(share.misc[2])):
        raise RuntimeError('SharedPass: recorded by a different network / mode')
      if x is not None and share.input is not None and (x.data_ptr(), tuple(x.shape)) != share.input:
        raise RuntimeError('SharedPass: the second pass was handed another input than the recorded one')
      ctx.saved, ctx.misc = share.saved, share.misc
      ctx.save_for_backward(*params)
      return share.saved[-1][2].detach()
    if share is not None and training:
      training = 2                     # (this pass stands for two: the running statistics move twice)
    x = x.contiguous()
    N, H, W, Cin = x.shape
    saved = []
    src = nhwc_src(x)
    h, w = H, W
    inorm = isinstance(bns, str) and bns == 'instance'    # InstanceNorm2d: no parameters either
    nonorm = bns is None or inorm
    for i, (k, cout, stride, pad) in enumerate(specs):
      if nonorm:
        Wp, bias = params[2 * i:2 * i + 2]
      elif i == 0:
        Wp, bias = params[0:2]
      else:
        Wp, bias = params[2 + 4 * (i - 1) + 2:2 + 4 * (i - 1) + 4]
      d = conv_desc([src], N, h, w, k, k, stride, pad)
      last = i + 1 == len(specs)
      st = ypre = ist = None
      y = _new(x, N, d.out_h, d.out_w, cout)
      if not last and not nonorm:      # conv + the statistics of the BatchNorm behind it, same launches
        st = ops.conv2d_forward_bn(d, _cl_weight(Wp), cout, bias, y, cout, bns[i], training, BN_EPS, BN_MOMENTUM,
                                   count=None if count is None else (count[0], count[1] * d.out_h * d.out_w))
      else:
        ops.conv2d_forward(d, _cl_weight(Wp), cout, bias, y, cout, slope if (nonorm and not inorm and not last) else 1.0)
      if inorm and not last:           # materialised norm + activation; y becomes the activated tensor
        ypre, ist = y, ops.instnorm_stats(y, BN_EPS)
        y = ops.instnorm_act_forward(ypre, ist, slope, _new(x, N, d.out_h, d.out_w, cout))
      saved.append((src, d, y, st, h, w, ypre, ist))
      h, w = d.out_h, d.out_w
      if st is not None:
        src = nhwc_src(y, 0, st.scale, st.shift, slope)
      elif nonorm:
        src = nhwc_src(y)
    # (padded batches under instance normalisation need nothing: the norm is per sample, the dummy crops are
    # independent samples and the counted losses hand them a zero gradient)
    ctx.saved, ctx.misc = saved, (specs, slope, training, tuple(x.shape), nonorm, inorm, count)
    if share is not None:
      share.saved, share.misc, share.input = saved, ctx.misc, (x.data_ptr(), tuple(x.shape))
    ctx.save_for_backward(*params)
    return saved[-1][2]

  @staticmethod
  def backward(ctx, g):
    params = ctx.saved_tensors
    specs, slope, training, xshape, nonorm, inorm, count = ctx.misc
    saved = ctx.saved
    N = xshape[0]
    ni = ctx.needs_input_grad[7:]
    grads = [None] * len(params)
    dy = g.contiguous()
    for i in range(len(specs) - 1, -1, -1):
      k, cout, stride, pad = specs[i]
      src, d, y, st, h, w = saved[i][:6]
      cin = src.channels
      if nonorm:
        wi = 2 * i
        Wp = params[wi]
      elif i == 0:
        Wp, wi = params[0], 0
      else:
        wi = 2 + 4 * (i - 1) + 2
        Wp = params[wi]
      # followed by a batch-statistics BN or an instance norm: the bias is cancelled by the mean
      shadowed = i + 1 < len(specs) and (inorm or (training and not nonorm))
      grads[wi], grads[wi + 1] = _conv_param_grads(d, dy, cout, (cout, k, k, cin), ni[wi], ni[wi + 1] and not shadowed,
                                                   Wp, params[wi + 1])
      if shadowed:
        grads[wi + 1] = _shadowed_bias_grad(params[wi + 1], ni[wi + 1])
      if i == 0:
        dx = None
        if ctx.needs_input_grad[0]:
          dx = _new(g, *xshape)
          ops.conv2d_backward_data(d, _cl_weight(Wp), cout, dy, cout, 0, cin, dx, cin)
        ctx.saved = None
        return (dx, None, None, None, None, None, None) + tuple(grads)
      gz = _new(g, N, h, w, cin)
      ops.conv2d_backward_data(d, _cl_weight(Wp), cout, dy, cout, 0, cin, gz, cin)
      yp, stp = saved[i - 1][2], saved[i - 1][3]
      if nonorm:                                          # yp is the activated output of conv i-1
        dy = ops.act_backward(_fptr(gz), cin, 0, N, h, w, yp, cin, cin, slope, gz)
        if inorm:
          ops.instnorm_backward(dy, saved[i - 1][6], saved[i - 1][7], dy)
        continue
      gi = 2 + 4 * (i - 1)
      dgam, dbet, accb, grads[gi], grads[gi + 1] = _bn_grad_bufs(g, cin, params[gi], params[gi + 1], ni[gi],
                                                                 ni[gi + 1])
      dy = ops.bn_act_backward(_fptr(gz), cin, 0, N, h, w, yp, cin, cin, params[gi], stp, slope, training, gz, dgam,
                               dbet, accb, count=None if count is None else (count[0], count[1] * h * w))


# ---- layer-by-layer pieces for build_cnn architecture strings with R / U / P / FC tokens
# (reference sg2im/layers.py:129-213).  The default 'C' only strings run through DiscCnnFn above.

class Conv2dFn(Function):
  """one nn.Conv2d (square kernel, same stride / padding on both axes) on an NHWC tensor"""

  @staticmethod
  def _desc(x, W, stride, pad):
    N, H, Wd, _ = x.shape
    return conv_desc([nhwc_src(x)], N, H, Wd, W.size(2), W.size(3), stride, pad)

  @staticmethod
  def forward(ctx, x, W, b, stride, pad, shadowed=False):
    x = x.contiguous()
    d = Conv2dFn._desc(x, W, stride, pad)
    cout = W.size(0)
    y = ops.conv2d_forward(d, _cl_weight(W), cout, b, _new(x, x.size(0), d.out_h, d.out_w, cout), cout)
    ctx.save_for_backward(x, W, b)
    ctx.geom = (stride, pad, shadowed)
    return y

  @staticmethod
  def backward(ctx, g):
    x, W, b = ctx.saved_tensors
    stride, pad, shadowed = ctx.geom
    ni = ctx.needs_input_grad
    g = g.contiguous()
    cout, cin, k = W.size(0), W.size(1), W.size(2)
    d = Conv2dFn._desc(x, W, stride, pad)
    dW, db = _conv_param_grads(d, g, cout, (cout, k, k, cin), ni[1], ni[2] and not shadowed, W, b)
    if shadowed:
      db = _shadowed_bias_grad(b, ni[2])
    dx = None
    if ni[0]:
      dx = ops.conv2d_backward_data(d, _cl_weight(W), cout, g, cout, 0, cin, _new(g, *x.shape), cin)
    return dx, dW, db, None, None, None


class InstNormAct(Function):
  """leaky_slope(InstanceNorm2d(y)) on an NHWC tensor"""

  @staticmethod
  def forward(ctx, y, slope):
    y = y.contiguous()
    st = ops.instnorm_stats(y, BN_EPS)
    z = ops.instnorm_act_forward(y, st, slope, _new(y, *y.shape))
    ctx.save_for_backward(y, st, z)
    ctx.slope = slope
    return z

  @staticmethod
  def backward(ctx, g):
    y, st, z = ctx.saved_tensors
    N, H, W, C = y.shape
    g = g.contiguous()
    dyn = ops.act_backward(_fptr(g), C, 0, N, H, W, z, C, C, ctx.slope, _new(g, *y.shape))
    return ops.instnorm_backward(dyn, y, st, dyn), None


class LeakyFn(Function):
  """a LeakyReLU on its own (any dense tensor)"""

  @staticmethod
  def forward(ctx, x, slope):
    x = x.contiguous()
    z = ops.leaky_forward(x, slope, _new(x, *x.shape))
    ctx.save_for_backward(z)
    ctx.slope = slope
    return z

  @staticmethod
  def backward(ctx, g):
    z, = ctx.saved_tensors
    C = z.size(-1)
    g = g.contiguous()
    return ops.act_backward(_fptr(g), C, 0, z.numel() // C, 1, 1, z, C, C, ctx.slope, _new(g, *z.shape)), None


class UpsampleFn(Function):
  """nn.Upsample(scale_factor=f, mode='nearest') on NHWC"""

  @staticmethod
  def forward(ctx, x, f):
    x = x.contiguous()
    N, H, W, C = x.shape
    ctx.f, ctx.shape = f, tuple(x.shape)
    return ops.resample_up(x, f, 1.0, _new(x, N, H * f, W * f, C))

  @staticmethod
  def backward(ctx, g):
    return ops.pool_sum(g.contiguous(), ctx.f, 1.0, _new(g, *ctx.shape)), None


class AvgPoolFn(Function):
  """nn.AvgPool2d(f, f) on NHWC (floor(H/f) x floor(W/f) windows)"""

  @staticmethod
  def forward(ctx, x, f):
    x = x.contiguous()
    N, H, W, C = x.shape
    ctx.f, ctx.shape = f, tuple(x.shape)
    return ops.pool_sum(x, f, 1.0 / (f * f), _new(x, N, H // f, W // f, C))

  @staticmethod
  def backward(ctx, g):
    return ops.resample_up(g.contiguous(), ctx.f, 1.0 / (ctx.f * ctx.f), _new(g, *ctx.shape)), None


class MaxPoolFn(Function):
  """nn.MaxPool2d(f, f) on NHWC"""

  @staticmethod
  def forward(ctx, x, f):
    x = x.contiguous()
    N, H, W, C = x.shape
    ctx.f = f
    ctx.save_for_backward(x)
    return ops.maxpool_forward(x, f, _new(x, N, H // f, W // f, C))

  @staticmethod
  def backward(ctx, g):
    x, = ctx.saved_tensors
    return ops.maxpool_backward(x, g.contiguous(), ctx.f, _new(g, *x.shape)), None


class AddFn(Function):
  """a + b (the residual sum of reference sg2im/layers.py:117)"""

  @staticmethod
  def forward(ctx, a, b):
    a, b = a.contiguous(), b.contiguous()
    return ops.add_forward(a, b, _new(a, *a.shape))

  @staticmethod
  def backward(ctx, g):
    return g, g


class GapFn(Function):
  """GlobalAvgPool over NHWC (reference sg2im/layers.py:83-86)"""

  @staticmethod
  def forward(ctx, x):
    N, H, W, C = x.shape
    ctx.hw = H * W
    ctx.shape = tuple(x.shape)
    return ops.gap_forward(x.contiguous(), _new(x, N, C))

  @staticmethod
  def backward(ctx, g):
    return ops.gap_backward(g.contiguous(), ctx.hw, _new(g, *ctx.shape))


# ----------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------

class _LossFn(Function):
  @staticmethod
  def _finish(ctx, loss, grad):
    ctx.save_for_backward(grad)
    return loss.view(())

  @staticmethod
  def backward(ctx, g):
    grad, = ctx.saved_tensors
    if ops.is_unit(g):                 # d(total)/d(this term) is the Trainer's literal 1.0 (ops.unit)
      return (grad,) + (None,) * 3
    out = ops.scale_by_scalar(grad, g.contiguous().view(1), torch.empty_like(grad))
    return (out,) + (None,) * 3


class SumScalars(Function):
  """total = t0 + t1 + ... (left to right) of up to 8 loss terms in one launch; the upstream
  gradient is handed to every term as is"""

  @staticmethod
  def forward(ctx, *terms):
    ctx.n = len(terms)
    return ops.sum_scalars([t.contiguous() for t in terms], _new(terms[0], 1)).view(())

  @staticmethod
  def backward(ctx, g):
    return (g,) * ctx.n


class L1Loss(_LossFn):
  @staticmethod
  def forward(ctx, pred, target, weight, _unused=None):
    pred, target = pred.contiguous(), target.contiguous()
    grad = torch.empty_like(pred) if ctx.needs_input_grad[0] else None
    loss = ops.l1_loss(pred, target, weight, grad)
    return _LossFn._finish(ctx, loss, grad)


class MseLoss(_LossFn):
  @staticmethod
  def forward(ctx, pred, target, weight, count=None):
    pred, target = pred.contiguous(), target.contiguous()
    grad = torch.empty_like(pred) if ctx.needs_input_grad[0] else None
    loss = ops.mse_loss(pred, target, weight, grad, count)
    return _LossFn._finish(ctx, loss, grad)


class BceLogitsLoss(_LossFn):
  @staticmethod
  def forward(ctx, x, target, weight, count=None):
    x = x.contiguous()
    grad = torch.empty_like(x) if ctx.needs_input_grad[0] else None
    loss = ops.bce_logits_loss(x, target, weight, grad, count)
    return _LossFn._finish(ctx, loss, grad)


class GanScoreLoss(_LossFn):
  """kind 1: WGAN mean term `target * mean(x)`, kind 2: LSGAN mse(sigmoid(x), target)"""
  @staticmethod
  def forward(ctx, x, kind, target, weight, count=None):
    x = x.contiguous()
    grad = torch.empty_like(x) if ctx.needs_input_grad[0] else None
    loss = ops.gan_score_loss(x, kind, target, weight, grad, count)
    return _LossFn._finish(ctx, loss, grad)

  @staticmethod
  def backward(ctx, g):
    return _LossFn.backward(ctx, g) + (None,)


class BceProbLoss(_LossFn):
  """F.binary_cross_entropy on probabilities (mask loss, scripts/train.py:407-410)"""
  @staticmethod
  def forward(ctx, prob, target, weight, count=None):
    prob, target = prob.contiguous(), target.contiguous().float()
    grad = torch.empty_like(prob) if ctx.needs_input_grad[0] else None
    loss = ops.bce_prob_loss(prob, target, weight, grad, count)
    return _LossFn._finish(ctx, loss, grad)


class CrossEntropyLoss(_LossFn):
  @staticmethod
  def forward(ctx, scores, labels, weight, count=None):
    scores = scores.contiguous()
    grad = torch.empty_like(scores) if ctx.needs_input_grad[0] else None
    loss = ops.cross_entropy_loss(scores, labels, weight, grad, count)
    return _LossFn._finish(ctx, loss, grad)
