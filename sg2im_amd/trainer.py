"""One G + D training iteration of the reference loop body (scripts/train.py:524-592) on
MI355X, data parallel over whole images with RCCL.

Differences from the reference that do not change the mathematics:
  * images are converted to NHWC once per step; every kernel works in NHWC;
  * the discriminators are frozen while the generator's loss is back-propagated - the
    reference deposits gradients in the D parameters there (train.py:559) and throws them
    away with the next zero_grad (train.py:577,590);
  * each discriminator's pass over the generated images is computed once per iteration and used by both the
    generator loss (train.py:544-548) and the discriminator's own step (train.py:566-568, 581-583: same images,
    same weights, same batch statistics; the BatchNorm running statistics still move twice) - functional.SharedPass;
  * losses stay on the device: no ``.item()`` per loss (train.py:145,552; utils.py:88);
  * ``num_images`` is passed down, removing the host sync of layout.py:143;
  * under data parallelism the gradient of every network is summed with one all-reduce
    over its flat gradient arena and scaled by 1/world_size inside the Adam kernel.
"""
import collections
import itertools
import math
import os
import time

import torch

from . import _lib
from . import functional as HF
from . import losses as L
from . import ops
from .bucketing import Bucketer, StaticBatch, pad_batch
from .discriminators import AcCropDiscriminator, PatchDiscriminator
from .distributed import GradReducer
from .model import Sg2ImModel
from .optim import FlatAdam, FlatParams

GENERATOR_DEFAULTS = dict(         # reference scripts/train.py:94-104
  image_size=(64, 64), embedding_dim=128, gconv_dim=128, gconv_hidden_dim=512, gconv_num_layers=5,
  mlp_normalization='none', refinement_dims=(1024, 512, 256, 128, 64), normalization='batch',
  activation='leakyrelu-0.2', mask_size=16, layout_noise_dim=32)
D_OBJ_DEFAULTS = dict(arch='C4-64-2,C4-128-2,C4-256-2', normalization='batch', activation='leakyrelu-0.2',
                      padding='valid', object_size=32)           # train.py:117-125
D_IMG_DEFAULTS = dict(arch='C4-64-2,C4-128-2,C4-256-2', normalization='batch', activation='leakyrelu-0.2',
                      padding='valid')                           # train.py:117-130
LOSS_WEIGHTS = dict(l1_pixel_loss_weight=1.0, bbox_pred_loss_weight=10.0, predicate_pred_loss_weight=0.0,
                    mask_loss_weight=0.0, discriminator_loss_weight=0.01, d_obj_weight=1.0, d_img_weight=1.0,
                    ac_loss_weight=0.1)                          # train.py:108-131


SHARE_FAKE_PASS = os.environ.get('SG2IM_SHARE_FAKE_PASS', '1') != '0'      # (A/B knob)
EARLY_ADAM_REST = os.environ.get('SG2IM_EARLY_ADAM_REST', '1') != '0'      # (A/B knob, Trainer._seg_generator_backward)

# 'thread_local': other threads (the RCCL watchdog polls events) may call into HIP while this
# thread captures; the default 'global' mode treats that as a capture error
_CAPTURE_MODE = 'thread_local'


def _set_requires_grad(module, flag):
  for p in module.parameters():
    p.requires_grad_(flag)


def generator_buckets(model, flat_g):
  """Early gradient buckets of the generator's arena for the in-graph exchange (module level so that the slicing can
  be unit-tested without a GPU): a list of (begin, end, ids), in the order in which the deferred weight gradients
  complete them (ops.release_deferred issues the refinement network's weight gradients first module first) -
    [module 0]  (1024-channel convolutions, 43.7 MB at the default architecture),
    [module 1]  (512 channels, 31.3 MB),
    [module 2 ... the output convolutions]  (11.6 MB),
  each a contiguous slice of the arena; ``ids`` = data_ptr() of the convolution WEIGHTS whose weight gradients are
  the last writes into the slice (their biases ride in the same launches, the BatchNorm gradients were written by
  the data-gradient chain the release waits for).  What is left of the arena (graph convolutions, embeddings, heads:
  26 MB) is exchanged after the backward pass - four buckets in all, each sent as soon as it is complete.
  [] when the refinement network has fewer than three modules or carries no BatchNorm (its backward then releases
  nothing early), or when a group is not contiguous in the arena."""
  net = model.refinement_net
  mods = getattr(net, 'refinement_modules', None)
  if mods is None or len(mods) < 3 or net.normalization != 'batch':
    return []
  groups = [[mods[0]], [mods[1]], list(mods[2:]) + [net.output_conv]]
  nb = int(os.environ.get('SG2IM_DP_BUCKETS', '3'))        # (A/B knob: 1 = rounds 3-4's single early bucket, modules 0 + 1)
  if nb == 1:
    groups = [[mods[0], mods[1]]]
  elif nb == 2:
    groups = [[mods[0]], [mods[1]]]
  out = []
  for grp in groups:
    ps = [p for m in grp for p in m.parameters()]
    mine = {id(p) for p in ps}
    offs = [(off, p.numel()) for p, off in zip(flat_g.params, flat_g.offsets) if id(p) in mine]
    if len(offs) != len(mine):
      return []
    a = min(o for o, _ in offs)
    b = max(o + (n + 3) // 4 * 4 for o, n in offs)
    inside = sum(1 for p, off in zip(flat_g.params, flat_g.offsets) if a <= off < b)
    if inside != len(mine):           # (not contiguous: keep one bucket)
      return []
    ids = frozenset(p.data_ptr() for p in ps if p.dim() == 4)
    out.append((a, min(b, flat_g.numel), ids))
  out.sort(key=lambda t: t[0])
  if any(out[i][1] > out[i + 1][0] for i in range(len(out) - 1)):
    return []
  return out


def generator_bucket(model, flat_g):
  """the first two refinement modules as ONE slice (rounds 3-4's early bucket; kept for tests / tools)"""
  bk = generator_buckets(model, flat_g)
  if len(bk) < 2 or bk[0][1] != bk[1][0]:
    return None
  return bk[0][0], bk[1][1], bk[0][2] | bk[1][2]


def complement(slices, n):
  """the gaps [(begin, end)] that sorted-or-not, non-overlapping ``slices`` leave in [0, n)"""
  gaps, at = [], 0
  for a, b in sorted(slices):
    if a > at:
      gaps.append((at, a))
    at = max(at, b)
  if at < n:
    gaps.append((at, n))
  return gaps


def refinement_slice(model, flat_g):
  """(begin, end) of the refinement network's parameters in the generator's arena, None when they are not one
  contiguous run"""
  ps = {id(p) for p in model.refinement_net.parameters()}
  offs = [(off, p.numel()) for p, off in zip(flat_g.params, flat_g.offsets) if id(p) in ps]
  if not offs:
    return None
  a = min(o for o, _ in offs)
  b = max(o + (n + 3) // 4 * 4 for o, n in offs)
  inside = sum(1 for p, off in zip(flat_g.params, flat_g.offsets) if a <= off < b)
  return (a, min(b, flat_g.numel)) if inside == len(ps) else None


class Trainer(object):
  def __init__(self, vocab, device, generator_kwargs=None, d_obj_kwargs=None, d_img_kwargs=None,
               loss_weights=None, learning_rate=1e-4, world_size=1, seed=None, use_graphs=False,
               gan_loss_type='gan', overlap_d=None, bucket='auto', max_graphs=32, rank=0, align_corners=False,
               compute_dtype='f32', dp_schedule=None, verify_replicas=True):
    """use_graphs: replay one captured hipGraph per batch-shape BUCKET instead of launching ~480
    kernels from Python.  bucket = (object multiple, triple multiple): the object / triple axes of
    every batch are padded to those multiples with exactly neutral rows (sg2im_amd/bucketing.py);
    'auto' = (32, 64) in graph mode and no padding in eager mode; an explicit bucket also pads the
    eager launches, which then run the same kernels on the same shapes as the graph (bit-identical
    results).  Batches of any (O, T) are accepted either way."""
    self.gan_g_loss, self.gan_d_loss = L.get_gan_losses(gan_loss_type)     # train.py:467
    # compute_dtype 'bf16' (BASELINE.json configs[2..4]): the spatial convolutions of the refinement
    # network, the discriminators and mask_net multiply bf16-rounded operands on the bf16 matrix cores
    # with fp32 accumulation (sg2im_conv_desc.compute_dtype); every tensor in memory - activations,
    # gradients, master weights, BatchNorm statistics, Adam state - stays fp32
    if compute_dtype not in ('f32', 'bf16'):
      raise ValueError('compute_dtype must be "f32" or "bf16"')
    self.compute_dtype = compute_dtype
    # bf16 mode: the generator's 3x3 convolutions read their weights from a bfloat16 mirror of the parameter arena,
    # refreshed by ONE launch at the start of every iteration (FlatParams.refresh_mirror; SG2IM_WEIGHT_MIRROR=0: A/B knob)
    self.weight_mirror = compute_dtype == 'bf16' and os.environ.get('SG2IM_WEIGHT_MIRROR', '1') != '0'
    self.device = device
    self.world_size = world_size
    self.rank = rank
    if seed is not None:
      torch.manual_seed(seed)            # identical initial weights on every rank
    gk = dict(GENERATOR_DEFAULTS)
    gk.update(generator_kwargs or {})
    dok = dict(D_OBJ_DEFAULTS)
    dok.update(d_obj_kwargs or {})
    dik = dict(D_IMG_DEFAULTS)
    dik.update(d_img_kwargs or {})
    if align_corners:            # torch-0.4 sampling convention for layout + crops (see Sg2ImModel)
      gk['align_corners'] = True
      dok['align_corners'] = True
    self.model_kwargs = dict(gk, vocab=vocab)
    self.d_obj_kwargs = dict(dok, vocab=vocab)
    self.d_img_kwargs = dik
    self.w = dict(LOSS_WEIGHTS)
    self.w.update(loss_weights or {})
    self.model = Sg2ImModel(vocab, **gk).to(device)
    # build_obj_discriminator / build_img_discriminator (train.py:194-228): a zero weight means
    # the discriminator, its optimiser and its loss terms do not exist
    dw = self.w['discriminator_loss_weight']
    self.d_obj = AcCropDiscriminator(vocab, **dok).to(device) if dw != 0 and self.w['d_obj_weight'] != 0 else None
    self.d_img = PatchDiscriminator(**dik).to(device) if dw != 0 and self.w['d_img_weight'] != 0 else None
    for m in (self.model, self.d_obj, self.d_img):
      if m is not None:
        m.train()
    self.flat_g = FlatParams(self.model)
    self.flat_do = FlatParams(self.d_obj) if self.d_obj is not None else None
    self.flat_di = FlatParams(self.d_img) if self.d_img is not None else None
    self.opt_g = FlatAdam(self.flat_g, lr=learning_rate)
    self.opt_do = FlatAdam(self.flat_do, lr=learning_rate) if self.d_obj is not None else None
    self.opt_di = FlatAdam(self.flat_di, lr=learning_rate) if self.d_img is not None else None
    # (bf16 mode: the gradient arenas travel as bfloat16, sg2im_amd/distributed.py; SG2IM_GRAD_PAYLOAD overrides)
    self.reducer = GradReducer(world_size, payload=os.environ.get('SG2IM_GRAD_PAYLOAD', 'bf16' if compute_dtype == 'bf16' else 'f32'),
                               exchange=os.environ.get('SG2IM_DP_EXCHANGE', 'allreduce'))
    if world_size > 1:
      # replicas must start from identical weights / buffers whatever the seeds were, and draw
      # different layout noise (model.py:164-168) per rank
      self.broadcast_state()
      if seed is not None:
        torch.cuda.manual_seed(seed * 1000003 + 7919 * (rank + 1))
    self.use_graphs = use_graphs
    # data-parallel graph schedule (see _capture): 2 = ONE graph with the all-reduces recorded inside it, bucketed and
    # overlapped; 0 = one iteration graph, exchange, Adam graph; 1 = segmented, the D_obj step replayed while the
    # generator's all-reduce is in flight
    # Default (round 6): 2 - decided by a probe, not by caution.  Whether THIS stack can capture the schedule-2 stream
    # pattern with RCCL inside is asked in a subprocess (sg2im_amd/capture_probe.py: the one failure seen so far was a
    # segfault at the end of a capture, not an error); every rank probes its own device, the verdicts are AND-ed over
    # the ranks, and a failure selects schedule 1 (graph segments, the exchanges between them: the D_obj step still
    # hides the generator's all-reduce).  What the probe cannot see - replicas that drift apart - is what
    # check_replicas watches for (steps 1, 2, 4, ... then every 1024th), falling back to schedule 0.  Projected 8-GPU
    # efficiencies of the three schedules: DESIGN.md section 6.  SG2IM_DP_SCHEDULE / dp_schedule= pin a schedule.
    if dp_schedule is None:
      dp_schedule = os.environ.get('SG2IM_DP_SCHEDULE', '2')
    self.dp_schedule = int(dp_schedule)
    if world_size > 1 and use_graphs and self.dp_schedule == 2:
      from .capture_probe import choose_dp_schedule, probe
      idx = device.index if device.index is not None else torch.cuda.current_device()

      def agree(ok):
        import torch.distributed as dist
        t = torch.tensor([1.0 if ok else 0.0], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)
      # (the children of all ranks form their own RCCL group: a multi-rank collective inside a graph is tried THERE first)
      chosen = choose_dp_schedule(2, self.reducer.capturable(), lambda: probe(idx, rank=rank, world_size=world_size),
                                  agree if self.reducer.capturable() else None)
      if chosen != 2 and rank == 0:
        print('[sg2im_amd] dp_schedule 2 (all-reduces inside the iteration graph) is not available here: running schedule %d' % chosen,
              flush=True)
      self.dp_schedule = chosen
    # Under dp_schedule 2 the replicas' parameter arenas are compared every now and then (replicas_in_sync: two tiny
    # all-reduces of checksums + a host sync; steps 1, 2, 4, 8 ... 1024, then every 1024th - step counts are the
    # same on every rank, which a "first replay of a new graph" trigger would not be).  If they ever differ - the
    # in-graph exchange has not run on more than one rank on the hardware this was developed on - the Trainer warns,
    # falls back to schedule 0 (iteration graph -> exposed all-reduces -> Adam graph), re-broadcasts rank 0's state
    # and re-captures.  verify_replicas=False: the caller does it (bench.py, around its timed loop).
    self.verify_replicas = bool(verify_replicas)
    self.replica_checks = {'checked': 0, 'diverged': 0}
    self._comm = None
    self._aux2 = None
    if bucket == 'auto':
      bucket = (32, 64) if use_graphs else None
    self.bucketer = Bucketer(*bucket) if bucket else None
    self.max_graphs = max_graphs
    self.graph_stats = {'captures': 0, 'replays': 0, 'invalidated': 0, 'evicted': 0}
    self.launch_stats = {}          # kernels per captured iteration (set by the last capture)
    self._cap_stream = None
    self._wgrad_stream = None
    self._lane_handles = set()
    # single-GPU graph mode: capture the whole iteration as ONE graph in which the two
    # discriminator steps run on a side stream concurrently with the generator's backward
    # (they only need imgs_pred; their small kernels fill the CUs the GCN / MLP backward leaves idle)
    self.overlap_d = True if overlap_d is None else bool(overlap_d)
    self.host_seconds = {'stage_batch': 0.0, 'graph_launch': 0.0}     # host time spent issuing replays (bench.py)
    # the last replays' individual hipGraphLaunch durations: their MEDIAN is what a launch costs the host; the mean
    # of a long run also contains the back-pressure stalls of a host that runs several iterations ahead
    self.host_launch_samples = collections.deque(maxlen=512)
    self._side = None
    self._graphs = collections.OrderedDict()
    self.t = 0

  # -- data-parallel helpers --------------------------------------------------
  def broadcast_state(self, src=0):
    """rank ``src``'s parameters, optimiser moments and BatchNorm buffers to every rank (after
    construction and after a checkpoint restore)"""
    import torch.distributed as dist
    from .distributed import broadcast
    if self.world_size <= 1 or not dist.is_initialized():
      return
    for flat, opt in ((self.flat_g, self.opt_g), (self.flat_do, self.opt_do), (self.flat_di, self.opt_di)):
      if flat is not None:
        for t in (flat.flat, opt.exp_avg, opt.exp_avg_sq, opt.state):
          broadcast(t, src)
    for m in (self.model, self.d_obj, self.d_img):
      if m is not None:
        for b in m.buffers():
          broadcast(b, src)

  def replicas_in_sync(self):
    """every rank applied the same (all-reduced) gradients <=> the parameter arenas are still bit-identical.
    Collective: every rank must call it at the same step.  Synchronises with the host."""
    import torch.distributed as dist
    if self.world_size <= 1 or not dist.is_initialized():
      return True
    from .distributed import _host_staged
    c = torch.stack([f.flat.double().sum() for f in (self.flat_g, self.flat_do, self.flat_di) if f is not None])
    if _host_staged(c):
      c = c.cpu()
    hi, lo = c.clone(), c.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    return bool(torch.equal(hi, lo))

  def check_replicas(self):
    """compare the replicas now; on divergence under the in-graph exchange fall back to schedule 0 (see __init__).
    Returns True when they were in sync."""
    self.replica_checks['checked'] += 1
    if self.replicas_in_sync():
      return True
    self.replica_checks['diverged'] += 1
    if self.rank == 0:
      print('WARNING: data-parallel replicas diverged at step %d under dp_schedule %d%s' % (
        self.t, self.dp_schedule, '; falling back to dp_schedule 0 and re-broadcasting rank 0' if self.dp_schedule == 2 else ''),
        flush=True)
    if self.use_graphs and self.dp_schedule == 2:
      self.dp_schedule = 0
      self._graphs.clear()
    self.broadcast_state()
    return False

  def _maybe_check_replicas(self):
    if not self.verify_replicas or self.world_size <= 1 or not self.use_graphs or self.dp_schedule != 2:
      return
    t = self.t
    if (t <= 1024 and t & (t - 1) == 0) or t % 1024 == 0:
      self.check_replicas()

  def set_generator_eval(self):
    """reference scripts/train.py:509-512: eval-mode BN for G and a fresh Adam"""
    self.model.eval()
    self.opt_g.reset_state()
    self._graphs.clear()           # captured segments baked in the training-mode kernels

  # -- one iteration ----------------------------------------------------------
  # The iteration is four segments separated by the points where a data-parallel exchange
  # is started: generator fwd+bwd | D_obj fwd+bwd | D_img fwd+bwd | the three Adam updates.
  def _seg_generator(self, batch, st):
    self._seg_generator_forward(batch, st)
    self._seg_generator_backward(st)

  def _seg_generator_forward(self, batch, st):
    self._seg_generator_model(batch, st)
    self._seg_generator_losses(batch, st)

  def _seg_generator_model(self, batch, st):
    """train.py:524-530: the generator itself; `imgs_fake` is all the discriminator steps need"""
    imgs, objs, boxes, masks, triples, obj_to_img = batch[:6]
    ops.mark('start')
    st['imgs_nhwc'] = HF.NchwToNhwc.apply(imgs)
    # captured iteration: whatever is not on the path to the image leaves the critical path (Sg2ImModel.forward_nhwc)
    aux = self._side[0] if (self._side is not None and torch.cuda.is_current_stream_capturing() and
                            os.environ.get('SG2IM_AUX', '1') != '0' and not ops.SINGLE_STREAM) else None      # (A/B knob)
    # (the bf16 weight mirror: 28 us over the 120 MB arena, next to the graph-convolution phase on the aux stream -
    # forward_nhwc issues it there behind the head of the step and joins it in front of the layout, long before the
    # first convolution that reads the mirror; without an aux stream: right away)
    late = os.environ.get('SG2IM_MIRROR_LATE', '1') != '0'      # (A/B knob: 0 = refreshed first thing, as before)
    if self.weight_mirror and not late:
      if aux is not None:
        aux.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(aux):
          self.flat_g.refresh_mirror()
      else:
        self.flat_g.refresh_mirror()
    w = self.w
    st['gen_out'] = self.model.forward_nhwc(objs, triples, obj_to_img, boxes_gt=boxes, masks_gt=masks,
                                            num_images=imgs.size(0), obj_count=st.get('ocnt'),
                                            triple_count=st.get('tcnt'), aux_stream=aux,
                                            aux_work=self.flat_g.refresh_mirror if self.weight_mirror and late else None,
                                            detach_masks=masks is not None and not w['mask_loss_weight'] > 0,
                                            detach_rel=not w['predicate_pred_loss_weight'] > 0)
    st['imgs_fake'] = st['gen_out'][0].detach()
    ops.mark('g_fwd_done')

  def _seg_generator_losses(self, batch, st):
    imgs, objs, boxes, masks, triples, obj_to_img = batch[:6]
    w = self.w
    imgs_pred, boxes_pred, masks_pred, rel_scores = st.pop('gen_out')
    if torch.cuda.is_current_stream_capturing():
      ev = torch.cuda.Event()
      ev.record()                      # (imgs_pred exists: the side stream may start on it, see below)
      st['imgs_pred_ready'] = ev
    # padded batch (graph mode): the means over objects / triples run over the real rows only
    oc, tc = st.get('ocnt'), st.get('tcnt')
    cnt = lambda c, unit: None if c is None else (c[0], c[1] * unit)
    # train.py:531-560.  The discriminators are frozen here (see module docstring).
    for d in (self.d_obj, self.d_img):
      if d is not None:
        _set_requires_grad(d, False)
    losses = {}          # the generator's own terms (st['losses'] may already hold a D step's)
    losses['L1_pixel_loss'] = L.l1_loss(imgs_pred, st['imgs_nhwc'], w['l1_pixel_loss_weight'])
    losses['bbox_pred'] = L.mse_loss(boxes_pred, boxes, w['bbox_pred_loss_weight'], cnt(oc, 4))
    if w['predicate_pred_loss_weight'] > 0:                       # train.py:402-405
      losses['predicate_pred'] = L.cross_entropy(rel_scores, triples[:, 1].contiguous(),
                                                 w['predicate_pred_loss_weight'], cnt(tc, 1))
    if w['mask_loss_weight'] > 0 and masks is not None and masks_pred is not None:   # train.py:407-410
      losses['mask_loss'] = L.binary_cross_entropy(masks_pred, masks, w['mask_loss_weight'],
                                                   cnt(oc, masks_pred.size(1) * masks_pred.size(2)))
    gi = None
    # the discriminators' passes over the generated images are needed twice (here and in their own steps, with
    # unchanged weights): computed once, functional.SharedPass
    share = st['shared_fake'] = {'obj': HF.SharedPass(), 'img': HF.SharedPass()} if SHARE_FAKE_PASS else {}
    par = (self.d_img is not None and self.d_obj is not None and self._aux2 is not None and
           torch.cuda.is_current_stream_capturing() and os.environ.get('SG2IM_PAR_DIMG', '1') != '0' and
           not ops.SINGLE_STREAM)      # (A/B knob)
    ev = st.pop('imgs_pred_ready', None)
    if par:
      # captured iteration: the image discriminator's pass over the fake images runs on a stream of its own NEXT
      # TO the object discriminator's (both only need imgs_pred) - and so does its backward, which autograd
      # executes on the stream of the forward: two dependent small-kernel chains between the refinement
      # network's forward and backward instead of one twice as long.  (Issued first: a replay issues the
      # nodes in capture order.  Not the discriminator steps' side stream: the backward of this branch would
      # queue up behind them.)
      main, side = torch.cuda.current_stream(), self._aux2
      side.wait_event(ev)
      with torch.cuda.stream(side):
        gi = self.gan_g_loss(self.d_img.forward_nhwc(imgs_pred, share.get('img')),
                             weight=w['discriminator_loss_weight'] * w['d_img_weight'])
    if self.d_obj is not None:
      # (loss weights are folded into the loss kernels; the terms are summed by ONE launch and the
      # backward pass is seeded with ops.unit, so no term pays a multiply / scaling launch)
      scores_fake, ac_loss = self.d_obj.forward_nhwc(imgs_pred, objs, boxes, obj_to_img, w['ac_loss_weight'], oc,
                                                     share.get('obj'))
      losses['ac_loss'] = ac_loss
      losses['g_gan_obj_loss'] = self.gan_g_loss(scores_fake, weight=w['discriminator_loss_weight'] * w['d_obj_weight'],
                                                 count=oc)
    if self.d_img is not None:
      if par:
        main.wait_stream(side)
      else:
        gi = self.gan_g_loss(self.d_img.forward_nhwc(imgs_pred, share.get('img')),
                             weight=w['discriminator_loss_weight'] * w['d_img_weight'])
      losses['g_gan_img_loss'] = gi
    total = HF.SumScalars.apply(*losses.values())
    losses['total_loss'] = total
    st['losses'].update(losses)
    st['total'] = total
    # NaN guard of train.py:553-555 without a host sync: every optimiser of this iteration
    # skips its update when the generator loss is not finite (on any rank).
    st['guard'] = total.detach().reshape(1).clone()
    ops.mark('g_losses_done')
    for d in (self.d_obj, self.d_img):
      if d is not None:
        _set_requires_grad(d, True)

  def _seg_generator_backward(self, st):
    self.opt_g.zero_grad()
    ops.mark('g_bwd_start')
    defer = ops.DEFER_WGRAD and torch.cuda.is_current_stream_capturing() and not ops.SINGLE_STREAM
    ops.DEFERRED = [] if defer else None
    try:
      st.pop('total').backward(ops.unit(self.device))
    finally:
      lanes, ops.DEFERRED = ops.DEFERRED, None
    ops.mark('g_bwd_done')
    early = st.get('g_adam_early')
    if early is not None and st.get('g_adam_prepared') and EARLY_ADAM_REST:
      # every gradient outside the refinement network's slice is complete HERE (the slice itself is updated at the end
      # of the weight-gradient lane): update the rest now, under the lane's remaining weight gradients, instead of
      # behind the join (two of the four Adam launches that used to end the iteration)
      gs = self.reducer.grad_scale
      self.opt_g.apply_guarded(0, early[0], gs)
      self.opt_g.apply_guarded(early[1], self.flat_g.numel, gs)
      st['g_adam_done'] = True
    for lane in lanes or ():
      lane.join()

  def _seg_d_obj(self, batch, st):
    ops.mark('d_obj_start')
    self._seg_d_obj_forward(batch, st)
    self._seg_d_obj_backward(batch, st)
    ops.mark('d_obj_done')

  def _seg_d_obj_forward(self, batch, st):
    imgs, objs, boxes, masks, triples, obj_to_img = batch[:6]
    losses = st['losses']
    # train.py:566-579
    oc = st.get('ocnt')
    sf, ac_fake = self.d_obj.forward_nhwc(st['imgs_fake'], objs, boxes, obj_to_img, obj_count=oc,
                                          share=st.get('shared_fake', {}).get('obj'))
    sr, ac_real = self.d_obj.forward_nhwc(st['imgs_nhwc'], objs, boxes, obj_to_img, obj_count=oc)
    gan_terms = self.gan_d_loss.terms(sr, sf, oc)
    losses['d_obj_gan_loss'] = HF.SumScalars.apply(*gan_terms).detach()      # (reported value only)
    losses['d_ac_loss_real'], losses['d_ac_loss_fake'] = ac_real, ac_fake
    st['d_obj_total'] = HF.SumScalars.apply(*gan_terms, ac_real, ac_fake)

  def _seg_d_obj_backward(self, batch, st):
    self.opt_do.zero_grad()
    st.pop('d_obj_total').backward(ops.unit(self.device))

  def _seg_d_img(self, batch, st):
    losses = st['losses']
    ops.mark('d_img_start')
    # train.py:581-592
    sf = self.d_img.forward_nhwc(st['imgs_fake'], st.get('shared_fake', {}).get('img'))
    sr = self.d_img.forward_nhwc(st['imgs_nhwc'])
    losses['d_img_gan_loss'] = self.gan_d_loss(sr, sf)
    self.opt_di.zero_grad()
    losses['d_img_gan_loss'].backward(ops.unit(self.device))
    ops.mark('d_img_done')

  def _seg_adam(self, st):
    # all three updates at the end: same values as the reference's in-order updates because
    # no forward/backward above reads another network's *updated* parameters
    gs, guard = self.reducer.grad_scale, st['guard']
    early = st.get('g_adam_early')
    if st.get('g_adam_done'):        # (captured iteration: the whole arena was updated inside the backward segment)
      pass
    elif early is not None:          # (captured iteration: [a, b) was updated under the weight gradients already)
      self.opt_g.apply_guarded(0, early[0], gs)
      self.opt_g.apply_guarded(early[1], self.flat_g.numel, gs)
    else:
      if st.get('g_adam_prepared'):
        self.opt_g.apply_guarded(0, self.flat_g.numel, gs)
      else:
        self.opt_g.step_guarded(guard, gs)
    if self.opt_do is not None:
      self.opt_do.step_guarded(guard, gs)
    if self.opt_di is not None:
      self.opt_di.step_guarded(guard, gs)
    st['out'] = {k: v.detach() for k, v in st['losses'].items()}
    ops.mark('adam_done')

  def _run_segments(self, batch, st, run):
    """run(name, fn) executes (or replays) one segment; exchanges are started in between.
    The 112 MB generator all-reduce is only waited for after both discriminator passes
    (they never read G's parameters), see sg2im_amd/distributed.py."""
    red = self.reducer
    if self.use_graphs and self.overlap_d and self.d_img is not None:
      # graph segments of the data-parallel step: the D_img step runs on the side stream INSIDE
      # the generator segment (next to the generator backward), the D_obj step is its own segment
      # and executes while the generator's all-reduce is in flight
      run('g+di', lambda: self._seg_generator_with_d_img(batch, st))
      red.start(self.flat_g.grad)
      red.start(st['guard'])
      red.start(self.flat_di.grad)
    else:
      run('g', lambda: self._seg_generator(batch, st))
      red.start(self.flat_g.grad)
      red.start(st['guard'])
      if self.d_img is not None:
        run('di', lambda: self._seg_d_img(batch, st))
        red.start(self.flat_di.grad)
    if self.d_obj is not None:
      run('do', lambda: self._seg_d_obj(batch, st))
      red.start(self.flat_do.grad)
    red.finish()
    run('adam', lambda: self._seg_adam(st))

  def _seg_generator_with_d_img(self, batch, st):
    """generator forward, then fork: generator backward on the current stream, the D_img step on
    the side stream (own workspace lane), joined at the end"""
    if self._side is None:
      self._n_side = 1
      self._side = (torch.cuda.Stream(),)
    side, main = self._side[0], torch.cuda.current_stream()
    self._seg_generator_forward(batch, st)
    side.wait_stream(main)
    with torch.cuda.stream(side):
      self._seg_d_img(batch, st)
    self._seg_generator_backward(st)
    main.wait_stream(side)

  def step(self, batch):
    """batch: (imgs (N,3,H,W), objs, boxes, masks | None, triples, obj_to_img) on the device.
    Returns a dict of 0-dim device tensors (no host sync)."""
    self.t += 1
    keep, ops.CONV_COMPUTE = ops.CONV_COMPUTE, (1 if self.compute_dtype == 'bf16' else 0)
    keep_mirror, ops.WEIGHT_MIRROR = ops.WEIGHT_MIRROR, self.weight_mirror
    keep_bwd = ops.GCN_PERSISTENT_BACKWARD
    if ops.GCN_PERSISTENT_BACKWARD_MODE == 'auto':
      ops.GCN_PERSISTENT_BACKWARD = self._gcn_backward_mode(batch)
    try:
      out = self._step(batch)
    finally:
      ops.CONV_COMPUTE = keep
      ops.WEIGHT_MIRROR = keep_mirror
      ops.GCN_PERSISTENT_BACKWARD = keep_bwd
    self._maybe_check_replicas()
    return out

  def _gcn_backward_mode(self, batch):
    """The GraphTripleConv stack's backward as ONE co-resident persistent launch ('low') or layer by layer (False) -
    whichever lane ends the step decides (both forms are parity-tested; measured in round 5).  When
    mask_net is trained through the layout (no ground-truth masks in the batch: the VG style of BASELINE configs[2..4];
    or a mask loss) its backward lengthens the main lane's small-kernel tail, the tail ends the step, and the 74 fewer
    dependent launches pay: -2.8 % fp32 / -3.8 % bf16 at VG-64.  In the COCO-style configuration the refinement network's
    weight-gradient lane ends the step and resident workgroups polling grid barriers cost it 0.13 ms: layer by layer."""
    masks = batch[3] if len(batch) > 3 else None
    trains_mask_net = self.model.mask_net is not None and (masks is None or self.w['mask_loss_weight'] > 0)
    # (round 5 also took the one-launch form for every bf16 step; since the GEMM epilogues got cheap - round 6 - the
    # layer-by-layer launches win in the COCO style in both modes: bf16 3.82-3.83 -> 3.69-3.75 ms; VG style, one launch:
    # fp32 8.34 vs 8.57, bf16 4.39 vs 4.56 ms)
    if trains_mask_net:
      # (the persistent kernel's 32 x 32 tiles are built for a few hundred rows; at the 256 x 256 shape - 700-960 objects,
      # <= 3 200 triples - its low-footprint form takes 2.1 ms and the implicit-GEMM family's 64 x 64+ tiles win:
      # fp32 53.0 -> 52.0 ms, bf16 20.3 -> 19.7 ms, profiles/r6_gcn_backward_staged_ab.txt)
      triples = batch[4] if len(batch) > 4 else None
      return 'low' if triples is None or not hasattr(triples, 'size') or triples.size(0) <= 1024 else False
    # (round 6, COCO style under the bf16 mode: the small-kernel tail ends that step too, by ~0.13 ms - the STAGED form, the
    # persistent kernel's 25 stages as 25 ordinary launches with the weight gradients riding as extra tiles, shortens the tail
    # without resident workgroups that poll: 3.73 -> 3.68 ms; fp32: the weight-gradient lane ends the step, 7.41 vs 7.45)
    return 'staged' if self.compute_dtype == 'bf16' else False

  def _step(self, batch):
    if self.use_graphs:
      if self.bucketer is None:
        self.bucketer = Bucketer(32, 64)
      return self._graph_step(batch)
    st = {'losses': {}}
    if self.bucketer is not None:       # eager launches on the padded batch (what a graph would replay)
      o_pad, t_pad = self.bucketer.bucket(batch[1].numel(), batch[4].size(0))
      batch, counts = pad_batch(batch, o_pad, t_pad)
      st['ocnt'], st['tcnt'] = (counts[0:1], 1), (counts[1:2], 1)
    self._run_segments(batch, st, lambda name, fn: fn())
    return st['out']

  # -- hipGraph replay, one graph per batch-shape bucket ---------------------------
  def _prepare_lanes(self, scratch_floats, workspace_bytes=0):
    """Streams and work buffers of every execution lane a capture uses, created EAGERLY (outside
    any capture, so that no captured graph's private memory pool ends up owning them): the capture
    stream, the side stream of the discriminator steps and the side stream the refinement network's
    weight gradients run on (ops.SideLane)."""
    dev = self.device
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    # torch.cuda.Stream() hands out a pool of 32 streams per device round-robin: in a process that builds many
    # Trainers a new stream object may BE a stream this Trainer already holds (or the weight-gradient stream an
    # earlier Trainer cached under the same capture-stream handle) - two lanes of the captured iteration would
    # then be one stream.  Every lane of this Trainer gets a handle none of its other lanes has.
    taken = self._lane_handles

    def fresh():
      for _ in range(64):
        s = torch.cuda.Stream(device=idx)
        if s.cuda_stream not in taken:
          taken.add(s.cuda_stream)
          return s
      raise RuntimeError('no unused stream left for a lane of the captured iteration')
    if self._cap_stream is None:
      _lib.init()
      self._cap_stream = fresh()
      self._n_side = 1               # (a second side stream for D_img measured slower: 11.1 vs 10.65 ms)
      self._side = (fresh(),)
      self._wgrad_stream = fresh()
    if self._comm is None and (self.world_size > 1 or self.reducer.force):
      self._comm = fresh()           # gradient exchange inside the captured iteration
    if self._aux2 is None:
      self._aux2 = fresh()           # the generator loss' pass through the image discriminator
    # (ops.SideLane looks the weight-gradient stream up by the capture stream's handle: this Trainer's, whatever
    # another Trainer with the same pooled handle left there)
    ops._wgrad_streams[(idx, self._cap_stream.cuda_stream)] = self._wgrad_stream
    lanes = [self._cap_stream, self._side[0], self._aux2, self._wgrad_stream] + ([self._comm] if self._comm is not None else [])
    assert len(set(s.cuda_stream for s in lanes)) == len(lanes), 'two lanes of the captured iteration share a stream'
    for s in (self._cap_stream, self._side[0], self._aux2, self._wgrad_stream):
      with torch.cuda.stream(s):
        ops.workspace(dev, workspace_bytes)
        ops.scratch(dev, scratch_floats)
        ops.sync_area(dev)           # (grid-barrier state of the persistent GraphTripleConv kernels)
    if self.reducer.payload == 'bf16' and (self.world_size > 1 or self.reducer.force):
      # the bfloat16 staging buffers of the gradient exchange: born here, not inside a capture (a buffer from one
      # graph's private pool must not be used by the graph of another shape bucket)
      for flat in (self.flat_g, self.flat_do, self.flat_di):
        if flat is not None:
          self.reducer.staging(flat.grad)
      # (every slice that ANY subset of the early buckets can leave for the final exchange: a bucket whose weight
      # gradients are not all among the released launches is never reported complete - ops.release_deferred - and its
      # range then travels with the remainder)
      early = [(a, b) for a, b, _ in self._generator_buckets()]
      seen = set()
      for k in range(len(early) + 1):
        for sent in itertools.combinations(early, k):
          for a, b in list(sent) + complement(list(sent), self.flat_g.numel):
            if (a, b) not in seen:
              seen.add((a, b))
              self.reducer.staging(self.flat_g.grad[a:b])
    if self.weight_mirror and self.flat_g.mirror is None:
      self.flat_g.refresh_mirror()   # (allocates the mirror outside the capture)
    ops.unit(dev)                    # (cached process-wide: must not be born inside a capture)
    ops.marks_init(dev)

  def _graph_step(self, batch):
    """The eager step costs ~12 ms of Python/ctypes launch time for ~480 kernels - more than the
    kernels themselves.  In graph mode the object / triple axes of the batch are padded to a bucket
    size (sg2im_amd/bucketing.py: exactly neutral padding, true sizes in device memory), the whole
    iteration of a bucket is captured ONCE as a hipGraph (torch.cuda.CUDAGraph records the launches
    our C ABI makes on the capture stream) and every later batch of that bucket is copied into the
    graph's static input buffers and replayed.  A new bucket is captured directly - no eager warm-up
    steps: sg2im_init() and _prepare_lanes() did everything a first launch would do lazily.
    Data parallel: the RCCL collectives are recorded inside the graph (dp_schedule 2, the default) or issued
    between an iteration graph and an Adam graph (schedules 0 / 1), see _capture.

    Safety net: round 1 reported that an EAGER launch from this library after a graph was instantiated
    made the next replay of that graph fault; 26 probe variants in round 2 could not reproduce it
    (profiles/r2_graph_fault_probes.log), so it is NOT a known property of the runtime.  The check stays
    because it is free on the training path: eager launches through the binding bump ``_lib.EAGER_EPOCH``
    (captured ones do not); a graph whose epoch is stale is dropped and re-captured at its next use, so a
    validation pass between training steps costs one re-capture per bucket and nothing otherwise."""
    imgs, objs, masks, triples = batch[0], batch[1], batch[3], batch[4]
    o_pad, t_pad = self.bucketer.bucket(objs.numel(), triples.size(0))
    key = (o_pad, t_pad, tuple(imgs.shape), None if masks is None else (masks.dtype,) + tuple(masks.shape[1:]))
    if self._graphs:
      stale = [k for k, e in self._graphs.items() if e[3] != _lib.EAGER_EPOCH]
      for k in stale:                # the library ran eagerly since these were instantiated
        del self._graphs[k]
      self.graph_stats['invalidated'] += len(stale)
    ent = self._graphs.get(key)
    if ent is None:
      sb = StaticBatch(batch, o_pad, t_pad)
      try:
        ent = self._capture(sb)
      except Exception as e:    # capture unsupported here: stay eager, loudly
        print('WARNING: hipGraph capture failed (%s: %s); falling back to eager launches' % (type(e).__name__, e))
        _lib.CAPTURING = False
        self.use_graphs = False
        self.bucketer = None          # (later steps run eagerly on the unpadded batch)
        torch.cuda.synchronize()
        st = {'losses': {}}
        self._run_segments(batch, st, lambda name, fn: fn())
        return st['out']
      self._graphs[key] = ent
      self.graph_stats['captures'] += 1
      while len(self._graphs) > self.max_graphs:
        self._graphs.popitem(last=False)
        self.graph_stats['evicted'] += 1
    else:
      self._graphs.move_to_end(key)
      t0 = time.perf_counter()
      ent[0].load(batch)
      self.host_seconds['stage_batch'] += time.perf_counter() - t0
    sb, graphs, st, _ = ent
    self.graph_stats['replays'] += 1
    if 'all' in graphs:
      t0 = time.perf_counter()
      graphs['all'].replay()
      dt = time.perf_counter() - t0
      self.host_seconds['graph_launch'] += dt
      self.host_launch_samples.append(dt)
      if 'adam' in graphs:          # data parallel: gradient exchange between the two graphs
        self._exchange_all(st)
        graphs['adam'].replay()
      return st['out']
    self._run_segments(sb.tensors(), st, lambda name, fn: graphs[name].replay())
    return st['out']

  def _capture(self, sb):
    """Capture the iteration for the bucket whose static buffers are ``sb``; returns the cache entry
    (static batch, graphs, state dict with the output tensors, launch epoch)."""
    imgs = sb.imgs
    # reduction scratch the crop backward wants: one image-sized plane per (padded) object
    # ... and the layout backward's per-tile partials (0.5 GB at 256 x 256 with ~900 objects: the lane workspaces grow
    # here, before the capture - ops.workspace)
    H, W = self.model.image_size
    ws_need = int(_lib.load().sg2im_layout_backward_workspace(int(sb.o_pad), int(self.model_kwargs['gconv_dim']), int(H), int(W)))
    self._prepare_lanes(max(1 << 24, sb.o_pad * imgs.size(2) * imgs.size(3) * imgs.size(1) + (1 << 20)), ws_need)
    st = {'losses': {}, 'ocnt': sb.obj_count, 'tcnt': sb.triple_count}
    static = sb.tensors()
    dp = self.world_size > 1 or self.reducer.force
    # Data parallel, two forms:
    #  SG2IM_DP_SCHEDULE=2 (default, RCCL only): ONE graph with the all-reduces recorded inside it
    #    (_capture_overlapped); falls back to schedule 0 for a process group that cannot be captured (gloo).
    #  SG2IM_DP_SCHEDULE=0: ONE iteration graph (discriminator steps on the side stream next
    #    to the generator backward, as at N = 1) -> the four all-reduces -> Adam graph.  The exchange
    #    (119.7 MB) is exposed, but the iteration graph is the short one.
    #  SG2IM_DP_SCHEDULE=1: [G fwd + bwd | D_img] graph -> all-reduce(G, guard, D_img) started ->
    #    [D_obj step] graph replayed while they are in flight -> all-reduce(D_obj) -> Adam graph.
    #    Hides the exchange behind the D_obj step, but that step then no longer runs next to the
    #    generator backward: measured 11.55 ms/step of compute against 10.4 for the one-graph form on
    #    one GPU (bench.py --force_dist), i.e. it only wins if the exchange costs > 1.1 ms.
    segmented = dp and self.dp_schedule == 1
    torch.cuda.synchronize()
    _lib.CAPTURING = True
    lib = _lib.load()
    n0, g0 = lib.sg2im_launch_count(0), lib.sg2im_launch_count(1)
    try:
      if self.overlap_d and not segmented:
        graphs = self._capture_overlapped(static, st, dp)
      else:
        graphs, pool = {}, [None]

        def capture(name, fn):
          g = torch.cuda.CUDAGraph()
          with torch.cuda.graph(g, pool=pool[0], stream=self._cap_stream, capture_error_mode=_CAPTURE_MODE):
            fn()
          if pool[0] is None:
            pool[0] = g.pool()
          graphs[name] = g
        mute, self.reducer.mute = self.reducer.mute, True       # (no collectives while capturing)
        try:
          self._run_segments(static, st, capture)
        finally:
          self.reducer.mute = mute
    finally:
      _lib.CAPTURING = False
    # kernels of this library in one captured iteration (torch's own - batch staging, layout noise, arena
    # zeroing - not counted): all / implicit-GEMM family incl. split-K finishes
    self.launch_stats = {'launches_per_step': int(lib.sg2im_launch_count(0) - n0),
                         'gemm_launches_per_step': int(lib.sg2im_launch_count(1) - g0)}
    return (sb, graphs, st, _lib.EAGER_EPOCH)

  def _generator_buckets(self):
    """the early buckets of the generator's gradient arena, see generator_buckets"""
    return generator_buckets(self.model, self.flat_g)

  def _generator_bucket(self):
    return generator_bucket(self.model, self.flat_g)

  @staticmethod
  def _log_schedule(msg):
    if not Trainer._schedule_logged:
      Trainer._schedule_logged = True
      print('[sg2im_amd] data-parallel schedule: ' + msg, flush=True)

  _schedule_logged = False

  def _exchange_all(self, st):
    red = self.reducer
    red.start(self.flat_g.grad)
    red.start(st['guard'])
    if self.flat_do is not None:
      red.start(self.flat_do.grad)
    if self.flat_di is not None:
      red.start(self.flat_di.grad)
    red.finish()

  def _capture_overlapped(self, static, st, dp=False):
    """One graph for the whole iteration: generator forward, then a fork - the generator's backward
    on the capture stream, the discriminator steps on a side stream (own split-K workspace /
    scratch, which ops keys by stream) - joined before the three Adam updates.  Data parallel
    (dp): the Adam updates are a second graph and the four all-reduces are issued between the two
    replays (the exchange is then not hidden behind compute, but the overlapped graph is 1.3 ms
    shorter than the sequential segments that could hide it)."""
    # dp_schedule 2: the gradient exchange is part of the graph.  Collectives are issued on the comm stream in one
    # fixed order (guard, D_img, D_obj, generator bucket A, the rest of the generator) as soon as their gradients
    # are complete: the discriminators' right after their steps on the side stream; bucket A = the weights of the
    # first two refinement modules (1024- and 512-channel 3x3 convolutions: ~2/3 of the generator's 112.6 MB for
    # 1/10 of its weight-gradient time) right after their weight gradients, which are the FIRST the deferred
    # release issues - it travels while the remaining weight gradients (~1.5 ms) run; only the rest (the graph
    # convolution / MLP / small-conv gradients, ~1/3 of the bytes) is exchanged after the backward pass.
    ingraph = dp and self.dp_schedule == 2 and self.reducer.capturable() and not self.reducer.mute
    comm, red = self._comm, self.reducer

    packed = []                      # bf16 payload: (tensor) reduced through its staging buffer, not yet widened back

    def reduce_after(stream, *tensors):
      live = red.live()
      for t in tensors:
        if live and red.packs(t):            # (bf16 payload: rounded on the producer's stream, see GradReducer.pack)
          with torch.cuda.stream(stream):
            red.pack(t)
      comm.wait_stream(stream)
      with torch.cuda.stream(comm):
        for t in tensors:
          if live and red.packs(t):
            red.reduce_packed(t)
            packed.append(t)
          else:
            red.reduce_here(t)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=self._cap_stream, capture_error_mode=_CAPTURE_MODE):
      main = torch.cuda.current_stream()
      side = self._side[0]
      self._seg_generator_forward(static, st)
      if ingraph:
        reduce_after(main, st['guard'])

      def on_side(seg, grads=None):
        if ops.SINGLE_STREAM:
          seg(static, st)
          return
        side.wait_stream(main)
        with torch.cuda.stream(side):
          seg(static, st)
        if ingraph and grads is not None:
          reduce_after(side, grads)
      # Schedule (measured in round 2).  The generator backward is: the refinement
      # network's data-gradient chain (big kernels), then the layout / mask / graph-convolution backward
      # (small dependent launches) with the refinement network's eleven weight gradients released underneath
      # them as background launches (ops.DEFER_WGRAD).  Both discriminator steps go on the side stream right
      # after the generator forward, next to the data-gradient chain: 9.66 ms against 9.77-9.81 for the
      # variants that hold the D_obj step back until the backward reaches the layout.  The ORDER OF CAPTURE
      # matters, not only the dependencies: a replay issues the nodes in capture order, and branches only
      # overlap with what is issued around the same time - capturing the discriminator steps AFTER the
      # generator backward (same dependencies) costs 10.9 ms.
      # (the image discriminator's step on a stream of its own, next to the object discriminator's step instead
      # of in front of it: measured, no gain - 8.31 vs 8.31 ms, round 3)
      if self.d_img is not None:
        on_side(self._seg_d_img, self.flat_di.grad)
      if self.d_obj is not None:
        on_side(self._seg_d_obj, self.flat_do.grad)
      # Generator exchange in FOUR buckets, each sent as soon as it is complete (generator_buckets): refinement module
      # 0, module 1, modules 2.. + output convolutions - reported by the tags of the released weight gradients, in
      # the order the release issues them - and, after the backward pass, whatever the arena holds besides.
      buckets = self._generator_buckets() if ingraph else []
      sent = []
      if buckets:
        def early_cb(a, b):
          def cb(stream):
            reduce_after(stream, self.flat_g.grad[a:b])
            sent.append((a, b))
          return cb
        ops.AFTER_DEFERRED = [(ids, early_cb(a, b)) for a, b, ids in buckets]
      # One GPU: the Adam update of the refinement network's parameters (3/4 of the generator) at the END OF THE
      # WEIGHT-GRADIENT LANE, right behind its last weight gradient - every gradient of that slice is complete there
      # (convolutions: the lane itself; BatchNorm: the data-gradient chain the release waited for) - while the main
      # lane finishes the graph-convolution / embedding backward; only the rest is left for the final update.  No new
      # stream or fork / join edge (an update on a stream of its own, under the remaining weight gradients, re-mapped
      # the branches onto the hardware queues and cost 0.7 ms: profiles/r4_early_adam_ab.txt).  Same values: the
      # update is element-wise (optim.FlatAdam.apply_guarded).
      # NOT under data parallelism.  Round 5 built it (the three early buckets are exactly that slice: the lane - or
      # the discriminator lane - waits for the comm stream, then updates) and hipStreamEndCapture SEGFAULTS on it: any
      # captured stream other than the capture's origin that waits for the comm stream (whose only operations are
      # waits and the collectives' own fork / join) kills clr at the end of the capture, also with a kernel on the
      # comm stream in front of the wait (profiles/r5_dp_early_adam_capture_crash.txt: three placements, all dump
      # core; without the early update the same 4-bucket graph captures and runs).  The alternatives - the collective
      # of the last early bucket on the lane itself, or the update on the comm stream - either depend on torch's
      # choice of NCCL stream or put an elementwise kernel on the comm stream (+1.7 ms when measured in round 4).
      early_adam = not dp and not ops.SINGLE_STREAM and ops.DEFER_WGRAD and os.environ.get('SG2IM_EARLY_ADAM', '1') != '0'
      crn = refinement_slice(self.model, self.flat_g) if early_adam else None
      if crn is not None:
        a, b = crn
        self.opt_g.prepare_guarded(st['guard'])        # (main stream: the release fork orders it before the lane)
        st['g_adam_prepared'] = True

        def early_update(stream):
          self.opt_g.apply_guarded(a, b, self.reducer.grad_scale)     # (current stream = the weight-gradient lane)
          st['g_adam_early'] = (a, b)
        ops.AFTER_ALL_DEFERRED = early_update
      try:
        self._seg_generator_backward(st)
      finally:
        ops.AFTER_DEFERRED = None
        ops.AFTER_ALL_DEFERRED = None
      if ingraph:
        rest = complement(sent, self.flat_g.numel)
        reduce_after(main, *[self.flat_g.grad[a:b] for a, b in rest])
      if dp and self.rank == 0:
        self._log_schedule(
          ('2: all-reduces recorded inside the iteration graph, generator in %d bucket(s)%s' % (
            len(sent) + len(rest), '')) if ingraph else
          ('%d requested, running 0 (iteration graph -> exposed all-reduces -> Adam graph): in-graph collectives need '
           'RCCL and an unmuted reducer' % self.dp_schedule) if self.dp_schedule == 2 else
          '0: iteration graph -> exposed all-reduces -> Adam graph')
      main.wait_stream(side)
      if ingraph:
        main.wait_stream(comm)
        for t in packed:                     # (bf16 payload: back into the fp32 arenas Adam reads)
          red.unpack(t)
      if not dp or ingraph:
        self._seg_adam(st)
    graphs = {'all': g}
    if dp and not ingraph:
      ga = torch.cuda.CUDAGraph()
      with torch.cuda.graph(ga, pool=g.pool(), stream=self._cap_stream, capture_error_mode=_CAPTURE_MODE):
        self._seg_adam(st)
      graphs['adam'] = ga
    return graphs

  @staticmethod
  def losses_to_host(losses):
    """one host sync for a whole dict (call at print_every, train.py:594-609); also the place where a persistent
    kernel whose grid barrier ever timed out is reported (ops.persistent_kernels_check: raises)"""
    out = {k: float(v) for k, v in losses.items()}
    ops.persistent_kernels_check()
    if not math.isfinite(out.get('total_loss', 0.0)):
      print('WARNING: Got loss = NaN')       # the reference skips the update (train.py:553-555)
    return out
