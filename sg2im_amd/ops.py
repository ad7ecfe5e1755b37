"""Thin tensor-level wrappers over the C ABI (include/sg2im_hip.h).

PyTorch is used here only as plumbing: device memory (torch tensors), the current HIP
stream and shapes.  Every function enqueues HIP kernels from libsg2im_hip.so on
``torch.cuda.current_stream()`` and returns immediately.  Inputs must live on the GPU;
anything else raises - there is no CPU path.
"""
import ctypes
import os
from ctypes import byref, c_int, c_longlong, c_void_p

import torch

from . import _lib
from ._lib import BnBwd, BnFwd, ConvDesc, GconvGrads, GconvLayer, GconvStack, GconvStackGrads, call

WORKSPACE_BYTES = 256 << 20      # split-K partials / layout-backward partials
_ws = {}
_scratch = {}


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream():
  """raw hipStream_t of torch's current stream.  torch.cuda.current_stream() costs ~9 us of
  Python per call (x ~500 launches per step); the raw query is one C call."""
  if _raw_stream is not None:
    return c_void_p(_raw_stream(torch.cuda.current_device()))
  return c_void_p(torch.cuda.current_stream().cuda_stream)


def _f(t):
  """device pointer of an fp32 CUDA tensor (None -> NULL)"""
  if t is None:
    return None
  if not (t.is_cuda and t.dtype == torch.float32):
    raise TypeError('expected a float32 tensor on the GPU, got %s on %s' % (t.dtype, t.device))
  return c_void_p(t.data_ptr())


def _fb(t):
  """device pointer of an fp32 OR bfloat16 (storage) CUDA tensor (None -> NULL)"""
  if t is None:
    return None
  if not (t.is_cuda and t.dtype in (torch.float32, torch.bfloat16)):
    raise TypeError('expected a float32 / bfloat16 tensor on the GPU, got %s on %s' % (t.dtype, t.device))
  return c_void_p(t.data_ptr())


def _dt(t):
  """storage type code of the C ABI: 0 float32, 1 bfloat16"""
  return 1 if t.dtype == torch.bfloat16 else 0


def _i64(t):
  if t is None:
    return None
  if not (t.is_cuda and t.dtype == torch.int64 and t.is_contiguous()):
    raise TypeError('expected a contiguous int64 tensor on the GPU')
  return c_void_p(t.data_ptr())


def _i32(t):
  if t is None:
    return None
  if not (t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()):
    raise TypeError('expected a contiguous int32 tensor on the GPU')
  return c_void_p(t.data_ptr())


# Deferred weight gradients (Trainer, graph capture): the refinement network's backward queues its
# weight-gradient launches instead of interleaving them with the data-gradient chain, and releases
# them all at once on the side lane (ONE fork, last layer first, as "background" launches:
# sg2im_conv_desc.launch_hints) when that chain is through - they then run underneath the small
# dependent kernels of the layout / mask / graph-convolution backward, which otherwise have the
# GPU to themselves.  The Trainer sets DEFERRED to a list for the duration of the generator backward
# and joins the lanes it finds there before the optimiser step.  [measured: 10.3 -> 9.7 ms per step,
# profiles/r2_deferred_wgrad_ab.log]
DEFER_WGRAD = True
# probe knob: capture the whole iteration on ONE stream (a hipGraph without parallel branches replays on clr's
# single-stream fast path, tools/graph_order_probe.py)
SINGLE_STREAM = os.environ.get('SG2IM_SINGLE_STREAM', '0') == '1'
DEFERRED = None
# [(ids, callback(stream)), ...]: the early gradient buckets of the data-parallel Trainer.  `ids` = data_ptr() of the
# parameters whose weight gradients complete a bucket.  Every deferred launch is tagged with the parameters it
# completes; a bucket's callback runs on the weight-gradient stream right after the LAST of its launches was issued
# (never, if one of its parameters is not among the released launches - the Trainer then exchanges that slice with
# the rest of the arena after the backward pass).
AFTER_DEFERRED = None
# callback(stream): called on the weight-gradient stream after the LAST released weight gradient was issued
AFTER_ALL_DEFERRED = None
HINT_BACKGROUND = 1
# only the first BG_COUNT released weight gradients are issued as background launches (occupancy cap): the later
# ones run after the small-kernel tail is over and may have the whole GPU
BG_COUNT = int(os.environ.get('SG2IM_BG_COUNT', '6'))        # [measured, round 2: all 11: 8.90-8.94, 8: 8.84, 6: 8.81, 4: 8.82, 2: 8.85 ms]
def _lane(device):
  """Work buffers are per (device, stream): kernels of one in-order stream use them one after
  the other, concurrently running streams (the Trainer's side stream, autograd branches that
  ran their forward on it) never share them.  Inside graph capture the same holds per captured
  stream."""
  idx = device.index if device.index is not None else torch.cuda.current_device()
  if _raw_stream is not None:
    return (idx, _raw_stream(idx))
  return (idx, torch.cuda.current_stream(idx).cuda_stream)


_wgrad_streams = {}
# the four weight gradients of a GraphTripleConv layer as one grouped launch (sg2im_conv2d_backward_weight_group)
GROUP_WGRAD = True


# Schedule marks (diagnostics, SG2IM_MARKS=1): mark(name) stores the device clock when the current stream reaches
# that point of a (captured) iteration; marks_report() reads the last replay's values back.
MARKS = os.environ.get('SG2IM_MARKS', '0') == '1'
_marks = {'buf': None, 'names': {}}


def marks_init(device):
  if MARKS and _marks['buf'] is None:
    _marks['buf'] = torch.zeros(256, dtype=torch.int64, device=device)


def mark(name):
  if not MARKS or _marks['buf'] is None:
    return
  idx = _marks['names'].setdefault(name, len(_marks['names']))
  if idx >= _marks['buf'].numel():
    raise RuntimeError('too many distinct schedule marks (%d slots)' % _marks['buf'].numel())
  call('sg2im_timestamp', _marks['buf'].data_ptr() + 8 * idx, _stream())


def marks_report():
  """[(name, microseconds since the 'start' mark)] of the last replay, in time order"""
  if _marks['buf'] is None:
    return []
  v = _marks['buf'].cpu().tolist()
  t0 = v[_marks['names'].get('start', 0)]
  return sorted(((n, (v[i] - t0) / 100.0) for n, i in _marks['names'].items()), key=lambda r: r[1])


def release_deferred(queue, stream, pending=None):
  """issue the queued (fn, tags) launches, last queued first (the layers the data-gradient chain reached last first:
  8.80 vs 8.84-8.85 ms in queue order), the first BG_COUNT as background launches; each AFTER_DEFERRED bucket's
  callback runs right after the last launch that completes one of its parameters.  pending: the open buckets of a lane
  that releases its queue in several parts (SideLane.flush) - [[ids still to come, callback], ...], updated in place."""
  if pending is None:
    pending = [[set(ids), cb] for ids, cb in (AFTER_DEFERRED or ()) if ids]
  # (a bucket whose parameters are not all among the released launches is never reported complete: its set never empties)
  for k, (fn, tags) in enumerate(reversed(queue)):
    fn(k < BG_COUNT)
    for ent in pending:
      if ent[0]:
        ent[0] -= tags
        if not ent[0]:
          ent[1](stream)                   # (Trainer: this gradient bucket is complete on this stream)


class SideLane(object):
  """Runs the weight-gradient launches of a backward pass on a second stream.  They are leaves of
  the backward graph (nothing downstream reads dW before the optimiser), so the small kernels of
  the data-gradient chain on the calling stream - BatchNorm backward, split-K finishes - execute
  underneath them instead of alone on an otherwise idle GPU [measured: tools/overlap_probe.py].
  Every tensor a side launch reads is kept referenced until ``join`` so the caching allocator
  cannot hand its memory to the calling stream while the side stream still uses it."""

  def __init__(self, device):
    # only while a hipGraph is being captured: launched eagerly, the extra event / stream switches
    # cost more host time (the eager step is launch bound) than the overlap returns
    self.on = torch.cuda.is_current_stream_capturing() and not SINGLE_STREAM
    self.used = False
    self.keep = []
    self.queue = []
    self.deferring = False     # (deferred mode: no joins before the final one)
    self.pending = None        # (open gradient buckets between two flush() calls)
    if self.on:
      idx = device.index if device.index is not None else torch.cuda.current_device()
      self.main = torch.cuda.current_stream(idx)
      key = (idx, self.main.cuda_stream)
      if key not in _wgrad_streams:
        _wgrad_streams[key] = torch.cuda.Stream(device=idx)
      self.side = _wgrad_streams[key]

  def run(self, fn, *reads):
    """fn() on the side stream, ordered after everything launched so far on the calling stream"""
    if not self.on:
      return fn()
    ev = torch.cuda.Event()
    ev.record(self.main)
    self.side.wait_event(ev)
    with torch.cuda.stream(self.side):
      out = fn()
    self.keep.extend(reads)
    self.used = True
    return out

  def defer(self, fn, *reads, completes=()):
    """queue fn for flush(); completes: the parameters whose gradients this launch finishes (see AFTER_DEFERRED)"""
    self.deferring = True
    self.queue.append((fn, frozenset(p.data_ptr() for p in completes if p is not None)))
    self.keep.extend(reads)

  def flush(self, final=True):
    """all queued launches on the side stream, ordered after everything launched so far on the calling
    stream (one fork per call; final=False: more launches will be queued and released by a later call)"""
    if not self.queue and not (final and self.used):
      return
    if self.pending is None:
      self.pending = [[set(ids), cb] for ids, cb in (AFTER_DEFERRED or ()) if ids]
    ev = torch.cuda.Event()
    ev.record(self.main)
    self.side.wait_event(ev)
    with torch.cuda.stream(self.side):
      if not self.used:
        mark('wgrad_lane_start')
      release_deferred(self.queue, self.side, self.pending)
      if final:
        mark('wgrad_lane_done')
        if AFTER_ALL_DEFERRED is not None:
          AFTER_ALL_DEFERRED(self.side)
    self.queue = []
    self.used = True

  def barrier(self):
    """the calling stream waits for the side launches so far (the next big kernel runs alone)"""
    if self.on and self.used and not self.deferring:
      self.main.wait_stream(self.side)

  def join(self):
    if self.used:
      self.main.wait_stream(self.side)
      self.keep = []
      self.used = False


_ws_retired = []


def workspace(device, min_bytes=0):
  """Split-K / layout-backward work buffer of the current lane (device, stream): 256 MB, grown on demand (the
  256 x 256 configuration with ~900 objects needs 0.5 GB for the layout backward's per-tile partials).  Like
  `scratch`, an outgrown buffer is RETIRED, not freed (graphs captured earlier keep using its address), and it
  cannot grow inside a stream capture - Trainer._prepare_lanes sizes it before the capture starts."""
  key = _lane(device)
  w = _ws.get(key)
  if w is None or w.numel() * 4 < min_bytes:
    if w is not None:
      if torch.cuda.is_current_stream_capturing():
        raise _lib.Sg2imHipError('the lane workspace (%d bytes) would have to grow to %d bytes inside a stream capture'
                                 % (w.numel() * 4, min_bytes))
      _ws_retired.append(w)
    nbytes = max(WORKSPACE_BYTES, (int(min_bytes) + (1 << 20) - 1) >> 20 << 20)
    w = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
    _ws[key] = w
  return w


_scratch_retired = []


def scratch(device, nfloats):
  """Reduction scratch (per device and stream), grown on demand.  A buffer that is outgrown is RETIRED, not
  freed: hipGraphs captured earlier have its address baked in and keep writing into it on every replay - were it
  returned to the caching allocator, a replay of an older bucket's graph would silently corrupt whatever tensor
  owns that memory by then.  The retired buffers are small next to 288 GB of HBM."""
  key = _lane(device)
  s = _scratch.get(key)
  if s is None or s.numel() < nfloats:
    if s is not None:
      _scratch_retired.append(s)
    s = torch.empty(max(int(nfloats), 1 << 20), dtype=torch.float32, device=device)
    _scratch[key] = s
  return s


_sync_areas = {}


def sync_area(device):
  """grid-barrier state of the persistent kernels (per device and stream; zeroed by every launch on its stream).
  Like the other lane buffers it must exist before a capture starts (Trainer._prepare_lanes)."""
  key = _lane(device)
  s = _sync_areas.get(key)
  if s is None:
    s = torch.zeros(int(_lib.load().sg2im_gconv_stack_sync_bytes()) // 4, dtype=torch.int32, device=device)
    _sync_areas[key] = s
  return s


def rows_ld(t):
  """(pointer, ld) of a 2-D row matrix whose rows are contiguous (column slices allowed)."""
  if t.dim() != 2 or (t.size(1) > 1 and t.stride(1) != 1):
    raise ValueError('expected a row matrix with unit column stride')
  return _f(t), t.stride(0) if t.size(0) > 1 else max(t.stride(0), t.size(1))


# ----------------------------------------------------------------------------
# conv / linear
# ----------------------------------------------------------------------------

class SrcSpec(object):
  __slots__ = ('t', 'channels', 'ld', 'up', 'gather', 'scale', 'shift', 'slope')

  def __init__(self, t, channels, ld, up=0, gather=None, scale=None, shift=None, slope=1.0):
    self.t, self.channels, self.ld, self.up = t, int(channels), int(ld), int(up)
    self.gather, self.scale, self.shift, self.slope = gather, scale, shift, float(slope)


def halo_unsplit(batch, h, w, cols):
  """does the halo'd 3x3 kernel run this map with `cols` output columns without split-K (sg2im_conv_halo_unsplit)?"""
  return bool(_lib.load().sg2im_conv_halo_unsplit(int(batch), int(h), int(w), int(cols), 64))


def nhwc_src(t, up=0, scale=None, shift=None, slope=1.0):
  """source from a contiguous NHWC tensor"""
  assert t.dim() == 4 and t.is_contiguous()
  return SrcSpec(t, t.size(3), t.size(3), up, None, scale, shift, slope)


def rows_src(t, gather=None):
  """source from a row matrix (optionally row-gathered)"""
  _, ld = rows_ld(t)
  return SrcSpec(t, t.size(1), ld, 0, gather)


# Arithmetic of the SPATIAL convolutions (refinement network, discriminators, mask_net): 0 = fp32
# matrix cores, 1 = bf16 operands with fp32 accumulation (sg2im_conv_desc.compute_dtype).  Linear
# layers (row matrices) always compute in fp32.  Set by Trainer(compute_dtype='bf16') around its step.
CONV_COMPUTE = 0
# set by Trainer.step (bf16 mode) after FlatParams.refresh_mirror: convolutions may read bf16 weight mirrors
WEIGHT_MIRROR = False


def conv_desc(srcs, batch, in_h, in_w, kh=1, kw=1, stride=1, pad=0, compute=None, weight_channels=0, weight_bf16=None):
  d = ConvDesc()
  d.weight_channels = int(weight_channels)     # (weight rows wider than the sources: include/sg2im_hip.h)
  d.weight_bf16 = weight_bf16                  # (device address of the weights' bf16 mirror, or None)
  if compute is None:
    compute = CONV_COMPUTE if (kh * kw > 1 or in_h * in_w > 1) else 0
  d.compute_dtype = int(compute)
  d.nsrc = len(srcs)
  for i, s in enumerate(srcs):
    q = d.src[i]
    q.data = s.t.data_ptr()
    q.gather = s.gather.data_ptr() if s.gather is not None else None
    q.scale = s.scale.data_ptr() if s.scale is not None else None
    q.shift = s.shift.data_ptr() if s.shift is not None else None
    q.slope, q.channels, q.ld, q.upsample_log2 = s.slope, s.channels, s.ld, s.up
    if not (s.t.is_cuda and s.t.dtype in (torch.float32, torch.bfloat16)):
      raise TypeError('conv source must be a float32 (or bfloat16-storage) GPU tensor')
    q.dtype = _dt(s.t)              # (bfloat16 storage: the bf16 halo'd kernels only, include/sg2im_hip.h)
  d.batch, d.in_h, d.in_w = int(batch), int(in_h), int(in_w)
  d.kh, d.kw, d.stride, d.pad = int(kh), int(kw), int(stride), int(pad)
  d.out_h = (in_h + 2 * pad - kh) // stride + 1
  d.out_w = (in_w + 2 * pad - kw) // stride + 1
  d._keep = srcs            # keep the tensors alive as long as the descriptor
  return d


class KernelTimer(object):
  """Optional HIP-event timing of the implicit-GEMM launches (bench.py's roofline leg).
  Events are recorded on the stream the kernels are launched on (torch's current stream),
  so ``elapsed_time`` is the device-side duration of the launch (incl. its split-K finish)."""

  def __init__(self):
    self.records = []        # (kind, flops, start_event, end_event, tag)
    self.alg_bytes = {}      # kind -> algorithmic HBM bytes of its launches (operands read once, result written once)

  def summary(self, tag=None):
    """per-kind totals; tag: only the launches made inside that network (TIMER_TAG)"""
    torch.cuda.synchronize()
    out = {}
    for kind, flops, e0, e1, tg in self.records:
      if tag is not None and tg != tag:
        continue
      d = out.setdefault(kind, {'launches': 0, 'flops': 0.0, 'ms': 0.0})
      d['launches'] += 1
      d['flops'] += flops
      d['ms'] += e0.elapsed_time(e1)
    return out


TIMER = None     # set to a KernelTimer to time every conv / linear launch
TIMER_TAG = None # set by a network around its launches (e.g. 'crn') to attribute them


def _timed(kind, flops, fn, bn_finish=False, finish_bytes=0.0):
  """bn_finish: fn is a *_bn entry point - its BatchNorm finish launch (not a GEMM) is timed as 'hbm_bn_finish';
  finish_bytes: what that launch must move - the tile partials read once, the per-channel results written"""
  if TIMER is None:
    return fn()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  em, hit = None, c_int(0)
  if bn_finish:
    em = torch.cuda.Event(enable_timing=True)
    em.record()                     # (creates the underlying hipEvent_t; the library records it again)
    _lib.load().sg2im_debug_mark_gemm_end(c_void_p(em.cuda_event), byref(hit))
  e0.record()
  try:
    fn()
  finally:
    if bn_finish:
      _lib.load().sg2im_debug_mark_gemm_end(None, None)
  e1.record()
  if hit.value:
    TIMER.records.append((kind, flops, e0, em, TIMER_TAG))
    TIMER.records.append(('hbm_bn_finish', float(finish_bytes), em, e1, TIMER_TAG))
  else:
    TIMER.records.append((kind, flops, e0, e1, TIMER_TAG))


def _desc_k(desc):
  return desc.kh * desc.kw * sum(desc.src[i].channels for i in range(desc.nsrc))


def _desc_src_floats(desc):
  """stored elements behind the (virtual) input: an upsampled source is read at its own size"""
  n = 0
  for i in range(desc.nsrc):
    s = desc.src[i]
    n += desc.batch * (desc.in_h >> s.upsample_log2) * (desc.in_w >> s.upsample_log2) * s.channels
  return n


def _note_bytes(kind, nfloats):
  if TIMER is not None:
    TIMER.alg_bytes[kind] = TIMER.alg_bytes.get(kind, 0.0) + 4.0 * nfloats


WEIGHT_MIRROR_LOOKUP = None      # functional._weight_mirror (set at import): weight tensor -> mirror address | None


def _set_mirror(desc, weight):
  """the bf16 mirror of `weight` for the launches that can use it (bf16 operands, 3x3: the halo'd kernels)"""
  desc.weight_bf16 = (WEIGHT_MIRROR_LOOKUP(weight) if (WEIGHT_MIRROR and desc.compute_dtype == 1 and desc.kh == 3 and
                                                       WEIGHT_MIRROR_LOOKUP is not None) else None)


def conv2d_forward(desc, weight, cout, bias, out, ld_out, out_slope=1.0, accumulate=False):
  _set_mirror(desc, weight)
  ws = workspace(out.device)
  flops = 2.0 * desc.batch * desc.out_h * desc.out_w * cout * _desc_k(desc)
  _note_bytes('igemm_fwd', _desc_src_floats(desc) + cout * _desc_k(desc) + desc.batch * desc.out_h * desc.out_w * cout)
  desc.out_dtype = _dt(out)
  _timed('igemm_fwd', flops, lambda: call(
    'sg2im_conv2d_forward', byref(desc), _f(weight), int(cout), _f(bias), float(out_slope), _fb(out),
    int(ld_out), int(accumulate), _f(ws), ws.numel() * 4, _stream()))
  return out


def conv2d_backward_data(desc, weight, cout, dy, ld_dy, c_begin, c_count, dx, ld_dx, accumulate=False):
  _set_mirror(desc, weight)
  ws = workspace(dx.device)
  flops = 2.0 * desc.batch * desc.out_h * desc.out_w * cout * desc.kh * desc.kw * c_count
  _note_bytes('igemm_dgrad', desc.batch * desc.out_h * desc.out_w * cout + cout * desc.kh * desc.kw * c_count +
              desc.batch * desc.in_h * desc.in_w * c_count)
  desc.dy_dtype, desc.out_dtype = _dt(dy), _dt(dx)
  _timed('igemm_dgrad', flops, lambda: call(
    'sg2im_conv2d_backward_data', byref(desc), _f(weight), int(cout), _fb(dy), int(ld_dy), int(c_begin),
    int(c_count), _fb(dx), int(ld_dx), int(accumulate), _f(ws), ws.numel() * 4, _stream()))
  return dx


# A/B knob: 0 = the activation backward as a launch of its own behind the data gradient (rounds 1-4)
FUSE_ACT_BWD = os.environ.get('SG2IM_FUSE_ACT_BWD', '1') != '0'


def conv2d_backward_data_act(desc, weight, cout, dy, ld_dy, c_begin, c_count, dx, ld_dx, act, ld_act, slope):
  """conv2d_backward_data whose result is multiplied by leaky'_slope(act) - act = the ACTIVATED output of the layer
  the gradient flows into - in the data gradient's own launches (sg2im_conv2d_backward_data_act)"""
  if not FUSE_ACT_BWD:
    conv2d_backward_data(desc, weight, cout, dy, ld_dy, c_begin, c_count, dx, ld_dx)
    rows = dx.numel() // c_count
    return act_backward(c_void_p(dx.data_ptr()), ld_dx, 0, rows, 1, 1, act, ld_act, c_count, slope, dx)
  _set_mirror(desc, weight)
  ws = workspace(dx.device)
  flops = 2.0 * desc.batch * desc.out_h * desc.out_w * cout * desc.kh * desc.kw * c_count
  _note_bytes('igemm_dgrad', desc.batch * desc.out_h * desc.out_w * cout + cout * desc.kh * desc.kw * c_count +
              2 * desc.batch * desc.in_h * desc.in_w * c_count)
  desc.dy_dtype, desc.out_dtype = 0, 0
  _timed('igemm_dgrad', flops, lambda: call(
    'sg2im_conv2d_backward_data_act', byref(desc), _f(weight), int(cout), _f(dy), int(ld_dy), int(c_begin),
    int(c_count), _f(dx), int(ld_dx), _f(act), int(ld_act), float(slope), _f(ws), ws.numel() * 4, _stream()))
  return dx


def conv2d_backward_weight(desc, dy, ld_dy, cout, dweight, accumulate=False, dbias=None):
  """dbias (optional): the layer's bias gradient, produced in the same pass over dy."""
  ws = workspace(dweight.device)
  flops = 2.0 * desc.batch * desc.out_h * desc.out_w * cout * _desc_k(desc)
  _note_bytes('igemm_wgrad', _desc_src_floats(desc) + desc.batch * desc.out_h * desc.out_w * cout + cout * _desc_k(desc))
  desc.dy_dtype = _dt(dy)
  _timed('igemm_wgrad', flops, lambda: call(
    'sg2im_conv2d_backward_weight', byref(desc), _fb(dy), int(ld_dy), int(cout), _f(dweight),
    _f(dbias) if dbias is not None else None, int(accumulate), _f(ws), ws.numel() * 4, _stream()))
  return dweight


def conv2d_backward_weight_group(items, accumulate=True):
  """items: up to 4 tuples (desc, dy (M, cout) dense, cout, dweight, dbias | None) - one launch + one finish
  launch for all of them (include/sg2im_hip.h).  Returns False (nothing launched) when a problem does not
  qualify; the caller then uses conv2d_backward_weight per item."""
  n = len(items)
  if n < 1 or n > 4:
    return False
  ws = workspace(items[0][3].device)
  D_, P_, I_ = ctypes.POINTER(ConvDesc), c_void_p, ctypes.c_int
  descs = (D_ * n)(*[ctypes.pointer(it[0]) for it in items])
  dys = (P_ * n)(*[it[1].data_ptr() for it in items])
  lds = (I_ * n)(*[int(it[1].stride(0)) if it[1].size(0) > 1 else int(it[2]) for it in items])
  couts = (I_ * n)(*[int(it[2]) for it in items])
  dws = (P_ * n)(*[it[3].data_ptr() for it in items])
  dbs = (P_ * n)(*[(it[4].data_ptr() if it[4] is not None else None) for it in items])
  if not _lib._inited:
    _lib.init()
  lib = _lib.load()
  rc = lib.sg2im_conv2d_backward_weight_group(n, descs, dys, lds, couts, dws, dbs, int(accumulate), _f(ws),
                                              ws.numel() * 4, _stream())
  if rc == 1:                      # SG2IM_ERR_ARG: not groupable, nothing was launched
    return False
  if rc != 0:
    raise _lib.Sg2imHipError('sg2im_conv2d_backward_weight_group failed (%d)' % rc)
  if not _lib.CAPTURING:
    _lib.EAGER_EPOCH += 1
  return True


def cast_f32_to_bf16(src, dst, n):
  """dst[:n] = bfloat16(src[:n]) (sg2im_cast_f32_to_bf16)"""
  if not (src.is_cuda and src.dtype == torch.float32 and dst.is_cuda and dst.dtype == torch.bfloat16 and
          src.is_contiguous() and dst.is_contiguous() and src.numel() >= n and dst.numel() >= n):
    raise TypeError('cast_f32_to_bf16: contiguous float32 source / bfloat16 destination on the GPU')
  call('sg2im_cast_f32_to_bf16', c_void_p(src.data_ptr()), c_void_p(dst.data_ptr()), int(n), _stream())
  return dst


def column_sum(x_ptr, rows, cols, ld, out, accumulate=False):
  part = scratch(out.device, 2 * cols * 1024)
  call('sg2im_column_sum', x_ptr, int(rows), int(cols), int(ld), _f(out), int(accumulate), _f(part), _stream())
  return out


# ----------------------------------------------------------------------------
# graph
# ----------------------------------------------------------------------------

class Csr(object):
  """Stable CSR over destination rows (see sg2im_csr_build)."""
  __slots__ = ('row_ptr', 'entries', 'n_a', 'n_b', 'n_rows')

  def __init__(self, keys_a, keys_b, n_rows, live=None):
    """live: None or (int32 device scalar, 1) - only that many leading keys of each array are real
    (padded batches, sg2im_amd/bucketing.py)"""
    dev = keys_a.device
    self.n_a = keys_a.numel()
    self.n_b = keys_b.numel() if keys_b is not None else 0
    self.n_rows = int(n_rows)
    n = self.n_a + self.n_b
    self.row_ptr = torch.empty(self.n_rows + 1, dtype=torch.int32, device=dev)
    self.entries = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    tmp = torch.empty(self.n_rows + max(n, 1), dtype=torch.int32, device=dev)
    call('sg2im_csr_build', _i64(keys_a), self.n_a, _i64(keys_b), self.n_b, self.n_rows, _i32(self.row_ptr),
         _i32(self.entries), _i32(tmp), _count_args(live)[0], _stream())


def triples_csr(triples, n_rows, live=None):
  """(s, p, o, Csr(s, o, n_rows, live)) from the (T, 3) int64 triples tensor in one launch
  (sg2im_csr_build_triples): the three columns come back as contiguous views of one (3, T) tensor."""
  T = triples.size(0)
  if triples.dtype != torch.int64 or triples.dim() != 2 or triples.size(1) != 3:
    raise ValueError('triples_csr: an int64 tensor of shape (T, 3) is expected')
  triples = triples.contiguous()
  split = torch.empty(3, max(T, 1), dtype=torch.int64, device=triples.device)
  csr = Csr.__new__(Csr)
  csr.n_a = csr.n_b = T
  csr.n_rows = int(n_rows)
  csr.row_ptr = torch.empty(csr.n_rows + 1, dtype=torch.int32, device=triples.device)
  csr.entries = torch.empty(max(2 * T, 1), dtype=torch.int32, device=triples.device)
  tmp = torch.empty(csr.n_rows + max(2 * T, 1), dtype=torch.int32, device=triples.device)
  call('sg2im_csr_build_triples', _i64(triples), T, csr.n_rows, _i64(split), _i32(csr.row_ptr), _i32(csr.entries),
       _i32(tmp), _count_args(live)[0], _stream())
  return split[0, :T], split[1, :T], split[2, :T], csr


def segment_sum(src_a, src_b, csr, width, average, out, accumulate=False):
  pa, lda = rows_ld(src_a)
  pb, ldb = rows_ld(src_b) if src_b is not None else (None, 0)
  po, ldo = rows_ld(out)
  # algorithmic bytes: every entry row read once, every destination row written once, the CSR read
  n_entries = src_a.size(0) + (src_b.size(0) if src_b is not None else 0)
  nbytes = 4.0 * (n_entries * width + csr.n_rows * width + n_entries + csr.n_rows + 1)
  _timed('hbm_pool_segment_sum', nbytes, lambda: call(
    'sg2im_segment_sum', pa, lda, csr.n_a, pb, ldb, _i32(csr.row_ptr), _i32(csr.entries), csr.n_rows,
    int(width), int(average), int(accumulate), po, ldo, _stream()))
  return out


def two_heads_supported(k, n1, n2):
  return bool(_lib.load().sg2im_two_heads_supported(int(k), int(n1), int(n2)))


def two_heads_forward(x, W1, b1, W2, b2, y1, y2):
  """y1 = x W1^T + b1, y2 = x W2^T + b2 in one launch (sg2im_two_heads_forward)"""
  px, ldx = rows_ld(x)
  p1, ld1 = rows_ld(y1)
  p2, ld2 = rows_ld(y2)
  M, K = x.shape
  n1, n2 = W1.size(0), W2.size(0)
  _timed('igemm_fwd', 2.0 * M * K * (n1 + n2), lambda: call(
    'sg2im_two_heads_forward', px, ldx, M, K, _f(W1), _f(b1), n1, _f(W2), _f(b2), n2, p1, ld1, p2, ld2, _stream()))
  return y1, y2


def two_heads_backward_data(g1, g2, W1, W2, dx):
  """dx = g1 W1 + g2 W2 in one launch (sg2im_two_heads_backward_data)"""
  p1, ld1 = rows_ld(g1)
  p2, ld2 = rows_ld(g2)
  pd, ldd = rows_ld(dx)
  M, K = dx.shape
  n1, n2 = W1.size(0), W2.size(0)
  _timed('igemm_dgrad', 2.0 * M * K * (n1 + n2), lambda: call(
    'sg2im_two_heads_backward_data', p1, ld1, p2, ld2, M, K, _f(W1), n1, _f(W2), n2, pd, ldd, _stream()))
  return dx


def gather_rows(src, idx, out, csr_for_average=None):
  ps, lds = rows_ld(src)
  po, ldo = rows_ld(out)
  rp = _i32(csr_for_average.row_ptr) if csr_for_average is not None else None
  call('sg2im_gather_rows', ps, lds, _i64(idx), idx.numel(), out.size(1), rp, po, ldo, _stream())
  return out


def gconv_pool_backward(dpooled, s_idx, o_idx, csr_for_average, g_pred, new_t, hidden, dout, slope, out):
  """d(new_t) of one GraphTripleConv layer from d(pooled) and d(new_p): one launch (see the C header)"""
  pd, ldd = rows_ld(dpooled)
  pn, ldn = rows_ld(new_t)
  po, ldo = rows_ld(out)
  pg, ldg = rows_ld(g_pred) if g_pred is not None else (None, 0)
  rp = _i32(csr_for_average.row_ptr) if csr_for_average is not None else None
  call('sg2im_gconv_pool_backward', pd, ldd, _i64(s_idx), _i64(o_idx), s_idx.numel(), rp, pg, ldg, pn, ldn,
       int(hidden), int(dout), float(slope), po, ldo, _stream())
  return out


def _gconv_layer_struct(obj_vecs, pred_vecs, s_idx, o_idx, csr, avg, weights):
  """weights: (W1a, b1a, W1b, b1b, W2a, b2a, W2b, b2b)"""
  L = GconvLayer()
  po, ldo = rows_ld(obj_vecs)
  T = pred_vecs.size(0)
  L.obj_vecs, L.ld_obj = po.value, ldo
  if T > 0:
    pp, ldp = rows_ld(pred_vecs)
    L.pred_vecs, L.ld_pred = pp.value, ldp
    L.s_idx, L.o_idx = _i64(s_idx).value, _i64(o_idx).value
  L.row_ptr, L.entries = csr.row_ptr.data_ptr(), csr.entries.data_ptr()
  L.n_objs, L.n_triples, L.din = obj_vecs.size(0), T, obj_vecs.size(1)
  L.hidden, L.dout, L.average = weights[4].size(0), weights[6].size(0), int(bool(avg))
  for k, w in zip(('w1a', 'b1a', 'w1b', 'b1b', 'w2a', 'b2a', 'w2b', 'b2b'), weights):
    setattr(L, k, _f(w).value if w is not None else None)
  L._keep = (obj_vecs, pred_vecs, s_idx, o_idx, csr, weights)
  return L


def _gconv_flops(L):
  T, O, H = L.n_triples, L.n_objs, L.hidden
  return 2.0 * (T * (3 * L.din * H + H * (2 * H + L.dout)) + O * (H * H + H * L.dout))


def gconv_layer_forward(L, h1, new_t, pooled, h2, new_obj):
  """one GraphTripleConv layer, forward: ONE call into the library (sg2im_gconv_layer_forward)"""
  ws = workspace(pooled.device)
  _note_bytes('igemm_fwd', _gconv_flops(L) / 2.0 / max(L.n_triples + L.n_objs, 1))      # (weights dominate; rough)
  _timed('igemm_fwd', _gconv_flops(L), lambda: call(
    'sg2im_gconv_layer_forward', byref(L), _f(h1), _f(new_t), _f(pooled), _f(h2), _f(new_obj), _f(ws), ws.numel() * 4,
    _stream()))


def gconv_layer_backward(L, h1, new_t, pooled, h2, new_obj, g_obj, g_pred, d_triple, d_obj, grads, accumulate):
  """grads: 8 tensors or None, in the order (dW1a, db1a, dW1b, db1b, dW2a, db2a, dW2b, db2b)"""
  dev = pooled.device
  ws = workspace(dev)
  need = _lib.load().sg2im_gconv_layer_backward_scratch(L.n_objs, L.n_triples, L.din, L.hidden, L.dout)
  sc = scratch(dev, need // 4 + 16)
  G = GconvGrads()
  for k, t in zip(('dw1a', 'db1a', 'dw1b', 'db1b', 'dw2a', 'db2a', 'dw2b', 'db2b'), grads):
    setattr(G, k, _f(t).value if t is not None else None)
  G.accumulate = int(bool(accumulate))
  pg, ldg = rows_ld(g_pred) if g_pred is not None else (None, 0)
  _timed('igemm_dgrad', 2.0 * _gconv_flops(L), lambda: call(
    'sg2im_gconv_layer_backward', byref(L), _f(h1), _f(new_t), _f(pooled), _f(h2), _f(new_obj), _f(g_obj), pg, int(ldg),
    _f(d_triple), _f(d_obj), byref(G), _f(sc), sc.numel() * 4, _f(ws), ws.numel() * 4, _stream()))


GCN_PERSISTENT = os.environ.get('SG2IM_GCN_PERSIST', '1') != '0'      # (A/B knob: 0 = one call per layer)
# The one-launch backward needs every one of its workgroups resident at once (grid barriers).  Inside the captured
# training iteration it starts underneath the refinement network's released weight gradients, which occupy every CU.
# 'full' (round 4: ~390 registers, 98 KB of LDS per workgroup = whole CUs): its first barrier waited ~0.5 ms for
# residency, 8.32 vs 7.99 ms (profiles/r4_gcn_persistent_backward_ab.txt).  'low' (round 5: <= 168 registers, 41 KB
# of LDS, 32 x 32 tiles, sg2im_gconv_stack_grads.low_footprint): its workgroups fit next to the weight gradients'.
# SG2IM_GCN_PERSIST_BWD = auto | 0 | full | low | staged | staged_full (1 = full; staged: the low / full kernel launched once
# per stage - no grid barrier, nothing co-resident); the same form runs in eager mode too, so that eager and replayed
# iterations stay bit-identical.  Every form is tested against the layer-by-layer launches in sec_gconv_stack.
# 'auto' (the default): the Trainer picks per configuration (trainer.Trainer._gcn_backward_mode) - 'low' where the main
# lane's small-kernel tail ENDS the step (VG-style batches / a trained mask_net: its backward sits in that tail; measured
# 9.51 -> 9.24 ms fp32, 5.84 -> 5.61 ms bf16 at VG-64), layer-by-layer launches where the weight-gradient lane ends it
# (the COCO-style headline configuration: 7.75 vs 7.90 ms) - profiles/r5_gcn_persistent_backward_low_footprint_ab.txt.
GCN_PERSISTENT_BACKWARD_MODE = os.environ.get('SG2IM_GCN_PERSIST_BWD', 'auto')
GCN_PERSISTENT_BACKWARD = {'0': False, '': False, 'auto': False, '1': 'full', 'full': 'full', 'low': 'low', '2': 'low',
                           'staged': 'staged', 'staged_full': 'staged_full'}.get(GCN_PERSISTENT_BACKWARD_MODE, False)


def gconv_stack_backward_in_one_launch():
  return bool(GCN_PERSISTENT_BACKWARD)


def gconv_stack_supported(dims):
  """dims: [(din, hidden, dout)] per layer - can the persistent stack kernels run them?"""
  lib = _lib.load()
  return (GCN_PERSISTENT and 1 <= len(dims) <= _lib.SG2IM_GCONV_MAX_LAYERS and
          all(lib.sg2im_gconv_stack_supported(int(a), int(b), int(c)) for a, b, c in dims))


def gconv_stack_struct(obj_vecs, pred_vecs, s_idx, o_idx, csr, avg, weights, acts):
  """weights: per layer (W1a, b1a, W1b, b1b, W2a, b2a, W2b, b2b); acts: per layer (h1, new_t, pooled, h2, new_obj)"""
  S = GconvStack()
  po, ldo = rows_ld(obj_vecs)
  T = pred_vecs.size(0)
  S.obj_vecs, S.ld_obj = po.value, ldo
  if T > 0:
    pp, ldp = rows_ld(pred_vecs)
    S.pred_vecs, S.ld_pred = pp.value, ldp
    S.s_idx, S.o_idx = _i64(s_idx).value, _i64(o_idx).value
  S.row_ptr, S.entries = csr.row_ptr.data_ptr(), csr.entries.data_ptr()
  S.n_objs, S.n_triples, S.n_layers, S.average = obj_vecs.size(0), T, len(weights), int(bool(avg))
  flops = 0.0
  for l, (w, a) in enumerate(zip(weights, acts)):
    L = S.layer[l]
    for k, t in zip(('w1a', 'b1a', 'w1b', 'b1b', 'w2a', 'b2a', 'w2b', 'b2b'), w):
      setattr(L, k, _f(t).value if t is not None else None)
    for k, t in zip(('h1', 'new_t', 'pooled', 'h2', 'new_obj'), a):
      setattr(L, k, _f(t).value if (t is not None and t.numel() > 0) else None)
    L.din, L.hidden, L.dout = w[0].size(1) // 3, w[4].size(0), w[6].size(0)
    flops += 2.0 * (T * (3 * L.din * L.hidden + L.hidden * (2 * L.hidden + L.dout)) + S.n_objs * (L.hidden * L.hidden + L.hidden * L.dout))
  S._keep = (obj_vecs, pred_vecs, s_idx, o_idx, csr, weights, acts)
  S._flops = flops
  return S


def gconv_stack_forward(S, device):
  """the whole GraphTripleConv stack, forward: ONE persistent launch (sg2im_gconv_stack_forward)"""
  sy = sync_area(device)
  _note_bytes('igemm_fwd', S._flops / 2.0 / max(S.n_triples + S.n_objs, 1))
  _timed('igemm_fwd', S._flops, lambda: call('sg2im_gconv_stack_forward', byref(S), c_void_p(sy.data_ptr()), sy.numel() * 4,
                                             _stream()))


def gconv_stack_backward(S, g_obj, g_pred, d_triple, d_obj, grads, accumulate, device):
  """the whole stack, backward: ONE persistent launch (sg2im_gconv_stack_backward).  grads: per layer 8 tensors or
  None in the order (dW1a, db1a, dW1b, db1b, dW2a, db2a, dW2b, db2b)"""
  G = GconvStackGrads()
  G.g_obj = _f(g_obj).value if g_obj is not None else None
  if g_pred is not None:
    pg, ldg = rows_ld(g_pred)
    G.g_pred, G.ld_gpred = pg.value, ldg
  G.d_triple = _f(d_triple).value
  G.d_obj = _f(d_obj).value if d_obj is not None else None
  need = int(_lib.load().sg2im_gconv_stack_backward_scratch(byref(S)))
  sc = scratch(device, need // 4 + 16)
  G.scratch, G.scratch_bytes = sc.data_ptr(), sc.numel() * 4
  for l, gl in enumerate(grads):
    for k, t in zip(('dw1a', 'db1a', 'dw1b', 'db1b', 'dw2a', 'db2a', 'dw2b', 'db2b'), gl):
      setattr(G.layer[l], k, _f(t).value if t is not None else None)
    G.layer[l].accumulate = int(bool(accumulate))
  G.low_footprint = {'low': 1, 'staged': 2, 'staged_full': 3}.get(GCN_PERSISTENT_BACKWARD, 0)
  sy = sync_area(device)
  _timed('igemm_dgrad', 2.0 * S._flops, lambda: call('sg2im_gconv_stack_backward', byref(S), byref(G), c_void_p(sy.data_ptr()),
                                                     sy.numel() * 4, _stream()))


def gconv_stack_check(device):
  """(debugging / tests; synchronises) raise if a grid barrier of the last persistent launch on this lane timed out"""
  sy = sync_area(device)
  host = sy.cpu().contiguous()
  if _lib.load().sg2im_gconv_stack_status(c_void_p(host.data_ptr())) != 0:
    raise _lib.Sg2imHipError('a grid barrier of the persistent GraphTripleConv kernel timed out (grid not resident?)')


def persistent_kernels_check():
  """Raise if ANY persistent launch on ANY lane of this process ever timed out in a grid barrier (the sticky word of
  every sync area, include/sg2im_hip.h).  A timed-out launch carries on with incomplete data - e.g. when another
  process shares the GPU and the grid is not fully resident - so its outputs are garbage; the per-launch error word
  is gone with the next launch's memset, the sticky one is not.  Costs one small device-to-host copy per lane: called
  where the host synchronises anyway (Trainer.losses_to_host, i.e. every --print_every iterations; bench.py after
  its timed loop)."""
  for key, sy in list(_sync_areas.items()):
    with torch.cuda.device(key[0]):
      host = sy.cpu().contiguous()
    st = int(_lib.load().sg2im_gconv_stack_status(c_void_p(host.data_ptr())))
    if st != 0:
      raise _lib.Sg2imHipError('a grid barrier of a persistent GraphTripleConv launch timed out (%d timed-out spins so far on lane %s): '
                               'its grid was not fully resident - another process on this GPU, or a second persistent kernel in '
                               'flight - and every result since is suspect.  SG2IM_GCN_PERSIST=0 selects the layer-by-layer launches.'
                               % (st >> 1, key))


def gconv_stack_stamps(device):
  """(diagnostics; synchronises) microseconds since kernel start of workgroup 0's stamps of the last persistent launch on
  this lane: [start = 0, before barrier 1, after barrier 1, ..., end]"""
  import ctypes
  host = sync_area(device).cpu().contiguous()
  out = (ctypes.c_ulonglong * 256)()
  n = _lib.load().sg2im_gconv_stack_stamps(c_void_p(host.data_ptr()), ctypes.cast(out, c_void_p), 256)
  return [(out[i] - out[0]) / 100.0 for i in range(n)]


def copy_2d(src, out):
  ps, lds = rows_ld(src)
  po, ldo = rows_ld(out)
  call('sg2im_copy_2d', ps, lds, po, ldo, src.size(0), src.size(1), _stream())
  return out


# ----------------------------------------------------------------------------
# layout / crops
# ----------------------------------------------------------------------------

def _mask_args(masks):
  if masks is None:
    return None, None, 0
  if masks.dtype == torch.int64:
    return None, _i64(masks.contiguous()), masks.size(1)
  return _f(masks.contiguous()), None, masks.size(1)


def layout_forward(vecs, boxes, masks, img_csr, n_images, H, W, align_corners, out):
  """out: NHWC (N,H,W,ld) tensor; channels [0, D) are written."""
  pv, ldv = rows_ld(vecs)
  mf, mi, M = _mask_args(masks)
  O, D = vecs.size(0), vecs.size(1)
  # algorithmic bytes: the layout written once + vectors, boxes and masks read once
  nbytes = 4.0 * (n_images * H * W * D + O * (D + 4 + M * M))
  _timed('hbm_layout_fwd', nbytes, lambda: call(
    'sg2im_layout_forward', pv, ldv, _f(boxes), mf, mi, M, _i32(img_csr.row_ptr), _i32(img_csr.entries),
    int(n_images), O, D, int(H), int(W), int(align_corners), _f(out), out.size(3), _stream()))
  return out


LAYOUT_PYRAMID = os.environ.get('SG2IM_LAYOUT_PYRAMID', '1') != '0'   # A/B knob: fused layout + noise + pyramid


def layout_pyramid_forward(vecs, boxes, masks, img_csr, n_images, H, W, align_corners, noise, levels):
  """levels[0]: NHWC (N,H,W,D+nd) full resolution; levels[l]: (N, H>>l, W>>l, D+nd) - all written."""
  pv, ldv = rows_ld(vecs)
  mf, mi, M = _mask_args(masks)
  O, D = vecs.size(0), vecs.size(1)
  nd = noise.size(1) if noise is not None else 0
  ptrs = (c_void_p * len(levels))(*[t.data_ptr() for t in levels])
  px = sum(t.size(1) * t.size(2) for t in levels)
  nbytes = 4.0 * (n_images * px * (D + nd) + n_images * H * W * nd + O * (D + 4 + M * M))
  _timed('hbm_layout_fwd', nbytes, lambda: call(
    'sg2im_layout_pyramid_forward', pv, ldv, _f(boxes), mf, mi, M, _i32(img_csr.row_ptr), _i32(img_csr.entries),
    int(n_images), D, _f(noise) if noise is not None else None, nd, int(H), int(W), int(align_corners),
    len(levels) - 1, ptrs, levels[0].size(3), _stream()))
  return levels


def layout_backward(dlayout, vecs, boxes, masks, obj_to_img, img_csr, n_images, H, W, align_corners,
                    d_vecs, d_masks, d_boxes=None):
  pv, ldv = rows_ld(vecs)
  mf, mi, M = _mask_args(masks)
  O, D = vecs.size(0), vecs.size(1)
  need = _lib.load().sg2im_layout_backward_workspace(O, D, int(H), int(W))
  ws = workspace(vecs.device, need)
  pd, ldd = rows_ld(d_vecs) if d_vecs is not None else (None, 0)
  call('sg2im_layout_backward', _f(dlayout), dlayout.size(3), pv, ldv, _f(boxes), mf, mi, M, _i64(obj_to_img),
       _i32(img_csr.row_ptr), _i32(img_csr.entries), int(n_images), O, D, int(H), int(W), int(align_corners),
       pd, ldd, _f(d_masks), _f(d_boxes), _f(ws), _stream())


def layout_backward_vecs_levels(levels, factors, vecs, boxes, masks, img_csr, n_images, H, W, align_corners, d_vecs):
  """d_vecs of the layout straight from the per-level layout gradients (sg2im_layout_backward_vecs_levels)"""
  mf, mi, M = _mask_args(masks)
  O, D = vecs.size(0), vecs.size(1)
  need = _lib.load().sg2im_layout_backward_workspace(O, D, int(H), int(W))
  ws = workspace(vecs.device, need)
  n = len(levels)
  ptrs = (c_void_p * n)(*[t.data_ptr() for t in levels])
  fs = (c_int * n)(*factors)
  ls = (c_longlong * n)(*[t.size(3) for t in levels])
  pd, ldd = rows_ld(d_vecs)
  nbytes = 4.0 * (sum(t.size(0) * t.size(1) * t.size(2) * D for t in levels) + 2 * O * D)
  _timed('hbm_layout_bwd_vecs', nbytes, lambda: call(
    'sg2im_layout_backward_vecs_levels', ptrs, fs, ls, n, _f(boxes), mf, mi, M, _i32(img_csr.row_ptr), _i32(img_csr.entries),
    int(n_images), O, D, int(H), int(W), int(align_corners), pd, ldd, _f(ws), _stream()))


def layout_backward_maps_levels(levels, factors, vecs, boxes, masks, img_csr, n_images, H, W, align_corners, d_masks, d_boxes):
  """d_masks / d_boxes of the layout straight from the per-level layout gradients (sg2im_layout_backward_maps_levels);
  False (nothing launched) when the shapes do not qualify - the caller then materialises the gradient"""
  O, D = vecs.size(0), vecs.size(1)
  if D % 4 or D > 128 or vecs.stride(1) != 1 or vecs.stride(0) % 4 or vecs.data_ptr() % 16:
    return False
  mf, mi, M = _mask_args(masks)
  need = _lib.load().sg2im_layout_backward_workspace(O, D, int(H), int(W))
  ws = workspace(vecs.device, need)
  n = len(levels)
  ptrs = (c_void_p * n)(*[t.data_ptr() for t in levels])
  fs = (c_int * n)(*factors)
  ls = (c_longlong * n)(*[t.size(3) for t in levels])
  call('sg2im_layout_backward_maps_levels', ptrs, fs, ls, n, _f(vecs), vecs.stride(0), _f(boxes), mf, mi, M,
       _i32(img_csr.row_ptr), _i32(img_csr.entries), int(n_images), O, D, int(H), int(W), int(align_corners),
       _f(d_masks), _f(d_boxes), _f(ws), _stream())
  return True


def crop_forward(imgs_nhwc, boxes, obj_to_img, size, align_corners, out):
  N, H, W, C = imgs_nhwc.shape
  call('sg2im_crop_forward', _f(imgs_nhwc), C, N, H, W, C, _f(boxes), _i64(obj_to_img), boxes.size(0), int(size),
       int(align_corners), _f(out), _stream())
  return out


def crop_backward(d_crops, boxes, obj_to_img, size, align_corners, d_imgs_nhwc):
  N, H, W, C = d_imgs_nhwc.shape
  O = boxes.size(0)
  ws = scratch(d_crops.device, max(1, O * H * W * C))        # per-object partial planes
  call('sg2im_crop_backward', _f(d_crops), N, H, W, C, _f(boxes), _i64(obj_to_img), O, int(size),
       int(align_corners), _f(d_imgs_nhwc), C, _f(ws), _stream())
  return d_imgs_nhwc


# ----------------------------------------------------------------------------
# batch norm / pooling / conversions
# ----------------------------------------------------------------------------

class BnState(object):
  """Per-call BatchNorm statistics + the folded affine the next conv loader applies."""
  __slots__ = ('mean', 'invstd', 'scale', 'shift')

  def __init__(self, C, device):
    buf = torch.empty(4, C, dtype=torch.float32, device=device)
    self.mean, self.invstd, self.scale, self.shift = buf[0], buf[1], buf[2], buf[3]


def _count_args(count):
  """count: None or (int32 device tensor [1], unit): only the first count[0] * unit rows / elements of
  a padded batch are real (sg2im_amd/bucketing.py).  -> (pointer, unit) for the C ABI."""
  if count is None:
    return None, 1
  t, unit = count
  if not (t.is_cuda and t.dtype == torch.int32):
    raise TypeError('row counts must be int32 tensors on the GPU')
  return c_void_p(t.data_ptr()), int(unit)


def bn_stats(x, rows, C, ld, bn, training, eps=1e-5, momentum=0.1, unbiased_rows=0, count=None):
  """x: any float tensor viewed as [rows][ld]; bn: a module holding weight/bias/running_*."""
  st = BnState(C, x.device)
  cp, cu = _count_args(count)
  part = scratch(x.device, 2 * C * 1024)
  nbt = bn.num_batches_tracked
  call('sg2im_bn_stats', _f(x), int(rows), int(C), int(ld), _f(bn.weight), _f(bn.bias), float(eps),
       float(momentum), int(training), _f(bn.running_mean), _f(bn.running_var),
       c_void_p(nbt.data_ptr()) if nbt is not None else None, int(unbiased_rows), _f(st.mean), _f(st.invstd), _f(st.scale),
       _f(st.shift), _f(part), cp, cu, _stream())
  return st


def _ptr(t):
  return t.data_ptr() if t is not None else None


def conv2d_forward_bn(desc, weight, cout, bias, out, ld_out, bn, training, eps=1e-5, momentum=0.1, out_slope=1.0,
                      unbiased_rows=0, count=None):
  """conv2d_forward followed by bn_stats of its output, with the statistics' reductions riding in the
  convolution's own launches (sg2im_conv2d_forward_bn).  Returns the BnState."""
  _set_mirror(desc, weight)
  ws = workspace(out.device)
  st = BnState(cout, out.device)
  M = desc.batch * desc.out_h * desc.out_w
  nfl = max(3 * cout * ((M + 63) // 64), 3 * cout * min((M + 7) // 8, 2048), 2 * cout * 1024)
  part = scratch(out.device, nfl)
  a = BnFwd()
  a.gamma, a.beta = _ptr(bn.weight), _ptr(bn.bias)
  a.eps, a.momentum, a.training = float(eps), float(momentum), int(training)
  a.running_mean, a.running_var = _ptr(bn.running_mean), _ptr(bn.running_var)
  a.num_batches_tracked = _ptr(bn.num_batches_tracked)
  a.unbiased_rows = int(unbiased_rows)
  a.mean, a.invstd, a.scale, a.shift = st.mean.data_ptr(), st.invstd.data_ptr(), st.scale.data_ptr(), st.shift.data_ptr()
  a.partial, a.partial_floats = part.data_ptr(), part.numel()
  cp, cu = _count_args(count)
  a.count, a.count_unit = (cp.value if cp is not None else None), cu
  flops = 2.0 * M * cout * _desc_k(desc)
  _note_bytes('igemm_fwd', _desc_src_floats(desc) + cout * _desc_k(desc) + M * cout)
  desc.out_dtype = _dt(out)
  _timed('igemm_fwd', flops, lambda: call(
    'sg2im_conv2d_forward_bn', byref(desc), _f(weight), int(cout), _f(bias), float(out_slope), _fb(out), int(ld_out),
    _f(ws), ws.numel() * 4, byref(a), _stream()), bn_finish=True,
    # (3 partial planes per 128-row tile and channel read once; mean / invstd / scale / shift + 2 running statistics written)
    finish_bytes=4.0 * (3 * cout * ((M + 127) // 128) + 6 * cout))
  return st


def conv2d_backward_data_bn(desc, weight, cout, dy, ld_dy, c_begin, c_count, dx, ld_dx, y, ld_y, pool2, gamma, st,
                            slope, training, dgamma, dbeta, accumulate=False, count=None):
  """conv2d_backward_data whose result dx is the gradient w.r.t. the ACTIVATED output of a BatchNorm'd layer
  (pre-norm output y, statistics st), together with that BatchNorm's backward reductions
  (sg2im_conv2d_backward_data_bn).  Returns the coefficient tensor for bn_backward_apply."""
  _set_mirror(desc, weight)
  ws = workspace(dx.device)
  rows_dx = desc.batch * desc.in_h * desc.in_w
  nfl = max(2 * c_count * ((rows_dx + 63) // 64), 2 * c_count * min((rows_dx + 7) // 8, 2048), 2 * c_count * 1024)
  part = scratch(dx.device, nfl)
  coef = torch.empty(3 * c_count, dtype=torch.float32, device=dx.device)
  a = BnBwd()
  a.y, a.ld_y, a.pool2 = y.data_ptr(), int(ld_y), int(pool2)
  a.y_dtype = _dt(y)
  desc.dy_dtype, desc.out_dtype = _dt(dy), _dt(dx)
  a.gamma = _ptr(gamma)
  a.mean, a.invstd, a.scale, a.shift = st.mean.data_ptr(), st.invstd.data_ptr(), st.scale.data_ptr(), st.shift.data_ptr()
  a.slope, a.training = float(slope), int(training)
  a.dgamma, a.dbeta, a.accumulate = _ptr(dgamma), _ptr(dbeta), int(accumulate)
  a.coef, a.partial, a.partial_floats = coef.data_ptr(), part.data_ptr(), part.numel()
  cp, cu = _count_args(count)
  a.count, a.count_unit = (cp.value if cp is not None else None), cu
  flops = 2.0 * desc.batch * desc.out_h * desc.out_w * cout * desc.kh * desc.kw * c_count
  _note_bytes('igemm_dgrad', desc.batch * desc.out_h * desc.out_w * cout + cout * desc.kh * desc.kw * c_count +
              rows_dx * c_count)
  _timed('igemm_dgrad', flops, lambda: call(
    'sg2im_conv2d_backward_data_bn', byref(desc), _f(weight), int(cout), _fb(dy), int(ld_dy), int(c_begin), int(c_count),
    _fb(dx), int(ld_dx), _f(ws), ws.numel() * 4, byref(a), _stream()), bn_finish=True,
    finish_bytes=4.0 * (2 * c_count * ((rows_dx + 127) // 128) + 5 * c_count))     # (2 planes read; coef[3] + dgamma + dbeta written)
  return coef


def bn_backward_apply(g, ld_g, pool2, batch, h, w, y, ld_y, C, st, slope, coef, dy, count=None, g_dtype=0):
  """dy = coef[0] * du + coef[1] * y + coef[2] (the third pass of bn_act_backward; g is a raw pointer, g_dtype its
  storage type).  Any of g / y / dy in bfloat16 storage: sg2im_bn_backward_apply_ex."""
  if g_dtype or _dt(y) or _dt(dy):
    if count is not None:
      raise ValueError('bfloat16 storage has no padded-batch form of the BatchNorm backward')
    call('sg2im_bn_backward_apply_ex', g, int(ld_g), int(pool2), int(batch), int(h), int(w), _fb(y), int(ld_y), int(C),
         _f(st.scale), _f(st.shift), float(slope), _f(coef), _fb(dy), int(g_dtype), _dt(y), _dt(dy), _stream())
    return dy
  cp, cu = _count_args(count)
  call('sg2im_bn_backward_apply', g, int(ld_g), int(pool2), int(batch), int(h), int(w), _f(y), int(ld_y), int(C),
       _f(st.scale), _f(st.shift), float(slope), _f(coef), _f(dy), cp, cu, _stream())
  return dy


def bn_act_backward(g, ld_g, pool2, batch, h, w, y, ld_y, C, gamma, st, slope, training, dy, dgamma, dbeta,
                    accumulate=False, count=None):
  part = scratch(y.device, 2 * C * 1024 + 3 * C)
  cp, cu = _count_args(count)
  call('sg2im_bn_act_backward', g, int(ld_g), int(pool2), int(batch), int(h), int(w), _f(y), int(ld_y), int(C),
       _f(gamma), _f(st.mean), _f(st.invstd), _f(st.scale), _f(st.shift), float(slope), int(training), _f(dy),
       _f(dgamma), _f(dbeta), int(accumulate), _f(part), cp, cu, _stream())
  return dy


def affine_act_forward(x, st, slope, out):
  """out = leaky_slope(st.scale * x + st.shift) for row matrices"""
  px, ldx = rows_ld(x)
  po, ldo = rows_ld(out)
  call('sg2im_affine_act_forward', px, ldx, x.size(0), x.size(1), _f(st.scale), _f(st.shift), float(slope), po, ldo,
       _stream())
  return out


def resample_up(x, factor, alpha, out):
  """out (N, Ho, Wo, C) = alpha * nearest-upsample_factor(x), zero beyond the scaled input"""
  N, H, W, C = x.shape
  call('sg2im_resample_nearest_up', _f(x), N, H, W, C, int(factor), out.size(1), out.size(2), float(alpha), _f(out),
       _stream())
  return out


def pool_sum(x, factor, alpha, out):
  N, H, W, C = x.shape
  call('sg2im_pool_sum_forward', _f(x), N, H, W, C, int(factor), float(alpha), _f(out), _stream())
  return out


def maxpool_forward(x, factor, out):
  N, H, W, C = x.shape
  call('sg2im_maxpool_forward', _f(x), N, H, W, C, int(factor), _f(out), _stream())
  return out


def maxpool_backward(x, dy, factor, dx):
  N, H, W, C = x.shape
  call('sg2im_maxpool_backward', _f(x), _f(dy), N, H, W, C, int(factor), _f(dx), _stream())
  return dx


def leaky_forward(x, slope, out):
  call('sg2im_leaky_forward', _f(x), x.numel(), float(slope), _f(out), _stream())
  return out


def add_forward(a, b, out):
  call('sg2im_add_forward', _f(a), _f(b), a.numel(), _f(out), _stream())
  return out


def instnorm_stats(x, eps=1e-5):
  """x: dense NHWC; returns (scale, shift), each (N, C): the InstanceNorm2d affine of every image"""
  N, H, W, C = x.shape
  st = torch.empty(2, N, C, dtype=torch.float32, device=x.device)
  call('sg2im_instnorm_stats', _f(x), N, H * W, C, float(eps), _f(st[0]), _f(st[1]), _stream())
  return st


def instnorm_act_forward(x, st, slope, out):
  N, H, W, C = x.shape
  call('sg2im_instnorm_act_forward', _f(x), N, H * W, C, _f(st[0]), _f(st[1]), float(slope), _f(out), _stream())
  return out


def instnorm_backward(dyn, x, st, dx):
  N, H, W, C = x.shape
  call('sg2im_instnorm_backward', _f(dyn), _f(x), N, H * W, C, _f(st[0]), _f(st[1]), _f(dx), _stream())
  return dx


def act_backward(g, ld_g, pool2, batch, h, w, y, ld_y, C, slope, dx):
  call('sg2im_act_backward', g, int(ld_g), int(pool2), int(batch), int(h), int(w), _f(y), int(ld_y), int(C),
       float(slope), _f(dx), _stream())
  return dx


def avgpool_forward(x, factor, out):
  N, H, W, C = x.shape
  call('sg2im_avgpool_forward', _f(x), N, H, W, C, int(factor), _f(out), _stream())
  return out


def pyramid_backward(levels, factors, lds, batch, H, W, C, out):
  n = len(levels)
  ptrs = (c_void_p * n)(*[t.data_ptr() for t in levels])
  fs = (c_int * n)(*factors)
  ls = (c_longlong * n)(*lds)
  call('sg2im_pyramid_backward', ptrs, fs, ls, n, int(batch), int(H), int(W), int(C), _f(out), out.size(3),
       _stream())
  return out


def nchw_to_nhwc(src, out, c_offset=0):
  N, C, H, W = src.shape
  call('sg2im_nchw_to_nhwc', _f(src), N, C, H, W, _f(out), out.size(3), int(c_offset), _stream())
  return out


def nhwc_to_nchw(src, out, c_offset=0):
  N, C, H, W = out.shape
  call('sg2im_nhwc_to_nchw', _f(src), src.size(3), int(c_offset), N, C, H, W, _f(out), _stream())
  return out


def gap_forward(x, out):
  N, H, W, C = x.shape
  call('sg2im_gap_forward', _f(x), N, H * W, C, _f(out), _stream())
  return out


def gap_backward(dout, hw, dx):
  N, C = dout.shape
  call('sg2im_gap_backward', _f(dout), N, int(hw), C, _f(dx), _stream())
  return dx


def sigmoid_forward(x, out):
  call('sg2im_sigmoid_forward', _f(x), x.numel(), _f(out), _stream())
  return out


def sigmoid_backward(y, dy, dx):
  call('sg2im_sigmoid_backward', _f(y), _f(dy), y.numel(), _f(dx), _stream())
  return dx


# ----------------------------------------------------------------------------
# losses / optimiser
# ----------------------------------------------------------------------------

def _loss_out(device):
  return torch.empty(1, dtype=torch.float32, device=device)


def l1_loss(pred, target, weight, grad):
  loss = _loss_out(pred.device)
  call('sg2im_l1_loss', _f(pred), _f(target), pred.numel(), float(weight), _f(loss), _f(grad),
       _f(scratch(pred.device, 256)), _stream())
  return loss


def mse_loss(pred, target, weight, grad, count=None):
  loss = _loss_out(pred.device)
  cp, cu = _count_args(count)
  call('sg2im_mse_loss', _f(pred), _f(target), pred.numel(), float(weight), _f(loss), _f(grad),
       _f(scratch(pred.device, 256)), cp, cu, _stream())
  return loss


def bce_logits_loss(x, target, weight, grad, count=None):
  loss = _loss_out(x.device)
  cp, cu = _count_args(count)
  call('sg2im_bce_logits_loss', _f(x), x.numel(), float(target), float(weight), _f(loss), _f(grad),
       _f(scratch(x.device, 256)), cp, cu, _stream())
  return loss


def gan_score_loss(x, kind, target, weight, grad, count=None):
  loss = _loss_out(x.device)
  cp, cu = _count_args(count)
  call('sg2im_gan_score_loss', _f(x), x.numel(), int(kind), float(target), float(weight), _f(loss), _f(grad),
       _f(scratch(x.device, 256)), cp, cu, _stream())
  return loss


def bce_prob_loss(prob, target, weight, grad, count=None):
  loss = _loss_out(prob.device)
  cp, cu = _count_args(count)
  call('sg2im_bce_prob_loss', _f(prob), _f(target), prob.numel(), float(weight), _f(loss), _f(grad),
       _f(scratch(prob.device, 256)), cp, cu, _stream())
  return loss


def cross_entropy_loss(scores, labels, weight, grad, count=None):
  loss = _loss_out(scores.device)
  R, C = scores.shape
  cp, cu = _count_args(count)
  call('sg2im_cross_entropy_loss', _f(scores), R, C, _i64(labels), float(weight), _f(loss), _f(grad),
       _f(scratch(scores.device, max(256, R))), cp, cu, _stream())
  return loss


_units = {}


def unit(device):
  """the cached 0-dim 1.0 a Trainer seeds ``backward`` with: a loss Function that sees exactly this
  tensor as its upstream gradient hands its stored gradient on without a scaling launch"""
  idx = device.index if device.index is not None else torch.cuda.current_device()
  u = _units.get(idx)
  if u is None:
    u = _units[idx] = torch.ones((), dtype=torch.float32, device=torch.device('cuda', idx))
  return u


def is_unit(g):
  u = _units.get(g.device.index)
  return u is not None and g.data_ptr() == u.data_ptr()


def sum_scalars(terms, out):
  arr = (c_void_p * len(terms))(*[t.data_ptr() for t in terms])
  for t in terms:
    _f(t)                              # (type / device check)
  call('sg2im_sum_scalars', arr, len(terms), _f(out), _stream())
  return out


def scale_by_scalar(x, a_dev, out):
  call('sg2im_scale_by_scalar', _f(x), _f(a_dev), x.numel(), _f(out), _stream())
  return out


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, grad_scale=1.0):
  call('sg2im_adam_step', _f(param), _f(grad), _f(exp_avg), _f(exp_avg_sq), param.numel(), float(lr),
       float(beta1), float(beta2), float(eps), int(step), float(grad_scale), _stream())


def adam_prepare_guarded(lr, beta1, beta2, state, guard):
  call('sg2im_adam_prepare_guarded', float(lr), float(beta1), float(beta2), _f(state), _f(guard), _stream())


def adam_apply_guarded(param, grad, exp_avg, exp_avg_sq, beta1, beta2, eps, state, grad_scale=1.0):
  """the update of a slice of the arena (after adam_prepare_guarded of the same step)"""
  _timed('hbm_adam', 28.0 * param.numel(), lambda: call(
    'sg2im_adam_apply_guarded', _f(param), _f(grad), _f(exp_avg), _f(exp_avg_sq), param.numel(), float(beta1), float(beta2),
    float(eps), float(grad_scale), _f(state), _stream()))


def adam_step_guarded(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, state, guard, grad_scale=1.0):
  """Adam with the step counter in ``state`` (float[4] on the device); skipped entirely
  when ``guard`` (a device scalar, e.g. the generator loss) is not finite."""
  # algorithmic bytes: p, g, m, v read + p, m, v written
  _timed('hbm_adam', 28.0 * param.numel(), lambda: call(
    'sg2im_adam_step_guarded', _f(param), _f(grad), _f(exp_avg), _f(exp_avg_sq), param.numel(), float(lr),
    float(beta1), float(beta2), float(eps), float(grad_scale), _f(state), _f(guard), _stream()))
