"""Data parallelism over whole images: one process per GPU, RCCL
(torch.distributed backend "nccl") over xGMI.

The reference has no distributed code at all.  The exchange here is deliberately minimal:
each network keeps all parameter gradients in ONE flat fp32 arena (sg2im_amd.optim), so a
step needs exactly one SUM all-reduce per network (generator 112.6 MB, D_obj 4.4 MB, D_img
2.6 MB) plus a 1-element reduce of the NaN guard.  Default (Trainer dp_schedule 2): the collectives are
RECORDED INSIDE the captured iteration on a comm stream, each as soon as its gradients are complete - the
discriminators' right after their steps, the generator's in two buckets (the first two refinement modules'
weight gradients, ~2/3 of the bytes, travel under the remaining weight gradients; the rest after the backward
pass) - and the Adam updates wait for the comm stream (sg2im_amd/trainer.py::_capture_overlapped).  Eager mode
and schedules 0 / 1 start the all-reduces asynchronously between (segments of) the step and wait before Adam.
The 1/world_size factor is folded into the fused Adam kernel (``grad_scale``), not a separate
pass over the arena.  Replicas are brought in line by a broadcast of parameters, optimiser
moments and BatchNorm buffers at construction, after a checkpoint restore and after rank 0's
validation pass (Trainer.broadcast_state, scripts/train.py); within a training step the BatchNorm
batch statistics are per replica (the reference's batch-32 semantics on every rank) and the running
statistics drift apart between two broadcasts - a checkpoint holds rank 0's.

Two forms of the exchange (GradReducer(exchange=...), SG2IM_DP_EXCHANGE): 'allreduce' - one library SUM all-reduce per
arena (slice), RCCL's choice of ring / tree - and 'direct' - the reduce-scatter / all-gather over ALL links that
SURVEY.md section 5 specifies, spelled out: an all-to-all of the W shards of the arena (every rank sends 1/W of its
gradients to every peer: 7 concurrent xGMI links on an 8-GPU node instead of one ring direction), a LOCAL sum of the W
received shards in fp32 (also with the bfloat16 payload: one rounding of each shard and one of the finished sum,
instead of W - 1 roundings of a bfloat16 running sum inside the library's ring), an all-gather of the reduced shards.
Every rank ends up with bit-identical arenas.  'direct' has never run on more than one GPU (1-rank RCCL and 2-rank gloo
tests); it is opt-in, eager / schedule 0 / 1 only (the in-graph schedule keeps the comm stream free of elementwise
kernels, which the local sum is).

Gradient semantics (tested in tests/test_dp_gloo.py): every rank computes the reference's
per-shard loss (means over ITS objects / pixels), so the applied gradient is the mean over
ranks of the per-shard gradients - not the gradient of the concatenated batch.
"""
import torch
import torch.distributed as dist


def _host_staged(tensor, group=None):
  """gloo has no device collectives in this build: a GPU tensor is reduced / broadcast through a host
  copy.  Only used when the process group's backend is gloo - the single-GPU test of the data-parallel
  schedules (two ranks sharing one MI355X, tests/test_dp_rccl.py); RCCL ("nccl") works on the arena
  in place."""
  return tensor.is_cuda and dist.get_backend(group) == 'gloo'


def broadcast(tensor, src=0, group=None):
  if _host_staged(tensor, group):
    host = tensor.detach().cpu()
    dist.broadcast(host, src, group=group)
    tensor.copy_(host)
  else:
    dist.broadcast(tensor, src, group=group)


class _Done(object):
  def wait(self):
    pass


def all_reduce_sum_async(tensor, group=None):
  """in-place SUM all-reduce; returns a handle with ``wait()``"""
  if _host_staged(tensor, group):
    host = tensor.detach().cpu()
    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
    tensor.copy_(host)
    return _Done()
  return dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group, async_op=True)



class _Widen(object):
  """handle of a bfloat16-payload reduction started by GradReducer.start: ``wait()`` makes the current stream wait for
  the collective and only then widens the staging buffer back into the fp32 arena (the copy is deferred to
  GradReducer.finish, so the stream that started the exchange is not blocked behind it: the D_obj step keeps
  running while the generator's 60 MB are in flight - ADVICE r4)"""

  def __init__(self, handle, buf, tensor, after):
    self.handle, self.buf, self.tensor, self.after = handle, buf, tensor, after

  def wait(self):
    self.handle.wait()
    self.after(self.buf)
    self.tensor.copy_(self.buf)


class _Then(object):
  def __init__(self, handle, tensor, after):
    self.handle, self.tensor, self.after = handle, tensor, after

  def wait(self):
    self.handle.wait()
    self.after(self.tensor)


class _Join(object):
  """handle of a direct exchange started on the reducer's own stream"""

  def __init__(self, stream, device):
    self.stream, self.device = stream, device

  def wait(self):
    torch.cuda.current_stream(self.device).wait_stream(self.stream)


class GradReducer(object):
  """Sum-reduces flat gradient arenas across ranks, asynchronously."""

  def __init__(self, world_size=None, group=None, force=False, payload='f32', exchange='allreduce'):
    if world_size is None:
      world_size = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    self.world_size = world_size
    self.group = group
    self.pending = []
    # force: issue the collectives even in a 1-rank group (exercises the RCCL path on one GPU)
    self.force = force
    # mute: skip the collectives (while a graph is being captured; bench.py's "step without the
    # exchange" timing leg)
    self.mute = False
    # payload 'bf16' (the Trainer's choice for compute_dtype 'bf16', BASELINE configs[2..4]): the arena travels as
    # bfloat16 - rounded (RNE) into a staging buffer, summed by RCCL in bfloat16, widened back into the fp32 arena -
    # i.e. half the xGMI bytes (59.9 instead of 119.7 MB per step) for two extra elementwise passes over the arena;
    # Adam's moments and the parameters stay fp32.  'f32': the arena itself is reduced in place.
    # Accuracy of the bfloat16 SUM over ranks: tests/test_dp_gloo.py::test_bf16_sum_of_eight_bf16_shards (CPU).
    if payload not in ('f32', 'bf16'):
      raise ValueError('payload must be "f32" or "bf16"')
    self.payload = payload
    if exchange not in ('allreduce', 'direct'):
      raise ValueError('exchange must be "allreduce" or "direct"')
    self.exchange = exchange
    self._staging = {}               # (arena address, elements) -> bfloat16 buffer (allocated once, see staging())
    self._direct = {}                # (arena address, elements) -> the buffers of the direct exchange (see _direct_bufs)
    self._xstream = None             # the stream the direct exchange runs on (start ... finish)

  @property
  def grad_scale(self):
    return 1.0 / self.world_size

  def _reduced(self, tensor):
    """hook: called on the stream that carries a collective right after it was issued on ``tensor`` (the fp32 arena
    slice, or the bfloat16 staging buffer that travelled in its place).  Nothing here; tests/hip_harness.py's
    GainReducer uses it to prove on ONE GPU that every gradient element goes through exactly one reduction."""

  def live(self):
    return (self.world_size > 1 or self.force) and not self.mute

  def capturable(self):
    """can the collectives be recorded into a hipGraph?  RCCL all-reduces: yes (tools/rccl_capture_probe.py,
    profiles/r3_rccl_capture_probe.log); the host-staged gloo form: no; the direct exchange: not offered (its local
    sum would be an elementwise kernel on the comm stream - module docstring)"""
    return ((self.world_size > 1 or self.force) and dist.is_initialized() and dist.get_backend(self.group) == 'nccl' and
            self.exchange == 'allreduce')

  # ---- the direct exchange: all-to-all of shards, local fp32 sum, all-gather ------------------------------------
  def _direct_bufs(self, tensor):
    """(send, recv, reduced, out, n): payload-typed buffers of W * s elements (s = ceil(n / W); the pad is zero and
    never written), an fp32 + payload-typed reduced shard.  One set per distinct arena (slice), kept."""
    key = (tensor.data_ptr(), tensor.numel())
    bufs = self._direct.get(key)
    if bufs is None:
      W, n = self._group_size(), tensor.numel()
      s_ = -(-n // W)
      dt = torch.bfloat16 if (self.payload == 'bf16' and n > 1) else torch.float32
      dev = torch.device('cpu') if _host_staged(tensor, self.group) else tensor.device
      mk = lambda m, d: torch.zeros(m, dtype=d, device=dev)
      bufs = (mk(W * s_, dt), mk(W * s_, dt), mk(s_, torch.float32), mk(s_, dt), mk(W * s_, dt))
      self._direct[key] = bufs
    return bufs

  def _group_size(self):
    return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

  def _exchange_direct(self, tensor):
    """SUM over the ranks of ``tensor`` (in place), ordered on the current stream"""
    send, recv, red32, red, out = self._direct_bufs(tensor)
    W, n = self._group_size(), tensor.numel()
    send[:n].copy_(tensor)                                        # (rounds to bfloat16 with that payload)
    dist.all_to_all_single(recv, send, group=self.group)          # recv[i * s : (i + 1) * s] = rank i's shard `me`
    shards = recv.view(W, -1)
    red32.copy_(shards[0])                                        # fp32 accumulation in RANK ORDER, spelled out: the sum
    for r in range(1, W):                                         # is a defined function of the shards (a library
      red32.add_(shards[r])                                       # reduction's order is not), W - 1 small kernels
    if red.dtype == torch.float32:
      red = red32
    else:
      red.copy_(red32)                                            # (ONE rounding of the finished sum)
    dist.all_gather_into_tensor(out, red, group=self.group)
    self._reduced(out)
    tensor.copy_(out[:n])

  def reduce_here(self, tensor):
    """SUM all-reduce of ``tensor`` ordered on the CURRENT stream (which waits for it; the collective itself
    runs on the process group's own stream) - the form that is recorded into a stream capture"""
    if self.live():
      if self.exchange == 'direct':
        self._exchange_direct(tensor)
      elif self.payload == 'bf16' and tensor.numel() > 1:
        buf = self.staging(tensor)
        buf.copy_(tensor)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        self._reduced(buf)
        tensor.copy_(buf)
      else:
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group)
        self._reduced(tensor)

  # The bfloat16 payload inside a captured iteration, in three steps on three streams, so that the comm stream carries
  # nothing but the collective (an elementwise kernel on it re-maps the graph's branches onto the hardware queues:
  # 4.69 -> 6.45 ms, profiles/r4_bf16_payload_ab.txt): pack (the PRODUCER's stream: round the finished gradients
  # into the staging buffer), reduce_packed (comm stream), unpack (the CONSUMER's stream, before Adam).
  def pack(self, tensor):
    self.staging(tensor).copy_(tensor)

  def reduce_packed(self, tensor):
    if self.live():
      buf = self.staging(tensor)
      dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
      self._reduced(buf)

  def unpack(self, tensor):
    tensor.copy_(self.staging(tensor))

  def packs(self, tensor):
    """does this tensor travel through the staging buffer?"""
    return self.payload == 'bf16' and tensor.numel() > 1

  def staging(self, tensor):
    """the bfloat16 staging buffer of an arena (slice): one per distinct (address, elements), allocated at first use
    and kept for the reducer's lifetime.  The Trainer calls this for every arena and bucket slice in _prepare_lanes,
    i.e. OUTSIDE any stream capture: a buffer born inside one graph's private pool must not be written by the graph
    of another shape bucket.  ``drop_staging`` frees them (a Trainer that rebuilds its arenas)."""
    key = (tensor.data_ptr(), tensor.numel())
    buf = self._staging.get(key)
    if buf is None:
      if torch.cuda.is_available() and tensor.is_cuda and torch.cuda.is_current_stream_capturing():
        raise RuntimeError('GradReducer.staging: first use of a staging buffer inside a stream capture - create it '
                           'eagerly first (Trainer._prepare_lanes does)')
      buf = torch.empty(tensor.numel(), dtype=torch.bfloat16, device=tensor.device)
      self._staging[key] = buf
    return buf

  def drop_staging(self):
    self._staging = {}
    self._direct = {}

  def start(self, tensor):
    """begin an all-reduce (SUM) of ``tensor`` in place; returns immediately"""
    if self.live():
      if self.exchange == 'direct':
        # the whole exchange on a stream of the reducer's own, behind everything issued so far on the current one:
        # the caller keeps launching (the D_obj step while the generator's arena travels), finish() joins
        if tensor.is_cuda:
          if self._xstream is None:
            self._xstream = torch.cuda.Stream(device=tensor.device)
          self._xstream.wait_stream(torch.cuda.current_stream(tensor.device))
          with torch.cuda.stream(self._xstream):
            self._exchange_direct(tensor)
          self.pending.append(_Join(self._xstream, tensor.device))
        else:
          self._exchange_direct(tensor)
      elif self.payload == 'bf16' and tensor.numel() > 1:
        buf = self.staging(tensor)
        buf.copy_(tensor)
        self.pending.append(_Widen(all_reduce_sum_async(buf, self.group), buf, tensor, self._reduced))
      else:
        self.pending.append(_Then(all_reduce_sum_async(tensor, self.group), tensor, self._reduced))

  def finish(self):
    """make the current stream wait for every started reduction (and widen the bfloat16 payloads back)"""
    for h in self.pending:
      h.wait()
    self.pending = []
