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



class GradReducer(object):
  """Sum-reduces flat gradient arenas across ranks, asynchronously."""

  def __init__(self, world_size=None, group=None, force=False, payload='f32'):
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
    if payload not in ('f32', 'bf16'):
      raise ValueError('payload must be "f32" or "bf16"')
    self.payload = payload
    self._staging = {}               # arena data_ptr -> bfloat16 buffer of the arena's size (allocated once)
    # test instrumentation (tests/test_gpu_parity.py::test_in_graph_exchange_reduces_every_gradient_exactly_once): a
    # 1-rank SUM is the identity, so a gradient slice that is reduced twice, never, or BEFORE its last writer ran
    # would go unnoticed on a one-GPU box.  With test_gain = g every reduction is followed by an in-place x g on
    # the same stream and grad_scale becomes 1/g: the arena x grad_scale equals the plain gradient exactly
    # (g a power of two) if and only if every element went through exactly one reduction after it was complete.
    self.test_gain = None

  @property
  def grad_scale(self):
    if self.test_gain is not None:
      return 1.0 / (self.world_size * self.test_gain)
    return 1.0 / self.world_size

  def capturable(self):
    """can the collectives be recorded into a hipGraph?  RCCL: yes (tools/rccl_capture_probe.py,
    profiles/r3_rccl_capture_probe.log); the host-staged gloo form: no"""
    return (self.world_size > 1 or self.force) and dist.is_initialized() and dist.get_backend(self.group) == 'nccl'

  def reduce_here(self, tensor):
    """SUM all-reduce of ``tensor`` ordered on the CURRENT stream (which waits for it; the collective itself
    runs on the process group's own stream) - the form that is recorded into a stream capture"""
    if (self.world_size > 1 or self.force) and not self.mute:
      if self.payload == 'bf16' and tensor.numel() > 1:
        buf = self.staging(tensor)
        buf.copy_(tensor)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        tensor.copy_(buf)
      else:
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group)
      if self.test_gain is not None:
        tensor.mul_(self.test_gain)

  # The bfloat16 payload inside a captured iteration, in three steps on three streams, so that the comm stream carries
  # nothing but the collective (an elementwise kernel on it re-maps the graph's branches onto the hardware queues:
  # 4.69 -> 6.45 ms, profiles/r4_bf16_payload_ab.txt): pack (the PRODUCER's stream: round the finished gradients
  # into the staging buffer), reduce_packed (comm stream), unpack (the CONSUMER's stream, before Adam).
  def pack(self, tensor):
    self.staging(tensor).copy_(tensor)

  def reduce_packed(self, tensor):
    if (self.world_size > 1 or self.force) and not self.mute:
      buf = self.staging(tensor)
      dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
      if self.test_gain is not None:
        buf.mul_(self.test_gain)

  def unpack(self, tensor):
    tensor.copy_(self.staging(tensor))

  def packs(self, tensor):
    """does this tensor travel through the staging buffer?"""
    return self.payload == 'bf16' and tensor.numel() > 1

  def staging(self, tensor):
    """the bfloat16 staging buffer of an arena (slice): one per distinct (address, size), allocated at first use -
    for a captured iteration that is inside the capture, i.e. in the graph's private pool, like every other tensor
    the graph creates"""
    key = (tensor.data_ptr(), tensor.numel())
    buf = self._staging.get(key)
    if buf is None:
      buf = torch.empty(tensor.numel(), dtype=torch.bfloat16, device=tensor.device)
      self._staging[key] = buf
    return buf

  def start(self, tensor):
    """begin an all-reduce (SUM) of ``tensor`` in place; returns immediately"""
    if (self.world_size > 1 or self.force) and not self.mute:
      if self.payload == 'bf16' and tensor.numel() > 1:
        buf = self.staging(tensor)
        buf.copy_(tensor)
        h = all_reduce_sum_async(buf, self.group)
        h.wait()
        tensor.copy_(buf)
        h = _Done()
      else:
        h = all_reduce_sum_async(tensor, self.group)
      if self.test_gain is not None:
        h.wait()
        tensor.mul_(self.test_gain)
        h = _Done()
      self.pending.append(h)

  def finish(self):
    """make the current stream wait for every started reduction"""
    for h in self.pending:
      h.wait()
    self.pending = []
