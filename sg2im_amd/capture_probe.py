"""Can THIS software stack digest the stream pattern of the data-parallel iteration graph with the RCCL all-reduces
recorded inside it (Trainer dp_schedule 2)?  Asked in a SUBPROCESS, because the failure mode seen so far is not an
error code: hipStreamEndCapture segfaulted inside clr on one wait pattern (profiles/r5_dp_early_adam_capture_crash.txt).

  python -m sg2im_amd.capture_probe      exit status 0 = captured, replayed twice, values as expected

The child builds an RCCL group on the device SG2IM_PROBE_DEVICE - one rank, or, in a data-parallel job, one rank per
job rank: the children of all ranks form their own group (probe(world_size=...)) - and captures a small replica of the schedule-2
pattern - origin stream, a side stream (the discriminator steps), a lane forked later (the released weight gradients)
and a comm stream that only ever waits and carries the collectives, in the order the Trainer issues them - with
element-wise kernels in place of the real ones.  ``probe()`` runs it with a timeout and caches the verdict; the Trainer
falls back to dp_schedule 1 (graph segments with the exchanges issued between them - nothing of RCCL inside a capture)
when it fails (Trainer.__init__, choose_dp_schedule).  SG2IM_PROBE_FORCE_FAIL=1 makes the child abort (tests)."""
import os
import socket
import subprocess
import sys

_verdict = {}


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def group_port(master_port, offset=29):
  """the rendezvous port of the children's own process group: the same on every rank (derived from the parent group's
  port, which every rank knows), different from the parent's"""
  p = int(master_port) + offset
  return p if p < 65000 else int(master_port) - offset


def probe(device_index=0, timeout=180, rank=0, world_size=1, master_addr=None, master_port=None):
  """True when the child process captured and replayed the pattern; False on a non-zero exit status, a signal or a
  timeout.  Cached per (process, device).

  world_size > 1 (every rank of a data-parallel job calls this at the same point): the children of all ranks form
  THEIR OWN world_size-rank RCCL group (rendezvous on group_port(master_port)) and capture the pattern with real
  cross-rank all-reduces inside - the question the 1-rank form cannot answer: does a multi-rank collective recorded in a
  graph hang here?  A child that hangs is killed by the timeout and the job runs schedule 1; the parent group is never
  exposed to the experiment."""
  if device_index in _verdict:
    return _verdict[device_index]
  env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'GROUP_RANK', 'ROLE_RANK',
                                                          'LOCAL_WORLD_SIZE', 'TORCHELASTIC_RUN_ID')}
  if int(world_size) > 1:
    addr = master_addr or os.environ.get('MASTER_ADDR', '127.0.0.1')
    port = group_port(master_port if master_port is not None else os.environ.get('MASTER_PORT', '29500'))
    env.update(MASTER_ADDR=str(addr), MASTER_PORT=str(port), SG2IM_PROBE_RANK=str(int(rank)),
               SG2IM_PROBE_WORLD=str(int(world_size)))
  else:
    env.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    env.pop('SG2IM_PROBE_RANK', None); env.pop('SG2IM_PROBE_WORLD', None)
  env.update(SG2IM_PROBE_DEVICE=str(int(device_index)))
  env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env['PYTHONPATH'] = root + os.pathsep + env.get('PYTHONPATH', '')
  try:
    out = subprocess.run([sys.executable, '-m', 'sg2im_amd.capture_probe'], env=env, cwd=root, timeout=timeout,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    ok = out.returncode == 0
    tail = out.stdout.decode(errors='replace')[-400:]
  except subprocess.TimeoutExpired:
    ok, tail = False, 'timed out after %d s' % timeout
  if not ok:
    print('[sg2im_amd] in-graph RCCL capture probe FAILED on device %d: %s' % (device_index, tail.strip().replace('\n', ' | ')),
          flush=True)
  _verdict[device_index] = ok
  return ok


def choose_dp_schedule(requested, capturable, probe_fn, agree_fn=None):
  """The schedule a data-parallel Trainer runs: ``requested`` unless it is 2 (collectives inside the graph) and either
  the process group cannot be captured at all (-> 0, as before: gloo) or the subprocess probe fails on ANY rank (-> 1:
  the D_obj step still overlaps the generator's exchange, nothing of RCCL is captured).  agree_fn(ok) -> bool reduces
  the verdict over the ranks (logical AND) so that every rank takes the same decision."""
  requested = int(requested)
  if requested != 2:
    return requested
  if not capturable:
    return 0
  ok = bool(probe_fn())
  if agree_fn is not None:
    ok = bool(agree_fn(ok))
  return 2 if ok else 1


def expected_values(world):
  """(work[0], work[-1], d_arena[0], guard[0]) after the child's two replays: every all-reduce sums `world` equal
  copies - one before the capture, one per replay (world = 1: 12, 18, 4, 1)"""
  w = float(world)
  return (6.0 + 2.0 * w ** 2 + 4.0 * w ** 3, 6.0 + 3.0 * w ** 2 + 9.0 * w ** 3, 4.0 * w ** 3, w ** 3)


def _child():
  if os.environ.get('SG2IM_PROBE_FORCE_FAIL') == '1':
    os.abort()
  import torch
  import torch.distributed as dist
  idx = int(os.environ.get('SG2IM_PROBE_DEVICE', '0'))
  world, rank = int(os.environ.get('SG2IM_PROBE_WORLD', '1')), int(os.environ.get('SG2IM_PROBE_RANK', '0'))
  dev = torch.device('cuda', idx)
  torch.cuda.set_device(dev)
  import datetime
  dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=120))
  n = 1 << 20
  g_arena, d_arena, guard = torch.ones(n, device=dev), torch.ones(n, device=dev), torch.ones(1, device=dev)
  work = torch.zeros(n, device=dev)
  for t in (g_arena, d_arena, guard):
    dist.all_reduce(t)                         # communicator set-up outside any capture
  torch.cuda.synchronize()
  cap, side, lane, comm = (torch.cuda.Stream() for _ in range(4))
  graph = torch.cuda.CUDAGraph()
  with torch.cuda.graph(graph, stream=cap, capture_error_mode='thread_local'):
    main = torch.cuda.current_stream()
    work.add_(1.0)                             # "generator forward"
    comm.wait_stream(main)
    with torch.cuda.stream(comm):
      dist.all_reduce(guard)
    side.wait_stream(main)
    with torch.cuda.stream(side):
      d_arena.mul_(2.0)                        # "discriminator step"
    comm.wait_stream(side)
    with torch.cuda.stream(comm):
      dist.all_reduce(d_arena)
    work.add_(1.0)                             # "data-gradient chain"
    lane.wait_stream(main)
    with torch.cuda.stream(lane):
      g_arena[:n // 2].mul_(2.0)               # "released weight gradients", first bucket complete
      comm.wait_stream(lane)
      with torch.cuda.stream(comm):
        dist.all_reduce(g_arena[:n // 2])
      g_arena[n // 2:].mul_(3.0)
    work.add_(1.0)                             # "small-kernel tail"
    main.wait_stream(lane)
    comm.wait_stream(main)
    with torch.cuda.stream(comm):
      dist.all_reduce(g_arena[n // 2:])
    main.wait_stream(side)
    main.wait_stream(comm)
    work.add_(g_arena)                         # "Adam"
  graph.replay()
  graph.replay()
  torch.cuda.synchronize()
  got = (float(work[0]), float(work[n - 1]), float(d_arena[0]), float(guard[0]))
  want = expected_values(world)
  dist.destroy_process_group()
  if got != want:
    print('capture probe: wrong values %r (expected %r)' % (got, want))
    sys.exit(3)
  print('capture probe: ok')


if __name__ == '__main__':
  _child()
