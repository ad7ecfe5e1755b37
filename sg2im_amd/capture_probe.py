"""Can THIS software stack digest the stream pattern of the data-parallel iteration graph with the RCCL all-reduces
recorded inside it (Trainer dp_schedule 2)?  Asked in a SUBPROCESS, because the failure mode seen so far is not an
error code: hipStreamEndCapture segfaulted inside clr on one wait pattern (profiles/r5_dp_early_adam_capture_crash.txt).

  python -m sg2im_amd.capture_probe      exit status 0 = captured, replayed twice, values as expected

The child builds a 1-rank RCCL group on the device SG2IM_PROBE_DEVICE and captures a small replica of the schedule-2
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


def probe(device_index=0, timeout=180):
  """True when the child process captured and replayed the pattern; False on a non-zero exit status, a signal or a
  timeout.  Cached per (process, device)."""
  if device_index in _verdict:
    return _verdict[device_index]
  env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'GROUP_RANK', 'ROLE_RANK',
                                                          'LOCAL_WORLD_SIZE', 'TORCHELASTIC_RUN_ID')}
  env.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), SG2IM_PROBE_DEVICE=str(int(device_index)))
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


def _child():
  if os.environ.get('SG2IM_PROBE_FORCE_FAIL') == '1':
    os.abort()
  import torch
  import torch.distributed as dist
  idx = int(os.environ.get('SG2IM_PROBE_DEVICE', '0'))
  dev = torch.device('cuda', idx)
  torch.cuda.set_device(dev)
  dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
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
  want = (3.0 + 2.0 + 3.0 + 4.0, 3.0 + 3.0 + 3.0 + 9.0, 4.0, 1.0)
  dist.destroy_process_group()
  if got != want:
    print('capture probe: wrong values %r (expected %r)' % (got, want))
    sys.exit(3)
  print('capture probe: ok')


if __name__ == '__main__':
  _child()
