"""Can an RCCL all-reduce be recorded into a hipGraph here?  1-rank group, collective on a side stream inside the
capture (as a data-parallel iteration graph would issue it), three replays.   timeout 120 python tools/rccl_capture_probe.py"""
import os
import sys
import torch
import torch.distributed as dist

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29577')
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
t = torch.ones(1 << 22, device=dev)
u = torch.zeros(1 << 22, device=dev)
dist.all_reduce(t)                       # eager warm-up: communicator set-up outside any capture
torch.cuda.synchronize()
print('eager all-reduce ok', float(t[0]))
cap, comm = torch.cuda.Stream(), torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
try:
  with torch.cuda.graph(g, stream=cap, capture_error_mode='thread_local'):
    t.mul_(2.0)
    comm.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(comm):
      dist.all_reduce(t)
    u.add_(1.0)                          # compute next to the collective
    torch.cuda.current_stream().wait_stream(comm)
    u.add_(t)
  print('capture ok')
  for i in range(3):
    g.replay()
  torch.cuda.synchronize()
  print('replays ok: t[0] = %g (expect 8), u[0] = %g (expect 3 + 2 + 4 + 8 = 17)' % (float(t[0]), float(u[0])))
except Exception as e:
  print('RCCL capture FAILED: %s: %s' % (type(e).__name__, e))
finally:
  try:
    dist.destroy_process_group()
  except Exception as e:
    print('destroy failed', e)
