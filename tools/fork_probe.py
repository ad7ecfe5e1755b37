import torch
dev = torch.device('cuda', 0)
a = torch.zeros(1 << 20, device=dev); b = torch.zeros(1 << 20, device=dev); c = torch.zeros(64, device=dev)
cap, lane, fin = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
import sys
variant = sys.argv[1] if len(sys.argv) > 1 else 'a'
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=cap, capture_error_mode='thread_local'):
  main = torch.cuda.current_stream()
  c.add_(1)
  lane.wait_stream(main)
  with torch.cuda.stream(lane):
    for k in range(4):
      a.add_(1)
      fin.wait_stream(lane)
      with torch.cuda.stream(fin):
        b.add_(1)
    if variant == 'a':
      lane.wait_stream(fin)
  c.add_(1)
  main.wait_stream(lane)
  if variant == 'b':
    main.wait_stream(fin)
  c.add_(1)
g.replay(); torch.cuda.synchronize()
print('variant', variant, 'ok', float(a[0]), float(b[0]), float(c[0]))
