import sys; sys.path.insert(0,'/root/repo')
import torch
from sg2im_amd import ops
from sg2im_amd.synthetic import synthetic_batch
D=torch.device('cuda',0)
b=synthetic_batch(32, seed=0)
imgs,objs,boxes,masks,triples,o2i,_=[t.to(D) if torch.is_tensor(t) else t for t in b]
O=boxes.size(0)
g=torch.randn(O,32,32,3,device=D)
d=torch.empty(32,64,64,3,device=D)
def run():
  ops.crop_backward(g, boxes, o2i, 32, False, d)
run(); torch.cuda.synchronize()
a,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): run()
e.record(); torch.cuda.synchronize()
print('crop_backward %.1f us' % (a.elapsed_time(e)/20*1e3), 'O', O)
