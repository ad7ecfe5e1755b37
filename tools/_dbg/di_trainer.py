import sys, os, torch
sys.path.insert(0, os.getcwd())
from oracle import sg2im_oracle as orc
from sg2im_amd.synthetic import make_vocab, synthetic_batch
from sg2im_amd.trainer import Trainer, GENERATOR_DEFAULTS, D_OBJ_DEFAULTS, D_IMG_DEFAULTS
from tests import hip_harness as hh
dev = hh.dev()
vocab = make_vocab(184, 7)
bs = 8
gk, dk = dict(normalization='none'), dict(normalization='none')
gcfg = dict(GENERATOR_DEFAULTS, vocab=vocab, **gk)
docfg, dicfg = dict(D_OBJ_DEFAULTS, vocab=vocab, **dk), dict(D_IMG_DEFAULTS, **dk)
PG = orc.init_generator_params(gcfg, 21); PDo = orc.init_ac_discriminator_params(docfg, 22); PDi = orc.init_patch_discriminator_params(dicfg, 23)
tr = Trainer(vocab, dev, seed=0, generator_kwargs=gk, d_obj_kwargs=dk, d_img_kwargs=dk)
hh.load_params(tr.model, PG); hh.load_params(tr.d_obj, PDo); hh.load_params(tr.d_img, PDi)
cpu_batch = synthetic_batch(bs, seed=61)
noise = torch.randn(bs, 32, 64, 64, generator=torch.Generator().manual_seed(62))
otr = hh.OracleRefs(PG, PDo, PDi, gcfg, docfg, dicfg)
batch = tuple(t.to(dev) if torch.is_tensor(t) else t for t in cpu_batch)
# grab the HIP fake image
keep = {}
orig = tr._seg_generator_model
def wrap(b, st):
  orig(b, st); keep['fake'] = st['imgs_fake'].detach().clone()
tr._seg_generator_model = wrap
with hh.fixed_noise(noise):
  tr.step(batch)
otr.step(tuple(cpu_batch[:6]), noise)
fake = keep['fake'].permute(0, 3, 1, 2).cpu().double()
# float64 D_img step on HIP's fake image
P = hh.oracle_leafs({k: v.double() if v.is_floating_point() else v for k, v in PDi.items()})
sr = orc.patch_discriminator(P, dicfg, cpu_batch[0].double()); sf = orc.patch_discriminator(P, dicfg, fake)
(orc.bce_loss(sr, torch.ones_like(sr)) + orc.bce_loss(sf, torch.zeros_like(sf))).backward()
for k, p in tr.d_img.named_parameters():
  g64 = otr.o64.PDi[k].grad
  if g64 is None: continue
  den = float(g64.abs().max()); got = p.grad.cpu().double()
  print('%-14s hip-vs-f64 %.2e  hip-vs-f64(on hip image) %.2e  o32-vs-f64 %.2e  f64(hip image)-vs-f64 %.2e' % (
    k, float((got - g64).abs().max()) / den, float((got - P[k].grad).abs().max()) / den,
    float((otr.o32.PDi[k].grad.double() - g64).abs().max()) / den, float((P[k].grad - g64).abs().max()) / den))
