import sys, os, torch
sys.path.insert(0, os.getcwd())
from oracle import sg2im_oracle as orc
from sg2im_amd.trainer import D_IMG_DEFAULTS
from sg2im_amd import losses as L
from tests import hip_harness as hh
dev = hh.dev()
for norm in ('none', 'batch'):
  dicfg = dict(D_IMG_DEFAULTS, normalization=norm)
  P = orc.init_patch_discriminator_params(dicfg, 23)
  g = torch.Generator().manual_seed(1)
  xr, xf = torch.randn(8, 3, 64, 64, generator=g), torch.randn(8, 3, 64, 64, generator=g) * 0.5
  for which in ('both', 'real', 'fake'):
    res = {}
    for dt in (torch.float32, torch.float64):
      Pd = hh.oracle_leafs({k: v.to(dt) if v.is_floating_point() else v for k, v in P.items()})
      sr = orc.patch_discriminator(Pd, dicfg, xr.to(dt)); sf = orc.patch_discriminator(Pd, dicfg, xf.to(dt))
      loss = (orc.bce_loss(sr, torch.ones_like(sr)) if which != 'fake' else 0) + (orc.bce_loss(sf, torch.zeros_like(sf)) if which != 'real' else 0)
      loss.backward()
      res[dt] = {k: v.grad for k, v in Pd.items() if v.requires_grad and v.grad is not None}
    D = hh.build_d_img(dicfg, P).train()
    from sg2im_amd.functional import NchwToNhwc
    sr = D.forward_nhwc(NchwToNhwc.apply(xr.to(dev))); sf = D.forward_nhwc(NchwToNhwc.apply(xf.to(dev)))
    gl = L.get_gan_losses('gan')[1]
    ones = lambda s: s
    from sg2im_amd import functional as HF
    terms = gl.terms(sr, sf, None)
    loss = terms[0] + terms[1] if which == 'both' else (terms[0] if which == 'real' else terms[1])
    loss.backward()
    for k, p in D.named_parameters():
      if k in res[torch.float64]:
        g64 = res[torch.float64][k]; den = float(g64.abs().max())
        if den < 1e-12: continue
        print(norm, which, '%-16s' % k, 'hip %.2e  ref32 %.2e  max|g| %.2e' % (float((p.grad.cpu().double() - g64).abs().max()) / den,
              float((res[torch.float32][k].double() - g64).abs().max()) / den, den))
