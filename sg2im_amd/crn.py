"""Cascaded Refinement Network (reference sg2im/crn.py) on the HIP path."""
import torch.nn as nn

from . import functional as HF
from .layers import activation_slope, get_activation, get_normalization_2d, to_channels_last


class RefinementModule(nn.Module):
  """Parameter container with the reference's layout (sg2im/crn.py:35-51): ``net`` is
  Sequential(conv3x3, norm, act, conv3x3, norm, act).  The compute happens in
  RefinementNetwork.forward_nhwc, which runs all modules as one fused HIP sequence."""

  def __init__(self, layout_dim, input_dim, output_dim, normalization='instance', activation='leakyrelu'):
    super(RefinementModule, self).__init__()
    layers = [
      nn.Conv2d(layout_dim + input_dim, output_dim, kernel_size=3, padding=1),
      get_normalization_2d(output_dim, normalization),
      get_activation(activation),
      nn.Conv2d(output_dim, output_dim, kernel_size=3, padding=1),
      get_normalization_2d(output_dim, normalization),
      get_activation(activation),
    ]
    layers = [l for l in layers if l is not None]
    for l in layers:
      if isinstance(l, nn.Conv2d):
        nn.init.kaiming_normal_(l.weight)
    self.net = nn.Sequential(*layers)


class RefinementNetwork(nn.Module):
  """reference sg2im/crn.py:68-111.  ``forward(layout NCHW) -> image NCHW`` keeps the
  reference signature; ``forward_nhwc`` is the zero-conversion entry the model uses."""

  def __init__(self, dims, normalization='instance', activation='leakyrelu'):
    super(RefinementNetwork, self).__init__()
    if normalization not in ('batch', 'none', 'instance'):
      raise ValueError('Unrecognized normalization type "%s"' % normalization)
    self.normalization = normalization
    layout_dim = dims[0]
    self.slope = activation_slope(activation)
    self.refinement_modules = nn.ModuleList()
    for i in range(1, len(dims)):
      input_dim = 1 if i == 1 else dims[i - 1]
      self.refinement_modules.append(
        RefinementModule(layout_dim, input_dim, dims[i], normalization=normalization, activation=activation))
    out_layers = [
      nn.Conv2d(dims[-1], dims[-1], kernel_size=3, padding=1),
      get_activation(activation),
      nn.Conv2d(dims[-1], 3, kernel_size=1, padding=0),
    ]
    nn.init.kaiming_normal_(out_layers[0].weight)
    nn.init.kaiming_normal_(out_layers[2].weight)
    self.output_conv = nn.Sequential(*out_layers)
    to_channels_last(self)

  def forward_nhwc(self, layout_nhwc, layout_grad_channels=None, link=None):
    """link: the functional.LayoutLink of the LayoutFn call that produced ``layout_nhwc`` when this network is
    that tensor's ONLY consumer (Sg2ImModel.forward_nhwc): pyramid and per-level gradients travel through it"""
    o0, o2 = self.output_conv[0], self.output_conv[2]
    if self.normalization in ('none', 'instance'):      # no norm parameters, no running statistics
      convs = []
      for mod in self.refinement_modules:
        c0, c1 = [m for m in mod.net if isinstance(m, nn.Conv2d)]
        convs += [c0.weight, c0.bias, c1.weight, c1.bias]
      params = convs + [o0.weight, o0.bias, o2.weight, o2.bias]
      return HF.RefinementNoNormFn.apply(layout_nhwc, len(self.refinement_modules), self.slope,
                                         layout_grad_channels, self.normalization == 'instance', link, *params)
    convs, bnps, bns = [], [], []
    for mod in self.refinement_modules:
      c0, n0, _, c1, n1, _ = mod.net
      convs += [c0.weight, c0.bias, c1.weight, c1.bias]
      bnps += [n0.weight, n0.bias, n1.weight, n1.bias]
      bns.append((n0, n1))
    o0, o2 = self.output_conv[0], self.output_conv[2]
    params = convs + [o0.weight, o0.bias, o2.weight, o2.bias] + bnps
    return HF.RefinementFn.apply(layout_nhwc, bns, self.slope, self.training, layout_grad_channels, link, *params)

  def forward(self, layout):
    return HF.NhwcToNchw.apply(self.forward_nhwc(HF.NchwToNhwc.apply(layout)))
