"""Layer factories with the reference's names and state_dict layout
(reference sg2im/layers.py), backed by the HIP Functions in sg2im_amd.functional.

torch.nn modules are used only as *parameter containers* (so state_dict keys, default
initialisers and ``load_state_dict`` behave exactly like the reference); their own
``forward`` is never called - the containers' parents dispatch to HIP kernels.
"""
import torch
import torch.nn as nn

from . import functional as HF


def to_channels_last(module):
  """Store every 4-D conv weight physically as [Cout][KH][KW][Cin] (torch channels_last):
  the layout the implicit-GEMM kernels read.  Shapes / state_dict are unchanged."""
  for m in module.modules():
    if isinstance(m, nn.Conv2d):
      m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
  return module


def get_normalization_2d(channels, normalization):
  """reference sg2im/layers.py:22-31"""
  if normalization == 'batch':
    return nn.BatchNorm2d(channels)
  if normalization == 'none':
    return None
  if normalization == 'instance':
    return nn.InstanceNorm2d(channels)      # parameter-free marker; computed by sg2im_instnorm_*
  raise ValueError('Unrecognized normalization type "%s"' % normalization)


def activation_slope(name):
  """Slope of the LeakyReLU the reference builds for ``name``.  sg2im/layers.py:39
  overwrites the name with 'leakyrelu', so *every* string gives a LeakyReLU; only
  'leakyrelu-<s>' changes the slope from 0.01."""
  slope = 0.01
  if name.lower().startswith('leakyrelu') and '-' in name:
    slope = float(name.split('-')[1])
  return slope


def get_activation(name):
  """reference sg2im/layers.py:33-46 (container only)"""
  return nn.LeakyReLU(negative_slope=activation_slope(name))


class GlobalAvgPool(nn.Module):
  """reference sg2im/layers.py:83-86 on an NHWC tensor"""

  def forward(self, x_nhwc):
    return HF.GapFn.apply(x_nhwc)


class Mlp(nn.Sequential):
  """build_mlp result (reference sg2im/layers.py:216-232) with activation='relu',
  batch_norm='none', final_nonlinearity=True: Linear, ReLU, Linear, ReLU, ..."""

  def linears(self):
    return [m for m in self if isinstance(m, nn.Linear)]

  def norms(self):
    return [m for m in self if isinstance(m, nn.BatchNorm1d)]

  def tail(self, y, i):
    """what follows Linear i: [BatchNorm1d +] ReLU.  Only needed with batch norm - without it the
    ReLU is fused into the GEMM epilogue by the callers."""
    bn = self.norms()[i]
    return HF.BnActRows.apply(y, bn, self.training, bn.weight, bn.bias)

  def forward(self, x):
    lin = self.linears()
    if self.norms():                   # Linear, BatchNorm1d, ReLU, ...
      for i, l in enumerate(lin):
        x = self.tail(HF.LinearAct.apply(x, l.weight, l.bias, 1.0, self.training), i)
      return x
    if len(lin) == 2:
      return HF.Mlp2.apply(x, lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias)
    for l in lin:
      x = HF.LinearAct.apply(x, l.weight, l.bias, 0.0)
    return x


def build_mlp(dim_list, activation='relu', batch_norm='none', dropout=0, final_nonlinearity=True):
  """reference sg2im/layers.py:216-232: Linear [, BatchNorm1d], ReLU per layer"""
  if batch_norm not in ('none', 'batch') or dropout > 0 or activation != 'relu' or not final_nonlinearity:
    raise NotImplementedError('only the build_mlp configurations the model uses are on the HIP path '
                              '(relu, batch_norm none/batch, no dropout, final nonlinearity)')
  layers = []
  for i in range(len(dim_list) - 1):
    layers.append(nn.Linear(dim_list[i], dim_list[i + 1]))
    if batch_norm == 'batch':
      layers.append(nn.BatchNorm1d(dim_list[i + 1]))
    layers.append(nn.ReLU())
  return Mlp(*layers)


def _get_padding(K, mode):
  if mode == 'valid':
    return 0
  if mode == 'same':
    assert K % 2 == 1, 'Invalid kernel size %d for "same" padding' % K
    return (K - 1) // 2
  raise ValueError('Invalid padding "%s"' % mode)


class DiscCnn(nn.Sequential):
  """build_cnn result for 'CK-X-S' architectures (reference sg2im/layers.py:129-213):
  conv, [norm, act, conv]*.  Operates on NHWC tensors."""

  def configure(self, specs, slope):
    self.specs, self.slope = specs, slope
    return self

  def forward(self, x_nhwc):
    convs = [m for m in self if isinstance(m, nn.Conv2d)]
    bns = [m for m in self if isinstance(m, nn.BatchNorm2d)]
    if not bns:                       # 'none': conv, act, conv, ...; 'instance': conv, IN, act, conv, ...
      inorm = any(isinstance(m, nn.InstanceNorm2d) for m in self)
      params = []
      for cv in convs:
        params += [cv.weight, cv.bias]
      return HF.DiscCnnFn.apply(x_nhwc, 'instance' if inorm else None, self.specs, self.slope, self.training, *params)
    if len(bns) != len(convs) - 1:
      raise NotImplementedError('discriminator CNN with a partial set of normalization layers')
    params = [convs[0].weight, convs[0].bias]
    for bn, cv in zip(bns, convs[1:]):
      params += [bn.weight, bn.bias, cv.weight, cv.bias]
    return HF.DiscCnnFn.apply(x_nhwc, bns, self.specs, self.slope, self.training, *params)


def build_cnn(arch, normalization='batch', activation='relu', padding='same', pooling='max', init='default'):
  """Architecture-string CNN builder.  Only the 'I', 'CK-X' and 'CK-X-S' tokens are on the
  HIP path (the defaults of reference scripts/train.py:118-128); 'R', 'U', 'P', 'FC'
  raise NotImplementedError."""
  if isinstance(arch, str):
    arch = arch.split(',')
  cur_C = 3
  if len(arch) > 0 and arch[0][0] == 'I':
    cur_C = int(arch[0][1:])
    arch = arch[1:]
  if init != 'default':
    raise NotImplementedError('only the default conv initialisation is supported')
  layers, specs = [], []
  first = True
  for tok in arch:
    if tok[0] != 'C':
      raise NotImplementedError('arch token "%s" is not on the HIP path (SURVEY.md 8f rank 3)' % tok)
    vals = [int(v) for v in tok[1:].split('-')]
    K, next_C = vals[0], vals[1]
    stride = vals[2] if len(vals) == 3 else 1
    if not first:
      layers.append(get_normalization_2d(cur_C, normalization))
      layers.append(get_activation(activation))
    first = False
    P = _get_padding(K, padding)
    layers.append(nn.Conv2d(cur_C, next_C, kernel_size=K, padding=P, stride=stride))
    specs.append((K, next_C, stride, P))
    cur_C = next_C
  layers = [l for l in layers if l is not None]
  cnn = DiscCnn(*layers).configure(specs, activation_slope(activation))
  return to_channels_last(cnn), cur_C
