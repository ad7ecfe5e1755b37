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
    if x_nhwc.dim() == 2:              # after an FC token: x.view(N, C, -1).mean(2) is the identity
      return x_nhwc
    return HF.GapFn.apply(x_nhwc)


class Mlp(nn.Sequential):
  """build_mlp result (reference sg2im/layers.py:216-232) with activation='relu',
  batch_norm='none', final_nonlinearity=True: Linear, ReLU, Linear, ReLU, ..."""

  def linears(self):
    return [m for m in self if isinstance(m, nn.Linear)]

  def norms(self):
    return [m for m in self if isinstance(m, nn.BatchNorm1d)]

  def tail(self, y, i, count=None):
    """what follows Linear i: [BatchNorm1d +] ReLU.  Only needed with batch norm - without it the
    ReLU is fused into the GEMM epilogue by the callers.  count: (int32 device scalar, 1) - the real rows
    of a padded batch (sg2im_amd/bucketing.py); the batch statistics only see those."""
    bn = self.norms()[i]
    return HF.BnActRows.apply(y, bn, self.training, bn.weight, bn.bias, 0.0, count)

  def forward(self, x, count=None):
    lin = self.linears()
    if self.norms():                   # Linear, BatchNorm1d, ReLU, ...
      for i, l in enumerate(lin):
        x = self.tail(HF.LinearAct.apply(x, l.weight, l.bias, 1.0, self.training), i, count)
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

  def forward(self, x_nhwc, count=None, share=None):
    """count: None or (int32 device scalar, 1) - real entries of a padded batch (sg2im_amd/bucketing.py);
    share: None or a functional.SharedPass - this pass is needed by two autograd graphs (x_nhwc may be None once
    the pass is recorded)"""
    convs = [m for m in self if isinstance(m, nn.Conv2d)]
    bns = [m for m in self if isinstance(m, nn.BatchNorm2d)]
    if not bns:                       # 'none': conv, act, conv, ...; 'instance': conv, IN, act, conv, ...
      inorm = any(isinstance(m, nn.InstanceNorm2d) for m in self)
      params = []
      for cv in convs:
        params += [cv.weight, cv.bias]
      return HF.DiscCnnFn.apply(x_nhwc, 'instance' if inorm else None, self.specs, self.slope, self.training, count,
                                share, *params)
    if len(bns) != len(convs) - 1:
      raise NotImplementedError('discriminator CNN with a partial set of normalization layers')
    params = [convs[0].weight, convs[0].bias]
    for bn, cv in zip(bns, convs[1:]):
      params += [bn.weight, bn.bias, cv.weight, cv.bias]
    return HF.DiscCnnFn.apply(x_nhwc, bns, self.specs, self.slope, self.training, count, share, *params)


def _init_conv(layer, method):
  """reference sg2im/layers.py:48-56"""
  if not isinstance(layer, nn.Conv2d) or method == 'default':
    return
  if method == 'kaiming-normal':
    nn.init.kaiming_normal_(layer.weight)
  elif method == 'kaiming-uniform':
    nn.init.kaiming_uniform_(layer.weight)


class Flatten(nn.Module):
  """reference sg2im/layers.py:59-64: (N, C, H, W) -> (N, C*H*W).  The NHWC tensor is put back
  into the reference's channel-major order first so an FC layer sees the reference's feature order."""

  def forward(self, x):
    if x.dim() == 4:
      x = HF.NhwcToNchw.apply(x)
    return x.reshape(x.size(0), -1)

  def __repr__(self):
    return 'Flatten()'


def _norm_act(norm, act, x, training):
  """[normalization +] LeakyReLU in front of a conv that does not take them into its loader"""
  slope = act.negative_slope
  if isinstance(norm, nn.BatchNorm2d):
    return HF.BnActRows.apply(x, norm, training, norm.weight, norm.bias, slope)
  if isinstance(norm, nn.InstanceNorm2d):
    return HF.InstNormAct.apply(x, slope)
  return HF.LeakyFn.apply(x, slope)


def _run_layers(mods, x, training, bn_inputs=None):
  """Executes build_cnn's module list on an NHWC tensor, one HIP op per module.  A conv whose
  output goes straight into a batch-statistics / instance norm gets the analytic zero bias gradient.
  ``bn_inputs`` collects (BatchNorm2d, its input) pairs for ResidualBlock's second statistics pass."""
  mods = list(mods)
  i = 0
  while i < len(mods):
    m = mods[i]
    nxt = mods[i + 1] if i + 1 < len(mods) else None
    if isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d)):
      if bn_inputs is not None and isinstance(m, nn.BatchNorm2d):
        bn_inputs.append((m, x))
      x = _norm_act(m, nxt, x, training)            # a norm is always followed by its activation
      i += 2
      continue
    if isinstance(m, nn.LeakyReLU):
      x = HF.LeakyFn.apply(x, m.negative_slope)
    elif isinstance(m, nn.Conv2d):
      shadowed = isinstance(nxt, nn.InstanceNorm2d) or (training and isinstance(nxt, nn.BatchNorm2d))
      x = HF.Conv2dFn.apply(x, m.weight, m.bias, m.stride[0], m.padding[0], shadowed)
    elif isinstance(m, ResidualBlock):
      x = m(x)
    elif isinstance(m, nn.Upsample):
      x = HF.UpsampleFn.apply(x, int(m.scale_factor))
    elif isinstance(m, nn.AvgPool2d):
      x = HF.AvgPoolFn.apply(x, int(m.kernel_size))
    elif isinstance(m, nn.MaxPool2d):
      x = HF.MaxPoolFn.apply(x, int(m.kernel_size))
    elif isinstance(m, Flatten):
      x = m(x)
    elif isinstance(m, nn.Linear):
      x = HF.LinearAct.apply(x, m.weight, m.bias, 1.0)
    else:
      raise NotImplementedError('no HIP op for %r' % (m,))
    i += 1
  return x


class ResidualBlock(nn.Module):
  """reference sg2im/layers.py:88-117: x + net(x), net = [norm, act, conv, norm, act, conv].
  The reference evaluates ``self.net(x)`` twice per forward (:116-117) and keeps the second
  result: same output, but a training-mode BatchNorm moves its running statistics twice.
  Reproduced by a second statistics-only pass over the same inputs."""

  def __init__(self, channels, normalization='batch', activation='relu', padding='same', kernel_size=3,
               init='default'):
    super(ResidualBlock, self).__init__()
    K, C = kernel_size, channels
    P = _get_padding(K, padding)
    self.padding = P
    layers = [get_normalization_2d(C, normalization), get_activation(activation),
              nn.Conv2d(C, C, kernel_size=K, padding=P),
              get_normalization_2d(C, normalization), get_activation(activation),
              nn.Conv2d(C, C, kernel_size=K, padding=P)]
    layers = [l for l in layers if l is not None]
    for l in layers:
      _init_conv(l, init)
    self.net = to_channels_last(nn.Sequential(*layers))

  def forward(self, x_nhwc):
    if self.padding == 0:
      # the reference slices the shortcut with [P:-P] = [0:-0], an empty tensor, and fails in the add
      raise RuntimeError('ResidualBlock with "valid" padding is broken in the reference (layers.py:113-114)')
    seen = []
    y = _run_layers(self.net, x_nhwc, self.training, seen)
    if self.training:                    # the reference's discarded first evaluation (:116)
      from . import ops
      for bn, t in seen:
        C = t.size(-1)
        ops.bn_stats(t.detach().contiguous(), t.numel() // C, C, C, bn, True, HF.BN_EPS, HF.BN_MOMENTUM)
    return HF.AddFn.apply(x_nhwc, y)


class SeqCnn(nn.Sequential):
  """build_cnn result for architecture strings with R / U / P / FC tokens: run layer by layer"""

  def forward(self, x_nhwc):
    return _run_layers(self, x_nhwc, self.training)


def build_cnn(arch, normalization='batch', activation='relu', padding='same', pooling='max', init='default'):
  """Architecture-string CNN builder (reference sg2im/layers.py:129-213): I<C>, C<K>-<X>[-<S>], R,
  U<f>, P<f>, FC-<in>-<out>.  Strings made of 'C' tokens only (every default of scripts/train.py)
  become a DiscCnn, whose norm + activation ride in the next conv's loader; anything else becomes
  a SeqCnn that runs one HIP op per module.  Module order, and so the state_dict, is the reference's."""
  if isinstance(arch, str):
    arch = arch.split(',')
  cur_C = 3
  if len(arch) > 0 and arch[0][0] == 'I':
    cur_C = int(arch[0][1:])
    arch = arch[1:]
  layers, specs = [], []
  first_conv, flat = True, False
  for i, tok in enumerate(arch):
    if tok[0] == 'C':
      vals = [int(v) for v in tok[1:].split('-')]
      K, next_C = vals[0], vals[1]
      stride = vals[2] if len(vals) == 3 else 1
      if not first_conv:
        layers.append(get_normalization_2d(cur_C, normalization))
        layers.append(get_activation(activation))
      first_conv = False
      P = _get_padding(K, padding)
      layers.append(nn.Conv2d(cur_C, next_C, kernel_size=K, padding=P, stride=stride))
      _init_conv(layers[-1], init)
      specs.append((K, next_C, stride, P))
      cur_C = next_C
    elif tok[0] == 'R':
      layers.append(ResidualBlock(cur_C, normalization='none' if first_conv else normalization,
                                  activation=activation, padding=padding, init=init))
      first_conv = False
    elif tok[0] == 'U':
      layers.append(nn.Upsample(scale_factor=int(tok[1:]), mode='nearest'))
    elif tok[0] == 'P':
      f = int(tok[1:])
      layers.append(nn.MaxPool2d(kernel_size=f, stride=f) if pooling == 'max'
                    else nn.AvgPool2d(kernel_size=f, stride=f))
    elif tok[:2] == 'FC':
      _, Din, Dout = tok.split('-')
      if not flat:
        layers.append(Flatten())
      flat = True
      layers.append(nn.Linear(int(Din), int(Dout)))
      if i + 1 < len(arch):
        layers.append(get_activation(activation))
      cur_C = int(Dout)
    else:
      raise ValueError('Invalid layer "%s"' % tok)
  layers = [l for l in layers if l is not None]
  if all(tok[0] == 'C' for tok in arch):
    cnn = DiscCnn(*layers).configure(specs, activation_slope(activation))
  else:
    cnn = SeqCnn(*layers)
  return to_channels_last(cnn), cur_C
