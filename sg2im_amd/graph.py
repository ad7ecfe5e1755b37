"""Scene-graph convolution modules (reference sg2im/graph.py) on the HIP path."""
import torch.nn as nn

from . import functional as HF
from .layers import build_mlp


def _init_weights(module):
  if isinstance(module, nn.Linear):
    nn.init.kaiming_normal_(module.weight)       # reference sg2im/graph.py:26-29


class GraphTripleConv(nn.Module):
  """A single layer of scene graph convolution (reference sg2im/graph.py:32-120).

  ``forward(obj_vecs (O,Din), pred_vecs (T,Din), edges (T,2) int64)`` returns
  ``(new_obj_vecs (O,Dout), new_pred_vecs (T,Dout))``.  ``edges`` may be replaced by a
  prepared ``(s_idx, o_idx, csr)`` triple so the CSR over destination objects is built
  once per batch instead of once per layer.
  """

  def __init__(self, input_dim, output_dim=None, hidden_dim=512, pooling='avg', mlp_normalization='none'):
    super(GraphTripleConv, self).__init__()
    if output_dim is None:
      output_dim = input_dim
    self.input_dim, self.output_dim, self.hidden_dim = input_dim, output_dim, hidden_dim
    assert pooling in ['sum', 'avg'], 'Invalid pooling "%s"' % pooling
    self.pooling = pooling
    self.net1 = build_mlp([3 * input_dim, hidden_dim, 2 * hidden_dim + output_dim], batch_norm=mlp_normalization)
    self.net1.apply(_init_weights)
    self.net2 = build_mlp([hidden_dim, hidden_dim, output_dim], batch_norm=mlp_normalization)
    self.net2.apply(_init_weights)

  @staticmethod
  def prepare_edges(edges, num_objs):
    from . import ops
    s_idx = edges[:, 0].contiguous()
    o_idx = edges[:, 1].contiguous()
    return s_idx, o_idx, ops.Csr(s_idx, o_idx, num_objs)

  def forward(self, obj_vecs, pred_vecs, edges, counts=None):
    """counts: None or ((int32 device scalar, 1) objects, (..., 1) triples) - the real rows of a padded
    batch (sg2im_amd/bucketing.py); only the BatchNorm1d statistics of mlp_normalization='batch' need them"""
    if not isinstance(edges, tuple):
      edges = self.prepare_edges(edges, obj_vecs.size(0))
    s_idx, o_idx, csr = edges
    ocnt, tcnt = counts if counts is not None else (None, None)
    a, b = self.net1.linears()
    c, d = self.net2.linears()
    if self.net1.norms():              # mlp_normalization='batch': chained from the composable pieces
      H, Dout = self.hidden_dim, self.output_dim
      t = self.net1.tail(HF.TripleLinear.apply(obj_vecs, pred_vecs, s_idx, o_idx, csr, a.weight, a.bias,
                                                  self.training), 0, tcnt)
      new_t = self.net1.tail(HF.LinearAct.apply(t, b.weight, b.bias, 1.0, self.training), 1, tcnt)
      pooled, new_p = HF.TriplePool.apply(new_t, s_idx, o_idx, csr, self.pooling == 'avg', H, Dout,
                                          obj_vecs.size(0))
      return self.net2(pooled, ocnt), new_p
    return HF.GraphTripleConvFn.apply(obj_vecs, pred_vecs, s_idx, o_idx, csr, self.pooling == 'avg',
                                      a.weight, a.bias, b.weight, b.bias, c.weight, c.bias, d.weight, d.bias)


class GraphTripleConvNet(nn.Module):
  """A sequence of scene graph convolution layers (reference sg2im/graph.py:123-144)."""

  def __init__(self, input_dim, num_layers=5, hidden_dim=512, pooling='avg', mlp_normalization='none'):
    super(GraphTripleConvNet, self).__init__()
    self.num_layers = num_layers
    self.gconvs = nn.ModuleList([
      GraphTripleConv(input_dim=input_dim, hidden_dim=hidden_dim, pooling=pooling,
                      mlp_normalization=mlp_normalization) for _ in range(num_layers)])

  def forward(self, obj_vecs, pred_vecs, edges, counts=None):
    if not isinstance(edges, tuple):
      edges = GraphTripleConv.prepare_edges(edges, obj_vecs.size(0))
    for gconv in self.gconvs:
      obj_vecs, pred_vecs = gconv(obj_vecs, pred_vecs, edges, counts)
    return obj_vecs, pred_vecs
