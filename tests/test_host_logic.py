"""CPU: host-side logic that needs no kernel - batch layout contract, sharding, module /
state_dict surface, scene-graph encoding, flat parameter arenas."""
import os
import sys

import pytest
import torch

from oracle import sg2im_oracle as orc
from sg2im_amd.synthetic import make_vocab, shard_batch, synthetic_batch

REF = '/root/reference'


@pytest.mark.parametrize('style', ['coco', 'vg'])
def test_synthetic_batch_follows_collate_contract(style):
  N = 6
  imgs, objs, boxes, masks, triples, o2i, t2i = synthetic_batch(N, style=style, seed=4)
  O, T = objs.numel(), triples.size(0)
  assert imgs.shape == (N, 3, 64, 64) and boxes.shape == (O, 4) and triples.shape == (T, 3)
  assert (masks is None) == (style == 'vg')
  assert bool((o2i[1:] >= o2i[:-1]).all()) and bool((t2i[1:] >= t2i[:-1]).all())
  for n in range(N):
    idx = (o2i == n).nonzero().view(-1)
    assert int(objs[idx[-1]]) == 0 and bool((objs[idx[:-1]] > 0).all())      # __image__ last
    assert boxes[idx[-1]].tolist() == [0.0, 0.0, 1.0, 1.0]
    tri = triples[t2i == n]
    assert bool(((tri[:, 0] >= idx[0]) & (tri[:, 0] <= idx[-1]) & (tri[:, 2] >= idx[0]) & (tri[:, 2] <= idx[-1])).all())
    k = idx.numel() - 1
    tail = tri[-k:]                                                            # __in_image__ block
    assert bool((tail[:, 1] == 0).all()) and bool((tail[:, 2] == idx[-1]).all())
    assert tail[:, 0].tolist() == idx[:-1].tolist()
  w = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
  assert float(w.min()) > 0


def test_shard_batch_equals_collating_the_shard():
  full = synthetic_batch(4, seed=9)
  for r in range(2):
    sh = shard_batch(full, r, 2)
    imgs, objs, boxes, masks, triples, o2i, t2i = sh
    assert imgs.size(0) == 2 and int(o2i.min()) == 0 and int(o2i.max()) == 1
    assert int(triples[:, [0, 2]].min()) >= 0 and int(triples[:, [0, 2]].max()) < objs.numel()
    lo = r * 2
    sel = (full[5] >= lo) & (full[5] < lo + 2)
    assert torch.equal(objs, full[1][sel]) and torch.equal(boxes, full[2][sel])
    # a triple's endpoints keep their categories
    tsel = (full[6] >= lo) & (full[6] < lo + 2)
    assert torch.equal(objs[triples[:, 0]], full[1][full[4][tsel][:, 0]])


def test_generator_state_dict_matches_oracle_and_reference():
  from sg2im_amd.model import Sg2ImModel
  from sg2im_amd.trainer import GENERATOR_DEFAULTS
  vocab = make_vocab(184, 7)
  m = Sg2ImModel(vocab, **GENERATOR_DEFAULTS)
  sd = m.state_dict()
  P = orc.init_generator_params(dict(GENERATOR_DEFAULTS, vocab=vocab))
  assert set(sd) == set(P)
  assert all(tuple(sd[k].shape) == tuple(P[k].shape) for k in sd)
  assert sum(p.numel() for p in m.parameters()) == 28158223
  if os.path.isdir(REF):
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    from sg2im.model import Sg2ImModel as RefModel
    r = RefModel(vocab, **GENERATOR_DEFAULTS).state_dict()
    assert set(r) == set(sd) and all(tuple(r[k].shape) == tuple(sd[k].shape) for k in r)
    m.load_state_dict(r)                                                   # reference checkpoints load
  # conv weights are physically channels_last
  w = m.refinement_net.refinement_modules[1].net[0].weight
  assert w.permute(0, 2, 3, 1).is_contiguous()


def test_discriminator_state_dicts():
  from sg2im_amd.discriminators import AcCropDiscriminator, PatchDiscriminator
  from sg2im_amd.trainer import D_IMG_DEFAULTS, D_OBJ_DEFAULTS
  vocab = make_vocab(184, 7)
  do = AcCropDiscriminator(vocab, **D_OBJ_DEFAULTS).state_dict()
  di = PatchDiscriminator(**D_IMG_DEFAULTS).state_dict()
  Po = orc.init_ac_discriminator_params(dict(D_OBJ_DEFAULTS, vocab=vocab))
  Pi = orc.init_patch_discriminator_params(dict(D_IMG_DEFAULTS))
  assert set(do) == set(Po) and set(di) == set(Pi)
  assert sum(v.numel() for k, v in do.items() if 'running' not in k and 'num_batches' not in k) == 1112057
  assert sum(v.numel() for k, v in di.items() if 'running' not in k and 'num_batches' not in k) == 659521


def test_activation_quirk_and_unsupported_options_fail_loudly():
  from sg2im_amd import layers
  assert layers.activation_slope('relu') == 0.01 and layers.activation_slope('leakyrelu-0.2') == 0.2
  with pytest.raises(ValueError):
    layers.get_normalization_2d(8, 'bogus')
  inorm = layers.get_normalization_2d(8, 'instance')      # reference layers.py:27-28: no affine, no buffers
  assert isinstance(inorm, torch.nn.InstanceNorm2d) and not inorm.state_dict()
  cnn_i, _ = layers.build_cnn('I3,C4-8-2,C4-16-2', normalization='instance', padding='valid')
  assert sorted(cnn_i.state_dict()) == ['0.bias', '0.weight', '3.bias', '3.weight']
  with pytest.raises(ValueError):
    layers.build_cnn('C3-8,X2')                           # reference layers.py:207
  seq, c = layers.build_cnn('I3,C3-8,R,P2,C3-16-2,U2,R,FC-64-10,FC-10-4', activation='leakyrelu-0.2')
  assert c == 4 and isinstance(seq, layers.SeqCnn) and isinstance(seq[2], torch.nn.MaxPool2d)
  assert [k for k in seq.state_dict() if k.startswith('1.net.')][:2] == ['1.net.0.weight', '1.net.0.bias']
  assert '9.weight' in seq.state_dict() and '11.bias' in seq.state_dict()
  cnn, c = layers.build_cnn('I3,C4-64-2,C4-128-2,C4-256-2', padding='valid', activation='leakyrelu-0.2')
  assert c == 256 and cnn.specs == [(4, 64, 2, 0), (4, 128, 2, 0), (4, 256, 2, 0)]
  assert [k for k in cnn.state_dict()][:3] == ['0.weight', '0.bias', '1.weight']


def test_encode_scene_graphs_like_the_reference():
  import json
  from sg2im_amd.model import Sg2ImModel
  vocab = make_vocab(5, 3)
  vocab['object_name_to_idx'].update({'sheep': 1, 'grass': 2})
  vocab['pred_name_to_idx'].update({'standing on': 1})
  m = Sg2ImModel(vocab, image_size=(16, 16), embedding_dim=8, gconv_dim=8, gconv_hidden_dim=16,
                 gconv_num_layers=1, refinement_dims=(16, 8), mask_size=4)
  sg = {'objects': ['sheep', 'grass'], 'relationships': [[0, 'standing on', 1]]}
  objs, triples, o2i = m.encode_scene_graphs([sg])
  assert objs.tolist() == [1, 2, 0] and o2i.tolist() == [0, 0, 0]
  assert triples.tolist() == [[0, 1, 1], [0, 0, 2], [1, 0, 2]]
  assert sg['objects'][-1] == '__image__'                                  # mutates its input
  with pytest.raises(ValueError):
    m.encode_scene_graphs({'objects': ['cow'], 'relationships': []})
  path = os.path.join(REF, 'scene_graphs', 'figure_6_sheep.json')
  if os.path.exists(path):
    graphs = json.load(open(path))
    assert len(graphs) == 7


def test_flat_params_alias_parameters_and_grads():
  from sg2im_amd.optim import FlatParams
  from sg2im_amd.discriminators import PatchDiscriminator
  m = PatchDiscriminator(arch='C4-8-2,C4-16-2', padding='valid')
  before = {k: v.clone() for k, v in m.state_dict().items()}
  fp = FlatParams(m)
  for k, v in m.state_dict().items():
    assert torch.equal(v, before[k])
  w = m.cnn[0].weight
  assert w.permute(0, 2, 3, 1).is_contiguous()                              # channels_last kept
  fp.flat.mul_(2.0)
  assert torch.equal(m.cnn[0].weight.detach(), before['cnn.0.weight'] * 2)
  m.cnn[0].bias.grad.fill_(3.0)
  assert float(fp.grad.sum()) == 3.0 * m.cnn[0].bias.numel()
  fp.zero_grad()
  assert float(fp.grad.abs().sum()) == 0.0


def test_optimizer_state_round_trips_through_torch_adam_layout():
  """checkpoints carry ``optimizer.state_dict()`` (reference scripts/train.py:642-650): the arena
  optimiser must read a torch.optim.Adam state of the same module and write one torch can read"""
  import copy
  from sg2im_amd.optim import FlatParams, FlatAdam
  from sg2im_amd.discriminators import PatchDiscriminator
  torch.manual_seed(3)
  m = PatchDiscriminator(arch='C4-8-2,C4-16-2', padding='valid')
  ref = copy.deepcopy(m)
  ref_opt = torch.optim.Adam(ref.parameters(), lr=1e-4)
  for _ in range(3):                       # any gradients will do; the unused classifier gets none
    ref_opt.zero_grad()
    for name, p in ref.named_parameters():
      if not name.startswith('classifier'):
        p.grad = torch.randn_like(p)
    ref_opt.step()
  sd = ref_opt.state_dict()
  opt = FlatAdam(FlatParams(m), lr=5e-4)
  opt.load_state_dict(sd)
  assert opt.t == 3 and opt.lr == 1e-4
  names = [n for n, _ in m.named_parameters()]
  out = opt.state_dict()
  assert out['param_groups'][0]['params'] == list(range(len(names)))
  for i, n in enumerate(names):
    if i in sd['state']:
      for k in ('exp_avg', 'exp_avg_sq'):
        assert torch.equal(out['state'][i][k], sd['state'][i][k]), (n, k)
        assert out['state'][i][k].is_contiguous()
      assert float(out['state'][i]['step']) == 3.0
    else:                                  # never stepped by torch: zero moments here
      assert float(out['state'][i]['exp_avg'].abs().sum()) == 0.0
  fresh = torch.optim.Adam(copy.deepcopy(ref).parameters(), lr=1e-4)
  fresh.load_state_dict(out)               # torch accepts what we write
  assert FlatAdam(FlatParams(copy.deepcopy(ref))).state_dict()['state'] == {}


def test_train_script_keeps_the_reference_flag_surface():
  import re
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  mine = set(re.findall(r"add_argument\('(--\w+)'", open(os.path.join(root, 'scripts', 'train.py')).read()))
  assert len(mine) >= 63
  ref_path = os.path.join(REF, 'scripts', 'train.py')
  if os.path.exists(ref_path):
    ref = set(re.findall(r"add_argument\('(--\w+)'", open(ref_path).read()))
    assert not (ref - mine), sorted(ref - mine)


def test_launch_planner_choices_are_valid_without_a_gpu():
  """The launch planner is host code inside libsg2im_hip.so: tools/plan_dump.py drives the three conv entry
  points with dummy pointers (the launches themselves fail without a GPU) and SG2IM_PLAN_DEBUG prints the
  chosen tile / split-K per layer.  Every plan must use one of the four tiles, a split count within the
  reduction length, and the big CRN layers must not fall back to the 64x64 tile; the 3x3 stride-1 layers on maps
  of 16x16 and larger take the halo'd-tile kernels (csrc/conv_halo.h) forward and in the data gradient: a 128-pixel
  patch, 64- or 128-wide column tiles, split-K over whole 32-channel chunks; their weight gradients take the halo'd
  weight-gradient kernel (csrc/wgrad_halo.h: 64-pixel patches, 64 x 64-channel blocks, K split over whole patches)
  except on the widest concats."""
  import re
  import subprocess
  import sys
  tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools')
  out = subprocess.run([sys.executable, os.path.join(tools, 'plan_dump.py')], cwd=tools, capture_output=True,
                       text=True, timeout=300).stderr
  plans, halo, whalo = {}, {}, {}
  cur = None
  for line in out.splitlines():
    m = re.match(r'== (\S+) (\S+)', line)
    if m:
      cur = (m.group(1), m.group(2))
    m = re.match(r'\[sg2im plan\] M=(\d+) N=(\d+) iters=(\d+) -> (\d+)x(\d+) x(\d+)', line)
    if m and cur:
      plans[cur] = tuple(int(v) for v in m.groups())
    m = re.match(r'\[sg2im halo\] M=(\d+) N=(\d+) chunks=(\d+) -> patch (\d+)x(\d+) bn=(\d+) x(\d+)', line)
    if m and cur:
      halo[cur] = tuple(int(v) for v in m.groups())
    m = re.match(r'\[sg2im wgrad halo\] Cout=(\d+) Ctot=(\d+) patches=(\d+) -> (\S+) blocks (\d+)x(\d+) x(\d+)', line)
    if m and cur:
      whalo[cur] = m.groups()
  assert len(plans) + len(halo) + len(whalo) >= 60, (len(plans), len(halo), len(whalo))
  for (layer, what), (M, N, iters, bm, bn, ns) in plans.items():
    assert (bm, bn) in ((128, 128), (128, 64), (64, 64), (64, 128)), (layer, what, bm, bn)
    assert 1 <= ns <= max(1, iters), (layer, what, ns, iters)
  for (layer, what), (M, N, chunks, rt, ct, bn, ns) in halo.items():
    assert (rt, ct) in ((8, 16), (4, 32), (2, 64)) and bn in (64, 128) and 1 <= ns <= chunks and M % 128 == 0, (layer, what)
    assert what in ('fwd', 'dgrad')
  for layer in ('m2.conv0', 'm3.conv0', 'm4.conv0', 'm4.conv1', 'out.conv0', 'mask.c3'):
    assert (layer, 'fwd') in halo and (layer, 'dgrad') in halo, layer
  for (layer, what), (cout, ctot, patches, patch, ncb, nkb, ns) in whalo.items():
    assert what == 'wgrad' and patch in ('4x16', '8x8') and int(ctot) <= 512, (layer, what, patch, ctot)
    assert int(ncb) == (int(ctot) + 63) // 64 and int(nkb) == (int(cout) + 63) // 64 and 1 <= int(ns) <= int(patches) // 6
  for layer in ('m1.conv1', 'm2.conv1', 'm3.conv0', 'm3.conv1', 'm4.conv0', 'm4.conv1', 'out.conv0', 'mask.c3'):
    assert (layer, 'wgrad') in whalo, layer
  for layer in ('m1.conv0', 'm2.conv0', 'm3.conv0', 'm4.conv0'):
    for what in ('fwd', 'dgrad', 'wgrad'):
      if (layer, what) not in halo and (layer, what) not in whalo:
        bm, bn = plans[(layer, what)][3:5]
        assert bm * bn > 64 * 64, (layer, what, bm, bn)


def test_bucketing_pads_with_neutral_rows():
  """sg2im_amd/bucketing.py: structure of a padded batch, and - on the CPU oracle, which restates the
  reference's own arithmetic - that the dummy objects / triples leave the generated images and the
  real objects' boxes untouched (dummy boxes lie outside the image, dummy triples only touch a dummy
  object)."""
  from oracle import sg2im_oracle as orc
  from sg2im_amd.bucketing import Bucketer, pad_batch, FAR_BOX
  from sg2im_amd.synthetic import make_vocab, synthetic_batch
  b = Bucketer(32, 64)
  assert b.bucket(31, 64) == (32, 64) and b.bucket(32, 65) == (64, 128) and b.bucket(5, 0) == (32, 64)
  batch = synthetic_batch(3, seed=4)
  O, T = batch[1].numel(), batch[4].size(0)
  o_pad, t_pad = b.bucket(O, T)
  (imgs, objs, boxes, masks, triples, o2i), counts = pad_batch(batch, o_pad, t_pad)
  assert objs.shape == (o_pad,) and boxes.shape == (o_pad, 4) and masks.shape[0] == o_pad
  assert triples.shape == (t_pad, 3) and o2i.shape == (o_pad,) and counts.tolist() == [O, T]
  assert torch.equal(objs[:O], batch[1]) and torch.equal(triples[:T], batch[4]) and torch.equal(boxes[:O], batch[2])
  assert bool((objs[O:] == 0).all()) and bool((o2i[O:] == imgs.size(0) - 1).all()) and bool((masks[O:] == 0).all())
  assert boxes[O:].tolist() == [list(FAR_BOX)] * (o_pad - O)
  assert bool((triples[T:, 0] == o_pad - 1).all()) and bool((triples[T:, 2] == o_pad - 1).all())
  assert bool((o2i[1:] >= o2i[:-1]).all())                     # still sorted by image
  with pytest.raises(ValueError):
    pad_batch(batch, O, t_pad)                                 # no room for a dummy object
  # neutrality under the reference's arithmetic (generator in eval mode: mask_net's BatchNorm then
  # has no cross-object coupling; in training mode the HIP kernels take the true count instead)
  vocab = make_vocab(184, 7)
  from sg2im_amd.trainer import GENERATOR_DEFAULTS
  cfg = dict(GENERATOR_DEFAULTS, vocab=vocab, layout_noise_dim=0)
  P = orc.init_generator_params(cfg, 0)
  with torch.no_grad():
    a = orc.generator_forward(P, cfg, batch[1], batch[4], batch[5], boxes_gt=batch[2], masks_gt=batch[3],
                              training=False)
    c = orc.generator_forward(P, cfg, objs, triples, o2i, boxes_gt=boxes, masks_gt=masks, training=False)
  assert torch.equal(a[0], c[0])                               # images: bit-identical
  assert torch.equal(a[1], c[1][:O])                           # boxes of the real objects
  assert torch.equal(a[3], c[3][:T])                           # relationship scores of the real triples


def test_dp_reference_with_one_rank_is_the_oracle_step():
  """tests/dp_reference.dp_step (the CPU reference tests/test_dp_rccl.py checks two GPUs against) with
  a single rank must be OracleTrainer.step; with two ranks holding the same shard, too (the mean of two
  identical gradients)."""
  from tests.dp_reference import dp_step
  from sg2im_amd.trainer import GENERATOR_DEFAULTS, D_OBJ_DEFAULTS, D_IMG_DEFAULTS
  vocab = make_vocab(20, 5)
  gcfg = dict(GENERATOR_DEFAULTS, vocab=vocab, layout_noise_dim=0, image_size=(16, 16), refinement_dims=(32, 16),
              gconv_num_layers=2, gconv_hidden_dim=32, gconv_dim=16, embedding_dim=16, mask_size=4)
  docfg = dict(D_OBJ_DEFAULTS, vocab=vocab, arch='C4-8-2,C4-16-2', object_size=16)
  dicfg = dict(D_IMG_DEFAULTS, arch='C4-8-2,C4-16-2')
  batch = tuple(synthetic_batch(2, image_size=(16, 16), num_objs=20, num_preds=5, mask_size=4, seed=3)[:6])

  def make():
    return orc.OracleTrainer(orc.init_generator_params(gcfg, 0, randomize_bn=True),
                             orc.init_ac_discriminator_params(docfg, 2, randomize_bn=True),
                             orc.init_patch_discriminator_params(dicfg, 1, randomize_bn=True), gcfg, docfg, dicfg)
  ref = make()
  want = ref.step(batch)
  one = make()
  got = dp_step([one], [batch])[0]
  for k, v in want.items():
    assert abs(got[k] - v) <= 1e-6 * max(1.0, abs(v)), k
  two = [make(), make()]
  dp_step(two, [batch, batch])
  for P, Q, R, S in ((ref.PG, one.PG, two[0].PG, two[1].PG), (ref.PDo, one.PDo, two[0].PDo, two[1].PDo),
                     (ref.PDi, one.PDi, two[0].PDi, two[1].PDi)):
    for k in P:
      assert torch.allclose(P[k], Q[k], atol=1e-7), k
      assert torch.allclose(P[k], R[k], atol=2e-6) and torch.equal(R[k], S[k]), k


def test_static_batch_refill_equals_pad_batch_on_cpu():
  """StaticBatch.load (three multi-tensor copies, one per dtype) leaves exactly what pad_batch builds in the
  static buffers - for batches of different object / triple counts sharing a bucket, one without triples,
  refilled in any order (the GPU form of this test is in tests/test_gpu_parity.py)."""
  from sg2im_amd.bucketing import StaticBatch, pad_batch
  cpu = [tuple(synthetic_batch(4, seed=s, min_objs=lo, max_objs=hi)[:6]) for s, lo, hi in ((1, 3, 4), (2, 5, 6), (3, 7, 8))]
  empty = list(cpu[0])
  empty[4] = empty[4][:0]
  cpu.append(tuple(empty))
  o_pad, t_pad = 64, 128
  sb = StaticBatch(cpu[0], o_pad, t_pad)
  for b in cpu[1:] + cpu[:1] + cpu[2:3]:
    sb.load(b)
    want, counts = pad_batch(b, o_pad, t_pad)
    for got, ref in zip(sb.tensors(), want):
      assert (got is None) == (ref is None)
      if ref is not None:
        assert got.dtype == ref.dtype and torch.equal(got, ref)
    assert torch.equal(sb.counts, counts)
  with pytest.raises(ValueError):
    sb.load(tuple(synthetic_batch(16, seed=9)[:6]))           # another image count: not this bucket


def test_generator_gradient_buckets_partition_the_arena():
  """Data-parallel schedule 2 (sg2im_amd/trainer.py::_capture_overlapped) exchanges the generator's gradient arena in
  FOUR buckets: three early slices - refinement module 0, module 1, modules 2.. + the output convolutions, in the
  order the released weight gradients complete them - and, after the backward pass, the gaps they leave
  (trainer.complement).  Together they must cover every element exactly once, each early slice must hold exactly its
  modules' parameters, no parameter may straddle a boundary, the completion tags must be exactly the slice's
  convolution weights, and the three early slices together must be exactly the refinement network's slice - the
  condition for its early Adam update under data parallelism (VERDICT r3 weak #1d, r4 item 7)."""
  import torch
  from sg2im_amd.model import Sg2ImModel
  from sg2im_amd.optim import FlatParams
  from sg2im_amd.synthetic import make_vocab
  from sg2im_amd.trainer import GENERATOR_DEFAULTS, complement, generator_bucket, generator_buckets, refinement_slice
  model = Sg2ImModel(**dict(GENERATOR_DEFAULTS, vocab=make_vocab(184, 7), refinement_dims=(64, 32, 16, 16, 8), gconv_dim=32,
                            gconv_hidden_dim=64, embedding_dim=32))
  flat = FlatParams(model)
  try:
    bk = generator_buckets(model, flat)
    assert len(bk) == 3
    early = [(a, b) for a, b, _ in bk]
    rest = complement(early, flat.numel)
    cover = torch.zeros(flat.numel, dtype=torch.int32)
    for a, b in early + rest:
      assert 0 <= a < b <= flat.numel and a % 4 == 0
      cover[a:b] += 1
    assert int(cover.min()) == 1 and int(cover.max()) == 1
    net = model.refinement_net
    mods = net.refinement_modules
    groups = [[mods[0]], [mods[1]], list(mods[2:]) + [net.output_conv]]
    for (a, b, ids), grp in zip(bk, groups):                     # (arena order = completion order: module 0 first)
      mine = {p.data_ptr() for m in grp for p in m.parameters()}
      inside = {p.data_ptr() for p, off in zip(flat.params, flat.offsets) if a <= off < b}
      assert inside == mine
      assert ids == {p.data_ptr() for m in grp for p in m.parameters() if p.dim() == 4}
    for p, off in zip(flat.params, flat.offsets):                  # no parameter straddles a bucket boundary
      assert sum(1 for a, b in early + rest if a <= off and off + p.numel() <= b) == 1
    assert refinement_slice(model, flat) == (early[0][0], early[-1][1])
    assert all(early[i][1] == early[i + 1][0] for i in range(2))
    # the rounds-3/4 two-module slice is the union of the first two buckets
    a, b, ids = generator_bucket(model, flat)
    assert (a, b) == (early[0][0], early[1][1]) and len(ids) == 4
    # fewer than three modules: one bucket
    small = Sg2ImModel(**dict(GENERATOR_DEFAULTS, vocab=make_vocab(184, 7), refinement_dims=(16, 8), gconv_dim=32,
                              gconv_hidden_dim=64, embedding_dim=32, image_size=(8, 8)))
    fs = FlatParams(small)
    assert generator_buckets(small, fs) == [] and generator_bucket(small, fs) is None
    assert complement([], fs.numel) == [(0, fs.numel)]
    fs.close()
  finally:
    flat.close()
  assert complement([(8, 12), (0, 4)], 16) == [(4, 8), (12, 16)]
  assert complement([(0, 16)], 16) == []


def test_deferred_release_reports_each_bucket_only_when_all_its_weight_gradients_were_issued():
  """ops.SideLane.flush: a bucket's early-exchange callback fires right after the LAST tagged launch of that bucket,
  buckets fire in completion order, and a bucket with a tagged parameter that is not among the released launches
  never fires (no hard-coded count)."""
  import torch
  from sg2im_amd import ops

  ps = [torch.zeros(2) for _ in range(5)]
  issued, fired = [], []
  lane = ops.SideLane.__new__(ops.SideLane)              # (the queue without streams)
  lane.on, lane.used, lane.keep, lane.queue, lane.deferring = False, False, [], [], False
  for i, p in enumerate(ps):
    lane.defer(lambda bg, i=i: issued.append(i), completes=(p, None))
  try:
    ops.AFTER_DEFERRED = [(frozenset({ps[4].data_ptr(), ps[2].data_ptr()}), lambda stream: fired.append(('A', len(issued)))),
                          (frozenset({ps[1].data_ptr()}), lambda stream: fired.append(('B', len(issued)))),
                          (frozenset({ps[3].data_ptr(), torch.zeros(1).data_ptr()}), lambda stream: fired.append(('C', len(issued))))]
    ops.release_deferred(lane.queue, 'stream')
    assert issued == [4, 3, 2, 1, 0]      # released in reverse order
    # A right after ps[2]'s launch (the last of its bucket), B after ps[1]'s; C holds a parameter that is never
    # released: no early exchange for it
    assert fired == [('A', 3), ('B', 4)]
    del issued[:], fired[:]
    ops.AFTER_DEFERRED = None
    ops.release_deferred(lane.queue, 'stream')
    assert issued == [4, 3, 2, 1, 0] and fired == []
    # a queue released in two parts (SideLane.flush(final=False) then flush()): a bucket whose launches straddle the
    # two releases fires with the LAST of them, buckets are not re-armed in between
    del issued[:], fired[:]
    pending = [[{ps[4].data_ptr(), ps[1].data_ptr()}, lambda stream: fired.append(('A', len(issued)))],
               [{ps[0].data_ptr()}, lambda stream: fired.append(('B', len(issued)))]]
    ops.release_deferred(lane.queue[:2], 'stream', pending)         # launches 1, 0
    assert issued == [1, 0] and fired == [('B', 2)]
    ops.release_deferred(lane.queue[2:], 'stream', pending)         # launches 4, 3, 2
    assert issued == [1, 0, 4, 3, 2] and fired == [('B', 2), ('A', 3)]
  finally:
    ops.AFTER_DEFERRED = None


def test_layout_link_refuses_a_gradient_that_is_not_the_handed_over_tensor():
  """functional.LayoutLink: the refinement network returns an UNWRITTEN tensor as the layout's gradient
  and leaves the per-level gradients in the link; LayoutFn.backward may only interpret that tensor through the link
  if it IS the tensor that was handed over - a sum / copy made by autograd (second consumer, hook, retain_grad) or an
  in-place modification must raise instead of silently back-propagating uninitialised memory."""
  import torch
  from sg2im_amd import functional as HF
  link = HF.LayoutLink()
  layout, levels = torch.zeros(1, 4, 4, 8), [torch.zeros(1, 4, 4, 8), torch.zeros(1, 2, 2, 8)]
  assert link.take_pyramid(layout) is None                      # nothing offered yet
  link.offer(layout, levels)
  assert link.take_pyramid(torch.zeros(1, 4, 4, 8)) is None     # another tensor: not its pyramid
  assert link.take_pyramid(layout) is levels and link.taken
  assert link.take_pyramid(layout) is None                      # handed over once
  d = torch.empty(1, 4, 4, 8)
  glev = [torch.ones(1, 2, 2, 8)]
  link.leave_grad(d, glev, [2], 8)
  with pytest.raises(RuntimeError):
    link.leave_grad(d, glev, [2], 8)                            # a pending gradient that nobody consumed
  assert link.take_grad(d) == (glev, [2], 8)
  assert link.take_grad(d) is None                              # consumed
  link.leave_grad(d, glev, [2], 8)
  with pytest.raises(RuntimeError):
    link.take_grad(d + 0)                                       # what autograd would pass on with a second consumer
  link.leave_grad(d, glev, [2], 8)
  d.add_(1.0)
  with pytest.raises(RuntimeError):
    link.take_grad(d)                                           # modified in place after the hand-over


def test_shared_pass_adoption_needs_no_kernel_and_rejects_a_foreign_recording():
  """functional.SharedPass: the second user of a recorded discriminator pass (the discriminator step, over the
  generated images the generator loss already ran through the same weights) builds its autograd node around the
  recorded activations without launching anything - so this runs on the CPU - and a recording made by another
  network / mode is refused."""
  from sg2im_amd import functional as HF
  specs = [(4, 8, 2, 0), (4, 16, 2, 0)]
  y0, y1 = torch.randn(2, 7, 7, 8), torch.randn(2, 2, 2, 16)
  sp = HF.SharedPass()
  assert not sp.recorded
  sp.saved = [(None, None, y0, None, 16, 16, None, None), (None, None, y1, None, 7, 7, None, None)]
  sp.misc = (specs, 0.2, 2, (2, 16, 16, 3), True, False, None)       # (training = 2: recorded as a pass that counts twice)
  assert sp.recorded
  params = [torch.randn(8, 4, 4, 3, requires_grad=True), torch.randn(8, requires_grad=True),
            torch.randn(16, 4, 4, 8, requires_grad=True), torch.randn(16, requires_grad=True)]
  out = HF.DiscCnnFn.apply(None, None, specs, 0.2, True, None, sp, *params)
  assert out.data_ptr() == y1.data_ptr() and out.requires_grad and out.grad_fn is not None
  with pytest.raises(RuntimeError):
    HF.DiscCnnFn.apply(None, None, specs, 0.1, True, None, sp, *params)          # another activation slope
  with pytest.raises(RuntimeError):
    HF.DiscCnnFn.apply(None, None, specs[:1], 0.2, True, None, sp, *params[:2])  # another architecture
  with pytest.raises(RuntimeError):
    HF.DiscCnnFn.apply(None, None, specs, 0.2, False, None, sp, *params)         # eval mode


def test_one_launch_gcn_backward_is_chosen_where_mask_net_trains():
  """Trainer._gcn_backward_mode (SG2IM_GCN_PERSIST_BWD=auto): the low-footprint one-launch GraphTripleConv backward
  for the steps whose main-lane tail carries mask_net's backward - no ground-truth masks in the batch (VG style,
  model.py:146-157) or a mask loss (train.py:407-410) - and the launch sequence otherwise."""
  from types import SimpleNamespace
  from sg2im_amd.trainer import Trainer
  with_net = SimpleNamespace(model=SimpleNamespace(mask_net=object()), w={'mask_loss_weight': 0.0}, compute_dtype='f32')
  no_net = SimpleNamespace(model=SimpleNamespace(mask_net=None), w={'mask_loss_weight': 0.0}, compute_dtype='f32')
  coco = (0, 1, 2, torch.zeros(3, 16, 16), torch.zeros(400, 3), 5)
  vg = (0, 1, 2, None, torch.zeros(500, 3), 5)
  vg_large = (0, 1, 2, None, torch.zeros(2900, 3), 5)         # (the 256 x 256 shape: thousands of triples)
  mode = Trainer._gcn_backward_mode
  assert mode(with_net, coco) is False and mode(with_net, vg) == 'low' and mode(with_net, vg_large) is False
  assert mode(no_net, vg) is False and mode(no_net, coco) is False
  with_net.w['mask_loss_weight'] = 0.1
  assert mode(with_net, coco) == 'low'
  # bf16 mode, COCO style: the same stages as 25 ordinary launches (no resident workgroups) - measured, trainer.py
  no_net.compute_dtype = 'bf16'
  with_net.compute_dtype, with_net.w['mask_loss_weight'] = 'bf16', 0.0
  assert mode(no_net, coco) == 'staged' and mode(with_net, coco) == 'staged' and mode(with_net, vg) == 'low'


def test_layout_backward_takes_both_halves_from_the_level_gradients_or_materialises_once(monkeypatch):
  """LayoutFn.backward's routing (functional.py), with the kernels replaced by recorders: with the
  refinement network's per-level gradients in the link, d_vecs AND d_masks / d_boxes come straight from the levels; a
  shape the mask / box kernel declines materialises the sum ONCE and computes only what is still missing; without a
  link everything goes through the materialised path."""
  from types import SimpleNamespace
  from sg2im_amd import functional as HF
  from sg2im_amd import ops
  calls = []
  declines = {'maps': False}
  monkeypatch.setattr(ops, 'mark', lambda name: None)
  monkeypatch.setattr(ops, 'layout_backward_vecs_levels', lambda *a: calls.append(('vecs_levels',)))
  monkeypatch.setattr(ops, 'layout_backward_maps_levels',
                      lambda *a: (calls.append(('maps_levels', a[-2] is not None, a[-1] is not None)), not declines['maps'])[1])
  monkeypatch.setattr(ops, 'pyramid_backward', lambda *a: calls.append(('pyramid',)))
  monkeypatch.setattr(ops, 'layout_backward',
                      lambda g, v, b, m, o2i, csr, n, H, W, ac, dv, dm, db=None: calls.append(
                        ('layout_backward', dv is not None, dm is not None, db is not None)))
  N, H, D, O, M = 2, 16, 8, 3, 4
  vecs, boxes, masks = torch.randn(O, D), torch.rand(O, 4), torch.rand(O, M, M)
  o2i = torch.tensor([0, 0, 1])
  levels = [torch.randn(N, H >> l, H >> l, D) for l in range(2)]

  def run(needs, with_link):
    del calls[:]
    link = None
    g = torch.empty(N, H, H, D)
    if with_link:
      link = HF.LayoutLink()
      link.leave_grad(g, levels, [1, 2], D)
    ctx = SimpleNamespace(saved_tensors=(vecs, boxes, masks, o2i), geom=(N, H, H, False), needs_input_grad=needs,
                          link=link, img_csr=None)
    out = HF.LayoutFn.backward(ctx, g)
    assert (out[0] is not None, out[1] is not None, out[2] is not None) == tuple(needs[:3])
    return list(calls)

  vg = (True, False, True) + (False,) * 9            # VG style: vectors and soft masks need gradients
  assert run(vg, True) == [('vecs_levels',), ('maps_levels', True, False)]
  assert run((True, False, False) + (False,) * 9, True) == [('vecs_levels',)]          # COCO style
  assert run((True, True, True) + (False,) * 9, True) == [('vecs_levels',), ('maps_levels', True, True)]
  declines['maps'] = True                            # (e.g. a vector width the tiled kernel does not take)
  assert run(vg, True) == [('vecs_levels',), ('maps_levels', True, False), ('pyramid',),
                           ('layout_backward', False, True, False)]
  assert run(vg, False) == [('layout_backward', True, True, False)]


def test_dp_schedule_choice_follows_the_capture_probe():
  """sg2im_amd/capture_probe.py::choose_dp_schedule (Trainer.__init__ at world_size > 1): schedule 2 only when the
  process group can be captured AND the subprocess probe passed on every rank; a failed probe selects schedule 1, a
  group that cannot be captured (gloo) schedule 0; pinned schedules 0 / 1 never probe."""
  from sg2im_amd.capture_probe import choose_dp_schedule
  calls = []

  def probe_ok():
    calls.append('ok')
    return True

  def probe_bad():
    calls.append('bad')
    return False
  assert choose_dp_schedule(2, True, probe_ok) == 2
  assert choose_dp_schedule(2, True, probe_bad) == 1
  assert choose_dp_schedule(2, True, probe_ok, agree_fn=lambda ok: False) == 1       # another rank's probe failed
  assert choose_dp_schedule(2, True, probe_ok, agree_fn=lambda ok: ok) == 2
  n = len(calls)
  assert choose_dp_schedule(2, False, probe_bad) == 0 and choose_dp_schedule(0, True, probe_bad) == 0
  assert choose_dp_schedule(1, True, probe_bad) == 1 and len(calls) == n              # (no probe for those)


def test_capture_probe_expected_values_follow_the_replayed_pattern():
  """capture_probe.expected_values(world): the child's pattern (capture_probe._child) simulated on the host with an
  all-reduce that sums `world` equal copies - the values the multi-rank probe of a data-parallel job checks"""
  from sg2im_amd.capture_probe import expected_values, group_port
  for world in (1, 2, 8):
    g_lo = g_hi = d = guard = 1.0
    w_lo = w_hi = 0.0
    g_lo, g_hi, d, guard = g_lo * world, g_hi * world, d * world, guard * world      # communicator set-up reductions
    for _ in range(2):                                                               # two replays
      w_lo += 1; w_hi += 1
      guard *= world
      d = d * 2 * world
      w_lo += 1; w_hi += 1
      g_lo = g_lo * 2 * world
      g_hi = g_hi * 3
      w_lo += 1; w_hi += 1
      g_hi *= world
      w_lo += g_lo; w_hi += g_hi
    assert expected_values(world) == (w_lo, w_hi, d, guard)
  assert expected_values(1) == (12.0, 18.0, 4.0, 1.0)
  assert group_port(29500) == 29529 and group_port('64990') == 64961 and group_port(29500) != 29500


def test_capture_probe_child_that_dies_is_a_failed_probe_not_a_crash():
  """the probe runs in a subprocess precisely because the failure it guards against is a segfault: a child that aborts
  (SG2IM_PROBE_FORCE_FAIL=1, before it touches any GPU) must come back as False"""
  import os
  from sg2im_amd import capture_probe
  capture_probe._verdict.clear()
  os.environ['SG2IM_PROBE_FORCE_FAIL'] = '1'
  try:
    assert capture_probe.probe(0, timeout=120) is False
    capture_probe._verdict.clear()
    assert capture_probe.probe(0, timeout=120, rank=1, world_size=2, master_port=29500) is False     # (the multi-rank form)
  finally:
    del os.environ['SG2IM_PROBE_FORCE_FAIL']
    capture_probe._verdict.clear()
