"""CPU, world_size 2, gloo: the data-parallel exchange.  Each rank takes its shard of a
collated batch (sg2im_amd.synthetic.shard_batch), computes the reference's per-shard
gradients with the CPU oracle, packs them into the flat gradient arena of the product's own
module classes (sg2im_amd.optim.FlatParams) and runs the product's asynchronous reducer.
Expected: arena * grad_scale == mean over ranks of the per-shard
reference gradients, identical on every rank; the NaN guard is collective."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.util import load_golden


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def _worker(rank, world, port, ret):
  os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    from oracle import sg2im_oracle as orc
    from sg2im_amd.discriminators import PatchDiscriminator
    from sg2im_amd.distributed import GradReducer
    from sg2im_amd.optim import FlatParams
    from sg2im_amd.synthetic import shard_batch
    from tests.hip_harness import load_params
    torch.manual_seed(0)
    fix = load_golden('tiny_coco')
    dicfg = dict(fix['config']['d_img'])
    P = fix['state_before']['Di']
    full = fix['batch']
    # pad the 3-image golden batch to 4 images so it splits over 2 ranks
    imgs = torch.cat([full[0], full[0][:1]], 0)
    fake = torch.cat([fix['outputs']['imgs_pred'], fix['outputs']['imgs_pred'][:1]], 0)
    lo, hi = rank * 2, rank * 2 + 2

    def shard_grads(r):
      Pl = {k: v.clone().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in P.items()}
      sf = orc.patch_discriminator(Pl, dicfg, fake[r * 2:r * 2 + 2], True)
      sr = orc.patch_discriminator(Pl, dicfg, imgs[r * 2:r * 2 + 2], True)
      orc.gan_d_loss(sr, sf).backward()
      return {k: v.grad for k, v in Pl.items() if v.requires_grad and v.grad is not None}

    mine = shard_grads(rank)
    module = load_params(PatchDiscriminator(**dicfg), P)
    flat = FlatParams(module)
    for k, p in module.named_parameters():
      if k in mine:
        p.grad.copy_(mine[k])
    red = GradReducer()
    assert red.world_size == world
    red.start(flat.grad)
    guard = torch.tensor([float('nan') if rank == 1 else 1.0])
    red.start(guard)
    red.finish()
    want = [shard_grads(r) for r in range(world)]
    worst = 0.0
    for k, p in module.named_parameters():
      if k in mine:
        mean = sum(w[k] for w in want) / world
        worst = max(worst, float((p.grad * red.grad_scale - mean).abs().max()))
    # the same exchange with the bfloat16 payload (GradReducer(payload='bf16'): what the bf16 training mode sends):
    # every element within bfloat16 rounding of the exact mean - of each rank's value (2^-9 relative) and of the sum
    for k, p in module.named_parameters():
      if k in mine:
        p.grad.copy_(mine[k])
    redh = GradReducer(payload='bf16')
    redh.start(flat.grad)
    redh.start(guard.clone())             # (a one-element tensor - the NaN guard - travels as it is)
    redh.finish()
    worst_h = 0.0
    for k, p in module.named_parameters():
      if k in mine:
        mean = sum(w[k] for w in want) / world
        bound = sum(w[k].abs() for w in want) / world * 2.0 ** -7 + 1e-12
        worst_h = max(worst_h, float(((p.grad * redh.grad_scale - mean).abs() / bound).max()))
    assert flat.grad.dtype == torch.float32 and worst_h <= 1.0, worst_h
    # the DIRECT exchange (GradReducer(exchange='direct'): all-to-all of shards, local fp32 sum, all-gather - SURVEY.md
    # section 5's reduce-scatter / all-gather over all links): the same sums as the all-reduce, on every rank the same bits
    for k, p in module.named_parameters():
      if k in mine:
        p.grad.copy_(mine[k])
    redd = GradReducer(exchange='direct')
    redd.start(flat.grad)
    gd = guard.clone() if rank == 0 else torch.tensor([1.0])
    redd.start(gd)                         # (one element over two ranks: a padded shard)
    redd.finish()
    worst_d = 0.0
    for k, p in module.named_parameters():
      if k in mine:
        worst_d = max(worst_d, float((p.grad - sum(w[k] for w in want)).abs().max()))
    assert worst_d <= 1e-7 * max(1.0, float(flat.grad.abs().max())), worst_d
    assert not bool(torch.isfinite(gd).all())              # (rank 0's NaN reached everybody)
    # ... with the bfloat16 payload: each shard rounded once, summed in fp32, the sum rounded once - EXACTLY
    # bf16(sum_r fp32(bf16(shard_r))), a tighter statement than the all-reduce form's bound above; length not a multiple of 2
    g = torch.Generator().manual_seed(5)
    shards = [torch.randn(1001, generator=g) * 3.0 for _ in range(world)]
    x = shards[rank].clone()
    reddh = GradReducer(payload='bf16', exchange='direct')
    reddh.start(x)
    reddh.finish()
    exact = sum(sh.bfloat16().float() for sh in shards).bfloat16().float()
    assert torch.equal(x, exact), float((x - exact).abs().max())
    allx = [torch.empty_like(x) for _ in range(world)]
    dist.all_gather(allx, x)
    assert all(torch.equal(a, x) for a in allx)            # bit-identical replicas
    ret[rank] = (worst, bool(torch.isfinite(guard).all()))
    # shard_batch on the 4-image synthetic batch gives each rank 2 whole images
    from sg2im_amd.synthetic import synthetic_batch
    sh = shard_batch(synthetic_batch(4, seed=1), rank, world)
    assert sh[0].size(0) == 2 and int(sh[5].max()) == 1
  finally:
    dist.destroy_process_group()


def test_flat_gradient_allreduce_world_size_2():
  world, port = 2, _free_port()
  ret = mp.Manager().dict()
  mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
  assert len(ret) == world
  for rank in range(world):
    worst, guard_finite = ret[rank]
    assert worst <= 1e-7, (rank, worst)
    assert guard_finite is False          # a NaN loss on ONE rank makes EVERY rank skip the update


def test_bf16_sum_of_eight_bf16_shards():
  """The bfloat16 gradient payload of the bf16 training mode (sg2im_amd/distributed.py: GradReducer(payload='bf16'))
  is SUMMED IN bfloat16 by RCCL across the ranks - an 8-way bfloat16 accumulation per element (VERDICT r4 weak #8 i).
  Its error, stated a priori: every shard is rounded once (relative 2^-9) and each of the 7 ring additions rounds the
  running sum once more, so an element of the mean is off by at most ~ (1 + 7) x 2^-9 of the LARGEST partial sum of
  that element - for a tensor that is <= 1.6e-2 of its largest mean-gradient magnitude times the partial-sum growth.
  Checked on per-rank gradients with the structure data-parallel shards have (a common signal plus per-shard
  variation of the same size): the bfloat16 ring sum x 1/8 stays within rel-to-max 2e-2 / cosine 0.9999 of the fp32
  mean of the fp32 shards - an order of magnitude inside the matrix-gradient bound of the bf16 mode (rel 0.15,
  cos 0.99).  Adversarial case, shards of alternating sign (the mean is ~20x smaller than the partial sums): the
  error follows the derived bound world x 2^-9 x (largest partial sum / largest sum) and is still inside 0.15."""
  torch.manual_seed(0)
  world = 8
  for n, cancel in ((1 << 16, False), (1 << 16, True), (9 * 1024 * 64, False)):
    base = torch.randn(n) * torch.rand(n).pow(4)              # heavy-tailed magnitudes, like weight gradients
    shards = [base + torch.randn(n) * base.abs().mean() for _ in range(world)]
    if cancel:
      shards = [s * (1.0 if r % 2 == 0 else -0.9) for r, s in enumerate(shards)]
    want = torch.stack(shards).double().mean(0)
    # ring reduce-scatter order for every element: rank r's contribution added to the running bfloat16 sum in rank
    # order starting at an element-dependent rank - any fixed rotation has the same bound; two rotations are checked
    for start in (0, 3):
      acc = shards[start].bfloat16()
      exact = shards[start].double()
      growth = exact.abs().max()
      for k in range(1, world):
        acc = (acc + shards[(start + k) % world].bfloat16())          # (bfloat16 + bfloat16 -> rounded to bfloat16)
        exact = exact + shards[(start + k) % world].double()
        growth = torch.maximum(growth, exact.abs().max())
      got = acc.float().double() / world
      rel = float((got - want).abs().max() / want.abs().max())
      cos = float(torch.dot(got, want) / (got.norm() * want.norm()))
      derived = world * 2.0 ** -9 * float(growth / (want.abs().max() * world))
      assert rel <= derived, (n, cancel, start, rel, derived)
      if cancel:
        assert rel <= 0.15 and cos >= 0.999, (n, start, rel, cos)
      else:
        assert rel <= 2e-2 and cos >= 0.9999, (n, start, rel, cos)


def _direct_worker(rank, world, port, ret):
  os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    from sg2im_amd.distributed import GradReducer
    g = torch.Generator().manual_seed(17)
    ok = True
    for n in (1, 5, 8, 1003, 4096):                  # shorter than the group, not a multiple of it, a multiple
      shards = [torch.randn(n, generator=g) * (1.0 + r) for r in range(world)]
      for payload in ('f32', 'bf16'):
        x = shards[rank].clone()
        red = GradReducer(payload=payload, exchange='direct')
        red.start(x)
        red.finish()
        if payload == 'f32' or n == 1:              # (a one-element tensor - the NaN guard - travels as fp32)
          want = torch.zeros(n)
          for sh in shards:                          # rank order, fp32: the local sum's order
            want = want + sh
        else:
          acc = torch.zeros(n)
          for sh in shards:
            acc = acc + sh.bfloat16().float()
          want = acc.bfloat16().float()
        ok = ok and torch.equal(x, want)
        # a second use of the same reducer and tensor re-uses its buffers (the pad must still be zero)
        y = shards[rank].clone()
        x.copy_(y)
        red.start(x)
        red.finish()
        ok = ok and torch.equal(x, want)
    ret[rank] = ok
  finally:
    dist.destroy_process_group()


def test_direct_exchange_world_size_8():
  """GradReducer(exchange='direct') with EIGHT ranks (the node size the exchange is designed for): lengths below,
  across and at multiples of the group size; fp32: the rank-ordered sum, bit for bit on every rank; bfloat16 payload:
  exactly bf16(sum_r fp32(bf16(shard_r)))."""
  world, port = 8, _free_port()
  ret = mp.Manager().dict()
  mp.spawn(_direct_worker, args=(world, port, ret), nprocs=world, join=True)
  assert len(ret) == world and all(ret[r] for r in range(world)), dict(ret)
