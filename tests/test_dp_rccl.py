"""GPU, 2 ranks: two HIP training iterations of the data-parallel Trainer - one process per rank, each
on its own, differently shaped shard of a collated batch - against the CPU reference of
tests/dp_reference.py: per step the losses, after the FIRST step every gradient
arena x 1/world against the mean over ranks of the reference's per-shard gradients (catches a missing
reduce as well as a missing 1/world: Adam alone is blind to a constant factor), identical parameters on
both ranks - in the eager form and in both hipGraph schedules of Trainer._capture.

  * test_two_ranks_on_one_gpu_*: both ranks on THIS box's single MI355X, exchange over gloo (staged
    through the host, sg2im_amd/distributed.py) - so that broadcast_state, the split iteration / Adam
    graphs and the reducer execute on the hardware a 1-GPU box has;
  * test_two_rank_rccl_*: one GPU per rank over RCCL (skipped with fewer than two GPUs)."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def _worker(rank, world, port, backend, share_gpu, use_graphs, dp_schedule, ret, exchange='allreduce', nonorm=False):
  import torch.distributed as dist
  os.environ['SG2IM_DP_EXCHANGE'] = exchange
  os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
  os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  torch.set_num_threads(8)       # (two workers run CPU oracles side by side: no oversubscription)
  idx = 0 if share_gpu else rank
  torch.cuda.set_device(idx)
  dev = torch.device('cuda', idx)
  if backend == 'nccl':
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
  else:
    dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    from oracle import sg2im_oracle as orc
    from sg2im_amd.synthetic import make_vocab, shard_batch, synthetic_batch
    from sg2im_amd.trainer import Trainer, GENERATOR_DEFAULTS, D_OBJ_DEFAULTS, D_IMG_DEFAULTS
    from tests.dp_reference import dp_step
    from tests.hip_harness import check_grad_rows, grad_parity_rows3, load_params, oracle_trainer
    vocab = make_vocab(184, 7)
    gk = {'layout_noise_dim': 0}
    dk = {}
    per_shard = 4
    if nonorm:
      # the well-conditioned configuration (no normalisation layer anywhere: nothing amplifies fp32 rounding), 8 images
      # per shard: every gradient is held to 1e-4 of its tensor's max, no reference-noise term
      gk, dk, per_shard = dict(gk, normalization='none'), {'normalization': 'none'}, 8
    gcfg = dict(GENERATOR_DEFAULTS, vocab=vocab, **gk)
    docfg, dicfg = dict(D_OBJ_DEFAULTS, vocab=vocab, **dk), dict(D_IMG_DEFAULTS, **dk)
    lr = 1e-4
    P0 = (orc.init_generator_params(gcfg, 0, randomize_bn=True), orc.init_ac_discriminator_params(docfg, 2, randomize_bn=True),
          orc.init_patch_discriminator_params(dicfg, 1, randomize_bn=True))
    refs = [oracle_trainer(*P0, gcfg, docfg, dicfg, torch.float32, lr=lr) for _ in range(world)]
    # the EXACT mean-of-per-shard gradients (float64), first step only: see tests/hip_harness.py::assert_grad_parity
    refs64 = [oracle_trainer(*P0, gcfg, docfg, dicfg, torch.float64, lr=lr) for _ in range(world)]
    # a deliberately different seed per rank: Trainer.broadcast_state must bring the replicas in line
    tr = Trainer(vocab, dev, seed=100 + rank, generator_kwargs=gk, d_obj_kwargs=dk, d_img_kwargs=dk, learning_rate=lr,
                 world_size=world, rank=rank, use_graphs=use_graphs, bucket=(8, 16), dp_schedule=dp_schedule)
    if rank == 0:
      load_params(tr.model, refs[0].PG); load_params(tr.d_obj, refs[0].PDo); load_params(tr.d_img, refs[0].PDi)
    tr.broadcast_state()
    worst_loss, worst_grad, worst_ref, bad = 0.0, 0.0, 0.0, []
    for step in range(2):
      full = synthetic_batch(per_shard * world, seed=40 + step)       # (>= 4 images per shard: the fp32 oracle's own error, which
                                                              # sets the bound, is then <= 1e-2 - VERDICT r3 weak #1c)
      shards = [tuple(shard_batch(full, r, world)[:6]) for r in range(world)]
      assert len(set((s[1].numel(), s[4].size(0)) for s in shards)) == world        # differently shaped shards
      mine = tuple(t.to(dev) if torch.is_tensor(t) else t for t in shards[rank])
      got = Trainer.losses_to_host(tr.step(mine))
      want = dp_step(refs, shards)[rank]
      for k, v in want.items():
        worst_loss = max(worst_loss, abs(got[k] - v) / max(1.0, abs(v)))
      if step == 0:
        # arena (SUM over ranks) x 1/world == mean over ranks of the reference's per-shard gradients
        from tests.hip_harness import cast_batch
        dp_step(refs64, [cast_batch(s, torch.float64) for s in shards])
        rows = grad_parity_rows3(tr, refs[rank], refs64[rank], scale=tr.reducer.grad_scale)
        bad, summ = check_grad_rows(rows, cos_min=0.999999, strict=True, noise_factor=1.0) if nonorm else check_grad_rows(rows)
        # (the summary also holds '<net>:matrices' rows whose third field is a cosine: only the per-network rows
        # carry (worst e_hip64, tensor, E_ref, ...) - as hip_harness.assert_grad_parity reads them)
        nets = [v for k, v in summ.items() if ':' not in k]
        worst_grad = max(v[0] for v in nets)
        worst_ref = max(v[2] for v in nets)
    torch.cuda.synchronize()
    # identical parameters on every rank
    from sg2im_amd.distributed import broadcast
    flat = tr.flat_g.flat.clone()
    broadcast(flat, 0)
    same = bool(torch.equal(flat, tr.flat_g.flat))
    ret[rank] = (worst_loss, (worst_grad, worst_ref),
                 ['%s.%s e_hip64 %.3e e_ref %.3e e_hip32 %.3e max|g| %.3e' % r[:6] for r in bad[:10]], same, dict(tr.graph_stats))
  finally:
    dist.destroy_process_group()


def _run(backend, share_gpu, use_graphs, dp_schedule, exchange='allreduce', nonorm=False):
  import torch.multiprocessing as mp
  world, port = 2, _free_port()
  ret = mp.Manager().dict()
  mp.spawn(_worker, args=(world, port, backend, share_gpu, use_graphs, dp_schedule, ret, exchange, nonorm), nprocs=world, join=True)
  assert len(ret) == world
  for rank in range(world):
    worst_loss, worst_grad, bad, same, stats = ret[rank]
    line = 'dp 2 ranks %s%s graphs=%s schedule=%s' % (backend, ' (one GPU)' if share_gpu else '', use_graphs, dp_schedule) + (
      ' exchange=direct' if exchange == 'direct' else '') + (' no-normalization b8/shard (bound max(1e-4, E_ref), cos 0.999999)' if nonorm else '') + ' rank %d: worst loss rel %.3e, worst e_hip64 %.3e (float32 oracle: E_ref %.3e)' % (
      rank, worst_loss, worst_grad[0], worst_grad[1])
    print(line)
    try:
      os.makedirs('gpurun_out', exist_ok=True)
      with open(os.path.join('gpurun_out', 'grad_parity.log'), 'a') as f:
        f.write(line + '\n')
    except OSError:
      pass
    assert worst_loss <= 5e-3, (rank, worst_loss)       # (step 2 sees parameters after one +-lr Adam step per side)
    assert not bad, (rank, bad)
    assert same, rank
    if use_graphs:
      assert stats['captures'] >= 1 and stats['replays'] == 2, stats


# (schedule 2 over gloo: its collectives cannot be captured, so the Trainer runs - and logs - schedule 0: covered as the
# fallback.  Schedule 2 proper - RCCL inside the graph - runs in the two-GPU test below and, on one GPU, in
# tests/test_gpu_parity.py::test_in_graph_exchange_reduces_every_gradient_exactly_once)
@pytest.mark.parametrize('use_graphs,dp_schedule', [(False, 0), (True, 0), (True, 1), (True, 2)])
def test_two_ranks_on_one_gpu_match_the_dp_reference(use_graphs, dp_schedule):
  _run('gloo', True, use_graphs, dp_schedule)


@pytest.mark.parametrize('use_graphs,dp_schedule,exchange', [(False, 0, 'allreduce'), (True, 0, 'allreduce'), (True, 1, 'allreduce'),
                                                             (True, 0, 'direct')])
def test_two_ranks_on_one_gpu_well_conditioned_step_to_1e4(use_graphs, dp_schedule, exchange):
  """the same with `normalization='none'` in the generator and both discriminators and 8 images per shard.  Without batch
  statistics the only ill-conditioning left is a LeakyReLU whose pre-activation changes sign under a 1e-6 perturbation
  (measured: the fp32 oracle is 5e-3 from float64 on such a tensor, tools/_dbg/di_trainer.py shows the HIP gradient equal
  to 1e-7 to the float64 gradient evaluated on the HIP image), so the bound is: EVERY gradient (arena x 1/world), however
  small the tensor, within max(1e-4, E_ref) of the float64 mean-of-per-shard gradients - at least as close as the
  reference's own fp32 arithmetic, no 3x factor, no flipped-decision escape - and cosine >= 0.999999; both exchanges
  (all-reduce, direct), eager and graph schedules."""
  _run('gloo', True, use_graphs, dp_schedule, exchange=exchange, nonorm=True)


@pytest.mark.parametrize('use_graphs,dp_schedule', [(False, 0), (True, 0), (True, 1), (True, 2)])
def test_two_rank_rccl_training_matches_the_dp_reference(use_graphs, dp_schedule):
  if torch.cuda.device_count() < 2:
    pytest.skip('needs two GPUs')
  _run('nccl', False, use_graphs, dp_schedule)


@pytest.mark.parametrize('use_graphs', [False, True])
def test_two_ranks_on_one_gpu_direct_exchange(use_graphs):
  """GradReducer(exchange='direct') - all-to-all of arena shards, local fp32 sum, all-gather (SG2IM_DP_EXCHANGE=direct;
  sg2im_amd/distributed.py) - under the Trainer: eager segments and the iteration graph + exposed exchange + Adam graph"""
  _run('gloo', True, use_graphs, 0, exchange='direct')


@pytest.mark.parametrize('use_graphs', [False, True])
def test_two_rank_rccl_direct_exchange(use_graphs):
  if torch.cuda.device_count() < 2:
    pytest.skip('needs two GPUs')
  _run('nccl', False, use_graphs, 0, exchange='direct')
