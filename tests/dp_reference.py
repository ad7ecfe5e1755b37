"""CPU reference of ONE data-parallel training iteration: every rank runs the
reference's loop body (scripts/train.py:524-592, restated by oracle.OracleTrainer) on ITS shard with
per-replica BatchNorm, the gradients of the three networks are averaged over the ranks, and every
rank applies the same Adam update.  Used by tests/test_dp_rccl.py (2 ranks on 2 GPUs) and pinned on
the CPU by tests/test_host_logic.py (world 1 == OracleTrainer.step)."""
import torch


def _param_dicts(tr):
  return [P for P in (tr.PG, tr.PDo, tr.PDi) if P is not None]


def dp_step(trainers, shards, noises=None):
  """trainers: one oracle.OracleTrainer per rank, all holding IDENTICAL parameters (buffers may
  differ); shards: one 6-tuple batch per rank.  Returns the per-rank loss dicts."""
  world = len(trainers)
  noises = noises or [None] * world
  results = []
  for tr, sh, nz in zip(trainers, shards, noises):
    for opt in (tr.opt_g, tr.opt_do, tr.opt_di):
      if opt:
        opt.zero_grad()
    total, losses, out = tr.g_forward_loss(sh, nz)
    total.backward()
    # the generator's backward also deposits (discarded) gradients in the discriminators (train.py:559)
    for opt in (tr.opt_do, tr.opt_di):
      if opt:
        opt.zero_grad()
    fake = out[0].detach()
    res = {k: float(v) for k, v in losses.items()}
    if tr.PDo is not None:
      ld, parts = tr.d_obj_loss(sh, fake)
      ld.backward()
      res.update({k: float(v) for k, v in parts.items()})
    if tr.PDi is not None:
      li, parts = tr.d_img_loss(sh, fake)
      li.backward()
      res.update({k: float(v) for k, v in parts.items()})
    results.append(res)
  # mean over ranks of the per-shard gradients
  for dicts in zip(*[_param_dicts(tr) for tr in trainers]):
    for k in dicts[0]:
      leafs = [d[k] for d in dicts]
      if not leafs[0].requires_grad:
        continue
      have = [t.grad for t in leafs if t.grad is not None]
      if not have:
        continue
      mean = sum(have) / world          # (a rank without a gradient contributes zero, like the flat arena)
      for t in leafs:
        t.grad = mean.clone()
  for tr in trainers:
    for opt in (tr.opt_g, tr.opt_do, tr.opt_di):
      if opt:
        opt.step()
  return results
