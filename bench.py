"""bench.py - training images/sec of one full G+D step (reference scripts/train.py:524-592)
on MI355X.

  python bench.py --gpus 1 --steps 50 --warmup 10
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], SURVEY.md section 8d "C1"): 64x64 COCO-shape synthetic
scene graphs (3..8 objects + __image__ per image, <= 16 triples), batch 32 PER GPU (weak
scaling), fp32, generator kwargs = the reference's train.py defaults, both discriminators,
three Adam optimisers.  Inputs are resident in HBM before the timed region starts.  The timed
loop cycles through --n_batches (default 16) DIFFERENTLY SEEDED batches - distinct object /
triple counts every step, as a real loader produces (reference sg2im/data/coco.py:286-359) -
which the Trainer pads into shape buckets and replays as one hipGraph per bucket
(sg2im_amd/bucketing.py); the padding copies are inside the timed region.

Prints ONE JSON line on rank 0: the driver's contract fields plus
  roofline     - the implicit-GEMM (conv + linear) kernel family against the fp32 MFMA peak,
                 measured live with HIP events on the launch stream in a separate
                 instrumented pass of the same step (events would otherwise serialise the
                 timed region);
  cpu_baseline - the CPU oracle (oracle/sg2im_oracle.py, a torch-CPU restatement of the
                 reference loop body) timed on this host's cores on the same batch.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0     # same guide: v_mfma_f32_32x32x16_bf16 dense peak (2495 measured)


def parse():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=50)      # >= 50 warm steps
  ap.add_argument('--warmup', type=int, default=10)
  ap.add_argument('--batch_size', type=int, default=32, help='images per GPU (reference default, train.py:51)')
  ap.add_argument('--image_size', type=int, default=64)
  ap.add_argument('--cpu_baseline_steps', type=int, default=5,
                  help='timed steps of the CPU-oracle leg at the best thread count of its sweep; 0 disables the leg')
  ap.add_argument('--no_roofline', action='store_true')
  ap.add_argument('--no_graphs', action='store_true', help='launch every kernel eagerly instead of hipGraph replay')
  ap.add_argument('--seed', type=int, default=0)
  ap.add_argument('--n_batches', type=int, default=16,
                  help='distinct synthetic batches (seeds seed+rank+1000*i) cycled through by the timed loop')
  ap.add_argument('--bucket', default='32,64', help='object,triple padding multiples of the hipGraph shape buckets')
  ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'],
                  help="'bf16': spatial convolutions on the bf16 matrix cores (bf16-rounded operands, fp32 accumulation, bf16 weight "
                       "mirror, bfloat16 storage of the refinement chain) - BASELINE configs[2..4]; a SECONDARY line, the headline "
                       "metric is quoted in fp32")
  ap.add_argument('--style', default='coco', choices=['coco', 'vg'],
                  help="'vg': VG-shape graphs without GT masks (BASELINE configs[2] shape, fp32) instead of the COCO headline workload")
  ap.add_argument('--refinement_dims', default=None,
                  help="comma-separated CRN widths (reference --refinement_network_dims, train.py:99; default 1024,512,256,128,64). "
                       "BASELINE configs[3] 'deeper CRN' = 1024,512,256,128,64,64 (SURVEY.md 8d C3: seed map 4x4 at 128x128)")
  ap.add_argument('--min_objs', type=int, default=None, help='real objects per image, lower bound (default 3)')
  ap.add_argument('--max_objs', type=int, default=None, help='real objects per image, upper bound (default 8 coco / 10 vg; configs[4]: 29)')
  ap.add_argument('--extra_rels', type=int, default=6, help="vg style: relationships per image ~ U{1..k+extra_rels} (configs[4]: 60 -> <= 100 triples with the __in_image__ ones)")
  ap.add_argument('--eval_generator', action='store_true',
                  help='generator in eval() mode (running BatchNorm statistics), as the reference trains after '
                       '--eval_mode_after iterations (train.py:509-512); not the headline workload')
  ap.add_argument('--force_dist', action='store_true',
                  help='debug: 1-rank RCCL group with real all-reduces (exercises the N>1 code path on one GPU)')
  ap.add_argument('--dp_schedule', type=int, default=None, choices=[0, 1, 2],
                  help='data-parallel graph schedule (sg2im_amd/trainer.py): default 2 = RCCL all-reduces recorded inside the '
                       'iteration graph when the subprocess capture probe passes on every rank (else 1); 0 = iteration graph -> '
                       'exposed all-reduces -> Adam graph; 1 = graph segments with the exchanges between them')
  ap.add_argument('--launcher_selftest', action='store_true',
                  help='CPU-only check of the --gpus N self-launcher: every rank joins a gloo group, all-reduces its rank, '
                       'rank 0 prints one JSON line (tests/test_bench_launcher.py)')
  return ap.parse_args()


def _free_port():
  import socket
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def self_launch(n):
  """`python bench.py --gpus N` without a launcher around it (WORLD_SIZE unset): re-execute this command line under
  torch.distributed.run with one rank per GPU on 127.0.0.1 - what the driver's own command does - and hand its
  output and exit status through."""
  import subprocess
  env = dict(os.environ)
  env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # (dmabuf IPC: RCCL across processes needs it on this stack)
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
         '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
  return subprocess.call(cmd, env=env)


def launcher_selftest(world, rank):
  dist.init_process_group('gloo', rank=rank, world_size=world)
  t = torch.tensor([float(rank + 1)])
  dist.all_reduce(t)
  dist.barrier()
  dist.destroy_process_group()
  if rank == 0:
    print(json.dumps({'launcher_selftest': True, 'world': world, 'sum_of_ranks_plus_one': float(t.item())}), flush=True)


def host_cpu():
  """(physical cores available to this process, CPU model string)"""
  model = 'unknown'
  try:
    for line in open('/proc/cpuinfo'):
      if line.startswith('model name'):
        model = line.split(':', 1)[1].strip()
        break
  except OSError:
    pass
  cores = None
  try:
    import psutil
    cores = psutil.cpu_count(logical=False)
  except Exception:
    pass
  avail = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
  cores = min(cores, avail) if cores else avail
  return max(1, cores), model


def cpu_baseline(vocab, batch, steps):
  """The CPU path timed beside the GPU number, on the same first batch: the oracle (oracle/sg2im_oracle.py: kind
  'port' - the reference's arithmetic restated functionally; the reference tree does not exist on the GPU box).
  The thread count is SWEPT (8, 16, 32, 64, ... up to the physical cores): MKL-DNN convolutions at batch 32 stop
  scaling long before 128 threads, and an oversubscribed run understates the baseline - the
  best setting is reported, with its thread count as `cores`."""
  from oracle import sg2im_oracle as orc
  from sg2im_amd.trainer import GENERATOR_DEFAULTS, D_OBJ_DEFAULTS, D_IMG_DEFAULTS
  cores, cpu_model = host_cpu()
  gcfg = dict(GENERATOR_DEFAULTS, vocab=vocab)
  docfg = dict(D_OBJ_DEFAULTS, vocab=vocab)
  dicfg = dict(D_IMG_DEFAULTS)
  tr = orc.OracleTrainer(orc.init_generator_params(gcfg, 0), orc.init_ac_discriminator_params(docfg, 2),
                         orc.init_patch_discriminator_params(dicfg, 1), gcfg, docfg, dicfg)
  cpu_batch = tuple(batch[:6])
  counts = sorted(set(min(c, cores) for c in (8, 16, 32, 64, 128, cores) if c >= 1))
  sweep, budget_t0 = {}, time.time()
  for n in counts:                      # sweep: one warm-up + one timed step per thread count
    torch.set_num_threads(n)
    tr.step(cpu_batch)                  # warm-up at this thread count (MKL-DNN primitive caches, thread pool)
    t0 = time.time()
    tr.step(cpu_batch)
    sweep[n] = time.time() - t0
    if time.time() - budget_t0 > 60:    # (bounded: the whole leg stays within ~2 minutes)
      break
  best = min(sweep, key=sweep.get)
  torch.set_num_threads(best)           # the reported value: `steps` (>= 5 by default) warm steps at the best setting
  tr.step(cpu_batch)
  t0 = time.time()
  for _ in range(steps):
    tr.step(cpu_batch)
  dt = (time.time() - t0) / steps
  nimg = cpu_batch[0].size(0)
  return {'value': round(nimg / dt, 2), 'unit': 'images/sec', 'cores': best, 'cpu': cpu_model,
          'kind': 'port', 'physical_cores': cores,
          'thread_sweep_images_per_sec': {str(n): round(nimg / t, 2) for n, t in sweep.items()},
          'sample': '%d warm G+D steps of the first batch of the stream (batch %d, O=%d, T=%d) at the best thread count of a '
                    'one-step-per-setting sweep: %d threads, %.2f s/step' % (steps, nimg, cpu_batch[1].numel(), cpu_batch[4].size(0), best, dt)}


def main():
  args = parse()
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
    raise SystemExit(self_launch(args.gpus))                 # one rank per GPU, launched from here
  if world != args.gpus:
    raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d (or without a '
                     'launcher: bench.py starts its own ranks)' % (args.gpus, world, args.gpus))
  if args.launcher_selftest:
    return launcher_selftest(world, rank)
  if not torch.cuda.is_available():
    raise SystemExit('bench.py needs a GPU: the HIP path has no CPU fallback')
  torch.cuda.set_device(local_rank)
  device = torch.device('cuda', local_rank)
  use_dist = world > 1 or args.force_dist
  if use_dist:
    if 'MASTER_ADDR' not in os.environ:
      os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', '29533'
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)

  from sg2im_amd import ops
  from sg2im_amd.synthetic import make_vocab, synthetic_batch
  from sg2im_amd.trainer import Trainer

  S = args.image_size
  nb = max(1, args.n_batches)
  if args.style == 'vg':
    vocab = make_vocab(179, 46)
    cpu_batches = [synthetic_batch(args.batch_size, image_size=(S, S), num_objs=179, num_preds=46, min_objs=args.min_objs or 3,
                                   max_objs=args.max_objs or 10, mask_size=16, style='vg', extra_rels=args.extra_rels,
                                   seed=args.seed + rank + 1000 * i)
                   for i in range(nb)]
  else:
    vocab = make_vocab(184, 7)            # COCO-Stuff: 184 object ids incl. __image__, 7 predicates
    cpu_batches = [synthetic_batch(args.batch_size, image_size=(S, S), num_objs=184, num_preds=7, min_objs=args.min_objs or 3,
                                   max_objs=args.max_objs or 8, mask_size=16, style='coco', seed=args.seed + rank + 1000 * i)
                   for i in range(nb)]
  cpu_batch = cpu_batches[0]
  batches = [tuple(t.to(device) if torch.is_tensor(t) else t for t in b) for b in cpu_batches]
  batch = batches[0]
  bucket = tuple(int(v) for v in args.bucket.split(','))
  gkw = {'image_size': (S, S)}
  if args.refinement_dims:
    gkw['refinement_dims'] = tuple(int(v) for v in args.refinement_dims.split(','))
  trainer = Trainer(vocab, device, generator_kwargs=gkw, world_size=world, seed=1234,
                    use_graphs=not args.no_graphs, bucket=bucket, rank=rank, compute_dtype=args.dtype,
                    verify_replicas=False,      # (checked here, around the timed loop: Trainer.check_replicas)
                    dp_schedule=args.dp_schedule)
  peak = BF16_MFMA_PEAK_TFLOPS if args.dtype == 'bf16' else FP32_MFMA_PEAK_TFLOPS
  # (data parallel: the same overlapped graph without its Adam updates, the four all-reduces issued
  # eagerly, then an Adam graph - DESIGN.md section 6; --no_graphs selects the eager segments)
  trainer_graphs = trainer.use_graphs
  if args.eval_generator:
    trainer.set_generator_eval()
  if args.force_dist:
    trainer.reducer.force = True

  def sync():
    torch.cuda.synchronize()
    if use_dist:
      dist.barrier()
      torch.cuda.synchronize()

  # one-time setup outside warm-up/timing: one pass over the stream captures the hipGraph of every
  # shape bucket that occurs (sg2im_amd/trainer.py::_graph_step)
  for b in (batches if trainer.use_graphs else []):
    trainer.step(b)
  for i in range(args.warmup):
    trainer.step(batches[i % nb])
  sync()

  # every rank applied the same (all-reduced) gradients <=> the parameter arenas are still bit-identical.  If the
  # in-graph exchange (dp_schedule 2) let them diverge on this stack, Trainer.check_replicas has fallen back to
  # schedule 0 and re-broadcast rank 0's state: warm up again and re-check (a training run gets the same check from
  # the Trainer itself, at steps 1, 2, 4 ... and every 1024th)
  schedule0 = trainer.dp_schedule
  in_sync = trainer.check_replicas() if world > 1 else True
  if not in_sync and trainer.dp_schedule != schedule0:
    for b in batches:
      trainer.step(b)
    for i in range(args.warmup):
      trainer.step(batches[i % nb])
    sync()
    in_sync = trainer.replicas_in_sync()
  stats0 = dict(trainer.graph_stats)
  host0 = dict(trainer.host_seconds)
  trainer.host_launch_samples.clear()
  t0 = time.perf_counter()
  host_trace = [] if os.environ.get('SG2IM_HOST_TRACE') == '1' else None      # diagnostics: host time of every step() call
  for i in range(args.steps):
    t1 = time.perf_counter()
    losses = trainer.step(batches[(args.warmup + i) % nb])
    if host_trace is not None:
      bt = batches[(args.warmup + i) % nb]
      host_trace.append((time.perf_counter() - t1, trainer.bucketer.bucket(bt[1].numel(), bt[4].size(0)) if trainer.bucketer else None))
  host_issue = time.perf_counter() - t0          # (the host's share: how long issuing the K steps took)
  if host_trace is not None:
    print('[host] ' + ' '.join('%s:%.2f' % (b[0] if b else '-', dt * 1e3) for dt, b in host_trace), file=sys.stderr)
  sync()
  elapsed = time.perf_counter() - t0
  ops.persistent_kernels_check()         # (raises if a grid barrier of a persistent launch ever timed out)
  stats1 = dict(trainer.graph_stats)
  launch_samples = sorted(trainer.host_launch_samples)
  if os.environ.get('SG2IM_MARKS') == '1':       # diagnostics: where the lanes of the last replayed iteration were in time
    from sg2im_amd import ops as _ops
    for name, us in _ops.marks_report():
      print('[mark] %-18s %9.1f us' % (name, us), file=sys.stderr)
    # the persistent GraphTripleConv launches of the last replay: workgroup 0's (before, after)-barrier stamps
    import ctypes as _ct
    from sg2im_amd import _lib as _l
    for key, area in _ops._sync_areas.items():
      host = area.cpu().contiguous()
      buf = (_ct.c_ulonglong * 256)()
      n = _l.load().sg2im_gconv_stack_stamps(_ct.c_void_p(host.data_ptr()), _ct.cast(buf, _ct.c_void_p), 256)
      if n > 2:
        st = [(buf[i] - buf[0]) / 100.0 for i in range(n)]
        print('[gcn-stamps] lane %s: total %.1f us; compute/barrier per stage: %s' % (
          key, st[-1], ' '.join('%.1f/%.1f' % (st[i] - st[i - 1], st[i + 1] - st[i]) for i in range(1, n - 1, 2))), file=sys.stderr)
  launch_stats = dict(trainer.launch_stats)
  n_graphs = len(trainer._graphs)
  # the persistent GraphTripleConv forward of the LAST replayed iteration: workgroup 0's device-clock stamps
  # (kernel start, before / after every grid barrier, end) - latency-bound, reported in us
  gcn_stack = None
  try:
    from sg2im_amd import ops as _ops2
    for key in list(_ops2._sync_areas):
      if key[0] != (device.index or 0):
        continue
      with torch.cuda.device(device):
        host = _ops2._sync_areas[key].cpu().contiguous()
      import ctypes as _ct
      from sg2im_amd import _lib as _l2
      buf = (_ct.c_ulonglong * 256)()
      n = _l2.load().sg2im_gconv_stack_stamps(_ct.c_void_p(host.data_ptr()), _ct.cast(buf, _ct.c_void_p), 256)
      if n > 2:
        st_ = [(buf[i] - buf[0]) / 100.0 for i in range(n)]
        gcn_stack = {'kernel': 'gcn_stack_fwd_kernel (one persistent launch for all GraphTripleConv layers)',
                     'us': round(st_[-1], 1), 'stages': (n - 2) // 2 + 1, 'grid_barriers': (n - 2) // 2,
                     'barrier_wait_us_workgroup0': round(sum(st_[i + 1] - st_[i] for i in range(1, n - 1, 2)), 1)}
  except Exception:
    gcn_stack = None

  comm = None
  if use_dist:
    # the gradient exchange on its own (the four all-reduces of a step, back to back), and the step
    # with the collectives muted: exposed = step - step_without_exchange, overlap = 1 - exposed / exchange
    def timed(fn, n):
      sync()
      t = time.perf_counter()
      for i in range(n):
        fn(i)
      sync()
      dt = time.perf_counter() - t
      if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
      return dt / n * 1e3
    guard = torch.ones(1, device=device)

    def exchange(_):
      red = trainer.reducer
      red.start(trainer.flat_g.grad); red.start(guard)
      if trainer.flat_di is not None:
        red.start(trainer.flat_di.grad)
      if trainer.flat_do is not None:
        red.start(trainer.flat_do.grad)
      red.finish()
    exchange(0)
    ar_ms = timed(exchange, 10)
    trainer.reducer.mute = True
    if trainer_graphs:                  # (graphs with the collectives recorded inside: re-captured without them)
      trainer._graphs.clear()
      for b in batches:
        trainer.step(b)
    nocomm_ms = timed(lambda i: trainer.step(batches[i % nb]), max(8, args.steps // 2))
    trainer.reducer.mute = False
    if trainer_graphs:
      trainer._graphs.clear()
    step_ms = elapsed / args.steps * 1e3
    exposed = max(0.0, step_ms - nocomm_ms)
    payload = (2.0 if trainer.reducer.payload == 'bf16' else 4.0) * (
      trainer.flat_g.grad.numel() + (trainer.flat_di.grad.numel() if trainer.flat_di is not None else 0) +
      (trainer.flat_do.grad.numel() if trainer.flat_do is not None else 0))
    comm = {'allreduce_ms': round(ar_ms, 3), 'payload_mb': round(payload / 1e6, 1), 'payload_dtype': trainer.reducer.payload,
            'allreduce_bus_gb_per_s': round(payload * 2 * (world - 1) / max(world, 1) / (ar_ms * 1e-3) / 1e9, 1) if world > 1 else None,
            'step_ms_without_exchange': round(nocomm_ms, 3), 'exposed_ms': round(exposed, 3),
            'overlap_frac': round(min(1.0, max(0.0, 1.0 - exposed / ar_ms)), 3) if ar_ms > 0 else None,
            'schedule': ('eager segments, exchanges started after each backward' if not trainer_graphs else
                         'graphs [G fwd+bwd | D_img] -> all-reduce(G, guard, D_img) || graph [D_obj step] -> all-reduce(D_obj) -> graph [3x Adam]'
                         if trainer.dp_schedule == 1 else
                         'ONE graph with the RCCL all-reduces recorded inside: guard / D_img / D_obj right after their steps, the generator in '
                         'four buckets (refinement module 0, module 1, modules 2.. + output convolutions as their weight gradients complete, '
                         'the rest after the backward)'
                         if trainer.dp_schedule == 2 and trainer.reducer.capturable() else
                         'one iteration graph (D steps on a side stream) -> 4 all-reduces (exposed) -> Adam graph')}
  if use_dist:
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
  host_losses = Trainer.losses_to_host(losses)

  roofline = None
  if not args.no_roofline:
    # instrumented pass: HIP events around every implicit-GEMM launch of 3 more steps (every
    # rank runs it - the steps contain the gradient all-reduces - rank 0 reports)
    ops.TIMER = ops.KernelTimer()
    trainer.use_graphs = False          # events need the individual (eager) launches (same padded shapes)
    n_prof = 3
    for i in range(n_prof):
      trainer.step(batches[i % nb])
    summ = ops.TIMER.summary()
    crn = ops.TIMER.summary('crn')       # the launches of the refinement network alone
    alg_bytes = sum(v for k, v in ops.TIMER.alg_bytes.items() if k.startswith('igemm')) / n_prof
    # ALGORITHMIC work: the same batches without the bucket padding (FLOPs / bytes counted, not timed)
    padded_flops = sum(v['flops'] for k, v in summ.items() if not k.startswith('hbm_'))
    ops.TIMER = ops.KernelTimer()
    keep_bucketer, trainer.bucketer = trainer.bucketer, None
    for i in range(n_prof):
      trainer.step(batches[i % nb])
    trainer.bucketer = keep_bucketer
    alg = ops.TIMER.summary()
    alg_flops = sum(v['flops'] for k, v in alg.items() if not k.startswith('hbm_'))
    alg_crn_flops = sum(v['flops'] for k, v in ops.TIMER.summary('crn').items() if not k.startswith('hbm_'))
    alg_bytes = sum(v for k, v in ops.TIMER.alg_bytes.items() if k.startswith('igemm')) / n_prof
    for k, v in summ.items():            # per-kind FLOPs: the algorithmic (unpadded) count
      if k in alg and not k.startswith('hbm_'):
        v['flops'] = alg[k]['flops']
    ops.TIMER = None
    hbm = {k[4:]: v for k, v in summ.items() if k.startswith('hbm_')}       # the HBM-bound kernels
    summ = {k: v for k, v in summ.items() if not k.startswith('hbm_')}
    crn = {k: v for k, v in crn.items() if not k.startswith('hbm_')}
    flops = sum(v['flops'] for v in summ.values())
    ms = sum(v['ms'] for v in summ.values())
    launches = sum(v['launches'] for v in summ.values())
    achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    roofline = {
      'bound': 'mfma', 'kernel': 'implicit-GEMM family conv_fwd/dgrad/wgrad_kernel + conv_halo_kernel (3x3 stride-1 forward / data gradient) (%s) incl. split-K finish' % (
        'bf16 operands v_mfma_f32_32x32x16_bf16 for the spatial convolutions, fp32 v_mfma_f32_32x32x2_f32 for the linear layers; '
        'priced against the bf16 peak' if args.dtype == 'bf16' else 'fp32 v_mfma_f32_32x32x2_f32'),
      'achieved': round(achieved, 2), 'peak': peak, 'unit': 'TFLOP/s',
      'frac': round(achieved / peak, 4), 'traffic': None,
      'algorithmic_mb_per_step': round(alg_bytes / 1e6, 1),     # every operand read once, every result written once
      # launches of the graph-mode plan (counted by the library while the iteration was captured); the timed
      # eager pass below issues every weight gradient on its own
      'launches_per_step': launch_stats.get('gemm_launches_per_step', launches // n_prof),
      'launches_per_step_all_kernels': launch_stats.get('launches_per_step'),
      'launches_per_step_instrumented_eager_pass': launches // n_prof, 'gflop_per_step': round(flops / n_prof / 1e9, 1),
      'gflop_per_step_incl_bucket_padding': round(padded_flops / n_prof / 1e9, 1),
      'ms_per_step_in_kernel': round(ms / n_prof, 3),
      'crn_only': (lambda f, m: {'gflop_per_step': round(f / n_prof / 1e9, 1), 'ms_per_step': round(m / n_prof, 3),
                                 'tflops': round(f / (m * 1e-3) / 1e12, 2) if m > 0 else 0.0,
                                 'frac': round(f / (m * 1e-3) / 1e12 / peak, 4) if m > 0 else 0.0})(
                    alg_crn_flops, sum(v['ms'] for v in crn.values())),
      # second roofline: kernels bound by HBM bandwidth, algorithmic bytes / event time
      'hbm_bound': {k: {'launches_per_step': v['launches'] // n_prof, 'mbytes_per_launch': round(v['flops'] / v['launches'] / 1e6, 2),
                        'us_per_launch': round(v['ms'] / v['launches'] * 1e3, 1),
                        'gb_per_s': round(v['flops'] / (v['ms'] * 1e-3) / 1e9, 1) if v['ms'] > 0 else 0.0,
                        'frac_of_8TBps': round(v['flops'] / (v['ms'] * 1e-3) / 8e12, 4) if v['ms'] > 0 else 0.0}
                    for k, v in hbm.items()},
      'latency_bound': {'gcn_stack_forward': gcn_stack},
      'by_kind': {k: {'launches_per_step': v['launches'] // n_prof, 'gflop_per_step': round(v['flops'] / n_prof / 1e9, 1),
                      'ms_per_step': round(v['ms'] / n_prof, 3),
                      'tflops': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2) if v['ms'] > 0 else 0.0}
                  for k, v in summ.items()},
    }
    # memory-side traffic of the same family cannot be counted from inside this process: it comes from the
    # committed rocprofv3 --pmc passes over this workload (tools/pmc_step.sh -> profiles/r1_pmc_step_traffic.json)
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_step_traffic.json')))
    pmc_path = cands[-1] if cands else ''                      # the latest round's measurement
    if args.style == 'coco' and S == 64 and args.batch_size == 32 and args.dtype == 'f32' and os.path.exists(pmc_path):
      pmc = json.load(open(pmc_path))
      per_step = (pmc['fetch_mb_per_step'] + pmc['write_mb_per_step']) * 1e6
      roofline['traffic'] = round(per_step)       # HBM-side bytes per STEP (all launches of the family; algorithmic_mb_per_step is its counterpart)
      roofline['traffic_detail'] = {'bytes_per_gemm_launch': round(per_step / max(roofline['launches_per_step'], 1)),'source': 'profiles/' + os.path.basename(pmc_path) + ' (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, '
                                              'separate passes, FETCH_SIZE x2 per the gfx950 note; counts L2 misses, '
                                              'most of which the 256 MB Infinity Cache serves)',
                                    'measured': 'not in this run: read from the committed file named in source',
                                    'fetch_mb_per_step': pmc['fetch_mb_per_step'], 'write_mb_per_step': pmc['write_mb_per_step'],
                                    'over_algorithmic': round(per_step / alg_bytes, 2) if alg_bytes else None}
  if use_dist:
    dist.barrier()

  if rank == 0:
    cpu = None
    if world == 1 and args.cpu_baseline_steps > 0:
      cpu = cpu_baseline(vocab, cpu_batch, args.cpu_baseline_steps)
    imgs = args.batch_size * world * args.steps
    shapes = sorted(set((int(b[1].numel()), int(b[4].size(0))) for b in cpu_batches))
    buckets = sorted(set(trainer.bucketer.bucket(o, t) for o, t in shapes)) if trainer.bucketer else []
    out = {
      'metric': 'training images/sec (G+D step)', 'value': round(imgs / elapsed, 2), 'unit': 'images/sec',
      'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
      'ms_per_step': round(elapsed / args.steps * 1e3, 3),
      'host_issue_ms_per_step': round(host_issue / args.steps * 1e3, 3),
      # mean per step incl. the stalls of a host that runs ahead of the device (back-pressure inside hipGraphLaunch);
      # what ISSUING a step costs is the median of the individual hipGraphLaunch calls
      'host_issue_detail_ms': dict({k: round((trainer.host_seconds[k] - host0[k]) / args.steps * 1e3, 3) for k in host0},
                                   graph_launch_median=round(launch_samples[len(launch_samples) // 2] * 1e3, 3) if launch_samples else None,
                                   graph_launch_min=round(launch_samples[0] * 1e3, 3) if launch_samples else None),
      'higher_is_better': True, 'scaling': 'weak',
      'vs_baseline': None,
      'dtype': 'f32' if args.dtype == 'f32' else 'bf16 (matrix-core operands of the spatial convolutions, bf16 weight mirror, bfloat16 storage of the refinement '
               'chain on the unsplit levels; fp32 accumulation, statistics, losses, master weights and Adam)',
      'data': 'synthetic',
      'config': {'workload': ('COCO-%d synthetic scene graphs (%d-%d objects + __image__, <=%d triples per image), ' % (
                                S, args.min_objs or 3, args.max_objs or 8, 2 * (args.max_objs or 8))
                              if args.style == 'coco' else 'VG-%d synthetic scene graphs (%d-%d objects, <=%d triples per image, no GT masks), ' % (
                                S, args.min_objs or 3, args.max_objs or 10, 2 * (args.max_objs or 10) + args.extra_rels)) +
                             'batch %d per GPU, full G + D_obj + D_img step with 3x Adam' % args.batch_size +
                             (' (generator in eval mode)' if args.eval_generator else ''),
                 'global_batch': args.batch_size * world, 'image_size': S,
                 'refinement_dims': list(trainer.model_kwargs['refinement_dims']),
                 'batch_stream': {'distinct_batches': nb, 'distinct_object_triple_shapes': len(shapes),
                                  'objects_min_max': [shapes[0][0], shapes[-1][0]],
                                  'triples_min_max': [min(t for _, t in shapes), max(t for _, t in shapes)],
                                  'shape_buckets': [list(b) for b in buckets], 'graphs_alive': n_graphs,
                                  'captures_in_timed_loop': stats1['captures'] - stats0['captures'],
                                  'recaptures_in_timed_loop': stats1['invalidated'] - stats0['invalidated'],
                                  'replays_in_timed_loop': stats1['replays'] - stats0['replays']},
                 'parallelism': 'dp%d' % world, 'total_loss': round(host_losses['total_loss'], 5),
                 'launch': ('eager' if not trainer_graphs else
                            'hipGraph replay, one graph per shape bucket (D steps on a side stream inside the graph)' if not use_dist
                            else 'hipGraph replay per shape bucket: iteration graph + all-reduces + Adam graph')},
      'roofline': roofline, 'cpu_baseline': cpu,
    }
    if comm is not None:
      comm['replicas_bit_identical_after_warmup'] = in_sync
      comm['dp_schedule'] = trainer.dp_schedule
      out['gradient_exchange'] = comm
  if use_dist:
    dist.destroy_process_group()
  if rank == 0:
    # RCCL writes a banner through C stdio: flush it first so the JSON line is the LAST line
    import ctypes
    ctypes.CDLL(None).fflush(None)
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
  main()
