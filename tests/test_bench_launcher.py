"""bench.py --gpus N without a launcher around it must start its own ranks (a driver that runs
`python bench.py --gpus 8` used to get SystemExit).  CPU-only: the launcher is exercised with --launcher_selftest,
where every rank joins a gloo group on 127.0.0.1 and all-reduces rank + 1."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
  env = dict(os.environ)
  for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
    env.pop(k, None)
  return env


def test_bench_gpus_n_launches_its_own_ranks():
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--launcher_selftest'],
                       env=_clean_env(), capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stderr[-2000:]
  lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, out.stdout          # exactly ONE JSON line, from rank 0
  got = json.loads(lines[0])
  assert got == {'launcher_selftest': True, 'world': 2, 'sum_of_ranks_plus_one': 3.0}


def test_bench_rejects_a_world_size_that_contradicts_gpus():
  env = _clean_env()
  env.update(WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--launcher_selftest'],
                       env=env, capture_output=True, text=True, timeout=600)
  assert out.returncode != 0 and 'WORLD_SIZE=1' in out.stderr
