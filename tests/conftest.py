import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu)')


def pytest_collection_modifyitems(config, items):
  import torch
  if torch.cuda.is_available():
    return
  skip = pytest.mark.skip(reason='no GPU in this process')
  for it in items:
    if 'gpu' in it.keywords:
      it.add_marker(skip)


@pytest.fixture(autouse=True)
def _release_graphs_between_tests(request):
  """GPU tests build Trainers whose hipGraphs (some with recorded RCCL kernels of a process group the test has
  already destroyed) would otherwise be torn down by the cyclic garbage collector at an arbitrary allocation inside a
  LATER test - possibly between that test's capture and its replays.  Collect them at the test boundary, with the
  device idle."""
  yield
  if 'gpu' in request.keywords:
    import gc
    import torch
    if torch.cuda.is_available():
      torch.cuda.synchronize()
      gc.collect()
      torch.cuda.synchronize()
