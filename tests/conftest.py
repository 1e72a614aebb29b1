import os
import sys
import pathlib

import pytest

# the pipelined agent's stream-pair measurement (its first 48 pipelined train calls switch pairs
# and capture each pair's graphs) is off in the tests, which build many short-lived agents; the
# tests of the measurement itself force it (tests/test_learner_gpu.py, tests/test_dp_gpu.py)
os.environ.setdefault('DD_PIPE_TUNE', '0')

ROOT = pathlib.Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
  sys.path.insert(0, str(ROOT))


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def hip():
  """The HIP backend on cuda:0; GPU tests fail loudly if it cannot load."""
  import torch
  from daydreamer_amd import hipops
  assert torch.cuda.is_available(), 'gpu test collected without a GPU'
  # native frames next to faulthandler's Python ones if the process dies in the HIP runtime
  hipops.load_library().dd_install_crash_handler()
  from daydreamer_amd import graphs
  graphs.CHECK_CAPTURE_ALLOCS = True   # captured segments must not allocate
  return hipops.HipOps('cuda:0')


@pytest.fixture(scope='session')
def ref():
  from oracle import ref_ops
  return ref_ops.RefOps('cpu')
