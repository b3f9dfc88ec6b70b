import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def golden_dir():
  return os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(scope='session', autouse=True)
def _built():
  """Build the CPU oracle and (if missing) the HIP library once per session."""
  import oracle
  oracle.build()
  lib_path = os.path.join(ROOT, 'hybridbackend_amd', 'lib', 'libhbk_core.so')
  if not os.path.exists(lib_path):
    import __graft_entry__
    __graft_entry__.build()
