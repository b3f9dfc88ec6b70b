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
  test_lib = os.path.join(ROOT, 'hybridbackend_amd', 'lib', 'libhbk_testing.so')
  if not os.path.exists(lib_path) or not os.path.exists(test_lib):
    import __graft_entry__
    __graft_entry__.build()


@pytest.fixture
def hbk_option():
  """Set library options (hbk_set_option) for one test; restored afterwards."""
  from hybridbackend_amd import _lib
  saved = []

  def set_(name, value):
    saved.append((name, _lib.set_option(name, value)))
  yield set_
  for name, old in reversed(saved):
    _lib.set_option(name, old)
