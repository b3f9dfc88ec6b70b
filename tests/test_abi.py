"""The drop-in boundary without a GPU: the C-ABI library loads, exports every symbol
include/hbk.h declares, reports errors the way the reference does, and its divide-free
integer arithmetic matches Python's % and //."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from hybridbackend_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
  text = open(os.path.join(ROOT, 'include', 'hbk.h')).read()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  return sorted(set(re.findall(r'\b(hbk_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
  lib = _lib.lib()
  syms = _declared_symbols()
  assert len(syms) >= 10
  missing = [s for s in syms if not hasattr(lib, s)]
  assert not missing, f'declared in include/hbk.h but not exported: {missing}'


def test_version_and_error_channel():
  lib = _lib.lib()
  assert lib.hbk_version().decode().endswith('gfx950')
  # argument validation happens before any device work -> testable without a GPU
  rc = lib.hbk_group_lookup_fwd(-1, None, None)
  assert rc == _lib.INVALID_ARGUMENT
  assert 'n_cols' in lib.hbk_last_error().decode()
  with pytest.raises(_lib.InvalidArgumentError):
    _lib.check(rc)


def test_partition_argument_checks():
  lib = _lib.lib()
  lens = _lib.i64_array([0])
  null = _lib.ptr_array([None])
  rc = lib.hbk_partition_by_modulo_n(1, _lib.INT64, 0, null, lens, null, null, null,
                                     None, C.c_size_t(0), None)
  assert rc == _lib.INVALID_ARGUMENT and 'num_partitions' in lib.hbk_last_error().decode()
  rc = lib.hbk_partition_by_modulo_n(1, _lib.FLOAT, 4, null, lens, null, null, null,
                                     None, C.c_size_t(0), None)
  assert rc == _lib.INVALID_ARGUMENT
  rc = lib.hbk_partition_by_dual_modulo_n(1, _lib.INT64, 4, 2, 3, null, lens, null, null,
                                          null, None, C.c_size_t(0), None)
  assert rc == _lib.INVALID_ARGUMENT and 'stage' in lib.hbk_last_error().decode()
  assert lib.hbk_partition_workspace_bytes(2, _lib.i64_array([1024, 1025]), 8) == 3 * 8 * 4


def test_host_floormod_matches_python():
  lib = _lib.lib()
  lib.hbk_host_floormod_i64.restype = C.c_int64
  lib.hbk_host_floormod_i64.argtypes = [C.c_int64, C.c_int64]
  lib.hbk_host_fastdiv_u64.restype = C.c_uint64
  lib.hbk_host_fastdiv_u64.argtypes = [C.c_uint64, C.c_uint64]
  rng = np.random.RandomState(0)
  divisors = [1, 2, 3, 5, 7, 8, 10, 1000, 1 << 20, 1000000, 1000003, 100000000,
              (1 << 31) - 1, 1 << 31, (1 << 31) + 1, (1 << 32) + 15, (1 << 40) - 87,
              (1 << 62) + 1, (1 << 63) - 1]
  edge = [0, 1, -1, 2, -2, (1 << 63) - 1, -(1 << 63), (1 << 62), -(1 << 62), (1 << 32),
          -(1 << 32), (1 << 32) - 1]
  vals = edge + [int(v) for v in rng.randint(-2**63, 2**63 - 1, size=400, dtype=np.int64)]
  for d in divisors:
    for v in vals + [d, d - 1, d + 1, -d, -d - 1, 2 * d - 1 if 2 * d - 1 < 2**63 else d]:
      if not -2**63 <= v < 2**63:
        continue
      assert lib.hbk_host_floormod_i64(v, d) == v % d, (v, d)
  uvals = [0, 1, 2**64 - 1, 2**63, 2**63 - 1, 2**32, 2**32 - 1] + [
    int(v) for v in rng.randint(0, 2**63 - 1, size=400, dtype=np.int64)] + [
    int(v) + 2**63 for v in rng.randint(0, 2**63 - 1, size=400, dtype=np.int64)]
  for d in divisors + [2**63, 2**64 - 1, 2**63 + 12345]:
    for n in uvals + [d, d - 1, d + 1]:
      if not 0 <= n < 2**64:
        continue
      assert lib.hbk_host_fastdiv_u64(n, d) == n // d, (n, d)


def test_cpu_tensors_are_refused_not_silently_computed():
  import torch
  import hybridbackend_amd as hb
  table = torch.zeros(4, 4)
  ids = torch.zeros(3, dtype=torch.int64)
  with pytest.raises(hb.HbkError):
    hb.embedding.group_lookup([table], [ids])
  with pytest.raises(hb.HbkError):
    hb.distribute.partition_by_modulo(ids, 2)


def test_shard_sizing_rule_matches_oracle():
  # variables.py:93-123
  import torch
  import oracle
  import hybridbackend_amd as hb
  for bucket in (1, 7, 8, 9, 1000, 1000003, 100000000):
    for W in (1, 2, 3, 8):
      for batch in (0, 1000, 65536):
        rows = []
        for r in range(W):
          got = hb.embedding.sharded_bucket_size(bucket, W, r, batch)
          assert got == oracle.shard_rows(bucket, W, r, batch)
          rows.append(got[1])
        if got[0]:
          assert sum(rows) == bucket
  t = torch.arange(10).view(10, 1)
  assert [len(hb.embedding.shard_of_table(t, 4, r)) for r in range(4)] == \
    [hb.embedding.sharded_bucket_size(10, 4, r)[1] for r in range(4)]
  with pytest.raises(ValueError):
    hb.embedding.sharded_bucket_size(10, 4, 4)
