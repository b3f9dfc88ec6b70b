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
  assert lib.hbk_partition_workspace_bytes(2, _lib.i64_array([0, 0]), 8) == 0


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


def test_xcd_contiguous_mapping_is_a_bijection_with_contiguous_ranges():
  """The block -> work item mapping of the XCD-aware launches (csrc/common.h xcd_contiguous):
  for every launch size a bijection of [0, n), and the blocks of one XCD (b % 8 equal) take
  consecutive items in block order -- which is what lets a column's rows meet in one L2."""
  from hybridbackend_amd import _lib
  lib = _lib.lib()
  for n in list(range(1, 80)) + [255, 256, 257, 1000, 3822, 5486, 40003]:
    items = [lib.hbk_host_xcd_contiguous(b, n) for b in range(n)]
    assert sorted(items) == list(range(n)), n
    lo = 0
    for x in range(8):
      mine = items[x::8]                     # what XCD x runs, in dispatch order
      assert mine == list(range(lo, lo + len(mine))), (n, x)
      lo += len(mine)


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


def test_argument_checks_of_every_entry_family():
  """Each entry validates its arguments before any device work and answers with the reference's
  InvalidArgument code and a message naming the problem (partition_by_modulo_ops.cc:81-83 style),
  so the checks are testable without a GPU."""
  lib = _lib.lib()
  bad = _lib.INVALID_ARGUMENT

  def msg():
    return lib.hbk_last_error().decode()

  one_i64 = _lib.i64_array([4])
  null1 = _lib.ptr_array([None])
  # R1 / R6
  assert lib.hbk_floormod_n(1, _lib.FLOAT, null1, one_i64, one_i64, null1, None) == bad
  assert lib.hbk_floormod_n(1, _lib.INT64, null1, one_i64, None, null1, None) == bad
  assert lib.hbk_cast_n(1, _lib.INT32, _lib.HALF, null1, one_i64, null1, None) == bad
  assert 'float->half' in msg()
  # R7
  assert lib.hbk_unique_n(-1, None, None, None, None, None, None, C.c_size_t(0), None) == bad
  # R8-R10: one column with a bad field each
  col = (_lib.LookupColumn * 1)()
  col[0].dim, col[0].divisor = 0, 1
  assert lib.hbk_group_lookup_fwd(1, col, None) == bad and 'dim' in msg()
  col[0].dim, col[0].ids_dtype = 16, _lib.FLOAT
  assert lib.hbk_group_lookup_fwd(1, col, None) == bad and 'int32 or int64' in msg()
  col[0].ids_dtype, col[0].divisor = _lib.INT64, 0
  assert lib.hbk_group_lookup_fwd(1, col, None) == bad and 'divisor' in msg()
  col[0].divisor, col[0].combiner = 1, 7
  assert lib.hbk_group_lookup_fwd(1, col, None) == bad and 'combiner' in msg()
  gcol = (_lib.LookupGradColumn * 1)()
  gcol[0].dim, gcol[0].divisor, gcol[0].ids_dtype = 16, 1, _lib.INT64
  gcol[0].n_ids, gcol[0].n_segments = 4, 5        # no row_splits: segments must equal ids
  assert lib.hbk_group_lookup_bwd(1, gcol, C.c_float(0.0), None, C.c_size_t(0), None) == bad
  assert 'n_segments' in msg()
  assert lib.hbk_group_lookup_bwd_apply(1, gcol, 5, C.c_float(0.1), None, C.c_size_t(0),
                                        None) == bad and 'HBK_APPLY' in msg()
  # step only (no IndexedSlices buffers) needs a learning rate; the two buffers go together
  gcol[0].n_segments, gcol[0].rows = 4, 10
  gcol[0].n_unique = gcol[0].ids = gcol[0].grad_out = 64    # checked for NULL only, never read
  assert lib.hbk_group_lookup_bwd(1, gcol, C.c_float(0.0), None, C.c_size_t(0), None) == bad
  assert 'nothing to do' in msg() or 'apply_lr' in msg(), msg()
  gcol[0].unique_rows = 64
  assert lib.hbk_group_lookup_bwd(1, gcol, C.c_float(0.1), None, C.c_size_t(0), None) == bad
  assert 'together' in msg(), msg()
  gcol[0].unique_rows = None
  # R11
  assert lib.hbk_cache_probe(None, 4, 0, None, 0, None, None, None) == bad
  assert 'cache_slab_size' in msg()
  assert lib.hbk_cache_probe(None, 0, 32, None, 0, None, None, None) == bad
  assert lib.hbk_cache_lookup(None, 4, 32, None, 0, None, None, None, None, None, None,
                              C.c_size_t(0), None) == bad and 'counts' in msg()
  # R4 / R5 / aggregation: no communicator
  assert lib.hbk_alltoall_n(None, 1, _lib.INT32, 0, null1, one_i64, null1, None) == bad
  assert lib.hbk_alltoallv_n(None, 1, _lib.FLOAT, _lib.FLOAT, 0, one_i64, null1, None, null1,
                             None, None, C.c_size_t(0), None) == bad
  assert lib.hbk_allreduce_n(None, 1, _lib.FLOAT, 0, null1, one_i64, null1, C.c_float(1.0),
                             None, C.c_size_t(0), None) == bad
  assert lib.hbk_allgatherv(None, _lib.FLOAT, None, None, None, None) == bad
  handle = C.c_void_p()
  ident = (C.c_uint8 * _lib.COMM_ID_BYTES)()
  assert lib.hbk_comm_create(C.byref(handle), ident, 0, 1, 0) == bad and 'world_size' in msg()
  assert lib.hbk_comm_create(C.byref(handle), ident, 8, 3, 0) == bad and 'local_size' in msg()
  assert lib.hbk_comm_create(C.byref(handle), ident, 8, 8, 8) == bad and 'rank' in msg()
  # R12: plan creation and the host layout
  plan = C.c_void_p()
  assert lib.hbk_sharded_create(C.byref(plan), None, 1, None, _lib.FLOAT) == bad
  assert lib.hbk_sharded_lookup_fwd(None, None, None, None, None, None, None, None) == bad
  assert lib.hbk_sharded_prefetch(None, None, None, None) == bad
  assert lib.hbk_sharded_owned_ids(None, 0) == -1
  dims = np.array([16, 0], np.int32)
  sz = np.zeros(4, np.int32)
  p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
  assert lib.hbk_sharded_layout(2, 2, p(dims), p(sz), p(sz), *([None] * 9)) == bad
  # workspace queries answer 0 for nonsense instead of failing
  assert lib.hbk_unique_workspace_bytes(0, None) == 0
  assert lib.hbk_group_lookup_bwd_workspace_bytes(0, None) == 0
  assert lib.hbk_allreduce_workspace_bytes(1, one_i64, _lib.FLOAT) == 0   # one tensor: in place
  assert lib.hbk_cache_lookup_workspace_bytes(0) == 0


def test_product_library_carries_no_test_scaffolding():
  """The in-process test world lives in tests/support (libhbk_testing.so, over the public
  custom-transport hook); the product library exports nothing of it and reads the environment
  only once, for its option defaults."""
  import subprocess
  out = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True,
                       text=True, check=True).stdout
  exported = set(re.findall(r'\b(hbk_[a-z0-9_]+)\b', out))
  assert not [s for s in exported if 'local_world' in s or 'testing' in s or 'debug' in s]
  assert exported == set(_declared_symbols()), exported ^ set(_declared_symbols())
  undefined = subprocess.run(['nm', '-D', '--undefined-only', _lib.LIB_PATH], capture_output=True,
                             text=True, check=True).stdout
  assert 'getenv' in undefined                      # misc.cpp: option defaults, once
  src = os.path.join(ROOT, 'hybridbackend_amd', 'csrc')
  users = [f for f in sorted(os.listdir(src)) if os.path.isfile(os.path.join(src, f)) and
           'getenv(' in open(os.path.join(src, f)).read()]
  assert users == ['misc.cpp'], users


def test_options_roundtrip_and_unknown_names():
  lib = _lib.lib()
  old = _lib.set_option('bwd_buckets_log2', 3)
  try:
    v = C.c_int32()
    assert lib.hbk_get_option(b'bwd_buckets_log2', C.byref(v)) == 0 and v.value == 3
  finally:
    _lib.set_option('bwd_buckets_log2', old)
  assert lib.hbk_set_option(b'no_such_option', 1) == _lib.INVALID_ARGUMENT
  assert 'no_such_option' in lib.hbk_last_error().decode()
  assert lib.hbk_get_option(None, None) == _lib.INVALID_ARGUMENT


def test_every_option_named_in_the_header_exists_and_every_option_is_named_there():
  """include/hbk.h lists the option names in the comment above hbk_set_option; the library's table
  is csrc/misc.cpp.  Both directions: a documented name the library does not know, or an option
  nobody documented, fails here."""
  import re
  lib = _lib.lib()
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  header = open(os.path.join(root, 'include', 'hbk.h')).read()
  block = header[header.index(' * Names:'):header.index('int hbk_set_option(')]
  documented = set(re.findall(r'\b((?:bwd|fwd|unique|partition|sharded|sync)_[a-z0-9_]+)\b', block))
  table = open(os.path.join(root, 'hybridbackend_amd', 'csrc', 'misc.cpp')).read()
  known = set(re.findall(r'\{"([a-z0-9_]+)", "HBK_[A-Z0-9_]+", &Options::', table))
  assert known, 'option table not found in csrc/misc.cpp'
  v = C.c_int32(0)
  for name in sorted(documented):
    assert lib.hbk_get_option(name.encode(), C.byref(v)) == 0, f'{name}: in hbk.h, unknown to the library'
  missing = sorted(known - documented)
  assert not missing, f'options without a mention in include/hbk.h: {missing}'


def test_custom_transport_argument_checks():
  lib = _lib.lib()
  comm = C.c_void_p()
  assert lib.hbk_comm_create_custom(C.byref(comm), None, 2, 2, 0) == _lib.INVALID_ARGUMENT

  class Transport(C.Structure):
    _fields_ = [('ctx', C.c_void_p), ('exchange', C.c_void_p), ('allreduce', C.c_void_p),
                ('destroy', C.c_void_p)]
  t = Transport()
  assert lib.hbk_comm_create_custom(C.byref(comm), C.byref(t), 2, 2, 0) == _lib.INVALID_ARGUMENT
  t.exchange = 1   # any non-NULL pointer: only the shape of the world is checked here
  assert lib.hbk_comm_create_custom(C.byref(comm), C.byref(t), 4, 3, 0) == _lib.INVALID_ARGUMENT
  assert 'local_size' in lib.hbk_last_error().decode()
  assert lib.hbk_comm_create_custom(C.byref(comm), C.byref(t), 4, 2, 7) == _lib.INVALID_ARGUMENT


def _partition_host(ids, P, modulus=1, stage=0):
  lib = _lib.lib()
  code = {np.dtype(np.int32): _lib.INT32, np.dtype(np.int64): _lib.INT64,
          np.dtype(np.uint32): _lib.UINT32, np.dtype(np.uint64): _lib.UINT64}[ids[0].dtype]
  outs = [np.empty_like(i) for i in ids]
  sizes = [np.empty(P, np.int32) for _ in ids]
  idx = [np.empty(i.size, np.int32) for i in ids]
  ptrs = lambda arrs: _lib.ptr_array([a.ctypes.data for a in arrs])
  lens = _lib.i64_array([i.size for i in ids])
  if stage == 0:
    rc = lib.hbk_partition_by_modulo_host(len(ids), code, P, ptrs(ids), lens, ptrs(outs),
                                          ptrs(sizes), ptrs(idx))
  else:
    rc = lib.hbk_partition_by_dual_modulo_host(len(ids), code, P, modulus, stage, ptrs(ids), lens,
                                               ptrs(outs), ptrs(sizes), ptrs(idx))
  _lib.check(rc)
  return outs, sizes, idx


def test_host_partition_entries_match_the_cpu_functor():
  """The CPU kernels behind the non-N partition ops (host-memory twins of the device entries):
  the reference's KATs and property cases, every dtype, plain and dual modulo -- bit-equal to the
  oracle's restatement of the CPU functors."""
  import json
  import oracle
  g = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'partition.json')))
  for k in g['modulo']:
    o, s, i = _partition_host([np.array(k['input'], np.int64)], k['num_partitions'])
    assert (o[0].tolist(), s[0].tolist(), i[0].tolist()) == (k['output'], k['sizes'], k['indices'])
  for k in g['dual']:
    o, s, i = _partition_host([np.array(k['input'], np.int64)], k['num_partitions'],
                              k['modulus'], k['stage'])
    assert (o[0].tolist(), s[0].tolist(), i[0].tolist()) == (k['output'], k['sizes'], k['indices'])
  rng = np.random.RandomState(0)
  for dt in (np.int32, np.int64, np.uint32, np.uint64):
    info = np.iinfo(dt)
    cols = [rng.randint(max(info.min, -2**62), min(info.max, 2**62), size=n).astype(dt)
            for n in (10000, 0, 1, 777)]
    for P in (1, 2, 5, 8, 64, 1000):
      outs, sizes, idx = _partition_host(cols, P)
      for c, o, s, i in zip(cols, outs, sizes, idx):
        wo, ws, wi = oracle.partition_by_modulo(c, P)
        np.testing.assert_equal(o, wo)
        np.testing.assert_equal(s, ws)
        np.testing.assert_equal(i, wi)
        np.testing.assert_equal(o[i], c)          # the reference's own property (partition_test.py:40-59)
    for P, M, stage in ((2, 2, 1), (2, 2, 2), (4, 3, 1), (4, 3, 2), (8, 1, 2)):
      outs, sizes, idx = _partition_host(cols, P, M, stage)
      for c, o, s, i in zip(cols, outs, sizes, idx):
        wo, ws, wi = oracle.partition_by_dual_modulo(c, P, M, stage)
        np.testing.assert_equal(o, wo)
        np.testing.assert_equal(s, ws)
        np.testing.assert_equal(i, wi)
  lib = _lib.lib()
  null, lens = _lib.ptr_array([None]), _lib.i64_array([0])
  assert lib.hbk_partition_by_modulo_host(1, _lib.FLOAT, 4, null, lens, null, null, null) == \
      _lib.INVALID_ARGUMENT
  assert lib.hbk_partition_by_dual_modulo_host(1, _lib.INT64, 4, 2, 3, null, lens, null, null,
                                               null) == _lib.INVALID_ARGUMENT


def test_tf_shim_parses_and_type_checks():
  """integration/tf_shim/hb_ops_shim.cc -- the REGISTER_OP / REGISTER_KERNEL_BUILDER layer a
  maintainer compiles against TensorFlow -- is parsed and type-checked against include/hbk.h and
  a declaration stub of the TensorFlow symbols it uses (no TensorFlow in this image: a syntax
  check, it pins nothing about TensorFlow)."""
  import shutil
  import subprocess
  if shutil.which('hipcc') is None and not os.path.exists('/opt/rocm/bin/hipcc'):
    pytest.skip('no hipcc')
  r = subprocess.run(['make', '-s', '-C', os.path.join(ROOT, 'integration', 'tf_shim'), 'check'],
                     capture_output=True, text=True, timeout=300)
  assert r.returncode == 0, r.stdout + r.stderr


# "C-ABI surface the replacement must export ... the *set* is the contract" -- and
# north_star: "the C++ REGISTER_OP/REGISTER_KERNEL_BUILDER shim is the only FFI layer", so every
# entry of that set has to be REACHED by the shim (hbk_cache_lookup is the op-shaped form of
# hbk_cache_probe: four compacted lists instead of a per-key slot).
_SECTION_8B = ['hbk_partition_by_modulo_n', 'hbk_partition_by_dual_modulo_n', 'hbk_comm_get_id',
               'hbk_comm_create', 'hbk_comm_destroy', 'hbk_comm_check_async', 'hbk_alltoall_n',
               'hbk_alltoallv_n', 'hbk_cast_n', ('hbk_cache_probe', 'hbk_cache_lookup'),
               'hbk_group_lookup_fwd', 'hbk_group_lookup_bwd', 'hbk_group_lookup_bwd_apply',
               'hbk_sharded_create', 'hbk_sharded_destroy', 'hbk_sharded_lookup_fwd',
               'hbk_sharded_lookup_bwd', 'hbk_sharded_lookup_bwd_apply', 'hbk_sharded_owned_ids',
               'hbk_unique_n', 'hbk_allreduce_n', 'hbk_allgatherv', 'hbk_broadcast',
               'hbk_partition_workspace_bytes', 'hbk_group_lookup_bwd_workspace_bytes',
               'hbk_unique_workspace_bytes', 'hbk_alltoallv_wire_workspace_bytes',
               'hbk_allreduce_workspace_bytes', 'hbk_cache_lookup_workspace_bytes', 'hbk_last_error']


def test_tf_shim_registers_a_kernel_for_every_op_and_reaches_the_whole_c_abi():
  """After preprocessing (the op names of the partition / collective families are built by macros):
  every REGISTER_OP name has at least one REGISTER_KERNEL_BUILDER -- an op registration without a
  kernel reads as done and is not (VERDICT r05 "missing" 2: HbGroupLookupGrad was one) -- every
  kernel names a registered op, and every entry of SURVEY 8(b)'s C-ABI set is called."""
  import re
  import shutil
  import subprocess
  hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
  if not os.path.exists(hipcc):
    pytest.skip('no hipcc')
  shim = os.path.join(ROOT, 'integration', 'tf_shim')
  r = subprocess.run([hipcc, '-std=c++14', '-E', '-P', '-x', 'c++', '-I' + os.path.join(shim, 'check'),
                      '-I' + os.path.join(ROOT, 'include'), os.path.join(shim, 'hb_ops_shim.cc')],
                     capture_output=True, text=True, timeout=300)
  assert r.returncode == 0, r.stderr
  text = r.stdout[r.stdout.index('namespace hybridbackend'):]     # (past the headers)
  lit = r'((?:\s*"[^"]*")+)'

  def names(pattern):
    return [''.join(re.findall(r'"([^"]*)"', m)) for m in re.findall(pattern + r'\(' + lit, text)]
  ops = names(r'OpDefBuilderWrapper')
  kernels = names(r'\bName')
  assert len(ops) == len(set(ops)) and len(ops) >= 29, ops
  for must in ('HbPartitionByModuloN', 'HbPartitionByDualModuloStageTwoN', 'HbNcclAlltoallvN',
               'HbGroupLookup', 'HbGroupLookupGrad', 'HbGroupLookupGradApply', 'HbShardedGroupLookup',
               'HbShardedGroupLookupGrad', 'HbShardedGroupLookupGradApply', 'HbUniqueN', 'HbCastN',
               'HbLookup', 'HbNcclCollectiveHandleOp'):
    assert must in ops, must
  assert sorted(set(ops) - set(kernels)) == [], 'ops registered without a kernel'
  assert sorted(set(kernels) - set(ops)) == [], 'kernels for ops that are not registered'
  body = open(os.path.join(shim, 'hb_ops_shim.cc')).read()
  body = re.sub(r'//[^\n]*', '', body)                             # calls, not comments
  for entry in _SECTION_8B:
    alternatives = entry if isinstance(entry, tuple) else (entry,)
    assert any(re.search(r'\b' + e + r'\(', body) for e in alternatives), f'{entry} is never called'


def test_tables_layout_puts_every_table_on_a_2mb_boundary():
  """hbk_tables_layout (host arithmetic of hbk_tables_alloc / hb.embedding.allocate_tables): one
  slab, 2 MB-aligned offsets -- the policy profiles/r05_placement.txt measured as the fastest."""
  import hybridbackend_amd as hb
  lib = _lib.lib()
  two_mb = 2 << 20
  sizes = (C.c_size_t * 4)(100, two_mb, two_mb + 1, 0)
  offs = (C.c_size_t * 4)()
  total = lib.hbk_tables_layout(4, sizes, offs)
  assert list(offs) == [0, two_mb, 2 * two_mb, 4 * two_mb] and total == 4 * two_mb
  assert lib.hbk_tables_layout(0, None, None) == 0
  tables = hb.embedding.allocate_tables([(10, 4), (1000, 16), (3, 128)], device='cpu')
  assert [tuple(t.shape) for t in tables] == [(10, 4), (1000, 16), (3, 128)]
  assert all(t.data_ptr() % two_mb == 0 and t.dtype.is_floating_point for t in tables)
  tables[1].fill_(1.0)                     # disjoint views of one slab
  tables[0].zero_()
  tables[2].zero_()
  assert float(tables[1].sum()) == 16000.0
