"""ctypes binding of ``libhbk_core.so`` (the C ABI declared in ``include/hbk.h``).

There is no fallback: if the HIP library has not been built the import of any op
fails loudly with the build command.  PyTorch is used only for device memory and
streams; every compute call below goes through the C ABI.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (HBK_LIBRARY: another build of the same library -- the probe builds of tools/scratch/*.sh; the
# default is the in-tree one, and either way a missing file is an ImportError, never a fallback)
LIB_PATH = os.environ.get('HBK_LIBRARY') or os.path.join(_HERE, 'lib', 'libhbk_core.so')

# dtype codes of include/hbk.h
INT8, UINT8, INT32, UINT32, INT64, UINT64, HALF, FLOAT, DOUBLE = range(9)
APPLY_SGD, APPLY_ADAGRAD = 0, 2
GRAD_DETERMINISTIC = 1   # hbk_lookup_grad_column_t.flags
COMBINER_SUM, COMBINER_MEAN, COMBINER_SQRTN = 0, 1, 2
OK, INVALID_ARGUMENT, UNIMPLEMENTED, INTERNAL = 0, 3, 12, 13
COMM_ID_BYTES = 128


class HbkError(RuntimeError):
  """A non-zero status from the C ABI (code = TensorFlow error code)."""

  def __init__(self, code, message):
    super().__init__(f'[hbk status {code}] {message}')
    self.code = code


class InvalidArgumentError(HbkError, ValueError):
  pass


class LookupColumn(C.Structure):
  """hbk_lookup_column_t"""
  _fields_ = [('table', C.c_void_p), ('rows', C.c_int64), ('dim', C.c_int32),
              ('ids_dtype', C.c_int32), ('ids', C.c_void_p), ('n_ids', C.c_int64),
              ('row_splits', C.c_void_p), ('n_segments', C.c_int64),
              ('bucket', C.c_int64), ('divisor', C.c_int32),
              ('combiner', C.c_int32), ('out', C.c_void_p),
              ('run_start', C.c_void_p), ('run_base', C.c_void_p),
              ('n_runs', C.c_int32), ('out_stride', C.c_int32),
              ('hot_rows', C.c_int32), ('half_io', C.c_int32), ('out_slots', C.c_void_p)]


class LookupGradColumn(C.Structure):
  """hbk_lookup_grad_column_t"""
  _fields_ = [('table', C.c_void_p), ('rows', C.c_int64), ('dim', C.c_int32),
              ('ids_dtype', C.c_int32), ('ids', C.c_void_p), ('n_ids', C.c_int64),
              ('row_splits', C.c_void_p), ('n_segments', C.c_int64),
              ('bucket', C.c_int64), ('divisor', C.c_int32),
              ('combiner', C.c_int32), ('grad_out', C.c_void_p),
              ('unique_rows', C.c_void_p), ('grad_rows', C.c_void_p),
              ('n_unique', C.c_void_p), ('run_start', C.c_void_p), ('run_ids', C.c_void_p),
              ('run_grads', C.c_void_p), ('n_runs', C.c_int32), ('grad_stride', C.c_int32),
              ('accum', C.c_void_p), ('table_pitch', C.c_int32), ('flags', C.c_int32)]


class ShardedColumn(C.Structure):
  """hbk_sharded_column_t"""
  _fields_ = [('shard', C.c_void_p), ('rows_local', C.c_int64), ('dim', C.c_int32),
              ('combiner', C.c_int32), ('bucket', C.c_int64), ('accum', C.c_void_p),
              ('hot_rows', C.c_int32), ('dedup', C.c_int32)]


class StitchGradColumn(C.Structure):
  """hbk_stitch_grad_column_t"""
  _fields_ = [('dim', C.c_int32), ('combiner', C.c_int32), ('n_ids', C.c_int64),
              ('index', C.c_void_p), ('row_splits', C.c_void_p), ('n_segments', C.c_int64),
              ('grad_out', C.c_void_p), ('grad_rows', C.c_void_p),
              ('run_start', C.c_void_p), ('run_base', C.c_void_p), ('n_runs', C.c_int32),
              ('grad_stride', C.c_int32)]


_lib = None


def _declare(l):
  """Full prototypes of include/hbk.h: without them ctypes would pass Python ints as
  32-bit C ints and truncate device pointers."""
  vp, i32, i64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t
  protos = {
    'hbk_last_error': (C.c_char_p, []),
    'hbk_version': (C.c_char_p, []),
    'hbk_tables_layout': (C.c_size_t, [i32, vp, vp]),
    'hbk_tables_alloc': (C.c_int, [i32, vp, vp, vp]),
    'hbk_tables_free': (C.c_int, [vp]),
    'hbk_set_option': (C.c_int, [C.c_char_p, i32]),
    'hbk_get_option': (C.c_int, [C.c_char_p, vp]),
    'hbk_comm_rccl_ranks': (C.c_int, [vp]),
    'hbk_broadcast': (C.c_int, [vp, i32, vp, vp, i64, i32, vp]),
    'hbk_sharded_p2p_bind': (C.c_int, [vp, vp, vp, vp, vp]),
    'hbk_sharded_p2p_unbind': (C.c_int, [vp]),
    'hbk_sync_check': (C.c_int, []),
    'hbk_sync_check_stream': (C.c_int, [C.c_void_p]),
    'hbk_host_floormod_i64': (i64, [i64, i64]),
    'hbk_host_fastdiv_u64': (C.c_uint64, [C.c_uint64, C.c_uint64]),
    'hbk_host_crc32c': (C.c_uint32, [C.c_uint32, C.c_void_p, C.c_int64]),
    'hbk_host_xcd_contiguous': (i32, [i32, i32]),
    'hbk_floormod_n': (C.c_int, [i32, i32, vp, vp, vp, vp, vp]),
    'hbk_partition_workspace_bytes': (sz, [i32, vp, i32]),
    'hbk_partition_by_modulo_n': (C.c_int, [i32, i32, i32, vp, vp, vp, vp, vp, vp, sz, vp]),
    'hbk_partition_by_dual_modulo_n':
      (C.c_int, [i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, sz, vp]),
    'hbk_partition_by_modulo_host': (C.c_int, [i32, i32, i32, vp, vp, vp, vp, vp]),
    'hbk_partition_by_dual_modulo_host': (C.c_int, [i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]),
    'hbk_cast_n': (C.c_int, [i32, i32, i32, vp, vp, vp, vp]),
    'hbk_unique_workspace_bytes': (sz, [i32, vp]),
    'hbk_unique_n': (C.c_int, [i32, vp, vp, vp, vp, vp, vp, sz, vp]),
    'hbk_group_lookup_fwd': (C.c_int, [i32, vp, vp]),
    'hbk_group_lookup_bwd_workspace_bytes': (sz, [i32, vp]),
    'hbk_group_lookup_bwd': (C.c_int, [i32, vp, C.c_float, vp, sz, vp]),
    'hbk_group_lookup_bwd_apply': (C.c_int, [i32, vp, i32, C.c_float, vp, sz, vp]),
    'hbk_group_stitch_bwd': (C.c_int, [i32, vp, vp]),
    'hbk_cache_probe': (C.c_int, [vp, i64, i32, vp, i64, vp, vp, vp]),
    'hbk_cache_lookup_workspace_bytes': (sz, [i64]),
    'hbk_cache_lookup': (C.c_int, [vp, i64, i32, vp, i64, vp, vp, vp, vp, vp, vp, sz, vp]),
    'hbk_murmur3_hash32': (C.c_int, [vp, i64, vp, vp]),
    'hbk_comm_get_id': (C.c_int, [vp]),
    'hbk_comm_rccl_versions': (C.c_int, [vp, vp]),
    'hbk_comm_create': (C.c_int, [vp, vp, i32, i32, i32]),
    'hbk_comm_destroy': (C.c_int, [vp]),
    'hbk_comm_check_async': (C.c_int, [vp]),
    'hbk_comm_world_size': (C.c_int, [vp]),
    'hbk_comm_rank': (C.c_int, [vp]),
    'hbk_comm_stream': (vp, [vp]),
    'hbk_comm_active_ranks': (C.c_int, [vp, i32, vp]),
    'hbk_alltoall_n': (C.c_int, [vp, i32, i32, i32, vp, vp, vp, vp]),
    'hbk_alltoallv_wire_workspace_bytes': (sz, [i32, vp, vp, vp, i32]),
    'hbk_alltoallv_n':
      (C.c_int, [vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, sz, vp]),
    'hbk_allreduce_workspace_bytes': (sz, [i32, vp, i32]),
    'hbk_allreduce_n': (C.c_int, [vp, i32, i32, i32, vp, vp, vp, C.c_float, vp, sz, vp]),
    'hbk_allgatherv': (C.c_int, [vp, i32, vp, vp, vp, vp]),
    'hbk_comm_create_custom': (C.c_int, [vp, vp, i32, i32, i32]),
    'hbk_sharded_layout': (C.c_int, [i32, i32] + [vp] * 12),
    'hbk_sharded_create': (C.c_int, [vp, vp, i32, vp, i32]),
    'hbk_sharded_destroy': (C.c_int, [vp]),
    'hbk_sharded_set_hot_rows': (C.c_int, [vp, vp]),
    'hbk_sharded_lookup_fwd': (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp]),
    'hbk_sharded_prefetch_on': (C.c_int, [vp, vp, vp, vp]),
    'hbk_sharded_lookup_fwd_begin': (C.c_int, [vp, vp, vp, vp, vp, vp]),
    'hbk_sharded_lookup_fwd_end': (C.c_int, [vp, vp, vp, vp]),
    'hbk_sharded_prefetch': (C.c_int, [vp, vp, vp, vp]),
    'hbk_sharded_owned_ids': (i64, [vp, i32]),
    'hbk_sharded_last_host_us': (C.c_int, [vp, vp]),
    'hbk_sharded_lookup_bwd': (C.c_int, [vp, vp, vp, C.c_float, vp, vp, vp, vp]),
    'hbk_sharded_lookup_bwd_apply': (C.c_int, [vp, vp, vp, i32, C.c_float, vp, vp, vp, vp]),
  }
  for name, (res, args) in protos.items():
    fn = getattr(l, name)   # AttributeError here = header and library out of sync
    fn.restype = res
    fn.argtypes = args


def lib():
  """Load libhbk_core.so once.  Raises if it is missing -- never falls back."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise ImportError(
      f'{LIB_PATH} not found: the gfx950 HIP extension is not built. '
      'Run `python -c "import __graft_entry__ as g; g.build()"` or '
      '`make -C hybridbackend_amd/csrc`. There is no CPU fallback.')
  # torch first: its bundled libamdhip64.so.7 / librccl.so.1 must be the ones the
  # process uses, so the streams and pointers it hands us belong to the same runtime.
  import torch  # noqa: F401  pylint: disable=unused-import,import-outside-toplevel
  l = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
  _declare(l)
  _lib = l
  return l


_testing = None
TESTING_LIB_PATH = os.path.join(_HERE, 'lib', 'libhbk_testing.so')


def testing_lib():
  """tests/support's libhbk_testing.so (in-process world over hbk_comm_create_custom): test
  support, NOT part of the product; only ``Collective.local_world`` loads it."""
  global _testing
  if _testing is None:
    lib()
    if not os.path.exists(TESTING_LIB_PATH):
      raise ImportError(f'{TESTING_LIB_PATH} not found: run `make -C tests/support` '
                        '(__graft_entry__.build() does)')
    t = C.CDLL(TESTING_LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, i32 = C.c_void_p, C.c_int32
    t.hbk_testing_local_world_create.restype = C.c_int
    t.hbk_testing_local_world_create.argtypes = [vp, i32]
    t.hbk_testing_local_world_destroy.restype = C.c_int
    t.hbk_testing_local_world_destroy.argtypes = [vp]
    t.hbk_testing_set_wire.restype = C.c_int
    t.hbk_testing_set_wire.argtypes = [vp, C.c_double, C.c_double, C.c_double, i32]
    t.hbk_testing_comm_create.restype = C.c_int
    t.hbk_testing_comm_create.argtypes = [vp, vp, i32, i32]
    _testing = t
  return _testing


def set_option(name, value):
  """hbk_set_option: process-wide tuning / diagnostic knob; returns the previous value."""
  old = C.c_int32()
  check(lib().hbk_get_option(name.encode(), C.byref(old)))
  check(lib().hbk_set_option(name.encode(), int(value)))
  from hybridbackend_amd import _marshal
  _marshal.options_changed()   # cached workspace sizes depend on options
  return old.value


def get_option(name):
  """hbk_get_option: current value of a process-wide option."""
  v = C.c_int32()
  check(lib().hbk_get_option(name.encode(), C.byref(v)))
  return v.value


def check(status):
  if status != OK:
    msg = lib().hbk_last_error().decode('utf-8', 'replace')
    if status == INVALID_ARGUMENT:
      raise InvalidArgumentError(status, msg)
    raise HbkError(status, msg)


def ptr_array(ptrs):
  return (C.c_void_p * len(ptrs))(*ptrs)


def i64_array(vals):
  return (C.c_int64 * len(vals))(*vals)


def i32_array(vals):
  return (C.c_int32 * len(vals))(*vals)


_DTYPE_CODES = None


def torch_dtype_code(dtype):
  global _DTYPE_CODES
  if _DTYPE_CODES is None:
    import torch  # pylint: disable=import-outside-toplevel
    table = {torch.int8: INT8, torch.uint8: UINT8, torch.int32: INT32,
             torch.int64: INT64, torch.float16: HALF, torch.float32: FLOAT,
             torch.float64: DOUBLE}
    for name, code in (('uint32', UINT32), ('uint64', UINT64)):
      if hasattr(torch, name):
        table[getattr(torch, name)] = code
    _DTYPE_CODES = table
  code = _DTYPE_CODES.get(dtype)
  if code is None:
    raise InvalidArgumentError(INVALID_ARGUMENT, f'unsupported dtype {dtype}')
  return code


def current_stream(device=None):
  import torch  # pylint: disable=import-outside-toplevel
  return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_device_tensor(t, what, row_strided=False):
  """row_strided: a 2-D tensor may be a column block of a wider one (rows contiguous, row stride
  larger than the row) -- outputs / gradients of the fused lookup."""
  if not t.is_cuda:
    raise HbkError(
      INTERNAL,
      f'{what} must live in HBM (got a {t.device} tensor): the HIP path is the '
      'only path, there is no CPU fallback')
  if t.is_contiguous():
    return
  if row_strided and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]:
    return
  raise InvalidArgumentError(INVALID_ARGUMENT, f'{what} must be contiguous')
