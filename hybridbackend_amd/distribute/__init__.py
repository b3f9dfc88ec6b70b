"""Collective / partition ops of the sharded-embedding path
(host mirror of ``hybridbackend/tensorflow/distribute``)."""
from hybridbackend_amd.distribute.collective import Collective
from hybridbackend_amd.distribute.collective import Topology
from hybridbackend_amd.distribute.collective import aggregate_gradients
from hybridbackend_amd.distribute.collective import alltoallv_offsets
from hybridbackend_amd.distribute.collective import broadcast
from hybridbackend_amd.distribute.collective import compute_active_ranks
from hybridbackend_amd.distribute.partition import PartitionByModuloN
from hybridbackend_amd.distribute.partition import partition_by_dual_modulo_n
from hybridbackend_amd.distribute.partition import partition_by_dual_modulo_stage_one
from hybridbackend_amd.distribute.partition import partition_by_dual_modulo_stage_two
from hybridbackend_amd.distribute.partition import partition_by_modulo
from hybridbackend_amd.distribute.partition import partition_by_modulo_n


def cast_n(values, dst_dtype):
  """fp32 <-> fp16 wire casts for N tensors in one launch (functor::CastN,
  hybridbackend/tensorflow/common/cast.h:40-54)."""
  import ctypes as C  # pylint: disable=import-outside-toplevel
  import torch  # pylint: disable=import-outside-toplevel
  from hybridbackend_amd import _lib  # pylint: disable=import-outside-toplevel
  if not values:
    return []
  lib = _lib.lib()
  for v in values:
    _lib.require_device_tensor(v, 'value')
  outs = [torch.empty(v.shape, dtype=dst_dtype, device=v.device) for v in values]
  _lib.check(lib.hbk_cast_n(
    len(values), _lib.torch_dtype_code(values[0].dtype), _lib.torch_dtype_code(dst_dtype),
    _lib.ptr_array([v.data_ptr() for v in values]),
    _lib.i64_array([v.numel() for v in values]),
    _lib.ptr_array([o.data_ptr() for o in outs]),
    _lib.current_stream(values[0].device)))
  del C
  return outs
