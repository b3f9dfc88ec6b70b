"""Integer partition ("shuffle") of id vectors -- host mirror of
``hybridbackend/tensorflow/distribute/partition/ops.py:58-221`` over the C ABI.

Same names, argument meaning and error behaviour as the reference's Python ops:
``partition_by_modulo(ids, num_partitions)`` returns ``(output, sizes, indices)``
with ``output[indices] == ids`` (partition_test.py:57-59).  The N-ary forms are
what the reference's ``Pack`` graph pass turns K independent ops into
(``HbPartitionByModuloN``, graph/common/packing.cc:124-575): one launch group for
all columns.
"""
import ctypes as C

import numpy as np
import torch

from hybridbackend_amd import _lib
from hybridbackend_amd import _marshal


class _Workspace:
  """Grow-only device scratch, ONE PER STREAM: the partition kernels of two streams (prefetch
  thread and training thread, or the in-process ranks of a test) run concurrently and must not
  share their histograms."""

  def __init__(self):
    self._bufs = {}

  def get(self, nbytes, device, stream_handle=None):
    if nbytes == 0:
      return None, 0
    if stream_handle is None:
      stream_handle = torch.cuda.current_stream(device).cuda_stream
    key = (device, stream_handle)
    buf = self._bufs.get(key)
    if buf is None or buf.numel() < nbytes:
      buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
      self._bufs[key] = buf
    return buf, buf.numel()


_ws = _Workspace()


_INT_DTYPES = tuple(d for d in (torch.int32, torch.int64, getattr(torch, 'uint32', None),
                                 getattr(torch, 'uint64', None)) if d is not None)
_shape_plans = {}   # (n, P, lengths) -> (total, run offsets, workspace bytes): pure functions of the key


def _partition_n_fresh(ids_list, num_partitions, modulus, stage, lazy=False):
  """The functional form with tensors it has never seen (every training step): one pass over the
  inputs, three allocations, arguments by vector arithmetic, lazy per-column views
  (``_marshal``).  None: some input needs the detailed checks of the general path."""
  seen = _marshal.vector_pass(ids_list, _INT_DTYPES)
  if seen is None or num_partitions < 1:
    return None
  ptrs, lens, dtype = seen
  lib = _lib.lib()
  n = len(ids_list)
  device = ids_list[0].device
  # (the workspace size depends on runtime options -- partition_sub_tiles, partition_onepass --
  # so the options generation is part of the key)
  key = (n, num_partitions, tuple(lens), _marshal.options_generation())
  plan = _shape_plans.get(key)
  if plan is None:
    lens_np = np.asarray(lens, dtype=np.int64)
    offs = np.zeros(n, dtype=np.uint64)
    offs[1:] = np.cumsum(lens_np)[:-1]
    need = lib.hbk_partition_workspace_bytes(n, _lib.i64_array(lens), num_partitions)
    if len(_shape_plans) > 256:
      _shape_plans.clear()
    plan = _shape_plans[key] = (int(lens_np.sum()), offs, need,
                                np.arange(n, dtype=np.uint64) * np.uint64(4 * num_partitions))
  total, offs, need, size_offs = plan
  flat_out = torch.empty(total, dtype=dtype, device=device)
  flat_idx = torch.empty(total, dtype=torch.int32, device=device)
  sizes2d = torch.empty((n, num_partitions), dtype=torch.int32, device=device)
  blk, addr = _marshal.arg_block(n, 5)
  blk[0] = ptrs
  blk[1] = lens
  blk[2] = offs * np.uint64(flat_out.element_size()) + np.uint64(flat_out.data_ptr())
  blk[3] = size_offs + np.uint64(sizes2d.data_ptr())
  blk[4] = offs * np.uint64(4) + np.uint64(flat_idx.data_ptr())
  stream = torch.cuda.current_stream(device).cuda_stream
  ws, ws_bytes = _ws.get(need, device, stream)
  row = n * 8
  args = (addr, addr + row, addr + 2 * row, addr + 3 * row, addr + 4 * row,
          ws.data_ptr() if ws is not None else None, ws_bytes, C.c_void_p(stream))
  code = _lib.torch_dtype_code(dtype)
  if stage == 0:
    _lib.check(lib.hbk_partition_by_modulo_n(n, code, num_partitions, *args))
  else:
    _lib.check(lib.hbk_partition_by_dual_modulo_n(n, code, num_partitions, modulus, stage, *args))
  if lazy:
    return (_marshal.Runs(flat_out, lens), _marshal.Rows(sizes2d), _marshal.Runs(flat_idx, lens))
  return (list(torch.split(flat_out, lens)), list(sizes2d.unbind(0)),
          list(torch.split(flat_idx, lens)))


def _partition_n(ids_list, num_partitions, modulus, stage, outputs=None, lazy=False):
  lib = _lib.lib()
  n = len(ids_list)
  if n == 0:
    return [], [], []
  if outputs is None:
    res = _partition_n_fresh(ids_list, num_partitions, modulus, stage, lazy)
    if res is not None:
      return res
  dtype = ids_list[0].dtype
  device = ids_list[0].device
  for t in ids_list:
    _lib.require_device_tensor(t, 'ids')
    if t.dim() != 1:
      raise _lib.InvalidArgumentError(
        _lib.INVALID_ARGUMENT, 'Input must be a vector')  # partition_by_modulo_ops.cc:81-83
    if t.dtype != dtype:
      raise _lib.InvalidArgumentError(
        _lib.INVALID_ARGUMENT, 'all inputs of an N-ary partition share one dtype')
  code = _lib.torch_dtype_code(dtype)
  lens = [int(t.numel()) for t in ids_list]
  if outputs is None:
    # three allocations for all N columns (views per column), not 3 N
    total = sum(lens)
    flat_out = torch.empty(total, dtype=dtype, device=device)
    flat_idx = torch.empty(total, dtype=torch.int32, device=device)
    sizes2d = torch.empty((n, max(num_partitions, 0)), dtype=torch.int32, device=device)
    # the per-column views come from one split call each and their addresses from arithmetic:
    # building and querying 3 N tensor views one by one was most of the op's 100 us through Python
    outs = list(torch.split(flat_out, lens))
    idxs = list(torch.split(flat_idx, lens))
    sizes = list(sizes2d.unbind(0))
    offs = [0] * n
    for c in range(1, n):
      offs[c] = offs[c - 1] + lens[c - 1]
    p_out, p_idx, p_sz = flat_out.data_ptr(), flat_idx.data_ptr(), sizes2d.data_ptr()
    isz, row = flat_out.element_size(), 4 * max(num_partitions, 0)
    out_ptrs = [p_out + o * isz for o in offs]
    idx_ptrs = [p_idx + o * 4 for o in offs]
    size_ptrs = [p_sz + c * row for c in range(n)]
  else:
    outs, sizes, idxs = outputs
    out_ptrs = [t.data_ptr() for t in outs]
    idx_ptrs = [t.data_ptr() for t in idxs]
    size_ptrs = [t.data_ptr() for t in sizes]
  lens_a = _lib.i64_array(lens)
  need = lib.hbk_partition_workspace_bytes(n, lens_a, num_partitions)
  ws, ws_bytes = _ws.get(need, device)
  args = (_lib.ptr_array([t.data_ptr() for t in ids_list]), lens_a,
          _lib.ptr_array(out_ptrs), _lib.ptr_array(size_ptrs), _lib.ptr_array(idx_ptrs),
          ws.data_ptr() if ws is not None else None, ws_bytes,
          _lib.current_stream(device))
  if stage == 0:
    _lib.check(lib.hbk_partition_by_modulo_n(n, code, num_partitions, *args))
  else:
    _lib.check(lib.hbk_partition_by_dual_modulo_n(
      n, code, num_partitions, modulus, stage, *args))
  return outs, sizes, idxs


def partition_by_modulo(ids, num_partitions, name=None):
  r'''Shuffle IDs using floormod strategy (ops.py:88-105; op HbPartitionByModulo).

  Returns:
    output: A tensor with shuffled IDs.
    sizes: Size of each shard in output.
    indices: Indices for gathering back.
  '''
  del name
  o, s, i = _partition_n([ids], num_partitions, 1, 0)
  return o[0], s[0], i[0]


def partition_by_modulo_n(ids_list, num_partitions, name=None, lazy=False):
  r'''N-ary form (op HbPartitionByModuloN, partition_by_modulo_ops.cc:124-143).

  Returns three LISTS of tensors (outputs, sizes, indices), one entry per column.  ``lazy=True``:
  the three are lazy sequences instead (``_marshal.Runs``): the per-column views are only made
  when indexed -- ~35 us less Python per call on 26 columns for a caller that passes them on.'''
  del name
  return _partition_n(list(ids_list), num_partitions, 1, 0, lazy=lazy)


def partition_by_dual_modulo_stage_one(ids, num_partitions, modulus, name=None):
  r'''Stage 1 of the two-staged (local modulo, global modulo) shuffle
  (ops.py:108-164; op HbPartitionByDualModuloStageOne).'''
  del name
  o, s, i = _partition_n([ids], num_partitions, modulus, 1)
  return o[0], s[0], i[0]


def partition_by_dual_modulo_stage_two(ids, num_partitions, modulus, name=None):
  r'''Stage 2 (ops.py:167-221; op HbPartitionByDualModuloStageTwo).'''
  del name
  o, s, i = _partition_n([ids], num_partitions, modulus, 2)
  return o[0], s[0], i[0]


def partition_by_dual_modulo_n(ids_list, num_partitions, modulus, stage, name=None, lazy=False):
  r'''N-ary dual-modulo shuffle; stage is 1 or 2.  Returns lists (``lazy``: see
  ``partition_by_modulo_n``).'''
  del name
  return _partition_n(list(ids_list), num_partitions, modulus, stage, lazy=lazy)


class PartitionByModuloN:
  r'''A bound N-ary partition for loops that shuffle the same id buffers every step (resident
  input batches refilled in place): ``bind`` validates the tensors, allocates the three output
  tensors and the workspace and marshals the C-ABI arguments ONCE; ``launch`` is one foreign call
  (the functional ``partition_by_modulo_n`` spends ~50 us in Python per call on 26 columns, the
  kernel 15 us).  calling the object (``__call__(ids_list)``) re-binds only when it is handed other tensors.

  Args: as ``partition_by_modulo_n`` / ``partition_by_dual_modulo_n`` (``stage`` 0 = plain).
  '''

  def __init__(self, num_partitions, modulus=1, stage=0):
    self.num_partitions, self.modulus, self.stage = int(num_partitions), int(modulus), int(stage)
    self._bound = None

  def bind(self, ids_list, outputs=None):
    lib = _lib.lib()
    ids_list = list(ids_list)
    n = len(ids_list)
    if n == 0:
      raise _lib.InvalidArgumentError(_lib.INVALID_ARGUMENT, 'no inputs')
    dtype, device = ids_list[0].dtype, ids_list[0].device
    for t in ids_list:
      _lib.require_device_tensor(t, 'ids')
      if t.dim() != 1:
        raise _lib.InvalidArgumentError(_lib.INVALID_ARGUMENT, 'Input must be a vector')
      if t.dtype != dtype:
        raise _lib.InvalidArgumentError(
          _lib.INVALID_ARGUMENT, 'all inputs of an N-ary partition share one dtype')
    lens = [int(t.numel()) for t in ids_list]
    P = self.num_partitions
    if outputs is None:
      total = sum(lens)
      flat_out = torch.empty(total, dtype=dtype, device=device)
      flat_idx = torch.empty(total, dtype=torch.int32, device=device)
      sizes2d = torch.empty((n, max(P, 0)), dtype=torch.int32, device=device)
      outputs = (list(torch.split(flat_out, lens)), list(sizes2d.unbind(0)),
                 list(torch.split(flat_idx, lens)))
    outs, sizes, idxs = outputs
    lens_a = _lib.i64_array(lens)
    need = lib.hbk_partition_workspace_bytes(n, lens_a, P)
    ws = torch.empty(max(need, 8), dtype=torch.uint8, device=device)   # the plan's own
    self._bound = dict(
      key=tuple(id(t) for t in ids_list), ptrs=[t.data_ptr() for t in ids_list], lens=lens,
      keep=(ids_list, outs, sizes, idxs, ws), n=n, code=_lib.torch_dtype_code(dtype),
      device=device, outputs=(outs, sizes, idxs),
      args=(_lib.ptr_array([t.data_ptr() for t in ids_list]), lens_a,
            _lib.ptr_array([t.data_ptr() for t in outs]),
            _lib.ptr_array([t.data_ptr() for t in sizes]),
            _lib.ptr_array([t.data_ptr() for t in idxs]), ws.data_ptr(), ws.numel()))
    return self._bound['outputs']

  def launch(self):
    b = self._bound
    lib = _lib.lib()
    if self.stage == 0:
      rc = lib.hbk_partition_by_modulo_n(b['n'], b['code'], self.num_partitions, *b['args'],
                                         _lib.current_stream(b['device']))
    else:
      rc = lib.hbk_partition_by_dual_modulo_n(b['n'], b['code'], self.num_partitions,
                                              self.modulus, self.stage, *b['args'],
                                              _lib.current_stream(b['device']))
    _lib.check(rc)
    return b['outputs']

  def __call__(self, ids_list):
    b = self._bound
    if b is None or b['key'] != tuple(id(t) for t in ids_list) or any(
        t.data_ptr() != q or t.numel() != m for t, q, m in zip(ids_list, b['ptrs'], b['lens'])):
      self.bind(ids_list)
    return self.launch()
