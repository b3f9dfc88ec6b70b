"""Owner-side unique: ``array_ops.unique`` of
hybridbackend/tensorflow/embedding/sharding.py:186 for N columns in one launch group
(first-occurrence order, like TF)."""
import ctypes as C

import numpy as np
import torch

from hybridbackend_amd import _lib
from hybridbackend_amd import _marshal

_shape_plans = {}   # lengths -> (total, run offsets, workspace bytes)


def unique_n(ids_list, lazy=False):
  """Returns a list with, per column, ``(unique int64[len], index int32[len], n_unique int32[1])``;
  ``unique[:n_unique]`` are the distinct ids in first-occurrence order and
  ``unique[index] == ids``.  The count stays on the device (no host sync).  ``lazy=True``: a lazy
  sequence of the same tuples (``_marshal.Zipped``; views made when indexed)."""
  lib = _lib.lib()
  n = len(ids_list)
  if n == 0:
    return []
  dev = ids_list[0].device
  seen = _marshal.vector_pass(ids_list, (torch.int64,))
  if seen is not None:
    # fresh tensors every step: one pass over the inputs, three allocations, lazy per-column views
    ptrs, counts, _ = seen
    # (unique_buckets_log2 changes the workspace size: the options generation is part of the key)
    key = (tuple(counts), _marshal.options_generation())
    plan = _shape_plans.get(key)
    if plan is None:
      c_np = np.asarray(counts, dtype=np.int64)
      offs = np.zeros(n, dtype=np.uint64)
      offs[1:] = np.cumsum(c_np)[:-1]
      if len(_shape_plans) > 256:
        _shape_plans.clear()
      plan = _shape_plans[key] = (int(c_np.sum()), offs,
                                  lib.hbk_unique_workspace_bytes(n, _lib.i64_array(counts)),
                                  np.arange(n, dtype=np.uint64) * np.uint64(4))
    total, offs, need, n_offs = plan
    flat_u = torch.empty(total, dtype=torch.int64, device=dev)
    flat_i = torch.empty(total, dtype=torch.int32, device=dev)
    flat_n = torch.empty(n, dtype=torch.int32, device=dev)
    blk, addr = _marshal.arg_block(n, 5)
    blk[0] = ptrs
    blk[1] = counts
    blk[2] = offs * np.uint64(8) + np.uint64(flat_u.data_ptr())
    blk[3] = offs * np.uint64(4) + np.uint64(flat_i.data_ptr())
    blk[4] = n_offs + np.uint64(flat_n.data_ptr())
    stream = torch.cuda.current_stream(dev).cuda_stream
    ws = _workspace(max(need, 8), dev, stream)
    row = n * 8
    _lib.check(lib.hbk_unique_n(n, addr, addr + row, addr + 2 * row, addr + 3 * row,
                                addr + 4 * row, C.c_void_p(ws.data_ptr()), C.c_size_t(ws.numel()),
                                C.c_void_p(stream)))
    if lazy:
      return _marshal.Zipped(_marshal.Runs(flat_u, counts), _marshal.Runs(flat_i, counts),
                             _marshal.Runs(flat_n, [1] * n))
    return list(zip(torch.split(flat_u, counts), torch.split(flat_i, counts),
                    torch.split(flat_n, 1)))
  for t in ids_list:
    _lib.require_device_tensor(t, 'ids')
    if t.dtype != torch.int64 or t.dim() != 1:
      raise _lib.InvalidArgumentError(_lib.INVALID_ARGUMENT, 'ids must be an int64 vector')
  counts = [int(t.numel()) for t in ids_list]
  lens = _lib.i64_array(counts)
  # three allocations for all columns; views from one split call each, addresses by arithmetic
  # (every count is written by the call: no zero fill)
  total = sum(counts)
  flat_u = torch.empty(total, dtype=torch.int64, device=dev)
  flat_i = torch.empty(total, dtype=torch.int32, device=dev)
  flat_n = torch.empty(n, dtype=torch.int32, device=dev)
  uniq = torch.split(flat_u, counts)
  idx = torch.split(flat_i, counts)
  nu = torch.split(flat_n, 1)
  offs = [0] * n
  for c in range(1, n):
    offs[c] = offs[c - 1] + counts[c - 1]
  pu, pi, pn = flat_u.data_ptr(), flat_i.data_ptr(), flat_n.data_ptr()
  need = lib.hbk_unique_workspace_bytes(n, lens)
  ws = _workspace(max(need, 8), dev)
  _lib.check(lib.hbk_unique_n(
    n, _lib.ptr_array([t.data_ptr() for t in ids_list]), lens,
    _lib.ptr_array([pu + 8 * o for o in offs]),
    _lib.ptr_array([pi + 4 * o for o in offs]),
    _lib.ptr_array([pn + 4 * c for c in range(n)]),
    C.c_void_p(ws.data_ptr()), C.c_size_t(ws.numel()), _lib.current_stream(dev)))
  return list(zip(uniq, idx, nu))


class UniqueN:
  """A bound ``unique_n`` for loops over the same id buffers: outputs, workspace and the C-ABI
  arguments are set up once by ``bind``; ``launch`` is one foreign call; calling the object (``__call__(ids_list)``)
  re-binds only when it is handed other tensors.  Returns what ``unique_n`` returns."""

  def __init__(self):
    self._bound = None

  def bind(self, ids_list):
    lib = _lib.lib()
    ids_list = list(ids_list)
    n = len(ids_list)
    if n == 0:
      raise _lib.InvalidArgumentError(_lib.INVALID_ARGUMENT, 'no inputs')
    dev = ids_list[0].device
    for t in ids_list:
      _lib.require_device_tensor(t, 'ids')
      if t.dtype != torch.int64 or t.dim() != 1:
        raise _lib.InvalidArgumentError(_lib.INVALID_ARGUMENT, 'ids must be an int64 vector')
    counts = [int(t.numel()) for t in ids_list]
    lens = _lib.i64_array(counts)
    total = sum(counts)
    flat_u = torch.empty(total, dtype=torch.int64, device=dev)
    flat_i = torch.empty(total, dtype=torch.int32, device=dev)
    flat_n = torch.empty(n, dtype=torch.int32, device=dev)
    uniq, idx, nu = torch.split(flat_u, counts), torch.split(flat_i, counts), torch.split(flat_n, 1)
    need = lib.hbk_unique_workspace_bytes(n, lens)
    ws = torch.empty(max(need, 8), dtype=torch.uint8, device=dev)
    self._bound = dict(
      key=tuple(id(t) for t in ids_list), ptrs=[t.data_ptr() for t in ids_list], lens=counts,
      keep=(ids_list, flat_u, flat_i, flat_n, ws), n=n, device=dev,
      outputs=list(zip(uniq, idx, nu)),
      args=(_lib.ptr_array([t.data_ptr() for t in ids_list]), lens,
            _lib.ptr_array([t.data_ptr() for t in uniq]),
            _lib.ptr_array([t.data_ptr() for t in idx]),
            _lib.ptr_array([t.data_ptr() for t in nu]),
            C.c_void_p(ws.data_ptr()), C.c_size_t(ws.numel())))
    return self._bound['outputs']

  def launch(self):
    b = self._bound
    _lib.check(_lib.lib().hbk_unique_n(b['n'], *b['args'], _lib.current_stream(b['device'])))
    return b['outputs']

  def __call__(self, ids_list):
    b = self._bound
    if b is None or b['key'] != tuple(id(t) for t in ids_list) or any(
        t.data_ptr() != q or t.numel() != m for t, q, m in zip(ids_list, b['ptrs'], b['lens'])):
      self.bind(ids_list)
    return self.launch()


_ws = {}


def _workspace(nbytes, dev, stream_handle=None):
  """Grow-only scratch, one per (device, stream) like the partition op's."""
  if stream_handle is None:
    stream_handle = torch.cuda.current_stream(dev).cuda_stream
  key = (dev, stream_handle)
  buf = _ws.get(key)
  if buf is None or buf.numel() < nbytes:
    buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
    _ws[key] = buf
  return buf


def unique(ids):
  """``tf.unique`` semantics for one vector: (unique values, index); syncs the host once to
  size the result, as TF's kernel does."""
  u, idx, nu = unique_n([ids])[0]
  return u[:int(nu.item())], idx
