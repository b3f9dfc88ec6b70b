"""Owner-side unique: ``array_ops.unique`` of
hybridbackend/tensorflow/embedding/sharding.py:186 for N columns in one launch group
(first-occurrence order, like TF)."""
import ctypes as C

import torch

from hybridbackend_amd import _lib


def unique_n(ids_list):
  """Returns per column ``(unique int64[len], index int32[len], n_unique int32[1])``;
  ``unique[:n_unique]`` are the distinct ids in first-occurrence order and
  ``unique[index] == ids``.  The count stays on the device (no host sync)."""
  lib = _lib.lib()
  n = len(ids_list)
  if n == 0:
    return []
  dev = ids_list[0].device
  for t in ids_list:
    _lib.require_device_tensor(t, 'ids')
    if t.dtype != torch.int64 or t.dim() != 1:
      raise _lib.InvalidArgumentError(_lib.INVALID_ARGUMENT, 'ids must be an int64 vector')
  lens = _lib.i64_array([t.numel() for t in ids_list])
  uniq = [torch.empty_like(t) for t in ids_list]
  idx = [torch.empty(t.numel(), dtype=torch.int32, device=dev) for t in ids_list]
  nu = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in ids_list]
  need = lib.hbk_unique_workspace_bytes(n, lens)
  ws = torch.empty(max(need, 8), dtype=torch.uint8, device=dev)
  _lib.check(lib.hbk_unique_n(
    n, _lib.ptr_array([t.data_ptr() for t in ids_list]), lens,
    _lib.ptr_array([t.data_ptr() for t in uniq]),
    _lib.ptr_array([t.data_ptr() for t in idx]),
    _lib.ptr_array([t.data_ptr() for t in nu]),
    C.c_void_p(ws.data_ptr()), C.c_size_t(ws.numel()), _lib.current_stream(dev)))
  return list(zip(uniq, idx, nu))


def unique(ids):
  """``tf.unique`` semantics for one vector: (unique values, index); syncs the host once to
  size the result, as TF's kernel does."""
  u, idx, nu = unique_n([ids])[0]
  return u[:int(nu.item())], idx
