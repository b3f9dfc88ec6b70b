"""Cache probe -- host mirror of the reference's ``HbLookup`` op
(hybridbackend/tensorflow/embedding/lookup_ops.cc:38-145; Python caller
hybridbackend/tensorflow/embedding/service.py:153-283)."""
import ctypes as C

import torch

from hybridbackend_amd import _lib

EMPTY_KEY = -2**63  # service.py:87


def murmur3_hash32(keys):
  """murmur3_hash32<int64, seed 0> of every key (hybridbackend/common/murmur3.cu.h:32-77);
  returned as int64 holding the uint32 value."""
  lib = _lib.lib()
  _lib.require_device_tensor(keys, 'keys')
  out = torch.empty(keys.numel(), dtype=torch.int32, device=keys.device)
  _lib.check(lib.hbk_murmur3_hash32(
    C.c_void_p(keys.data_ptr()), C.c_int64(keys.numel()), C.c_void_p(out.data_ptr()),
    _lib.current_stream(keys.device)))
  return out.to(torch.int64) & 0xffffffff


def probe(keys_cache, keys, cache_slab_size=32):
  """Per-key probe result: ``hit_slot[i]`` = index into ``keys_cache`` holding ``keys[i]``
  or -1 for a miss, plus the device-side miss count."""
  lib = _lib.lib()
  for t, what in ((keys_cache, 'keys_cache'), (keys, 'keys')):
    _lib.require_device_tensor(t, what)
    if t.dtype != torch.int64 or t.dim() != 1:
      raise _lib.InvalidArgumentError(_lib.INVALID_ARGUMENT, f'{what} must be an int64 vector')
  if cache_slab_size < 1 or keys_cache.numel() % max(cache_slab_size, 1) != 0:
    raise _lib.InvalidArgumentError(
      _lib.INVALID_ARGUMENT, 'keys_cache must hold a whole number of slabs')
  hit_slot = torch.empty(keys.numel(), dtype=torch.int64, device=keys.device)
  n_miss = torch.zeros(1, dtype=torch.int32, device=keys.device)
  _lib.check(lib.hbk_cache_probe(
    C.c_void_p(keys_cache.data_ptr()), C.c_int64(keys_cache.numel() // cache_slab_size),
    C.c_int32(cache_slab_size), C.c_void_p(keys.data_ptr()), C.c_int64(keys.numel()),
    C.c_void_p(hit_slot.data_ptr()), C.c_void_p(n_miss.data_ptr()),
    _lib.current_stream(keys.device)))
  return hit_slot, n_miss


def lookup(keys_cache, keys, cache_slab_size=32):
  """``HbLookup`` outputs (lookup_ops.cc:38-58): hit_keys_indices, hit_cache_indices,
  miss_keys_indices, miss_keys -- key order preserved inside each list (``hbk_cache_lookup``).
  Syncs the host once to size the outputs, as the reference op does (lookup_ops.cc:118-121)."""
  lib = _lib.lib()
  for t, what in ((keys_cache, 'keys_cache'), (keys, 'keys')):
    _lib.require_device_tensor(t, what)
    if t.dtype != torch.int64 or t.dim() != 1:
      raise _lib.InvalidArgumentError(_lib.INVALID_ARGUMENT, f'{what} must be an int64 vector')
  if cache_slab_size < 1 or keys_cache.numel() % max(cache_slab_size, 1) != 0:
    raise _lib.InvalidArgumentError(
      _lib.INVALID_ARGUMENT, 'keys_cache must hold a whole number of slabs')
  n, dev = keys.numel(), keys.device
  hit_idx = torch.empty(n, dtype=torch.int32, device=dev)
  hit_cache = torch.empty(n, dtype=torch.int64, device=dev)
  miss_idx = torch.empty(n, dtype=torch.int32, device=dev)
  miss_keys = torch.empty(n, dtype=torch.int64, device=dev)
  counts = torch.zeros(2, dtype=torch.int32, device=dev)
  need = lib.hbk_cache_lookup_workspace_bytes(n)
  ws = torch.empty(max(need, 8), dtype=torch.uint8, device=dev)
  _lib.check(lib.hbk_cache_lookup(
    C.c_void_p(keys_cache.data_ptr()), C.c_int64(keys_cache.numel() // cache_slab_size),
    C.c_int32(cache_slab_size), C.c_void_p(keys.data_ptr()), C.c_int64(n),
    C.c_void_p(hit_idx.data_ptr()), C.c_void_p(hit_cache.data_ptr()),
    C.c_void_p(miss_idx.data_ptr()), C.c_void_p(miss_keys.data_ptr()),
    C.c_void_p(counts.data_ptr()), C.c_void_p(ws.data_ptr()), C.c_size_t(ws.numel()),
    _lib.current_stream(dev)))
  n_hit, n_miss = (int(x) for x in counts.tolist())   # the op's one host sync
  return hit_idx[:n_hit], hit_cache[:n_hit], miss_idx[:n_miss], miss_keys[:n_miss]
