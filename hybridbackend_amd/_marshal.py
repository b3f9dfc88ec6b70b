"""Cheap marshalling of N-column calls with FRESH tensors every step (round 4).

What a functional N-ary op costs in Python is not the foreign call (~5 us) but N x (attribute
reads + a view per output): creating 3 x 26 output views alone is ~48 us (0.6 us each) where the
partition kernel takes 15.  So:
  * every tensor is looked at ONCE (shape, dtype, device, contiguity, address in one pass); the
    detailed error message is produced by the slow checker only when that pass finds a problem;
  * outputs are carved from one allocation per kind; the per-column views are made by ONE
    ``torch.split`` call.  The public ops return PLAIN LISTS of tensors by default (round 5:
    ``torch.cat(outs)``, ``outs + [...]`` and ``isinstance(outs, list)`` must work whatever path
    a call took); ``lazy=True`` opts into the lazy sequences below (``Runs``: the flat tensor and
    the per-column extents are kept, the views are made the first time somebody indexes /
    iterates the sequence -- for callers that pass the result on without looking at it);
  * pointer / length arrays live in one per-thread numpy block per column count and are filled
    by vector arithmetic (run starts are offsets into the one allocation).
"""
import collections.abc
import threading

import numpy as np
import torch


class Runs(collections.abc.Sequence):
  """Per-column pieces of one flat tensor, materialised on first use.  ``flat``: the whole
  allocation; ``counts``: items per column (rows when ``row_shape`` gives every row's width)."""
  __slots__ = ('flat', 'counts', '_views')

  def __init__(self, flat, counts):
    self.flat = flat
    self.counts = counts
    self._views = None

  def _materialise(self):
    if self._views is None:
      self._views = list(torch.split(self.flat, self.counts))
    return self._views

  def __len__(self):
    return len(self.counts)

  def __getitem__(self, i):
    return self._materialise()[i]

  def __iter__(self):
    return iter(self._materialise())

  def __repr__(self):
    return f'Runs({len(self.counts)} columns of {self.flat.dtype})'


class Rows(collections.abc.Sequence):
  """Row c of a 2-D tensor as column c's piece (the [N, P] sizes of a partition), lazily."""
  __slots__ = ('block', '_views')

  def __init__(self, block):
    self.block = block
    self._views = None

  def _materialise(self):
    if self._views is None:
      self._views = list(self.block.unbind(0))
    return self._views

  def __len__(self):
    return self.block.shape[0]

  def __getitem__(self, i):
    return self._materialise()[i]

  def __iter__(self):
    return iter(self._materialise())


class Zipped(collections.abc.Sequence):
  """``zip`` of lazy sequences, itself lazy: element c is the tuple of the c-th pieces."""
  __slots__ = ('parts',)

  def __init__(self, *parts):
    self.parts = parts

  def __len__(self):
    return len(self.parts[0])

  def __getitem__(self, i):
    if isinstance(i, slice):
      return [tuple(p[k] for p in self.parts) for k in range(*i.indices(len(self)))]
    return tuple(p[i] for p in self.parts)


def plain(seq):
  """A lazy sequence as the plain list of tensors (or tuples of tensors) it stands for."""
  if isinstance(seq, Zipped):
    return list(zip(*[plain(p) for p in seq.parts]))
  if isinstance(seq, (Runs, Rows)):
    return seq._materialise()
  return list(seq)


_tls = threading.local()
_options_generation = [0]


def options_changed():
  """Called by ``_lib.set_option``: sizes cached from hbk_*_workspace_bytes depend on options."""
  _options_generation[0] += 1


def options_generation():
  return _options_generation[0]


def arg_block(n, rows):
  """A per-thread uint64 block [rows, n] (one row per pointer / length array of a call) and its
  address; row k is passed to the C ABI as address + k * n * 8.  Valid until the thread's next
  call with the same shape -- i.e. for the duration of one foreign call."""
  cache = getattr(_tls, 'blocks', None)
  if cache is None:
    cache = _tls.blocks = {}
  hit = cache.get((n, rows))
  if hit is None:
    blk = np.zeros((rows, n), dtype=np.uint64)
    hit = cache[(n, rows)] = (blk, blk.ctypes.data)
  return hit


def vector_pass(tensors, dtypes):
  """One pass over 1-D device tensors: returns (addresses, lengths, dtype) when every tensor is a
  contiguous device vector of ONE dtype out of `dtypes`, else None (the caller then runs its
  detailed checks to raise the right error)."""
  dt = tensors[0].dtype
  if dt not in dtypes:
    return None
  ptrs, lens = [], []
  for t in tensors:
    sh = t.shape
    if len(sh) != 1 or t.dtype is not dt or not t.is_cuda or not t.is_contiguous():
      return None
    lens.append(sh[0])
    ptrs.append(t.data_ptr())
  return ptrs, lens, dt


