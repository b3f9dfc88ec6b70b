"""GroupLookup: fused multi-table sparse embedding lookup (forward) -- the host side
of the new additive op ``HbGroupLookup`` (include/hbk.h) that replaces, for N columns
at once, what the reference builds per column out of stock TF ops:

  ``feature % embedding_size`` (docs/tutorial/ranking/data.py:179,186)
  -> ``tf.nn.embedding_lookup_sparse(weights, sp_ids, None)`` (data.py:189), whose
     ``embedding_lookup`` is patched by hybridbackend/tensorflow/embedding/sharding.py:171-205.

Ragged id lists use the values + row_splits layout HybridBackend's own data path
produces (hybridbackend/tensorflow/data/dataframe.py:366-376).
"""
import ctypes as C

import numpy as np
import torch

from hybridbackend_amd import _lib
from hybridbackend_amd import _marshal

_COMBINERS = {None: _lib.COMBINER_MEAN,  # embedding_lookup_sparse default
              'sum': _lib.COMBINER_SUM, 'mean': _lib.COMBINER_MEAN,
              'sqrtn': _lib.COMBINER_SQRTN}


def _combiner_code(c):
  if isinstance(c, int):
    return c
  if c not in _COMBINERS:
    raise _lib.InvalidArgumentError(
      _lib.INVALID_ARGUMENT, "combiner must be one of 'mean', 'sqrtn' or 'sum'")
  return _COMBINERS[c]


class GroupLookup:
  """A group of N embedding columns looked up with one launch.

  Args:
    tables: list of fp32 ``[rows, dim]`` device tensors (the local shard when sharded).
    buckets: per-column ``embedding_size`` for the fused bucketize, or None/0.
    combiners: per-column 'sum' | 'mean' | 'sqrtn' | None (= mean), or one for all.
    divisor: ``row = id // divisor`` after bucketize (owner side of a sharded table,
      sharding.py:189); 1 for a whole table.
    hot_rows: skewed ids expected (Zipf heads): wide one-id-per-sample columns (dim >= 64) fetch
      every row repeated inside a 256-sample tile once and serve the repeats from LDS -- the
      forward's counterpart of the reference's slab cache in front of the table
      (hbtf/embedding/lookup_functors.cu.cc:54-149).  One bool for all columns or one per column;
      ``'auto'``: the column starts without and follows the data -- every backward of a
      ``GroupLookupGrad`` over this lookup leaves the number of distinct rows of the step on the
      device; it is copied to pinned memory behind the backward (no host wait) and the next
      forward that finds the copy landed turns the staging on for the columns whose last batch
      named fewer than half as many distinct rows as ids, off for the others (the reference's
      cache in front of the table is switched by the user, service.py:87; here the engine looks).
  """

  def __init__(self, tables, buckets=None, combiners='sum', divisor=1, hot_rows=False):
    self._lib = _lib.lib()
    self.tables = list(tables)
    n = len(self.tables)
    for t in self.tables:
      _lib.require_device_tensor(t, 'embedding weights')
      if t.dtype != torch.float32 or t.dim() != 2:
        raise _lib.InvalidArgumentError(
          _lib.INVALID_ARGUMENT, 'embedding weights must be fp32 [rows, dim]')
    if buckets is None:
      buckets = [0] * n
    if isinstance(combiners, (str, int)) or combiners is None:
      combiners = [combiners] * n
    self.buckets = [int(b or 0) for b in buckets]
    self.combiners = [_combiner_code(c) for c in combiners]
    self.divisor = int(divisor)
    if isinstance(hot_rows, (bool, int, str)):
      hot_rows = [hot_rows] * n
    self._auto_hot = [c for c in range(n) if hot_rows[c] == 'auto']
    hot_rows = [False if h == 'auto' else h for h in hot_rows]
    self._auto_state = None   # (pinned counts, event, ids per column) of the last backward
    self._cols = (_lib.LookupColumn * n)()
    # the same descriptors as a numpy record array: a step's pointers / counts are written one
    # FIELD at a time for all columns (bind), not one ctypes attribute at a time
    self._cols_np = np.frombuffer(self._cols, dtype=np.dtype(_lib.LookupColumn)) if n else None
    self._dims = [int(t.shape[1]) for t in self.tables]
    for c, t in enumerate(self.tables):
      col = self._cols[c]
      col.hot_rows = 1 if hot_rows[c] else 0
      col.table = t.data_ptr()
      col.rows = t.shape[0]
      col.dim = t.shape[1]
      col.bucket = self.buckets[c]
      col.divisor = self.divisor
      col.combiner = self.combiners[c]

  def __len__(self):
    return len(self.tables)

  def bind(self, ids, row_splits=None, outs=None, lazy=False):
    """Point the column descriptors at this step's inputs/outputs; returns outs -- a LIST of the
    per-column ``[segments, dim]`` tensors (``lazy=True`` with ``outs=None``: a lazy sequence whose
    views are made when indexed, for callers that hand the result on untouched)."""
    # the descriptors change: what __call__ remembers of its last tensors no longer describes them
    # (__call__ sets its key again after a bind of its own)
    self._call_key = None
    n = len(self.tables)
    if len(ids) != n:
      raise _lib.InvalidArgumentError(
        _lib.INVALID_ARGUMENT, f'expected {n} id tensors, got {len(ids)}')
    fast = self._bind_fresh(ids, row_splits, outs, lazy=lazy) if n else None
    if fast is not None:
      return fast
    if row_splits is None:
      row_splits = [None] * n
    if outs is None:
      outs = [None] * n
    outs = list(outs)
    for c in range(n):
      i = ids[c]
      _lib.require_device_tensor(i, 'ids')
      if i.dtype not in (torch.int32, torch.int64) or i.dim() != 1:
        raise _lib.InvalidArgumentError(
          _lib.INVALID_ARGUMENT, 'ids must be an int32/int64 vector')
      s = row_splits[c]
      if s is not None:
        _lib.require_device_tensor(s, 'row_splits')
        if s.dtype != torch.int32 or s.dim() != 1 or s.numel() < 1:
          raise _lib.InvalidArgumentError(
            _lib.INVALID_ARGUMENT, 'row_splits must be an int32 vector [segments+1]')
      n_seg = i.numel() if s is None else s.numel() - 1
      if outs[c] is None:
        outs[c] = torch.empty((n_seg, self.tables[c].shape[1]), dtype=torch.float32,
                              device=self.tables[c].device)
      o = outs[c]
      _lib.require_device_tensor(o, 'output', row_strided=True)
      if o.dtype != torch.float32 or tuple(o.shape) != (n_seg, self.tables[c].shape[1]):
        raise _lib.InvalidArgumentError(
          _lib.INVALID_ARGUMENT, f'output {c} must be fp32 [{n_seg}, dim]')
      col = self._cols[c]
      col.ids_dtype = _lib.INT64 if i.dtype == torch.int64 else _lib.INT32
      col.ids = i.data_ptr()
      col.n_ids = i.numel()
      col.row_splits = s.data_ptr() if s is not None else None
      col.n_segments = n_seg
      col.out = o.data_ptr()
      # a column block of a wider [segments, sum of dims] tensor is written in place
      if not o.is_contiguous() and o.stride(1) != 1 and o.shape[1] > 1:
        raise _lib.InvalidArgumentError(
          _lib.INVALID_ARGUMENT, f'output {c} must be contiguous along its last dimension')
      col.out_stride = 0 if o.is_contiguous() else int(o.stride(0))
    self._keep = (list(ids), list(row_splits), outs)
    return outs

  def bind_block(self, ids, row_splits, block, offsets):
    """bind() with every column's output a column block of ONE ``[segments, pitch]`` fp32 tensor
    (``block``; column c starts at float ``offsets[c]`` of every row): what DenseFeatures writes.
    No per-column views are made -- the addresses are arithmetic.  Returns False when the inputs
    need bind()'s detailed checks (the caller then makes views and calls bind)."""
    self._call_key = None
    n = len(self.tables)
    if len(ids) != n:        # (the same refusal as bind(): never a numpy broadcast error)
      raise _lib.InvalidArgumentError(
        _lib.INVALID_ARGUMENT, f'expected {n} id tensors, got {len(ids)}')
    if (n == 0 or block.dtype is not torch.float32 or not block.is_cuda or block.dim() != 2 or
        block.stride(1) != 1 or block.stride(0) % 4 != 0):
      return False
    n_rows, pitch = block.shape[0], block.stride(0)
    if len(offsets) != n or any(offsets[c] < 0 or offsets[c] + self._dims[c] > block.shape[1]
                                for c in range(n)):
      return False          # (a column block outside the row: bind() on the views says so)
    base = block.data_ptr()
    return self._bind_fresh(ids, row_splits, None,
                            block=(n_rows, pitch, [base + 4 * o for o in offsets], block)) is not None

  def _bind_fresh(self, ids, row_splits, outs, block=None, lazy=False):
    """bind() for the common shapes -- id vectors of one dtype, contiguous outputs (or none: one
    allocation, lazy per-column views) -- with ONE pass over the tensors and the descriptors
    written field by field for all columns.  None: something needs bind()'s detailed checks
    (its error messages, row-strided output blocks, mixed id dtypes)."""
    n = len(self.tables)
    seen = _marshal.vector_pass(ids, (torch.int32, torch.int64))
    if seen is None:
      return None
    id_ptrs, n_ids, id_dtype = seen
    if row_splits is None:
      row_splits = [None] * n
      sp_ptrs, n_seg = 0, n_ids
    else:
      sp_ptrs, n_seg = [], []
      for c in range(n):
        sp = row_splits[c]
        if sp is None:
          sp_ptrs.append(0)
          n_seg.append(n_ids[c])
          continue
        sh = sp.shape
        if (len(sh) != 1 or sh[0] < 1 or sp.dtype is not torch.int32 or not sp.is_cuda or
            not sp.is_contiguous()):
          return None
        sp_ptrs.append(sp.data_ptr())
        n_seg.append(sh[0] - 1)
    dims = self._dims
    stride = 0
    if block is not None:
      n_rows, stride, o_ptrs, keep = block
      if any(k != n_rows for k in n_seg):
        return None
      outs = keep
    elif outs is None:
      counts = [n_seg[c] * dims[c] for c in range(n)]
      pad = [(k + 3) // 4 * 4 for k in counts]     # every column's block on a 16-byte boundary
      flat = torch.empty(sum(pad) + 4, dtype=torch.float32, device=self.tables[0].device)
      base = flat.data_ptr()
      o_ptrs, at = [], 0
      for k in pad:
        o_ptrs.append(base + 4 * at)
        at += k
      outs = _LazyOutputs(flat, pad, n_seg, dims)
      if not lazy:
        outs = outs._materialise()
    else:
      o_ptrs = []
      for c in range(n):
        o = outs[c]
        if (o.dtype is not torch.float32 or not o.is_cuda or not o.is_contiguous() or
            tuple(o.shape) != (n_seg[c], dims[c])):
          return None
        o_ptrs.append(o.data_ptr())
      outs = list(outs)
    rec = self._cols_np
    rec['ids_dtype'] = _lib.INT64 if id_dtype is torch.int64 else _lib.INT32
    rec['ids'] = id_ptrs
    rec['n_ids'] = n_ids
    rec['row_splits'] = sp_ptrs
    rec['n_segments'] = n_seg
    rec['out'] = o_ptrs
    rec['out_stride'] = stride
    self._keep = (list(ids), list(row_splits), outs)   # (own lists: the caller may refill his)
    return outs

  # ---- hot rows by observation (hot_rows='auto') ----------------------------------------------
  def note_backward(self, n_unique, n_ids):
    """Called by GroupLookupGrad behind a backward: `n_unique` (device int32 [N]) holds the
    distinct rows of every column's batch once the stream gets there, `n_ids` the ids."""
    if not self._auto_hot:
      return
    st = self._auto_state
    if st is None or st[0].numel() != n_unique.numel():
      st = self._auto_state = [torch.empty(n_unique.numel(), dtype=torch.int32).pin_memory(),
                               torch.cuda.Event(), None, False]
    st[0].copy_(n_unique, non_blocking=True)
    st[1].record()
    st[2] = list(n_ids)
    st[3] = True

  def _poll_auto_hot(self):
    st = self._auto_state
    if st is None or not st[3] or not st[1].query():
      return
    st[3] = False
    counts = st[0].tolist()
    for c in self._auto_hot:
      n = st[2][c]
      self._cols[c].hot_rows = 1 if n > 0 and 2 * counts[c] < n else 0

  def launch(self, stream=None):
    """Enqueue the bound lookup on `stream` (a torch stream; default: current)."""
    if self._auto_state is not None:
      self._poll_auto_hot()
    if stream is None:
      s = _lib.current_stream(self.tables[0].device if self.tables else None)
    else:
      s = C.c_void_p(stream.cuda_stream)
    _lib.check(self._lib.hbk_group_lookup_fwd(len(self.tables), self._cols, s))

  def __call__(self, ids, row_splits=None, outs=None, lazy=False):
    # handed the SAME tensors as the call before (resident buffers refilled in place, caller-owned
    # outputs): the descriptors are still right, the call is one foreign call
    if outs is not None:
      tensors = list(ids) + [x for x in (row_splits or []) if x is not None] + list(outs)
      # the key keeps WHICH column a row_splits tensor belongs to (None positions included)
      key = (tuple(id(t) for t in ids),
             tuple(None if x is None else id(x) for x in (row_splits or ())),
             tuple(id(t) for t in outs))
      cached = getattr(self, '_call_key', None)
      if cached is not None and cached[0] == key and all(
          t.data_ptr() == q and t.numel() == m for t, (q, m) in zip(tensors, cached[1])):
        self.launch()
        return cached[2]
      outs = self.bind(ids, row_splits, outs)
      self._call_key = (key, [(t.data_ptr(), t.numel()) for t in tensors], outs)
    else:
      outs = self.bind(ids, row_splits, outs, lazy=lazy)   # (bind clears the remembered call)
    self.launch()
    return outs


class _LazyOutputs(_marshal.collections.abc.Sequence):
  """The per-column [segments, dim] outputs of one allocation, made when first asked for."""
  __slots__ = ('flat', 'pad', 'n_seg', 'dims', '_views')

  def __init__(self, flat, pad, n_seg, dims):
    self.flat, self.pad, self.n_seg, self.dims, self._views = flat, pad, n_seg, dims, None

  def _materialise(self):
    if self._views is None:
      pieces = torch.split(self.flat[:sum(self.pad)], self.pad)
      self._views = [p[:r * d].view(r, d) for p, r, d in zip(pieces, self.n_seg, self.dims)]
    return self._views

  def __len__(self):
    return len(self.dims)

  def __getitem__(self, i):
    return self._materialise()[i]

  def __iter__(self):
    return iter(self._materialise())


def group_lookup(tables, ids, row_splits=None, buckets=None, combiners='sum', divisor=1,
                 outs=None):
  """Functional form: one fused launch over N columns; returns the list of outputs."""
  return GroupLookup(tables, buckets, combiners, divisor)(ids, row_splits, outs)


class GroupLookupGrad:
  """Backward of :class:`GroupLookup` (new additive op ``HbGroupLookupGrad``): from the
  gradient of every column's combiner output to the ``IndexedSlices`` (unique local rows,
  summed gradient rows) TF hands to the optimizer on the shard -- the chain
  SparseSegment*Grad -> UnsortedSegmentSum of SURVEY 3.4 -- optionally fused with the
  sparse SGD apply (``apply_lr``; sharded variables skip cross-rank aggregation,
  hybridbackend/tensorflow/training/gradient.py:193-217).
  """

  def __init__(self, lookup, accums=None, interleaved=None, workspace_of=None, deterministic=False):
    """deterministic: every row's gradient terms are summed in id order (``HBK_GRAD_DETERMINISTIC`` on
    every column): IndexedSlices and stepped tables have the same bits on every run, equal to the
    sequential fp32 sum -- TF's CPU ``UnsortedSegmentSum`` -- and the rows leave ascending.  What the
    process-wide option ``bwd_deterministic`` = 1 (``TF_DETERMINISTIC_OPS`` through the TF shim) does
    for every call, chosen for this object only.

    accums: per column the Adagrad accumulator table (fp32, same shape as the weights,
    filled with ``initial_accumulator_value``), needed for ``optimizer='adagrad'``.

    interleaved: per column an fp32 ``[rows, 2 * dim]`` tensor that holds every row's weights AND
    accumulator side by side (``[:, :dim]`` the weights, ``[:, dim:]`` the accumulator): the fused
    optimizer step then works on THAT storage instead of ``lookup.tables`` / ``accums``
    (``hbk_lookup_grad_column_t.table_pitch`` = 2 dim) -- for dim <= 16 a row's weights and
    accumulator share one 128-byte line.  The forward keeps reading ``lookup.tables``: a trainer
    that uses this keeps the weights there in sync itself (a probe of the layout).

    workspace_of: another GroupLookupGrad whose scratch memory this one uses too (objects bound to
    different resident batches that run one after the other on one stream need one workspace, not
    one each)."""
    self._lib = _lib.lib()
    self.lookup = lookup
    n = len(lookup)
    self.accums = list(accums) if accums is not None else None
    self.interleaved = list(interleaved) if interleaved is not None else None
    if self.interleaved is not None and self.accums is None:
      self.accums = [b[:, b.shape[1] // 2:] for b in self.interleaved]   # (views: "accumulators exist")
    self._cols = (_lib.LookupGradColumn * n)()
    self._cols_np = np.frombuffer(self._cols, dtype=np.dtype(_lib.LookupGradColumn)) if n else None
    for c, t in enumerate(lookup.tables):
      col = self._cols[c]
      if self.accums is not None and self.interleaved is None:
        a = self.accums[c]
        _lib.require_device_tensor(a, 'accumulator')
        if a.dtype != torch.float32 or a.shape != t.shape:
          raise _lib.InvalidArgumentError(
            _lib.INVALID_ARGUMENT, f'accumulator {c} must be fp32 {tuple(t.shape)}')
        col.accum = a.data_ptr()
      col.table = t.data_ptr()
      col.rows = t.shape[0]
      col.dim = t.shape[1]
      if interleaved is not None:
        b = interleaved[c]
        _lib.require_device_tensor(b, 'interleaved weights + accumulator')
        if b.dtype != torch.float32 or tuple(b.shape) != (t.shape[0], 2 * t.shape[1]):
          raise _lib.InvalidArgumentError(
            _lib.INVALID_ARGUMENT, f'interleaved {c} must be fp32 [{t.shape[0]}, {2 * t.shape[1]}]')
        col.table = b.data_ptr()
        col.accum = b.data_ptr() + 4 * t.shape[1]
        col.table_pitch = 2 * t.shape[1]
      col.bucket = lookup.buckets[c]
      col.divisor = lookup.divisor
      col.combiner = lookup.combiners[c]
      col.flags = _lib.GRAD_DETERMINISTIC if deterministic else 0
    self._ws_box = workspace_of._ws_box if workspace_of is not None else [None]

  @property
  def _ws(self):
    return self._ws_box[0]

  @_ws.setter
  def _ws(self, t):
    self._ws_box[0] = t

  def _bind_fresh(self, ids, grads, row_splits, emit, block):
    """The descriptors of a step with tensors never seen before, in one pass and written field by
    field (cf. GroupLookup._bind_fresh).  `block` = (tensor [segments, pitch], float offsets): the
    gradients are column blocks of one tensor (DenseFeatures), addressed by arithmetic.  False:
    the general path's checks are needed."""
    n = len(self.lookup)
    if len(ids) != n or len(row_splits) != n or (block is None and len(grads) != n):
      return False           # (the general path raises InvalidArgumentError with the counts)
    seen = _marshal.vector_pass(ids, (torch.int32, torch.int64))
    if seen is None:
      return False
    id_ptrs, n_ids, id_dtype = seen
    if all(s is None for s in row_splits):
      sp_ptrs, n_seg = 0, n_ids
    else:
      sp_ptrs, n_seg = [], []
      for c in range(n):
        sp = row_splits[c]
        if sp is None:
          sp_ptrs.append(0)
          n_seg.append(n_ids[c])
          continue
        sh = sp.shape
        if (len(sh) != 1 or sh[0] < 1 or sp.dtype is not torch.int32 or not sp.is_cuda or
            not sp.is_contiguous()):
          return False
        sp_ptrs.append(sp.data_ptr())
        n_seg.append(sh[0] - 1)
    dims = [int(t.shape[1]) for t in self.lookup.tables]
    if block is not None:
      g, offsets = block
      if (g.dtype is not torch.float32 or not g.is_cuda or g.dim() != 2 or g.stride(1) != 1 or
          g.stride(0) % 4 != 0 or any(k != g.shape[0] for k in n_seg) or len(offsets) != n or
          any(offsets[c] < 0 or offsets[c] + dims[c] > g.shape[1] for c in range(n))):
        return False
      base, stride = g.data_ptr(), g.stride(0)
      g_ptrs = [base + 4 * o for o in offsets]
    else:
      g_ptrs, stride = [], []
      for c in range(n):
        g = grads[c]
        sh = g.shape
        if (g.dtype is not torch.float32 or not g.is_cuda or len(sh) != 2 or sh[0] != n_seg[c] or
            sh[1] != dims[c] or (sh[1] > 1 and g.stride(1) != 1) or g.stride(0) < sh[1]):
          return False
        g_ptrs.append(g.data_ptr())
        stride.append(0 if g.is_contiguous() else g.stride(0))
    rec = self._cols_np
    rec['ids_dtype'] = _lib.INT64 if id_dtype is torch.int64 else _lib.INT32
    rec['ids'] = id_ptrs
    rec['n_ids'] = n_ids
    rec['row_splits'] = sp_ptrs
    rec['n_segments'] = n_seg
    rec['grad_out'] = g_ptrs
    rec['grad_stride'] = stride
    u_ptrs, g_out_ptrs, nu_ptrs = self._out_ptrs
    rec['unique_rows'] = u_ptrs if emit else 0
    rec['grad_rows'] = g_out_ptrs if emit else 0
    rec['n_unique'] = nu_ptrs
    return True

  def __call__(self, ids, grads, row_splits=None, apply_lr=0.0, optimizer='sgd', emit=True,
               grad_block=None):
    """Returns per column ``(unique_rows int64[n_ids], grad_rows f32[n_ids, dim],
    n_unique int32[1])``; only the first ``n_unique`` rows are meaningful, in unspecified
    order (device-side count: no host sync here).  The result buffers belong to this object
    and are reused by the next call with the same id counts.  ``emit=False`` (with ``apply_lr``):
    step only -- the rows are stepped, no IndexedSlices are written; only ``n_unique`` of each
    returned triple is meaningful."""
    if not emit and apply_lr == 0.0:
      raise _lib.InvalidArgumentError(_lib.INVALID_ARGUMENT, 'emit=False needs apply_lr != 0')
    n = len(self.lookup)
    if row_splits is None:
      row_splits = [None] * n
    dev = self.lookup.tables[0].device if n else None
    dims = [int(t.shape[1]) for t in self.lookup.tables]
    counts = tuple(int(i.numel()) for i in ids) if emit else ('step only',)
    if getattr(self, '_out_key', None) != counts and not emit:
      self._nu = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
      self._urows = self._grows = None
      self._views = [(None, None, self._nu[c:c + 1]) for c in range(n)]
      self._out_ptrs = (0, 0, [self._nu.data_ptr() + 4 * c for c in range(n)])
      self._out_key = counts
    if getattr(self, '_out_key', None) != counts:
      # three allocations for all columns, carved into per-column views
      self._urows = torch.empty(sum(counts), dtype=torch.int64, device=dev)
      # every column's block starts on a 16-byte boundary whatever the dims before it
      pad4 = lambda x: (x + 3) // 4 * 4   # noqa: E731
      self._grows = torch.empty(sum(pad4(k * d) for k, d in zip(counts, dims)) + 4,
                                dtype=torch.float32, device=dev)
      self._nu = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
      self._out_key = counts
      self._views = []
      o_r = o_g = 0
      for c in range(n):
        k, d = counts[c], dims[c]
        self._views.append((self._urows[o_r:o_r + k],
                            self._grows[o_g:o_g + k * d].view(k, d), self._nu[c:c + 1]))
        o_r += k
        o_g += pad4(k * d)
      self._out_ptrs = ([v[0].data_ptr() for v in self._views], [v[1].data_ptr() for v in self._views],
                        [self._nu.data_ptr() + 4 * c for c in range(n)])
    # handed the SAME tensors as the call before (resident buffers refilled in place): the
    # descriptors are still right -- validating and marshalling 26 columns is ~12 us of Python
    fresh = False
    if n > 0:
      # other tensors than the call before (told apart by the first id tensor): one pass over them.
      # The same first tensor twice in a row: the general path below, which remembers the step so
      # that a loop over resident buffers pays ~12 us from its third call on.
      bk = getattr(self, '_bound_key', None)
      maybe_same = (bk is not None and bk[0][1][:1] == (id(ids[0]),)) or \
          ids[0] is getattr(self, '_fresh_first', None)
      if grad_block is not None or not maybe_same:
        fresh = self._bind_fresh(ids, grads, row_splits, emit, grad_block)
        if fresh:
          self._bound_key = None
          self._fresh_first = ids[0]
    if grad_block is not None and not fresh:
      g, offsets = grad_block
      grads = [g[:, o:o + d] for o, d in zip(offsets, dims)]
    tensors = [] if fresh else list(ids) + list(grads) + [x for x in row_splits if x is not None]
    key = (emit, tuple(id(t) for t in tensors))
    cached = getattr(self, '_bound_key', None)
    same = fresh or (cached is not None and cached[0] == key and all(
      t.data_ptr() == q and t.numel() == m for t, (q, m) in zip(tensors, cached[1])))
    for c in range(0 if not same else n, n):
      i, g, s = ids[c], grads[c], row_splits[c]
      _lib.require_device_tensor(i, 'ids')
      _lib.require_device_tensor(g, 'grads', row_strided=True)
      if i.dtype not in (torch.int32, torch.int64) or i.dim() != 1:
        raise _lib.InvalidArgumentError(_lib.INVALID_ARGUMENT, 'ids must be an int32/int64 vector')
      if s is not None:
        _lib.require_device_tensor(s, 'row_splits')
        if s.dtype != torch.int32 or s.dim() != 1 or s.numel() < 1:
          raise _lib.InvalidArgumentError(
            _lib.INVALID_ARGUMENT, 'row_splits must be an int32 vector [segments+1]')
      n_seg = i.numel() if s is None else s.numel() - 1
      if g.dtype != torch.float32 or tuple(g.shape) != (n_seg, dims[c]):
        raise _lib.InvalidArgumentError(
          _lib.INVALID_ARGUMENT, f'grad {c} must be fp32 [{n_seg}, {dims[c]}]')
      urows, grows, nu = self._views[c]
      col = self._cols[c]
      col.ids_dtype = _lib.INT64 if i.dtype == torch.int64 else _lib.INT32
      col.ids = i.data_ptr()
      col.n_ids = i.numel()
      col.row_splits = s.data_ptr() if s is not None else None
      col.n_segments = n_seg
      col.grad_out = g.data_ptr()
      if not g.is_contiguous() and g.stride(1) != 1 and g.shape[1] > 1:
        raise _lib.InvalidArgumentError(
          _lib.INVALID_ARGUMENT, f'grad {c} must be contiguous along its last dimension')
      col.grad_stride = 0 if g.is_contiguous() else int(g.stride(0))
      col.unique_rows = urows.data_ptr() if emit else None
      col.grad_rows = grows.data_ptr() if emit else None
      col.n_unique = nu.data_ptr()
    if not same:
      self._bound_key = (key, [(t.data_ptr(), t.numel()) for t in tensors])
    if grad_block is not None:
      grads = grad_block[0]
    need = self._lib.hbk_group_lookup_bwd_workspace_bytes(n, self._cols)   # (depends on options too)
    if self._ws is None or self._ws.numel() < need:
      self._ws = torch.empty(max(need, 8), dtype=torch.uint8, device=dev)
    self._ws_bound = self._ws       # (launch(): the workspace this binding was sized for)
    self._keep = (ids, grads, row_splits)
    if optimizer not in ('sgd', 'adagrad'):
      raise _lib.InvalidArgumentError(_lib.INVALID_ARGUMENT, "optimizer must be 'sgd' or 'adagrad'")
    if optimizer == 'adagrad' and apply_lr != 0.0 and self.accums is None:
      raise _lib.InvalidArgumentError(
        _lib.INVALID_ARGUMENT, "optimizer='adagrad' needs GroupLookupGrad(lookup, accums=...)")
    _lib.check(self._lib.hbk_group_lookup_bwd_apply(
      n, self._cols, _lib.APPLY_ADAGRAD if optimizer == 'adagrad' else _lib.APPLY_SGD,
      C.c_float(apply_lr), C.c_void_p(self._ws.data_ptr()),
      C.c_size_t(self._ws.numel()), _lib.current_stream(dev)))
    if self.lookup._auto_hot:
      self.lookup.note_backward(self._nu, [int(i.numel()) for i in ids])
    self._bound_call = (emit, [int(i.numel()) for i in ids])
    return list(self._views)

  def launch(self, apply_lr=0.0, optimizer='sgd'):
    """The backward of the LAST call again, on the same tensors (a training loop over resident
    buffers that are refilled in place; bench.py): the descriptors and the workspace of that call are
    still right, so this is ONE foreign call -- no validation, no marshalling (the counterpart of
    ``GroupLookup.launch``).  Same emit mode as that call; returns the same result views."""
    bound = getattr(self, '_bound_call', None)
    if bound is None:
      raise _lib.HbkError(_lib.INTERNAL, 'launch() needs a call that bound the tensors first')
    emit, n_ids = bound
    if not emit and apply_lr == 0.0:
      raise _lib.InvalidArgumentError(_lib.INVALID_ARGUMENT, 'emit=False needs apply_lr != 0')
    if optimizer == 'adagrad' and apply_lr != 0.0 and self.accums is None:
      raise _lib.InvalidArgumentError(
        _lib.INVALID_ARGUMENT, "optimizer='adagrad' needs GroupLookupGrad(lookup, accums=...)")
    dev = self.lookup.tables[0].device if len(self.lookup) else None
    _lib.check(self._lib.hbk_group_lookup_bwd_apply(
      len(self.lookup), self._cols, _lib.APPLY_ADAGRAD if optimizer == 'adagrad' else _lib.APPLY_SGD,
      C.c_float(apply_lr), C.c_void_p(self._ws_bound.data_ptr()),
      C.c_size_t(self._ws_bound.numel()), _lib.current_stream(dev)))
    if self.lookup._auto_hot:
      self.lookup.note_backward(self._nu, n_ids)
    return list(self._views)
