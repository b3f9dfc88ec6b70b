"""Sharded group lookup -- the host-side pipeline driver (R12), mirror of
``hybridbackend/tensorflow/embedding/sharding.py:171-205`` for N columns at once (what the
reference's ``Pack`` pass makes of N per-column pipelines, graph/common/packing.cc:124-575).

Per rank and step (W = world size, owner of an id = ``id mod W``, local row = ``id // W``):

  1 bucketize + ``partition_by_modulo`` (sharding.py:182)    hbk_floormod_n, hbk_partition_by_modulo_n
  2 ``alltoall(ids_shards, sizes=ids_sizes)`` (:184)         ONE [N x W] size exchange + one
                                                             host sync for all N columns, then
                                                             hbk_alltoallv_n (int64 ids)
  3 ``unique`` / ``// W`` / gather / restore (:186-193)      hbk_group_lookup_fwd on the shard with
                                                             divisor = W (unique + restore are
                                                             value-transparent in the forward)
  4 ``alltoall(embeddings, sizes=shard_sizes)`` (:196)       hbk_alltoallv_n (fp32 or fp16 wire)
  5 ``gather(embeddings, shard_index)`` (:200) + combiner    hbk_group_lookup_fwd over the received
                                                             rows with ids = shard_index

The reference pays one host ``BlockHostUntilDone`` per exchange op (nccl_alltoallv.cc:316,533);
here the receive sizes of BOTH exchanges come from the single size exchange of step 2.

``__call__`` / ``backward`` run the whole step inside ONE C-ABI call each
(``hbk_sharded_lookup_fwd/_bwd``, csrc/sharded.hip: every exchange is one message per peer, host
cost tens of microseconds).  The compute phases also exist as separate methods so a test can
drive W virtual ranks through the kernels with its own transport.
"""
import ctypes as C
import threading

import torch

from hybridbackend_amd import _lib
from hybridbackend_amd.distribute import partition as _partition
from hybridbackend_amd.embedding.lookup import GroupLookup
from hybridbackend_amd.embedding.lookup import GroupLookupGrad


class _Step:
  """Per-step state carried between the phases (and kept for the backward)."""
  __slots__ = ('ids', 'row_splits', 'send_ids', 'send_sizes', 'shard_index', 'send_sizes_host',
               'recv_sizes_host', 'recv_ids', 'send_rows', 'recv_rows', 'outs')

  def __init__(self):
    for s in self.__slots__:
      setattr(self, s, None)


# plans read the process-wide sharded_* options when they are created: creating one under a
# temporarily changed option (PipelinedLookup) must not interleave with another thread's creation
# (in-process ranks are host threads)
_PLAN_LOCK = threading.RLock()


class _BoundStep:
  """One step's tensors marshalled for hbk_sharded_lookup_fwd (ShardedGroupLookup.bind)."""
  __slots__ = ('keep', 'outs', 'args', 'shapes')


class ShardedGroupLookup:
  """N row-sharded embedding tables looked up together.

  Args:
    shards: this rank's local rows per column, fp32 ``[rows_local, dim]``
      (``rows_local = R // W + (rank < R % W)``, variables.py:107-111).
    coll: a ``hybridbackend_amd.distribute.Collective`` (or None when only the phase
      methods are used).
    buckets: per-column ``embedding_size`` for the fused bucketize (ids are taken modulo it
      before the partition), or None when ids are already in ``[0, R)``.
    combiners: as for ``GroupLookup``.
    wire_dtype: ``torch.float16`` sends the embedding rows as fp16 (``comm_wire_dtype``,
      collective.py:291-296); None keeps fp32.
    hot_rows: as for ``GroupLookup`` (skewed ids: the owner-side gather of wide columns stages the
      rows repeated inside a tile in LDS); a bool or one per column.
    dedup: requester-side dedup, a bool or one per column: every DISTINCT id of the column's batch
      goes on the wire once -- what the reference's tutorials do in user code in front of the
      patched lookup (docs/tutorial/ranking/data.py:180-182: ``tf.unique`` -> lookup ->
      ``tf.gather``).  Same results; fewer ids out, fewer rows back and fewer gradient rows in the
      backward when ids repeat inside a batch (Zipf), at the price of a unique over the batch.
  """

  def __init__(self, shards, coll, buckets=None, combiners='sum', wire_dtype=None,
               world_size=None, accums=None, hot_rows=False, dedup=False):
    self.shards = list(shards)
    # Adagrad accumulators of the shards (same shapes), for backward(optimizer='adagrad')
    self.accums = list(accums) if accums is not None else None
    self.coll = coll
    self.world_size = int(world_size if world_size is not None else coll.world_size)
    n = len(self.shards)
    self.buckets = [int(b or 0) for b in (buckets or [0] * n)]
    self.combiners = combiners
    self.wire_dtype = wire_dtype
    self.device = self.shards[0].device if n else None
    self.dims = [int(t.shape[1]) for t in self.shards]
    if isinstance(hot_rows, (bool, int, str)):
      hot_rows = [hot_rows] * n
    # 'auto': off until a backward has shown how many distinct rows the owner was asked for
    self._auto_hot = [c for c in range(n) if hot_rows[c] == 'auto']
    self._auto_state = None
    self.hot_rows = [False if h == 'auto' else bool(h) for h in hot_rows]
    self.dedup = [bool(dedup)] * n if isinstance(dedup, (bool, int)) else [bool(d) for d in dedup]
    self._setup()

  def _setup(self):
    """Device-side state of the compute phases (the HIP path; there is no other)."""
    self._lib = _lib.lib()
    # owner-side gather: ids arrive bucketized, row = id // W (sharding.py:188-189)
    self._owner = GroupLookup(self.shards, None, 'sum', divisor=self.world_size,
                              hot_rows=self.hot_rows)
    self._owner_grad = GroupLookupGrad(self._owner)

  # ---- phase 1: bucketize + stable partition -------------------------------------
  def partition(self, ids, row_splits=None):
    st = _Step()
    n = len(self.shards)
    st.ids = list(ids)
    st.row_splits = list(row_splits) if row_splits is not None else [None] * n
    work = st.ids
    if any(self.buckets):
      work = [torch.empty_like(t) if b else t for t, b in zip(st.ids, self.buckets)]
      sel = [c for c in range(n) if self.buckets[c]]
      code = _lib.torch_dtype_code(st.ids[sel[0]].dtype)
      _lib.check(self._lib.hbk_floormod_n(
        len(sel), code, _lib.ptr_array([st.ids[c].data_ptr() for c in sel]),
        _lib.i64_array([st.ids[c].numel() for c in sel]),
        _lib.i64_array([self.buckets[c] for c in sel]),
        _lib.ptr_array([work[c].data_ptr() for c in sel]),
        _lib.current_stream(self.device)))
    W = self.world_size
    sizes = torch.empty((n, W), dtype=torch.int32, device=self.device)
    outs = [torch.empty_like(t) for t in work]
    idxs = [torch.empty(t.numel(), dtype=torch.int32, device=self.device) for t in work]
    _partition._partition_n(work, W, 1, 0, outputs=(outs, [sizes[c] for c in range(n)], idxs))
    st.send_ids, st.send_sizes, st.shard_index = outs, sizes, idxs
    return st

  # ---- phase 3: owner-side gather ------------------------------------------------------
  def owner_gather(self, st, recv_ids):
    st.recv_ids = recv_ids
    st.send_rows = self._owner(recv_ids)
    return st.send_rows

  # ---- phase 5: stitch + combine --------------------------------------------------------
  def stitch(self, st, recv_rows):
    st.recv_rows = recv_rows
    stitcher = GroupLookup(recv_rows, None, self.combiners, divisor=1)
    st.outs = stitcher(st.shard_index, st.row_splits)
    return st.outs

  # ---- the whole forward: ONE C-ABI call (hbk_sharded_lookup_fwd, csrc/sharded.hip) ----------
  def _plan(self):
    if getattr(self, '_plan_handle', None) is None:
      with _PLAN_LOCK:
        self._create_plan()
    return self._plan_handle

  def _create_plan(self):
    from hybridbackend_amd.embedding.lookup import _combiner_code
    n = len(self.shards)
    combs = self.combiners
    if isinstance(combs, (str, int)) or combs is None:
      combs = [combs] * n
    cols = (_lib.ShardedColumn * n)()
    for c, t in enumerate(self.shards):
      cols[c].shard = t.data_ptr()
      cols[c].rows_local = t.shape[0]
      cols[c].dim = t.shape[1]
      cols[c].combiner = _combiner_code(combs[c])
      cols[c].bucket = self.buckets[c]
      cols[c].hot_rows = 1 if self.hot_rows[c] else 0
      cols[c].dedup = 1 if self.dedup[c] else 0
      if self.accums is not None:
        cols[c].accum = self.accums[c].data_ptr()
    self._plan_handle = C.c_void_p()
    wire = _lib.HALF if self.wire_dtype == torch.float16 else _lib.FLOAT
    _lib.check(self._lib.hbk_sharded_create(
      C.byref(self._plan_handle), self.coll._handle, n, cols, wire))

  def last_host_us(self):
    """Host time of the last forward step in microseconds: (enqueueing the partition and the
    size exchange, waiting for the sizes -- the device, not host work --, enqueueing the rest)."""
    out = (C.c_float * 3)()
    _lib.check(self._lib.hbk_sharded_last_host_us(self._plan(), out))
    return tuple(float(v) for v in out)

  def p2p_bind(self, outs):
    """Register this rank's output tensors for the P2P FORM of the forward (round 5;
    ``hbk_sharded_p2p_bind``) -- a COLLECTIVE: every rank calls it with its own ``outs`` (fp32
    ``[n_ids[c], dim]`` per column, contiguous or column blocks of one wider tensor).  From then on
    every forward of this object must be handed exactly these tensors and one id per segment; a step
    sends (id, output row) pairs and the owner-side gather stores every row straight into the
    requester's output -- no reply buffer, no rows Alltoallv, no stitch: one random-row pass instead
    of two, and the gather itself is the exchange.  Returns True; False (on every rank) when some
    peer's memory cannot be mapped, in which case the object keeps the exchange form.  Not with
    ``dedup`` or the fp16 wire."""
    n = len(self.shards)
    if len(outs) != n:
      raise _lib.InvalidArgumentError(_lib.INVALID_ARGUMENT, f'p2p_bind: {len(outs)} outputs for {n} columns')
    for c, o in enumerate(outs):
      _lib.require_device_tensor(o, 'output', row_strided=True)
      if o.dtype != torch.float32 or o.dim() != 2 or o.shape[1] != self.dims[c]:
        raise _lib.InvalidArgumentError(_lib.INVALID_ARGUMENT, f'p2p_bind: output {c} must be fp32 [*, {self.dims[c]}]')
    strides = (C.c_int32 * n)(*[0 if o.is_contiguous() else int(o.stride(0)) for o in outs])
    rows = _lib.i64_array([int(o.shape[0]) for o in outs])
    rc = self._lib.hbk_sharded_p2p_bind(self._plan(), _lib.ptr_array([o.data_ptr() for o in outs]),
                                        strides, rows, _lib.current_stream(self.device))
    if rc == _lib.UNIMPLEMENTED:
      self._p2p_keep = None
      return False
    _lib.check(rc)
    self._p2p_keep = list(outs)
    return True

  def p2p_unbind(self):
    """Back to the exchange form (every rank must do the same before its next forward)."""
    _lib.check(self._lib.hbk_sharded_p2p_unbind(self._plan()))
    self._p2p_keep = None

  def close(self):
    if getattr(self, '_plan_handle', None) is not None:
      self._lib.hbk_sharded_destroy(self._plan_handle)
      self._plan_handle = None

  def bind(self, ids, row_splits=None, outs=None):
    """Validate one step's tensors and marshal them into the C-ABI argument arrays once; the
    returned object can be launched any number of times (``launch``) at the cost of a single
    foreign call -- what a training loop with resident input batches does (bench.py)."""
    n = len(self.shards)
    fast = self._bind_fresh(ids, row_splits, outs)
    if fast is not None:
      return fast
    if row_splits is None:
      row_splits = [None] * n
    n_seg = []
    for c in range(n):
      _lib.require_device_tensor(ids[c], 'ids')
      if ids[c].dtype != torch.int64 or ids[c].dim() != 1:
        raise _lib.InvalidArgumentError(_lib.INVALID_ARGUMENT, 'ids must be an int64 vector')
      s = row_splits[c]
      if s is not None:
        _lib.require_device_tensor(s, 'row_splits')
        if s.dtype != torch.int32 or s.dim() != 1 or s.numel() < 1:
          raise _lib.InvalidArgumentError(
            _lib.INVALID_ARGUMENT, 'row_splits must be an int32 vector [segments+1]')
      n_seg.append(ids[c].numel() if s is None else s.numel() - 1)
    if outs is None:
      outs = [torch.empty((n_seg[c], self.dims[c]), dtype=torch.float32, device=self.device)
              for c in range(n)]
    for c in range(n):
      _lib.require_device_tensor(outs[c], 'output', row_strided=True)
      if outs[c].dtype != torch.float32 or tuple(outs[c].shape) != (n_seg[c], self.dims[c]):
        raise _lib.InvalidArgumentError(
          _lib.INVALID_ARGUMENT, f'output {c} must be fp32 [{n_seg[c]}, {self.dims[c]}]')
    bound = _BoundStep()
    bound.shapes = [(n_seg[c], self.dims[c]) for c in range(n)]
    bound.keep = (list(ids), list(row_splits), list(outs))   # (own lists: the caller may refill his)
    bound.outs = outs
    # column blocks of one wider tensor are written in place (row stride != dim)
    bound.args = (
      _lib.ptr_array([t.data_ptr() for t in ids]), _lib.i64_array([t.numel() for t in ids]),
      _lib.ptr_array([None if s is None else s.data_ptr() for s in row_splits]),
      _lib.i64_array(n_seg), _lib.ptr_array([o.data_ptr() for o in outs]),
      (C.c_int32 * n)(*[0 if o.is_contiguous() else int(o.stride(0)) for o in outs]))
    return bound

  def _bind_fresh(self, ids, row_splits, outs):
    """bind() in one pass over the tensors for the common shapes (int64 id vectors, contiguous
    caller-owned outputs); None: bind()'s detailed checks / allocations are needed."""
    from hybridbackend_amd import _marshal
    n = len(self.shards)
    if outs is None or len(ids) != n or len(outs) != n:
      return None
    seen = _marshal.vector_pass(ids, (torch.int64,))
    if seen is None:
      return None
    id_ptrs, n_ids, _ = seen
    if row_splits is None:
      row_splits = [None] * n
      sp_ptrs, n_seg = [None] * n, n_ids
    else:
      sp_ptrs, n_seg = [], []
      for c in range(n):
        sp = row_splits[c]
        if sp is None:
          sp_ptrs.append(None)
          n_seg.append(n_ids[c])
          continue
        sh = sp.shape
        if (len(sh) != 1 or sh[0] < 1 or sp.dtype is not torch.int32 or not sp.is_cuda or
            not sp.is_contiguous()):
          return None
        sp_ptrs.append(sp.data_ptr())
        n_seg.append(sh[0] - 1)
    dims = self.dims
    o_ptrs = []
    for c in range(n):
      o = outs[c]
      if (o.dtype is not torch.float32 or not o.is_cuda or not o.is_contiguous() or
          tuple(o.shape) != (n_seg[c], dims[c])):
        return None
      o_ptrs.append(o.data_ptr())
    bound = _BoundStep()
    bound.shapes = [(n_seg[c], dims[c]) for c in range(n)]
    bound.keep = (list(ids), list(row_splits), list(outs))   # (own lists: the caller may refill his)
    bound.outs = outs
    bound.args = ((C.c_void_p * n)(*id_ptrs), (C.c_int64 * n)(*n_ids), (C.c_void_p * n)(*sp_ptrs),
                  (C.c_int64 * n)(*n_seg), (C.c_void_p * n)(*o_ptrs), (C.c_int32 * n)())
    return bound

  def launch(self, bound):
    """Enqueue a bound step on the current stream; returns its outputs."""
    self._keep = bound.keep
    self._last_shapes = bound.shapes   # what backward() differentiates
    st = self._auto_state
    if st is not None and st[3] and st[1].query():
      # the last backward's counts have landed: distinct local rows < half the ids asked for ->
      # the owner gather stages repeated rows in LDS from this step on (and stops when not)
      st[3] = False
      counts = st[0].tolist()
      for c in self._auto_hot:
        self.hot_rows[c] = st[2][c] > 0 and 2 * counts[c] < st[2][c]
      _lib.check(self._lib.hbk_sharded_set_hot_rows(
        self._plan(), (C.c_int32 * len(self.hot_rows))(*[int(h) for h in self.hot_rows])))
    _lib.check(self._lib.hbk_sharded_lookup_fwd(
      self._plan(), *bound.args, _lib.current_stream(self.device)))
    return bound.outs

  def prefetch_on_current_stream(self, bound):
    """``prefetch`` without a stream of the plan's own (``hbk_sharded_prefetch_on``): the partition +
    size exchange of ``bound`` are enqueued on the CURRENT stream; the caller puts this behind the
    end of the step before the last one begun on this object (PipelinedLookup: right behind a
    step's ``launch_begin``)."""
    a = bound.args
    _lib.check(self._lib.hbk_sharded_prefetch_on(self._plan(), a[0], a[1],
                                                 _lib.current_stream(self.device)))
    self._keep_prefetch = bound.keep

  def launch_begin(self, bound):
    """First half of a bound step (``hbk_sharded_lookup_fwd_begin``): partition (or its prefetched
    result), the one host wait, id exchange, owner-side gather.  ``launch_end`` finishes it."""
    self._keep = bound.keep
    self._last_shapes = bound.shapes
    a = bound.args
    _lib.check(self._lib.hbk_sharded_lookup_fwd_begin(
      self._plan(), a[0], a[1], a[2], a[3], _lib.current_stream(self.device)))

  def launch_end(self, bound):
    """Second half (``hbk_sharded_lookup_fwd_end``): rows exchange, stitch + combiner; returns the
    step's outputs."""
    a = bound.args
    _lib.check(self._lib.hbk_sharded_lookup_fwd_end(
      self._plan(), a[4], a[5], _lib.current_stream(self.device)))
    return bound.outs

  def prefetch(self, bound, ids_ready=None):
    """Pipelining hint: run bucketize + partition + size exchange of a FUTURE step on the plan's
    own stream now, overlapping the exchanges of the step that was just launched; the next
    forward over the same id tensors picks the result up.  ``bound``: a ``bind()`` result, or --
    the functional form, ids new every step -- the list of id tensors the next ``__call__`` will
    be handed (the step's one wait for the device, the sizes, then finds them there: W = 1
    180 -> ~135 us per step, tools/sweep.py case e).  ``ids_ready``: a ``torch.cuda.Event``
    recorded after the ids were produced (None: they are complete).  All ranks must prefetch the
    same steps."""
    ev = C.c_void_p(ids_ready.cuda_event) if ids_ready is not None else None
    if isinstance(bound, _BoundStep):
      id_ptrs, n_ids, keep = bound.args[0], bound.args[1], bound.keep
    else:
      from hybridbackend_amd import _marshal
      ids = list(bound)
      n = len(self.shards)
      if len(ids) != n:
        raise ValueError(f'prefetch: {len(ids)} id tensors for {n} columns')
      seen = _marshal.vector_pass(ids, (torch.int64,)) if n else ([], [], None)
      if seen is None:
        raise ValueError('prefetch: ids must be contiguous int64 device vectors')
      id_ptrs = (C.c_void_p * n)(*seen[0])
      n_ids = (C.c_int64 * n)(*seen[1])
      keep = (ids,)
    _lib.check(self._lib.hbk_sharded_prefetch(self._plan(), id_ptrs, n_ids, ev))
    self._keep_prefetch = keep

  def __call__(self, ids, row_splits=None, outs=None):
    """One forward step through the communicator.  ``ids[c]``: int64 device vector;
    ``row_splits[c]``: int32 device vector or None.  Returns the per-column outputs.
    Handed the SAME tensors as the step before (resident buffers refilled in place, caller-owned
    ``outs``) the marshalled arguments of that step are reused: validating and marshalling 26
    columns costs ~130 us of Python, the reuse check ~15."""
    cached = getattr(self, '_call_cache', None)
    # (a step with OTHER tensors is told apart by its first id tensor: the full comparison -- 3 N
    # tensors against the addresses / counts the bound step was marshalled with -- is only paid
    # when the step may really be the one before)
    if cached is not None and outs is not None and len(ids) and ids[0] is cached[0][0][0]:
      (p_ids, p_splits, p_outs), bound = cached
      a = bound.args
      sp = row_splits if row_splits is not None else [None] * len(ids)
      if len(ids) == len(p_ids) and all(
          ids[c] is p_ids[c] and sp[c] is p_splits[c] and outs[c] is p_outs[c] and
          ids[c].data_ptr() == a[0][c] and ids[c].numel() == a[1][c] and
          (sp[c] is None or sp[c].data_ptr() == a[2][c]) and outs[c].data_ptr() == a[4][c]
          for c in range(len(ids))):
        return self.launch(bound)
    bound = self.bind(ids, row_splits, outs)
    if outs is not None and len(ids):
      self._call_cache = (bound.keep, bound)
    return self.launch(bound)

  # ---- backward -----------------------------------------------------------------
  # phase B1: d(stitch + combiner): per-id gradient rows in the order of the partitioned ids
  def stitch_bwd(self, st, grads):
    n = len(self.shards)
    from hybridbackend_amd.embedding.lookup import _combiner_code
    combs = self.combiners
    if isinstance(combs, (str, int)) or combs is None:
      combs = [combs] * n
    cols = (_lib.StitchGradColumn * n)()
    outs = []
    for c in range(n):
      g = grads[c]
      _lib.require_device_tensor(g, 'grads')
      n_ids = st.shard_index[c].numel()
      sp = st.row_splits[c]
      n_seg = n_ids if sp is None else sp.numel() - 1
      if g.dtype != torch.float32 or tuple(g.shape) != (n_seg, self.dims[c]):
        raise _lib.InvalidArgumentError(
          _lib.INVALID_ARGUMENT, f'grad {c} must be fp32 [{n_seg}, {self.dims[c]}]')
      o = torch.empty((n_ids, self.dims[c]), dtype=torch.float32, device=self.device)
      col = cols[c]
      col.dim = self.dims[c]
      col.combiner = _combiner_code(combs[c])
      col.n_ids = n_ids
      col.index = st.shard_index[c].data_ptr()
      col.row_splits = sp.data_ptr() if sp is not None else None
      col.n_segments = n_seg
      col.grad_out = g.data_ptr()
      col.grad_rows = o.data_ptr()
      outs.append(o)
    _lib.check(self._lib.hbk_group_stitch_bwd(n, cols, _lib.current_stream(self.device)))
    return outs

  # phase B3: owner side: duplicate-row reduction (+ optional fused SGD on the shard)
  def owner_bwd(self, st, recv_grads, apply_lr=0.0):
    return self._owner_grad(st.recv_ids, recv_grads, None, apply_lr=apply_lr)

  def backward(self, grads, apply_lr=0.0, outs=None, optimizer='sgd', emit=True):
    """Backward of the LAST forward step (hbk_sharded_lookup_bwd).  grads[c]: gradient of
    column c's output [segments, dim].  Returns per column the IndexedSlices of the LOCAL
    shard ``(unique_rows, grad_rows, n_unique)``; with ``apply_lr`` the SGD update is applied
    to the shard in the same pass (sharded variables are not aggregated across ranks,
    training/gradient.py:193-217).  The exchange reuses the forward's sizes reversed
    (collective.py:334-347): no new size exchange, no host sync.  ``emit=False`` (with
    ``apply_lr``): step only, no IndexedSlices are written (only the ``n_unique`` counts)."""
    if not emit and apply_lr == 0.0:
      raise _lib.InvalidArgumentError(_lib.INVALID_ARGUMENT, 'emit=False needs apply_lr != 0')
    n = len(self.shards)
    plan = self._plan()
    if optimizer not in ('sgd', 'adagrad'):
      raise _lib.InvalidArgumentError(_lib.INVALID_ARGUMENT, "optimizer must be 'sgd' or 'adagrad'")
    res = []
    shapes = getattr(self, '_last_shapes', None)
    auto = bool(self._auto_hot) and outs is None
    # the counts of THIS call: one fresh tensor per call with emitted slices (IndexedSlices kept from
    # step k must not have their counts overwritten by step k + 1: ADVICE r04); step-only calls
    # write into one buffer of the driver (nothing is handed out that could be kept)
    nu_call = None
    if outs is None and emit:
      nu_call = torch.zeros(n, dtype=torch.int32, device=self.device)
    owned = []
    for c in range(n):
      _lib.require_device_tensor(grads[c], 'grads', row_strided=True)
      if grads[c].dtype != torch.float32 or (
          shapes is not None and tuple(grads[c].shape) != shapes[c]):
        raise _lib.InvalidArgumentError(
          _lib.INVALID_ARGUMENT,
          f'grad {c} must be fp32 {shapes[c] if shapes else "[segments, dim]"} '
          '(the shape of the last forward\'s output)')
      k = int(self._lib.hbk_sharded_owned_ids(plan, c))
      if k < 0:
        raise _lib.HbkError(_lib.INTERNAL, 'backward() needs a forward step first')
      owned.append(k)
      if outs is not None:
        # caller-owned (unique_rows, grad_rows, n_unique) with capacity >= owned ids
        if outs[c][0].numel() < k or outs[c][1].numel() < k * self.dims[c]:
          raise _lib.InvalidArgumentError(
            _lib.INVALID_ARGUMENT, f'backward outs[{c}] too small for {k} owned ids')
        res.append(tuple(outs[c]))
        continue
      if not emit:
        # the counts live in one buffer of the driver, written by every call
        if getattr(self, '_nu_step', None) is None:
          self._nu_step = torch.zeros(n, dtype=torch.int32, device=self.device)
        res.append((None, None, self._nu_step[c:c + 1]))
        continue
      res.append((torch.empty(k, dtype=torch.int64, device=self.device),
                  torch.empty((k, self.dims[c]), dtype=torch.float32, device=self.device),
                  nu_call[c:c + 1]))
    self._keep_bwd = (grads, res)
    strides = (C.c_int32 * n)(*[0 if g.is_contiguous() else int(g.stride(0)) for g in grads])
    if optimizer == 'adagrad' and apply_lr != 0.0 and self.accums is None:
      raise _lib.InvalidArgumentError(
        _lib.INVALID_ARGUMENT, "optimizer='adagrad' needs ShardedGroupLookup(..., accums=...)")
    _lib.check(self._lib.hbk_sharded_lookup_bwd_apply(
      plan, _lib.ptr_array([g.data_ptr() for g in grads]), strides,
      _lib.APPLY_ADAGRAD if optimizer == 'adagrad' else _lib.APPLY_SGD, C.c_float(apply_lr),
      _lib.ptr_array([r[0].data_ptr() for r in res]) if emit else None,
      _lib.ptr_array([r[1].data_ptr() for r in res]) if emit else None,
      _lib.ptr_array([r[2].data_ptr() for r in res]), _lib.current_stream(self.device)))
    if auto:
      st = self._auto_state
      if st is None:
        st = self._auto_state = [torch.empty(n, dtype=torch.int32).pin_memory(),
                                 torch.cuda.Event(), None, False]
      st[0].copy_(nu_call if nu_call is not None else self._nu_step, non_blocking=True)
      st[1].record()
      st[2], st[3] = owned, True
    return res


class PipelinedLookup:
  """Exchanges overlapped with the local gather ACROSS STEPS (round 5; north_star's "overlapped with
  local gather on a second HIP stream", the reference's mechanism is hbtf/common/stream.cc:83-142).

  Two (or more) :class:`ShardedGroupLookup` plans over the SAME tables and the SAME communicator,
  each on its own compute stream.  ``step(bound_next)`` begins the next step on the idle plan
  (partition result, id exchange, owner gather: ``launch_begin``) and only then ends the step begun
  before (rows exchange, stitch: ``launch_end``), so on the communicator the ids of step i + 1 travel
  AHEAD of the rows of step i: plan B gathers while plan A's rows are on the wire, plan A stitches
  while plan B's are.  A step costs ~max(wire, kernels) instead of their sum
  (profiles/r05_overlap_model.txt).  Forward-only use (inference, or a lookup that runs ahead of the
  trainer): nothing may update the tables between a step's begin and its end.  Every rank makes the
  same calls in the same order.

  ``plans``: the ShardedGroupLookup objects.  Their exchanges must run on the communicator's stream
  (option ``sharded_inline`` = 0, read when a plan is created; the library's default is 1): the
  constructor (re)creates every plan under that option and restores the caller's value, so a
  pipeline built with defaults overlaps instead of silently serialising.
  ``bind(k, ids, row_splits, outs)`` marshals a step for plan ``k``; ``step(bound)`` takes the bound
  steps of plans 0, 1, 0, 1 ...

  Stream order: every plan runs on a private stream.  ``step`` makes that stream wait
  for the caller's current stream first -- the ids, row splits and outputs handed to ``bind`` may
  have been produced there -- and marks the step's tensors as used on the private stream
  (``record_stream``), so the caching allocator cannot hand them out again before the step is done.
  """

  def __init__(self, plans, stream_exchanges=True):
    """stream_exchanges=False keeps the plans as they were created (e.g. inline exchanges: the
    pipeline is then correct but serialises)."""
    if len(plans) < 2:
      raise ValueError('PipelinedLookup needs at least two plans')
    self.plans = list(plans)
    if stream_exchanges:
      with _PLAN_LOCK:
        old = _lib.set_option('sharded_inline', 0)
        try:
          for p in self.plans:
            if getattr(p, '_p2p_keep', None) is None:
              p.close()      # (a plan with registered outputs keeps them: its step has no rows exchange)
            p._plan()        # pylint: disable=protected-access
        finally:
          _lib.set_option('sharded_inline', old)
    self.streams = [torch.cuda.Stream(device=p.device) for p in self.plans]
    self._open = None      # (plan index, bound step) begun and not ended
    self._next = 0
    self._done = [torch.cuda.Event() for _ in self.plans]

  def bind(self, k, ids, row_splits=None, outs=None):
    return self.plans[k].bind(ids, row_splits, outs)

  def next_plan(self):
    """Index of the plan the next ``step`` uses (its bound steps must come from that plan)."""
    return self._next

  def step(self, bound, prefetch=None):
    """Begin ``bound`` on the next plan, end the step begun by the previous call.  Returns the
    outputs of THAT step (None the first time); the current stream waits for them.  ``prefetch``:
    the bound step this plan will be handed next (its partition runs ahead)."""
    k = self._next
    self._next = (k + 1) % len(self.plans)
    caller = torch.cuda.current_stream(self.plans[k].device)
    self.streams[k].wait_stream(caller)       # the step's inputs may still be in flight there
    for b in (bound, prefetch):
      if b is not None:
        self._used_on(b, self.streams[k])
    with torch.cuda.stream(self.streams[k]):
      self.plans[k].launch_begin(bound)
      if prefetch is not None:
        # right behind the begin (the plan's partition state is double buffered): the partition of
        # this plan's NEXT step has len(plans) steps to finish before its begin waits for the sizes
        self.plans[k].prefetch_on_current_stream(prefetch)
    finished = self._finish()
    self._open = (k, bound)
    return finished

  @staticmethod
  def _used_on(bound, stream):
    for group in bound.keep:
      for t in group:
        if t is not None and t.is_cuda:
          t.record_stream(stream)

  def _finish(self):
    if self._open is None:
      return None
    k, bound = self._open
    self._open = None
    with torch.cuda.stream(self.streams[k]):
      outs = self.plans[k].launch_end(bound)
      self._done[k].record()
    torch.cuda.current_stream().wait_event(self._done[k])
    return outs

  def flush(self):
    """End the step that is still open; returns its outputs."""
    return self._finish()

  def close(self):
    self.flush()
    for p in self.plans:
      p.close()
