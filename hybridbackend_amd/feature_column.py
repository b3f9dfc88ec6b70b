"""Embedding feature columns -> one dense block: host mirror of what the reference's users build
with ``tf.feature_column.embedding_column`` + ``tf.keras.layers.DenseFeatures`` /
``hb.keras.layers.dense_features`` (hybridbackend/tensorflow/keras/layers/__init__.py:29-46,
docs/tutorial/ranking/taobao/train_keras.py:60-75) under ``hb.scope(sharding=True)``
(hybridbackend/tensorflow/embedding/variables.py:77-146, sharding.py:171-205).

All columns are looked up by ONE fused launch (``hbk_group_lookup_fwd``) or one sharded step
(``hbk_sharded_lookup_fwd``), and every column writes its block of the concatenated
``[batch, sum of dims]`` tensor in place (``out_stride``): there is no per-column output and no
concat pass.  The backward reads the blocks of the incoming gradient in place the same way.
"""
import torch

from hybridbackend_amd import _lib
from hybridbackend_amd.embedding.lookup import GroupLookup
from hybridbackend_amd.embedding.lookup import GroupLookupGrad
from hybridbackend_amd.embedding.sharded import ShardedGroupLookup
from hybridbackend_amd.embedding.variables import sharded_bucket_size


class EmbeddingColumn:
  """``embedding_column(categorical_column_with_identity/hash_bucket(key, num_buckets),
  dimension, combiner)``: ids are bucketized with floor-mod ``num_buckets``
  (docs/tutorial/ranking/data.py:179,186).  ``hot_rows``: the ids are skewed (Zipf heads) --
  a wide column (dimension >= 64, one id per sample) then fetches the rows repeated inside a
  256-sample tile once and serves the repeats from LDS (GroupLookup(hot_rows=)).  The default
  ``'auto'`` lets the layer decide per column from what the last backward saw (distinct rows <
  half the ids: on); True / False pin it.  ``dedup``: a sharded column sends every distinct id of
  a batch once (ShardedGroupLookup(dedup=); the tutorials' tf.unique in front of the lookup)."""

  def __init__(self, key, num_buckets, dimension, combiner='mean', hot_rows='auto', dedup=False):
    if num_buckets < 1 or dimension < 1:
      raise _lib.InvalidArgumentError(
        _lib.INVALID_ARGUMENT, 'num_buckets and dimension must be >= 1')
    self.key, self.num_buckets, self.dimension = key, int(num_buckets), int(dimension)
    self.combiner = combiner
    self.hot_rows = 'auto' if hot_rows == 'auto' else bool(hot_rows)
    self.dedup = bool(dedup)


class DenseFeatures:
  """N embedding columns -> ``[batch, sum of dims]``.

  Args:
    columns: list of :class:`EmbeddingColumn`.
    device: the GPU.
    coll: a ``hybridbackend_amd.distribute.Collective`` for sharded tables, or None.
    batch_size: local batch size used by the replicate-or-shard rule
      (``bucket_size <= num_shards or bucket_size <= batch_size`` keeps a table replicated,
      variables.py:93-104).
    init: ``init(column, rows, device) -> fp32 [rows, dim]`` for this rank's rows (rows
      ``rank, rank + W, ..`` of the logical table when sharded); default uniform(-1e-3, 1e-3)
      (docs/tutorial/ranking/criteo/train.py:84,91).
  """

  def __init__(self, columns, device, coll=None, batch_size=0, init=None,
               initial_accumulator_value=None):
    self.columns = list(columns)
    self.device = torch.device(device)
    self.coll = coll
    world = coll.world_size if coll is not None else 1
    rank = coll.rank if coll is not None else 0
    if init is None:
      def init(col, rows, dev):
        return torch.empty(rows, col.dimension, device=dev).uniform_(-1e-3, 1e-3)
    self.sharded, self.weights = [], []
    for col in self.columns:
      is_sharded, rows, _ = sharded_bucket_size(col.num_buckets, world, rank, batch_size)
      is_sharded = is_sharded and world > 1
      self.sharded.append(is_sharded)
      self.weights.append(init(col, rows if is_sharded else col.num_buckets, self.device))
    # Adagrad accumulators (tf.train.AdagradOptimizer: initial_accumulator_value = 0.1)
    self.accums = None
    if initial_accumulator_value is not None:
      self.accums = [torch.full_like(w, float(initial_accumulator_value)) for w in self.weights]
    self.offsets, off = [], 0
    for col in self.columns:
      self.offsets.append(off)
      off += col.dimension
    self.width = off
    self._rep = [c for c in range(len(self.columns)) if not self.sharded[c]]
    self._shd = [c for c in range(len(self.columns)) if self.sharded[c]]
    pick = lambda idx, xs: [xs[c] for c in idx]   # noqa: E731
    self._lookup = self._grad = self._sharded = None
    if self._rep:
      self._lookup = GroupLookup(pick(self._rep, self.weights),
                                 [self.columns[c].num_buckets for c in self._rep],
                                 [self.columns[c].combiner for c in self._rep],
                                 hot_rows=[self.columns[c].hot_rows for c in self._rep])
      self._grad = GroupLookupGrad(
        self._lookup, pick(self._rep, self.accums) if self.accums is not None else None)
    if self._shd:
      self._sharded = ShardedGroupLookup(pick(self._shd, self.weights), coll,
                                         buckets=[self.columns[c].num_buckets for c in self._shd],
                                         combiners=[self.columns[c].combiner for c in self._shd],
                                         hot_rows=[self.columns[c].hot_rows for c in self._shd],
                                         dedup=[self.columns[c].dedup for c in self._shd],
                                         accums=(pick(self._shd, self.accums)
                                                 if self.accums is not None else None))

  def _split(self, features):
    ids, splits, batch = [], [], None
    for col in self.columns:
      f = features[col.key]
      i, s = f if isinstance(f, (tuple, list)) else (f, None)
      n = i.numel() if s is None else s.numel() - 1
      if batch is not None and n != batch:
        raise _lib.InvalidArgumentError(
          _lib.INVALID_ARGUMENT, f'feature {col.key}: {n} samples, expected {batch}')
      batch = n
      ids.append(i)
      splits.append(s)
    return ids, splits, batch

  def __call__(self, features, cols_to_output_tensors=None):
    """features[key] = int64 ids ``[batch]`` (one id per sample) or ``(values, row_splits)``
    (the values + row_splits layout of hybridbackend/tensorflow/data/dataframe.py:366-376).
    Returns the dense block; ``cols_to_output_tensors`` (a dict) receives each column's view."""
    ids, splits, batch = self._split(features)
    # rows start on 16-byte boundaries (row stride padded to 4 floats) so that columns whose
    # offset is a multiple of 4 floats keep 16-byte accesses
    pitch = (self.width + 3) // 4 * 4
    out = torch.empty((batch or 0, pitch), dtype=torch.float32, device=self.device)[:, :self.width]
    pick = lambda idx, xs: [xs[c] for c in idx]   # noqa: E731
    views = None

    def col_views():
      return [out[:, self.offsets[c]:self.offsets[c] + self.columns[c].dimension]
              for c in range(len(self.columns))]
    if self._rep:
      # the blocks' addresses are arithmetic: no per-column views unless somebody asks for them
      # (26 views + their validation were ~100 us of Python per step)
      if batch and self._lookup.bind_block(pick(self._rep, ids), pick(self._rep, splits), out,
                                           pick(self._rep, self.offsets)):
        self._lookup.launch()
      else:
        views = col_views()
        self._lookup(pick(self._rep, ids), pick(self._rep, splits), pick(self._rep, views))
    if self._shd:
      views = views or col_views()
      self._sharded(pick(self._shd, ids), pick(self._shd, splits), pick(self._shd, views))
    self._last = (ids, splits)
    if cols_to_output_tensors is not None:
      views = views or col_views()
      for c, col in enumerate(self.columns):
        cols_to_output_tensors[col] = views[c]
    return out

  def prefetch(self, features, ids_ready=None):
    """The NEXT step's features, as soon as the loader has them on the device: the sharded
    columns' bucketize + partition + size exchange run on the plan's stream beside the step in
    flight (``ShardedGroupLookup.prefetch``); the forward over the same tensors picks them up.
    A no-op for a layer whose columns are all replicated.  All ranks prefetch the same steps."""
    if self._shd:
      ids, _, _ = self._split(features)
      self._sharded.prefetch([ids[c] for c in self._shd], ids_ready)

  def backward(self, grad, apply_lr=0.0, optimizer='sgd', emit=True):
    """grad: ``[batch, sum of dims]`` gradient of the last forward's output.  Returns per column
    the ``IndexedSlices`` ``(unique_rows, grad_rows, n_unique)`` of this rank's rows (local row
    numbers for sharded tables); with ``apply_lr`` the sparse optimizer step (``'sgd'``, or
    ``'adagrad'`` when the layer was built with ``initial_accumulator_value``) is applied in the
    same pass -- for the SHARDED tables.  Small tables are replicated on every rank: at W > 1
    their gradients must first be aggregated across ranks (``hb.distribute.aggregate_gradients``,
    hybridbackend/tensorflow/training/gradient.py:119-177) or the replicas diverge, so for them
    this method never applies the step at W > 1: it returns their IndexedSlices and the caller
    applies the aggregated gradient.  ``emit=False`` (with ``apply_lr``): the stepped tables
    write no IndexedSlices (step only; their entries are ``(None, None, n_unique)``)."""
    ids, splits = self._last
    if grad.dim() != 2 or grad.shape[1] != self.width or grad.dtype != torch.float32:
      raise _lib.InvalidArgumentError(
        _lib.INVALID_ARGUMENT, f'grad must be fp32 [batch, {self.width}]')
    if grad.stride(1) != 1 or grad.stride(0) % 4 != 0:
      # same pitch as the forward's block: rows on 16-byte boundaries (one extra copy)
      pitch = (self.width + 3) // 4 * 4
      padded = torch.empty((grad.shape[0], pitch), dtype=torch.float32,
                           device=grad.device)[:, :self.width]
      padded.copy_(grad)
      grad = padded
    pick = lambda idx, xs: [xs[c] for c in idx]   # noqa: E731
    res = [None] * len(self.columns)
    if self._rep:
      rep_lr = apply_lr if (self.coll.world_size if self.coll is not None else 1) <= 1 else 0.0
      # the gradient's column blocks are addressed by arithmetic (no per-column views)
      r = self._grad(pick(self._rep, ids), None, pick(self._rep, splits),
                     apply_lr=rep_lr, optimizer=optimizer, emit=emit or rep_lr == 0.0,
                     grad_block=(grad, pick(self._rep, self.offsets)))
      for k, c in enumerate(self._rep):
        res[c] = r[k]
    if self._shd:
      views = [grad[:, self.offsets[c]:self.offsets[c] + self.columns[c].dimension]
               for c in self._shd]
      r = self._sharded.backward(views, apply_lr=apply_lr, optimizer=optimizer,
                                 emit=emit)
      for k, c in enumerate(self._shd):
        res[c] = r[k]
    return res

  # ---- checkpoints (hybridbackend/tensorflow/training/saver.py:97-185) ----------------------------
  def variables(self):
    """``{name: tensor | ShardedSlice}`` of this rank: the embedding weights (TF naming:
    ``<key>_embedding/embedding_weights``; a sharded table is the slice ``part_<rank>`` of it,
    variables.py:112-141) and, when the layer keeps them, the Adagrad slots (``.../Adagrad``)."""
    from hybridbackend_amd.training.saver import ShardedSlice
    world = self.coll.world_size if self.coll is not None else 1
    rank = self.coll.rank if self.coll is not None else 0
    out = {}
    for c, col in enumerate(self.columns):
      name = f'{col.key}_embedding/embedding_weights'
      for suffix, tensors in (('', self.weights), ('/Adagrad', self.accums)):
        if tensors is None:
          continue
        t = tensors[c]
        out[name + suffix] = (ShardedSlice(t, col.num_buckets, world, rank)
                              if self.sharded[c] else t)
    return out

  def _saver(self, barrier):
    from hybridbackend_amd.training.saver import Saver
    world = self.coll.world_size if self.coll is not None else 1
    rank = self.coll.rank if self.coll is not None else 0
    if barrier is None and world > 1:
      import torch.distributed as dist   # pylint: disable=import-outside-toplevel
      if not dist.is_initialized():
        raise _lib.HbkError(_lib.INTERNAL, 'save/restore at W > 1 needs a barrier '
                                           '(torch.distributed is not initialized)')
      barrier = dist.barrier
    return Saver(rank, world, barrier)

  def save(self, prefix, barrier=None):
    """Every rank writes its shards, rank 0 also the replicated tables and the index; all ranks
    call this together.  The device work of the current stream is waited for first."""
    if self.device.type == 'cuda':
      torch.cuda.current_stream(self.device).synchronize()
    return self._saver(barrier).save(prefix, self.variables())

  def restore(self, prefix, barrier=None, layout='logical'):
    """Loads this rank's rows from a checkpoint written at ANY world size (``layout='reference'``:
    the reference's contiguous slicing instead, see training/saver.py)."""
    self._saver(barrier).restore(prefix, self.variables(), layout=layout)

  def restore_reference(self, prefix, names=None, barrier=None, layout='logical'):
    """Loads this rank's rows from a checkpoint the REFERENCE saved (TensorFlow tensor bundle,
    training/tf_bundle.py), written at any world size.  ``names``: ``{name here: tensor name in
    the checkpoint}`` for variables the model named differently (default: the TF names of
    ``variables()``).  layout: see ``Saver.restore_reference``."""
    self._saver(barrier).restore_reference(prefix, self.variables(), names=names, layout=layout)

  def close(self):
    if self._sharded is not None:
      self._sharded.close()


def dense_features(features, layer):
  """``hb.keras.layers.dense_features``: the per-column tensors, in column order."""
  m = {}
  layer(features, cols_to_output_tensors=m)
  return [m[c] for c in layer.columns]
