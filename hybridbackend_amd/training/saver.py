"""Sharded checkpoints -- host mirror of ``hybridbackend/tensorflow/training/saver.py:97-185``
(``HybridBackendSaverBuilder._AddShardedSaveOpsForV2``) and of the slice bookkeeping of
``hybridbackend/tensorflow/embedding/variables.py:114-141`` (``SaveSliceInfo``).

What the reference does, and what is kept:

* every rank writes the variables it owns into a temporary part file
  (``<prefix>_temp_<id>/part-<rank>-of-<world>``): rank 0 everything it holds, the other ranks
  only their shards of sharded variables (saver.py:120-152);
* a barrier; rank 0 merges the parts into the checkpoint ``<prefix>`` (index + data files) and
  removes the temporary directory (``merge_v2_checkpoints(..., delete_old_dirs=True)``,
  saver.py:154-180); a second barrier releases the other ranks;
* a sharded variable is recorded as a slice of its full tensor: ``full_name``, ``full_shape =
  [bucket_size, dim]``, ``var_offset = [bucket_offset, 0]``, ``var_shape = [rows_local, dim]``
  with the CONTIGUOUS ``bucket_offset`` of variables.py:118-123.

The reference's quirk, made explicit.  Ownership of rows is STRIDED (owner = ``id mod W``, local
row = ``id // W``, sharding.py:182,189) while the recorded offsets are contiguous: the "full
tensor" a reference checkpoint describes is the concatenation of the shards, i.e. a permutation
of the logical table, and restoring it at a different world size hands every rank rows of other
ids.  Here every slice additionally records its ownership (``stride = W``, ``phase = rank``), so

* ``restore`` re-shards correctly at ANY world size (a rank gathers its rows ``k W' + r'`` from
  the slices that hold them), and
* ``layout='reference'`` reproduces the reference's view (contiguous slices of the concatenated
  shards) for tools that expect it; ``load_full(prefix, name)`` returns the logical table.

The files are this library's own (``<prefix>.index`` is JSON, ``<prefix>.data-g<generation>-<r>-of-<W>``
raw little-endian tensors; a save never overwrites the files the current index points to and
switches over with one atomic rename of the index), not TensorFlow's tensor-bundle format.  The
bridge to the reference's files is ``training/tf_bundle.py`` (that format read and written on the
host, pinned by its published vectors only -- TensorFlow is not available to this build):
``Saver.restore_reference`` loads a checkpoint the reference wrote, ``export_reference`` rewrites
one of ours as the bundle the reference would have saved.  The host logic is device agnostic (CPU tensors work, which
is how the tests without a GPU drive it); GPU tensors travel through pinned memory.
"""
import json
import os
import shutil

import numpy as np
import torch

_DTYPES = {'float32': np.float32, 'float64': np.float64, 'int32': np.int32, 'int64': np.int64,
           'float16': np.float16, 'uint8': np.uint8, 'int8': np.int8, 'int16': np.int16,
           # numpy has no bfloat16: the bits travel as int16 and the index keeps the real name
           'bfloat16': np.int16}
_TORCH_DTYPES = {'bfloat16': torch.bfloat16}


class ShardedSlice:
  """One rank's shard of a row-sharded variable: rows ``rank, rank + W, ..`` of the logical
  ``[bucket_size, dim]`` table (``SaveSliceInfo`` of variables.py:133-141 plus the ownership)."""

  def __init__(self, tensor, bucket_size, world_size, rank):
    self.tensor = tensor
    self.bucket_size, self.world_size, self.rank = int(bucket_size), int(world_size), int(rank)
    rows = self.bucket_size // self.world_size + (self.rank < self.bucket_size % self.world_size)
    if tensor.dim() != 2 or tensor.shape[0] != rows:
      raise ValueError(f'shard of rank {rank}/{world_size} of a {bucket_size}-row table must '
                       f'have {rows} rows, got {tuple(tensor.shape)}')

  @property
  def var_offset(self):
    """The contiguous offset the reference records (variables.py:118-123)."""
    q, rem = divmod(self.bucket_size, self.world_size)
    return q * self.rank + min(self.rank, rem)


def _to_numpy(t):
  """(array, dtype name): bfloat16 tensors are written as their 16-bit patterns."""
  t = t.detach()
  if t.is_cuda:
    host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    host.copy_(t, non_blocking=False)
    t = host
  t = t.cpu().contiguous()
  if t.dtype == torch.bfloat16:
    return np.ascontiguousarray(t.view(torch.int16).numpy()), 'bfloat16'
  arr = np.ascontiguousarray(t.numpy())
  if str(arr.dtype) not in _DTYPES:
    raise ValueError(f'unsupported dtype {arr.dtype} in a checkpoint')
  return arr, str(arr.dtype)


class Saver:
  """``Saver(rank, world_size, barrier)``: ``barrier()`` must return once every rank has called
  it (``torch.distributed.barrier``, a ``threading.Barrier.wait`` for in-process ranks; not
  needed for ``world_size == 1``)."""

  def __init__(self, rank=0, world_size=1, barrier=None):
    self.rank, self.world_size = int(rank), int(world_size)
    if self.world_size > 1 and barrier is None:
      raise ValueError('a multi-rank Saver needs a barrier')
    self._barrier = barrier or (lambda: None)

  # -- save ------------------------------------------------------------------------------------
  def save(self, prefix, variables):
    """variables: ``{name: tensor | ShardedSlice}``.  Replicated tensors are written by rank 0
    only (saver.py:135-141), shards by every rank.  Returns ``prefix`` (saver.py:176-185)."""
    tmp_dir = f'{prefix}_temp'
    part = os.path.join(tmp_dir, f'part-{self.rank:05d}-of-{self.world_size:05d}')
    if self.rank == 0:
      os.makedirs(tmp_dir, exist_ok=True)
    self._barrier()
    entries, off = [], 0
    with open(part + '.data', 'wb') as f:
      for name in sorted(variables):
        v = variables[name]
        sharded = isinstance(v, ShardedSlice)
        if not sharded and self.rank != 0:
          continue                      # only sharded saveables for non-chief workers
        arr, dtype_name = _to_numpy(v.tensor if sharded else v)
        e = {'name': name, 'dtype': dtype_name, 'shape': list(arr.shape), 'offset': off,
             'nbytes': int(arr.nbytes)}
        if sharded:
          e.update(full_shape=[v.bucket_size, int(arr.shape[1])], var_offset=[v.var_offset, 0],
                   stride=v.world_size, phase=v.rank)
        f.write(arr.tobytes())
        off += arr.nbytes
        entries.append(e)
    with open(part + '.json', 'w') as f:
      json.dump(entries, f)
    self._barrier()                      # local barrier: every part is on disk
    if self.rank == 0:
      self._merge(prefix, tmp_dir)
    self._barrier()                      # global barrier: the checkpoint is complete
    return prefix

  def _merge(self, prefix, tmp_dir):
    """Crash safety: the data files of a save carry a generation tag in their names, so they
    never overwrite the files an existing index points to; every data file and the directory are
    fsync'ed, THEN the index is written to a temporary file, fsync'ed and moved into place (one
    atomic rename switches the checkpoint over; the directory is fsync'ed again).  The generation
    that was just replaced is kept until the next save."""
    index = {'format': 'hbk-sharded-checkpoint-1', 'world_size': self.world_size, 'variables': {}}
    folder = os.path.dirname(prefix) or '.'
    base = os.path.basename(prefix)
    old_files = set()
    if os.path.exists(prefix + '.index'):
      try:
        for var in _read_index(prefix)['variables'].values():
          old_files.update(s['file'] for s in var['slices'])
      except (ValueError, KeyError, json.JSONDecodeError):
        pass
    gen = 0
    while any(f'{base}.data-g{gen}-' in f for f in old_files):
      gen += 1
    for r in range(self.world_size):
      part = os.path.join(tmp_dir, f'part-{r:05d}-of-{self.world_size:05d}')
      data = f'{base}.data-g{gen}-{r:05d}-of-{self.world_size:05d}'
      os.replace(part + '.data', os.path.join(folder, data))
      _fsync_path(os.path.join(folder, data))   # durable BEFORE the index that names it
      for e in json.load(open(part + '.json')):
        var = index['variables'].setdefault(
          e['name'], {'dtype': e['dtype'], 'slices': [],
                      'full_shape': e.get('full_shape', e['shape'])})
        var['slices'].append({
          'file': data, 'offset': e['offset'], 'nbytes': e['nbytes'], 'var_shape': e['shape'],
          'var_offset': e.get('var_offset', [0] * len(e['shape'])),
          'stride': e.get('stride', 1), 'phase': e.get('phase', 0)})
    with open(prefix + '.index.tmp', 'w') as f:
      json.dump(index, f, indent=1, sort_keys=True)
      f.flush()
      os.fsync(f.fileno())
    _fsync_path(folder)                      # the new data files' directory entries
    os.replace(prefix + '.index.tmp', prefix + '.index')
    _fsync_path(folder)                      # the switch-over itself
    # The generation just replaced STAYS until the next save (a reader that has the old index
    # open still finds its data files); what goes now are the generations before it.
    new_files = {sl['file'] for v in index['variables'].values() for sl in v['slices']}
    for name in os.listdir(folder):
      if name.startswith(base + '.data-g') and name not in old_files and name not in new_files:
        try:
          os.remove(os.path.join(folder, name))
        except OSError:
          pass
    shutil.rmtree(tmp_dir, ignore_errors=True)    # delete_old_dirs=True

  # -- restore ---------------------------------------------------------------------------------
  def restore(self, prefix, variables, layout='logical'):
    """Fills ``variables`` (same mapping as ``save``) in place.  A sharded variable may be
    restored at any world size: with ``layout='logical'`` rank ``r'`` of ``W'`` receives the
    logical rows ``r', r' + W', ..`` wherever they were saved; ``layout='reference'`` slices the
    concatenation of the saved shards contiguously, as a reference restore at a different world
    size would (same W: identical to 'logical').  Names missing from the checkpoint are left
    untouched (saver.py:208-215: their initializer's value stays)."""
    if layout not in ('logical', 'reference'):
      raise ValueError("layout must be 'logical' or 'reference'")
    index = _read_index(prefix)
    for name, v in variables.items():
      if name not in index['variables']:
        continue
      meta = index['variables'][name]
      if isinstance(v, ShardedSlice):
        if meta['full_shape'][0] != v.bucket_size:
          raise ValueError(f'{name}: checkpoint holds {meta["full_shape"][0]} rows, '
                           f'the variable {v.bucket_size}')
        if layout == 'logical':
          rows = np.arange(v.rank, v.bucket_size, v.world_size, dtype=np.int64)
          arr = _gather_logical_rows(prefix, meta, rows)
        else:
          arr = _concatenated(prefix, meta)[v.var_offset:v.var_offset + v.tensor.shape[0]]
        target = v.tensor
      else:
        arr = _load_slice(prefix, meta, meta['slices'][0])
        target = v
      if tuple(arr.shape) != tuple(target.shape):
        raise ValueError(f'{name}: checkpoint shape {tuple(arr.shape)} != {tuple(target.shape)}')
      host = torch.from_numpy(np.array(arr, copy=True))
      if meta['dtype'] in _TORCH_DTYPES:
        host = host.view(_TORCH_DTYPES[meta['dtype']])
      target.copy_(host.to(target.dtype))
    self._barrier()


  def restore_reference(self, prefix, variables, names=None, layout='logical', verify=True):
    """``restore`` from a checkpoint the REFERENCE wrote (TensorFlow's tensor bundle,
    training/tf_bundle.py; saver.py:187-246 is the reference's own restore).  ``names``: variable
    name here -> tensor name in the checkpoint (default: the same).  A ``ShardedSlice`` takes
    its rows ``rank, rank + W', ..`` of the table; ``layout`` says what the checkpoint's slices
    mean: 'logical' -- the reference row-sharded the table over W ranks by ``id mod W``, slice r
    holds ids ``r, r + W, ..`` (what the reference's embedding weights are, variables.py:
    114-141) -- or 'reference' -- the full tensor as TensorFlow assembles it IS the table (a
    variable that was never sharded by id, or one saved by a single rank).  Names missing from
    the checkpoint are left untouched."""
    from hybridbackend_amd.training import tf_bundle
    reader = tf_bundle.BundleReader(prefix, verify=verify)
    names = names or {}
    for name, v in variables.items():
      ck = names.get(name, name)
      if ck not in reader.entries:
        continue
      sharded = isinstance(v, ShardedSlice)
      table = (tf_bundle.read_reference_table(reader, ck, layout) if sharded
               else reader.read(ck))
      target = v.tensor if sharded else v
      if sharded:
        if table.shape[0] != v.bucket_size:
          raise ValueError(f'{name}: checkpoint holds {table.shape[0]} rows, the variable '
                           f'{v.bucket_size}')
        table = table[v.rank::v.world_size]
      if tuple(table.shape) != tuple(target.shape):
        raise ValueError(f'{name}: checkpoint shape {tuple(table.shape)} != {tuple(target.shape)}')
      host = torch.from_numpy(np.ascontiguousarray(table))
      if reader.is_bfloat16(ck):
        host = host.view(torch.int16).view(torch.bfloat16)
      target.copy_(host.to(target.dtype))
    self._barrier()


def export_reference(src_prefix, dst_prefix):
  """A checkpoint of this library rewritten as the tensor bundle the reference would have saved
  at the same world size: every shard becomes a slice of its full tensor at the reference's
  CONTIGUOUS offset (variables.py:118-123), replicated variables are saved whole.  One process,
  on the host (the reference's chief merges the parts the same way, saver.py:154-180)."""
  from hybridbackend_amd.training import tf_bundle
  index = _read_index(src_prefix)
  # one data file per rank that saved, every shard in its rank's file: what MergeV2Checkpoints
  # leaves of the ranks' parts
  w = tf_bundle.BundleWriter(dst_prefix, num_shards=index['world_size'])
  for name, meta in sorted(index['variables'].items()):
    bf16 = meta['dtype'] == 'bfloat16'
    if len(meta['slices']) == 1 and meta['slices'][0]['stride'] == 1:
      w.add(name, np.asarray(_load_slice(src_prefix, meta, meta['slices'][0])), bfloat16=bf16)
      continue
    for s in sorted(meta['slices'], key=lambda s: s['var_offset'][0]):
      w.add_slice(name, meta['full_shape'], s['var_offset'],
                  np.asarray(_load_slice(src_prefix, meta, s)), bfloat16=bf16, shard=s['phase'])
  return w.finish()


def _read_index(prefix):
  with open(prefix + '.index') as f:
    index = json.load(f)
  if index.get('format') != 'hbk-sharded-checkpoint-1':
    raise ValueError(f'{prefix}.index is not a checkpoint of this library')
  return index


def _load_slice(prefix, meta, s):
  path = os.path.join(os.path.dirname(prefix) or '.', s['file'])
  if s['nbytes'] == 0:      # a rank without rows (bucket_size < world_size): nothing to map
    return np.empty(tuple(s['var_shape']), _DTYPES[meta['dtype']])
  return np.memmap(path, dtype=_DTYPES[meta['dtype']], mode='r', offset=s['offset'],
                   shape=tuple(s['var_shape']))


def _gather_logical_rows(prefix, meta, rows):
  """rows of the LOGICAL table, from whichever slices own them (slice with stride W and phase r
  holds logical row k W + r at its row k)."""
  slices = meta['slices']
  out = np.empty((rows.size,) + tuple(meta['full_shape'][1:]), _DTYPES[meta['dtype']])
  done = np.zeros(rows.size, bool)
  for s in slices:
    sel = (rows % s['stride']) == s['phase']
    if sel.any():
      out[sel] = _load_slice(prefix, meta, s)[rows[sel] // s['stride']]
      done |= sel
  if not done.all():
    raise ValueError('the checkpoint does not hold every requested row')
  return out


def _concatenated(prefix, meta):
  """The reference's view of a sharded variable: its shards back to back in var_offset order."""
  parts = sorted(meta['slices'], key=lambda s: s['var_offset'][0])
  return np.concatenate([np.asarray(_load_slice(prefix, meta, s)) for s in parts], axis=0)


def _fsync_path(path):
  """fsync a file or a directory by path (directories: so that renames / new entries are durable)."""
  fd = os.open(path, os.O_RDONLY)
  try:
    os.fsync(fd)
  finally:
    os.close(fd)


def load_full(prefix, name, layout='logical'):
  """The whole variable as a numpy array: the logical table (rows de-interleaved) or, with
  ``layout='reference'``, the concatenation a reference checkpoint calls the full tensor.
  bfloat16 variables come back as their int16 BIT PATTERNS (numpy has no bfloat16):
  ``torch.from_numpy(a).view(torch.bfloat16)`` gives the values."""
  meta = _read_index(prefix)['variables'][name]
  if layout == 'reference' or all(s['stride'] == 1 for s in meta['slices']):
    return _concatenated(prefix, meta)
  return _gather_logical_rows(prefix, meta, np.arange(meta['full_shape'][0], dtype=np.int64))
