"""``hb.data.ParquetDataset``: batches of tabular features straight into the layout the fused
lookup consumes -- host mirror of hybridbackend/tensorflow/data/tabular/dataset_v1.py:46-97,128-175
(reader: tabular/parquet.cc, table.cc over Arrow C++) and of the values + row_splits value
type ``DataFrame.Value`` (data/dataframe.py:283-377).

A scalar (int64 / int32) column becomes an id vector ``[batch]``; a ``list<int64>`` column becomes
``(values [n], row_splits int32 [batch + 1])`` -- no ``SparseTensor`` is ever materialised, the
pair goes into ``GroupLookup`` / ``DenseFeatures`` as it is.  Decoding is Arrow's own C++ reader
(pyarrow); batches are staged in pinned host buffers and copied to the GPU on a side stream by a
prefetch thread, so the copy of batch i + 1 overlaps the lookups of batch i
(the reference's prefetch + packed H2D transfer: data/prefetch, ops/transfer).
"""
import queue
import threading

import numpy as np
import torch


class ParquetDataset:
  """Iterable over batches ``{field: ids | (values, row_splits)}``.

  Args:
    filenames: one path or a list of paths.
    batch_size: samples per batch.
    fields: column names to read (default: all).
    partition_count / partition_index: this reader takes row groups
      ``partition_index, partition_index + partition_count, ..`` (dataset_v1.py:48-49; one
      partition per data-parallel rank).
    drop_remainder: only full batches.
    device: where the tensors go; ``None`` / ``'cpu'`` keeps them on the host.
    prefetch: batches decoded and copied ahead of the consumer.
  """

  def __init__(self, filenames, batch_size, fields=None, partition_count=1, partition_index=0,
               drop_remainder=False, device=None, prefetch=2):
    import pyarrow.parquet as pq  # pylint: disable=import-outside-toplevel
    self._pq = pq
    self.filenames = [filenames] if isinstance(filenames, str) else list(filenames)
    self.batch_size = int(batch_size)
    if self.batch_size < 1:
      raise ValueError('batch_size must be >= 1')
    if partition_count < 1 or not 0 <= partition_index < partition_count:
      raise ValueError(f'partition {partition_index} out of range for {partition_count}')
    self.fields = list(fields) if fields is not None else None
    self.partition_count, self.partition_index = int(partition_count), int(partition_index)
    self.drop_remainder = bool(drop_remainder)
    self.device = torch.device(device) if device is not None else torch.device('cpu')
    self.prefetch = max(int(prefetch), 1)

  # ---- decoding -------------------------------------------------------------------------------
  @staticmethod
  def _column(col, name):
    """One Arrow column of a record batch -> numpy ids or (values, row_splits)."""
    import pyarrow as pa  # pylint: disable=import-outside-toplevel
    if pa.types.is_list(col.type) or pa.types.is_large_list(col.type):
      if not pa.types.is_integer(col.type.value_type):
        raise TypeError(f'field {name}: list of {col.type.value_type} is not an id list')
      offsets = col.offsets.to_numpy(zero_copy_only=False).astype(np.int64)
      if col.null_count:
        # a null list is an empty list: its offset range is empty in a valid Arrow array
        valid = np.asarray(col.is_valid())
        lens = np.where(valid, np.diff(offsets), 0)
        starts = offsets[:-1]
        flat = col.values.to_numpy(zero_copy_only=False)
        values = np.concatenate([flat[s:s + n] for s, n in zip(starts, lens)] or
                                [flat[:0]]).astype(np.int64)
        splits = np.concatenate([[0], np.cumsum(lens)])
      else:
        flat = col.values.to_numpy(zero_copy_only=False)
        values = np.ascontiguousarray(flat[offsets[0]:offsets[-1]], np.int64)
        splits = offsets - offsets[0]
      if splits[-1] >= 2**31:
        raise ValueError(f'field {name}: more than 2^31 - 1 ids in one batch')
      return values, splits.astype(np.int32)
    if not pa.types.is_integer(col.type):
      raise TypeError(f'field {name}: {col.type} is not an id column')
    if col.null_count:
      raise ValueError(f'field {name}: null ids in a scalar column')
    return np.ascontiguousarray(col.to_numpy(zero_copy_only=False), np.int64)

  def _host_batches(self):
    """Decoded batches as numpy, across files and this partition's row groups, re-batched to
    exactly batch_size samples (a batch may span row groups and files)."""
    pending, have = [], 0

    def cut(chunks, n):
      # first n samples of the concatenation of chunks -> (head dict, remaining chunks)
      import pyarrow as pa  # pylint: disable=import-outside-toplevel
      table = pa.Table.from_batches(chunks).combine_chunks()
      head, rest = table.slice(0, n), table.slice(n)
      out = {name: self._column(head.column(name).chunk(0) if head.num_rows else
                                pa.array([], head.schema.field(name).type), name)
             for name in table.column_names}
      return out, (rest.to_batches() if rest.num_rows else [])

    for path in self.filenames:
      f = self._pq.ParquetFile(path)
      groups = list(range(self.partition_index, f.num_row_groups, self.partition_count))
      if not groups:
        continue
      for rb in f.iter_batches(batch_size=self.batch_size, row_groups=groups,
                               columns=self.fields):
        pending.append(rb)
        have += rb.num_rows
        while have >= self.batch_size:
          out, pending = cut(pending, self.batch_size)
          have -= self.batch_size
          yield out
    if have and not self.drop_remainder:
      out, _ = cut(pending, have)
      yield out

  # ---- host -> device, ahead of the consumer ------------------------------------------------------
  def _to_device(self, batch, stream):
    def put(a):
      if not a.flags.writeable:   # Arrow hands out read-only views
        a = a.copy()
      t = torch.from_numpy(a)
      if self.device.type == 'cpu':
        return t
      return t.pin_memory().to(self.device, non_blocking=True)
    if stream is None:
      return {k: (tuple(put(x) for x in v) if isinstance(v, tuple) else put(v))
              for k, v in batch.items()}, None
    with torch.cuda.stream(stream):
      out = {k: (tuple(put(x) for x in v) if isinstance(v, tuple) else put(v))
             for k, v in batch.items()}
      ready = torch.cuda.Event()
      ready.record(stream)
    return out, ready

  def __iter__(self):
    q = queue.Queue(maxsize=self.prefetch)
    on_gpu = self.device.type != 'cpu'
    stream = torch.cuda.Stream(self.device) if on_gpu else None
    stop = threading.Event()

    def worker():
      try:
        if on_gpu:
          torch.cuda.set_device(self.device)
        for b in self._host_batches():
          if stop.is_set():
            return
          q.put(self._to_device(b, stream))
        q.put(None)
      except BaseException as e:  # pylint: disable=broad-except
        q.put(e)

    t = threading.Thread(target=worker, daemon=True)
    t.start()
    try:
      while True:
        item = q.get()
        if item is None:
          return
        if isinstance(item, BaseException):
          raise item
        batch, ready = item
        if ready is not None:
          cur = torch.cuda.current_stream(self.device)
          cur.wait_event(ready)
          # the tensors were allocated on the copy stream: tell the caching allocator that the
          # consumer's stream uses them too, or their blocks could be handed to a later batch's
          # H2D copy while lookup kernels of this batch are still queued on the consumer's stream
          for v in batch.values():
            for t in (v if isinstance(v, tuple) else (v,)):
              if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(cur)
        yield batch
    finally:
      stop.set()
      while not q.empty():
        try:
          q.get_nowait()
        except queue.Empty:
          break
