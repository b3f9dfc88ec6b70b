"""Shard sizing of embedding tables -- host mirror of
``hybridbackend/tensorflow/embedding/variables.py:77-146`` (``build_sharded_weights``).

A ``[bucket_size, dim]`` table created under ``sharding=True`` becomes, on rank ``shard`` of
``num_shards``, the local shard ``[rows_local, dim]`` that holds the rows
``shard, shard + num_shards, shard + 2 * num_shards, ...`` of the logical table (owner of an
id = ``id mod W``, local row = ``id // W``: sharding.py:182,189).  Small tables stay
replicated (``bucket_size <= num_shards`` or ``bucket_size <= batch_size``, variables.py:93-104).
"""


def sharded_bucket_size(bucket_size, num_shards, shard, batch_size=0):
  """Returns ``(is_sharded, rows_local, save_slice_offset)``.

  ``rows_local = bucket_size // num_shards (+1 if shard < bucket_size % num_shards)``
  (variables.py:107-111); ``save_slice_offset`` is the contiguous offset the reference
  records in ``SaveSliceInfo`` (variables.py:118-123; note: contiguous although ownership is
  strided -- a checkpoint written this way is only self-consistent for the same W).
  """
  bucket_size, num_shards, shard = int(bucket_size), int(num_shards), int(shard)
  if num_shards < 1 or not 0 <= shard < num_shards:
    raise ValueError(f'shard {shard} out of range for {num_shards} shards')
  if bucket_size <= num_shards or bucket_size <= int(batch_size or 0):
    return False, bucket_size, 0
  rows = bucket_size // num_shards
  if shard < bucket_size % num_shards:
    rows += 1
  offset = (bucket_size // num_shards) * shard
  remained = bucket_size % num_shards
  offset += shard if shard < remained else remained
  return True, rows, offset


def shard_of_table(table, num_shards, shard):
  """The local shard (a strided view, rows ``shard::num_shards``) of a full table tensor;
  call ``.contiguous()`` to materialise it.  ``len(view) == sharded_bucket_size(...)[1]``."""
  return table[shard::num_shards]


def allocate_tables(shapes, device='cuda', dtype=None):
  """N embedding tables carved from ONE slab, each at a 2 MB-aligned offset (``hbk_tables_layout``):
  the allocation policy that was fastest in every run of tools/placement_probe
  (profiles/r05_placement.txt: -2 % on config 4's forward against one allocation per table, 2.1 x
  fewer address-translation misses).  ``shapes``: ``(rows, dim)`` per table; returns the list of
  fp32 ``[rows, dim]`` tensors (views of the slab, uninitialised)."""
  import ctypes as C
  import torch
  from hybridbackend_amd import _lib
  dtype = dtype or torch.float32
  item = torch.empty(0, dtype=dtype).element_size()
  n = len(shapes)
  sizes = (C.c_size_t * max(n, 1))(*[int(r) * int(d) * item for r, d in shapes])
  offs = (C.c_size_t * max(n, 1))()
  total = _lib.lib().hbk_tables_layout(n, sizes, offs)
  # (torch's allocator returns 2 MB-aligned blocks for large requests; the slack covers it if not)
  slab = torch.empty(int(total) + (2 << 20), dtype=torch.uint8, device=device)
  base = (-slab.data_ptr()) % (2 << 20)
  out = []
  for (r, d), o in zip(shapes, offs):
    nbytes = int(r) * int(d) * item
    out.append(slab[base + o:base + o + nbytes].view(dtype).view(int(r), int(d)))
  return out
