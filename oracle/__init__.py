"""CPU oracle for the sharded-embedding hot path -- TEST INFRASTRUCTURE ONLY.

numpy-facing wrappers over ``oracle/hbk_oracle.c`` (built to
``oracle/_build/liboracle.so`` by ``oracle/Makefile``) plus the in-process
multi-rank composition of the pipeline (R12).  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this package; nothing under ``hybridbackend_amd/`` does.

Each wrapper names the reference file:line its C body restates (paths relative
to /root/reference; hbtf/ = hybridbackend/tensorflow/).  Rows R1, R7-R10 live in
TensorFlow 1.15 (third party, absent): pinned to its documentation's worked examples,
to vectors derived from its documented rules and to second implementations, not to its
binary -- see the ``hbk_oracle.c`` header and DESIGN.md section 2.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, '_build', 'liboracle.so')
_REF_PATH = os.path.join(_HERE, '_ref', 'libref_murmur3.so')


def build(force=False):
  """Compile the C oracle (and oracle/_ref when /root/reference is mounted)."""
  if force or not os.path.exists(_LIB_PATH) or (
      os.path.getmtime(_LIB_PATH) <
      os.path.getmtime(os.path.join(_HERE, 'hbk_oracle.c'))):
    subprocess.check_call(['make', '-s', '-C', _HERE, '_build/liboracle.so'])
  if os.path.isdir('/root/reference') and (
      force or not os.path.exists(_REF_PATH)):
    subprocess.check_call(['make', '-s', '-C', _HERE, 'ref'])


_lib = None


def lib():
  global _lib
  if _lib is None:
    build()
    _lib = C.CDLL(_LIB_PATH)
    _lib.orc_unique_i64.restype = C.c_int64
    _lib.orc_murmur3_hash32_i64.restype = C.c_uint32
    _lib.orc_murmur3_hash32_i64.argtypes = [C.c_int64, C.c_uint32]
    _lib.orc_compute_active_ranks.restype = C.c_int32
    _lib.orc_compute_active_size.restype = C.c_int32
    _lib.orc_shard_rows.restype = C.c_int32
  return _lib


def ref_lib():
  """The reference's own murmur3 header compiled as-is, or None if not built."""
  if not os.path.exists(_REF_PATH):
    return None
  r = C.CDLL(_REF_PATH)
  r.ref_murmur3_hash32_i64.restype = C.c_uint32
  r.ref_murmur3_hash32_i64.argtypes = [C.c_longlong]
  return r


def _p(a):
  return a.ctypes.data_as(C.c_void_p)


_SUFFIX = {np.dtype(np.int32): 'i32', np.dtype(np.int64): 'i64',
           np.dtype(np.uint32): 'u32', np.dtype(np.uint64): 'u64'}

SUM, MEAN, SQRTN = 0, 1, 2
COMBINERS = {'sum': SUM, 'mean': MEAN, 'sqrtn': SQRTN}


# R1 -- docs/tutorial/ranking/data.py:179,186 (TF FloorMod)
def floormod(ids, m):
  ids = np.ascontiguousarray(ids)
  out = np.empty_like(ids)
  if ids.dtype == np.int64:
    lib().orc_floormod_i64(_p(ids), C.c_int64(ids.size), C.c_int64(m), _p(out))
  elif ids.dtype == np.int32:
    lib().orc_floormod_i32(_p(ids), C.c_int64(ids.size), C.c_int32(m), _p(out))
  else:
    raise TypeError(ids.dtype)
  return out


# R2 -- hbtf/distribute/partition/partition_by_modulo_functors.cc:39-70
def partition_by_modulo(ids, num_partitions):
  ids = np.ascontiguousarray(ids)
  out = np.empty_like(ids)
  sizes = np.zeros(num_partitions, np.int32)
  indices = np.empty(ids.size, np.int32)
  fn = getattr(lib(), 'orc_partition_by_modulo_' + _SUFFIX[ids.dtype])
  fn(C.c_int32(num_partitions), _p(ids), C.c_int32(ids.size), _p(out),
     _p(sizes), _p(indices))
  return out, sizes, indices


# R3 -- hbtf/distribute/partition/partition_by_dual_modulo_functors.cc:37-91
def partition_by_dual_modulo(ids, num_partitions, modulus, stage):
  assert stage in (1, 2)
  ids = np.ascontiguousarray(ids)
  out = np.empty_like(ids)
  sizes = np.zeros(num_partitions, np.int32)
  indices = np.empty(ids.size, np.int32)
  fn = getattr(lib(), 'orc_partition_by_dual_modulo_' + _SUFFIX[ids.dtype])
  fn(C.c_int32(num_partitions), C.c_int32(modulus), C.c_int32(stage), _p(ids),
     C.c_int32(ids.size), _p(out), _p(sizes), _p(indices))
  return out, sizes, indices


# R4 -- hbtf/distribute/collective.h:80-112
def compute_active_ranks(topology, world_size, local_size, rank):
  out = np.zeros(world_size, np.int32)
  k = lib().orc_compute_active_ranks(
    C.c_int32(topology), C.c_int32(world_size), C.c_int32(local_size),
    C.c_int32(rank), _p(out))
  return out[:k].tolist()


def compute_active_size(topology, world_size, local_size):
  return lib().orc_compute_active_size(
    C.c_int32(topology), C.c_int32(world_size), C.c_int32(local_size))


# R5 -- hbtf/distribute/nccl/nccl_collective.cc:250-288 (+ sizes :112-151)
def alltoallv_sim(send_values, send_sizes):
  """All ranks at once.  send_values[r]: array [sum(send_sizes[r]), ...common];
  send_sizes[r]: int32[W].  Returns (recv_values[r], recv_sizes[r]) lists."""
  world = len(send_values)
  ss = np.ascontiguousarray(np.asarray(send_sizes, np.int32).reshape(world, world))
  rs = np.zeros((world, world), np.int32)
  vals = [np.ascontiguousarray(v) for v in send_values]
  common = vals[0].shape[1:]
  row_bytes = int(np.prod(common, dtype=np.int64)) * vals[0].dtype.itemsize
  tot = ss.sum(axis=0)  # rows each rank receives
  recvs = [np.zeros((int(tot[r]),) + tuple(common), vals[0].dtype)
           for r in range(world)]
  sp = (C.c_void_p * world)(*[v.ctypes.data for v in vals])
  rp = (C.c_void_p * world)(*[v.ctypes.data for v in recvs])
  lib().orc_alltoallv_sim(C.c_int32(world), C.c_int64(row_bytes), sp, _p(ss), rp,
                          _p(rs))
  return recvs, [rs[r].copy() for r in range(world)]


# R6 -- hbtf/common/cast.cu.cc:37-42,60-65
def cast_f32_to_f16(x):
  x = np.ascontiguousarray(x, np.float32)
  out = np.empty(x.shape, np.uint16)
  lib().orc_cast_f32_to_f16(_p(x), C.c_int64(x.size), _p(out))
  return out.view(np.float16)


def cast_f16_to_f32(x):
  x = np.ascontiguousarray(x).view(np.uint16)
  out = np.empty(x.shape, np.float32)
  lib().orc_cast_f16_to_f32(_p(x), C.c_int64(x.size), _p(out))
  return out


# R7 -- hbtf/embedding/sharding.py:186 (TF Unique, first-occurrence order)
def unique(ids):
  ids = np.ascontiguousarray(ids, np.int64)
  uniq = np.empty(max(ids.size, 1), np.int64)
  idx = np.empty(ids.size, np.int32)
  u = lib().orc_unique_i64(_p(ids), C.c_int64(ids.size), _p(uniq), _p(idx))
  return uniq[:u].copy(), idx


# R8 -- hbtf/embedding/sharding.py:191,193,200 (TF GatherV2)
def gather(table, rows):
  table = np.ascontiguousarray(table, np.float32)
  rows = np.ascontiguousarray(rows, np.int64)
  out = np.empty((rows.size, table.shape[1]), np.float32)
  lib().orc_gather_f32(_p(table), C.c_int64(table.shape[0]),
                       C.c_int32(table.shape[1]), _p(rows), C.c_int64(rows.size),
                       _p(out))
  return out


# R9 -- TF sparse_segment_{sum,mean,sqrt_n}; call sites data.py:189-193
def segment_combine(emb, idx, splits, combiner, f64=False):
  emb = np.ascontiguousarray(emb, np.float32)
  splits = np.ascontiguousarray(splits, np.int32)
  n_seg = splits.size - 1
  dim = emb.shape[1]
  if idx is not None:
    idx = np.ascontiguousarray(idx, np.int32)
  ip = _p(idx) if idx is not None else None
  comb = COMBINERS.get(combiner, combiner)
  if f64:
    out = np.empty((n_seg, dim), np.float64)
    lib().orc_segment_combine_f64acc(_p(emb), C.c_int32(dim), ip, _p(splits),
                                     C.c_int64(n_seg), C.c_int32(comb), _p(out))
  else:
    out = np.empty((n_seg, dim), np.float32)
    lib().orc_segment_combine_f32(_p(emb), C.c_int32(dim), ip, _p(splits),
                                  C.c_int64(n_seg), C.c_int32(comb), _p(out))
  return out


# R10 -- TF SparseSegment*Grad + UnsortedSegmentSum; hbtf/distribute/collective.py:334-347
def segment_combine_grad(g_out, splits, combiner):
  g_out = np.ascontiguousarray(g_out, np.float32)
  splits = np.ascontiguousarray(splits, np.int32)
  n_seg = splits.size - 1
  dim = g_out.shape[1]
  n = int(splits[-1]) if splits.size else 0
  out = np.zeros((n, dim), np.float32)
  comb = COMBINERS.get(combiner, combiner)
  lib().orc_segment_combine_grad_f32(_p(g_out), C.c_int32(dim), _p(splits),
                                     C.c_int64(n_seg), C.c_int32(comb), _p(out))
  return out


def unsorted_segment_sum(g, idx, num_segments, f64=False):
  g = np.ascontiguousarray(g, np.float32)
  idx = np.ascontiguousarray(idx, np.int32)
  dim = g.shape[1]
  if f64:
    out = np.empty((num_segments, dim), np.float64)
    lib().orc_unsorted_segment_sum_f64acc(_p(g), C.c_int32(dim), _p(idx),
                                          C.c_int64(idx.size),
                                          C.c_int64(num_segments), _p(out))
  else:
    out = np.empty((num_segments, dim), np.float32)
    lib().orc_unsorted_segment_sum_f32(_p(g), C.c_int32(dim), _p(idx),
                                       C.c_int64(idx.size),
                                       C.c_int64(num_segments), _p(out))
  return out


def sparse_sgd_apply(table, rows, g_u, lr):
  """In place: table[rows[u]] -= lr * g_u[u]."""
  assert table.dtype == np.float32 and table.flags.c_contiguous
  rows = np.ascontiguousarray(rows, np.int64)
  g_u = np.ascontiguousarray(g_u, np.float32)
  lib().orc_sparse_sgd_apply_f32(_p(table), C.c_int64(table.shape[0]),
                                 C.c_int32(table.shape[1]), _p(rows), _p(g_u),
                                 C.c_int64(rows.size), C.c_float(lr))
  return table


def sparse_adagrad_apply(table, accum, rows, g_u, lr):
  """In place, TF1.15 AdagradOptimizer sparse apply on deduplicated slices (third-party TF:
  training_ops SparseApplyAdagrad; call site docs/tutorial/ranking/taobao/train.py:115):
  accum[r] += g * g; table[r] -= lr * g * (1 / sqrt(accum[r])), all in fp32, entries in order."""
  assert table.dtype == np.float32 and accum.dtype == np.float32
  g_u = np.ascontiguousarray(g_u, np.float32)
  lr = np.float32(lr)
  rows = np.asarray(rows, np.int64)
  if np.unique(rows).size == rows.size:
    # distinct rows: the entries do not interact, the same fp32 operations element by element
    a = accum[rows] + g_u * g_u
    accum[rows] = a
    table[rows] = table[rows] - (lr * g_u) * (np.float32(1.0) / np.sqrt(a))
    return table, accum
  for u, r in enumerate(rows):
    a = accum[r] + g_u[u] * g_u[u]
    accum[r] = a
    table[r] = table[r] - (lr * g_u[u]) * (np.float32(1.0) / np.sqrt(a))
  return table, accum


# R11 -- hybridbackend/common/murmur3.cu.h:32-77
def murmur3_hash32(keys):
  keys = np.ascontiguousarray(keys, np.int64)
  out = np.empty(keys.size, np.uint32)
  lib().orc_murmur3_hash32_i64_n(_p(keys), C.c_int64(keys.size), _p(out))
  return out


EMPTY_KEY = -2**63  # hbtf/embedding/service.py:87


# R11 -- hbtf/embedding/lookup_functors.cu.cc:54-149
def cache_probe(keys_cache, slab_size, keys):
  keys_cache = np.ascontiguousarray(keys_cache, np.int64)
  keys = np.ascontiguousarray(keys, np.int64)
  slab_count = keys_cache.size // slab_size
  out = np.empty(keys.size, np.int64)
  lib().orc_cache_probe_i64(_p(keys_cache), C.c_int64(slab_count),
                            C.c_int32(slab_size), _p(keys), C.c_int64(keys.size),
                            _p(out))
  return out


# R13 -- hbtf/embedding/variables.py:93-123
def shard_rows(bucket_size, num_shards, shard, batch_size=0):
  rows = C.c_int64()
  off = C.c_int64()
  sharded = lib().orc_shard_rows(C.c_int64(bucket_size), C.c_int32(num_shards),
                                 C.c_int32(shard), C.c_int64(batch_size),
                                 C.byref(rows), C.byref(off))
  return bool(sharded), rows.value, off.value


# --------------------------------------------------------------------------
# W = 1 group lookup through the C pipeline (the bench's cpu_baseline, kind "port")
class _LookupColumn(C.Structure):
  _fields_ = [('table', C.c_void_p), ('rows', C.c_int64), ('dim', C.c_int32),
              ('ids', C.c_void_p), ('n_ids', C.c_int64), ('splits', C.c_void_p),
              ('n_segments', C.c_int64), ('bucket', C.c_int64),
              ('combiner', C.c_int32), ('out', C.c_void_p)]


def group_lookup_fwd(tables, ids, splits, buckets, combiners, n_threads=1, repeat=1):
  """R12 at W=1: per column bucketize -> partition(P=1) -> unique -> gather ->
  restore/stitch -> combiner.  tables[c]: f32 [rows, dim]; ids[c]: int64 [n];
  splits[c]: int32 [S+1] or None (one id per segment)."""
  n_cols = len(tables)
  cols = (_LookupColumn * n_cols)()
  outs = []
  keep = []
  for c in range(n_cols):
    t = np.ascontiguousarray(tables[c], np.float32)
    i = np.ascontiguousarray(ids[c], np.int64)
    s = None if splits[c] is None else np.ascontiguousarray(splits[c], np.int32)
    n_seg = i.size if s is None else s.size - 1
    o = np.empty((n_seg, t.shape[1]), np.float32)
    keep += [t, i, s]
    outs.append(o)
    cols[c].table = t.ctypes.data
    cols[c].rows = t.shape[0]
    cols[c].dim = t.shape[1]
    cols[c].ids = i.ctypes.data
    cols[c].n_ids = i.size
    cols[c].splits = None if s is None else s.ctypes.data
    cols[c].n_segments = n_seg
    cols[c].bucket = int(buckets[c]) if buckets is not None else 0
    cols[c].combiner = COMBINERS.get(combiners[c], combiners[c])
    cols[c].out = o.ctypes.data
  lib().orc_group_lookup_fwd_repeat(cols, C.c_int32(n_cols), C.c_int32(n_threads),
                                    C.c_int32(repeat))
  return outs


# --------------------------------------------------------------------------
# R12 -- hbtf/embedding/sharding.py:171-205, all W ranks simulated in one process.
def make_shards(table, world):
  """Row-modulo shards: owner = id mod W, local row = id // W
  (sharding.py:182,189; shard length rule variables.py:107-111)."""
  return [np.ascontiguousarray(table[r::world]) for r in range(world)]


def sharded_lookup_fwd(shards, ids_per_rank, wire_f16=False, keep=False):
  """One column.  shards[r]: rank r's local rows [rows_local_r, D];
  ids_per_rank[r]: int64 ids requested by rank r (already bucketized).
  Returns per-rank embeddings [n_r, D] in requester order; with keep=True also
  the intermediates the backward needs."""
  world = len(shards)
  part = [partition_by_modulo(np.asarray(ids_per_rank[r], np.int64), world)
          for r in range(world)]                                   # :182
  shard_ids, shard_sizes = alltoallv_sim([p[0] for p in part],
                                         [p[1] for p in part])     # :184
  ctx = []
  send = []
  for r in range(world):
    uniq, uidx = unique(shard_ids[r])                              # :186
    rows = uniq // world                                           # :189
    emb = gather(shards[r], rows)                                  # :191
    emb = emb[uidx] if uidx.size else emb[:0]                      # :193
    if wire_f16:
      emb = cast_f32_to_f16(emb)
    send.append(np.ascontiguousarray(emb))
    ctx.append((uniq, uidx, rows))
  back, _ = alltoallv_sim(send, shard_sizes)                       # :196
  outs = []
  for r in range(world):
    e = back[r]
    if wire_f16:
      e = cast_f16_to_f32(e)
    outs.append(e[part[r][2]] if part[r][2].size else e[:0])       # :200
  if keep:
    return outs, dict(part=part, shard_sizes=shard_sizes, ctx=ctx)
  return outs


def _group_alltoallv(values, sizes, groups):
  """alltoallv inside disjoint rank groups (Topology.INTRA_NODE / INTER_NODE: active ranks of
  hbtf/distribute/collective.h:80-112): groups = list of rank lists; values[r] holds the chunks
  for the members of r's group in group order.  Returns (recv_values, recv_sizes) indexed by
  global rank."""
  world = len(values)
  recv_v, recv_s = [None] * world, [None] * world
  for g in groups:
    rv, rs = alltoallv_sim([values[r] for r in g], [sizes[r] for r in g])
    for k, r in enumerate(g):
      recv_v[r], recv_s[r] = rv[k], rs[k]
  return recv_v, recv_s


def hierarchical_lookup_fwd(shards, ids_per_rank, local_size):
  """One column through the two-staged lookup of multi-node jobs,
  hbtf/embedding/sharding.py:210-276: ids first travel inside the node to the GPU whose LOCAL
  index owns them (dual-modulo stage 1, intra-node alltoallv, unique), then across nodes to the
  owning rank (stage 2, inter-node alltoallv, unique), rows come back the same way.
  shards[r]: rank r's rows (id mod W == r, local row = id // W); W = len(shards),
  nodes = W // local_size.  Returns per-rank embeddings in requester order."""
  world = len(shards)
  L = int(local_size)
  assert world % L == 0
  M = world // L
  intra = [list(range(n * L, (n + 1) * L)) for n in range(M)]           # same node
  inter = [[m * L + l for m in range(M)] for l in range(L)]             # same local index
  # stage one: shard = (id mod (L*M)) mod L  (:224-228), intra-node exchange (:229-232)
  p0 = [partition_by_dual_modulo(np.asarray(ids_per_rank[r], np.int64), L, M, 1)
        for r in range(world)]
  s0_ids, s0_sizes = _group_alltoallv([p[0] for p in p0], [p[1] for p in p0], intra)
  u0 = [unique(s0_ids[r]) for r in range(world)]                        # :234-236
  # stage two: shard = (id mod (M*L)) div L = the node (:237-240), inter-node exchange (:241-244)
  p1 = [partition_by_dual_modulo(u0[r][0], M, L, 2) for r in range(world)]
  s1_ids, s1_sizes = _group_alltoallv([p[0] for p in p1], [p[1] for p in p1], inter)
  send = []
  for r in range(world):
    uniq, uidx = unique(s1_ids[r])                                      # :245-247
    assert np.all(uniq % world == r)                                    # this rank owns them
    emb = gather(shards[r], uniq // world)                              # :249-254
    send.append(np.ascontiguousarray(emb[uidx] if uidx.size else emb[:0]))   # :256-258
  back1, _ = _group_alltoallv(send, s1_sizes, inter)                    # :260-264
  send0 = []
  for r in range(world):
    e = back1[r][p1[r][2]] if p1[r][2].size else back1[r][:0]           # :265-267 s1 stitch
    send0.append(np.ascontiguousarray(e[u0[r][1]] if u0[r][1].size else e[:0]))  # :268-270
  back0, _ = _group_alltoallv(send0, s0_sizes, intra)                   # :272-276
  return [back0[r][p0[r][2]] if p0[r][2].size else back0[r][:0] for r in range(world)]


def sharded_lookup_bwd(kept, grads_per_rank, world):
  """Reverse of sharded_lookup_fwd for one column.  grads_per_rank[r]:
  [n_r, D] grads w.r.t. the stitched embeddings.  Returns per-rank
  (unique_local_rows int64[u], grad_rows f32[u, D]) = the IndexedSlices handed to
  the optimizer on the shard."""
  part, shard_sizes, ctx = kept['part'], kept['shard_sizes'], kept['ctx']
  send = []
  for r in range(world):
    g = np.ascontiguousarray(grads_per_rank[r], np.float32)
    n = g.shape[0]
    # d(stitch): gather by shard_index -> UnsortedSegmentSum (a permutation)
    send.append(unsorted_segment_sum(g, part[r][2], n))
  # d(XCHG #2): alltoallv of the grad with the exchanged sizes (collective.py:334-347)
  # XCHG #2 delivered part[r][1] rows per peer to requester r: those are the sizes the
  # gradient travels back with
  del shard_sizes
  back, _ = alltoallv_sim(send, [np.asarray(part[r][1], np.int32) for r in range(world)])
  out = []
  for r in range(world):
    uniq, uidx, rows = ctx[r]
    g_u = unsorted_segment_sum(back[r], uidx, uniq.size)  # d(restore): dup reduction
    out.append((rows, g_u))
  return out
