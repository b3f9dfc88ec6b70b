"""Randomised parity: hypothesis draws shapes, dims, id distributions, combiners and ragged
layouts; the fused lookup, its backward and the stable partition must agree with the CPU oracle
on every draw (bit-exact integers and in-order fp32 sums, 1e-5 against float64 for the backward)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests.support.tolerance import (WIRE16_FLOOR, WIRE16_REL, assert_sums_close,  # noqa: E402
                                     dense_sums, world_grad_sums)

hypothesis = pytest.importorskip('hypothesis')
from hypothesis import HealthCheck, given, settings  # noqa: E402
from hypothesis import strategies as st  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
# the committed runs are deterministic; HBK_FUZZ_RANDOM=1 HBK_FUZZ_SCALE=10 hunts with fresh draws
_RANDOM = os.environ.get('HBK_FUZZ_RANDOM') == '1'
_SCALE = int(os.environ.get('HBK_FUZZ_SCALE', '1'))


def _cfg(n):
  return settings(max_examples=n * _SCALE, deadline=None, derandomize=not _RANDOM, database=None,
                  suppress_health_check=list(HealthCheck))


def dev(x):
  return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


column = st.fixed_dictionaries({
  'dim': st.sampled_from([1, 3, 4, 6, 8, 16, 20, 32, 64, 128, 256]),
  'rows': st.sampled_from([1, 2, 7, 64, 1000, 65537]),
  'n_seg': st.integers(0, 700),
  'ragged': st.booleans(),
  'max_len': st.integers(0, 9),
  'combiner': st.sampled_from(['sum', 'mean', 'sqrtn']),
  'skew': st.sampled_from(['uniform', 'zipf', 'one', 'negative', 'int32']),
})


def _ids(rng, n, rows, skew):
  if skew == 'zipf':
    return (rng.zipf(1.3, size=n) % (4 * rows)).astype(np.int64)
  if skew == 'one':
    return np.full(n, 3, np.int64)
  if skew == 'negative':
    return rng.randint(-2**40, 2**40, size=n).astype(np.int64)
  if skew == 'int32':
    return rng.randint(-2**31, 2**31 - 1, size=n).astype(np.int32)
  return rng.randint(0, 2**40, size=n).astype(np.int64)


@_cfg(80)
@given(cols=st.lists(column, min_size=1, max_size=6), seed=st.integers(0, 2**31 - 1))
def test_group_lookup_forward_backward_random(cols, seed):
  _check_group_lookup(cols, seed)


big_column = st.fixed_dictionaries({
  'dim': st.sampled_from([4, 8, 16, 32, 64, 128]),
  'rows': st.sampled_from([200, 5000, 100000, 3000000]),
  'n_seg': st.sampled_from([3000, 9000, 20000, 50000]),
  'ragged': st.booleans(),
  'max_len': st.integers(1, 4),
  'combiner': st.sampled_from(['sum', 'mean']),
  'skew': st.sampled_from(['uniform', 'zipf', 'negative']),
})
plan_options = st.fixed_dictionaries({
  'bwd_dense': st.sampled_from([0, 1, 2, 3]),       # never / by policy / bitmap / row-sorted buckets wherever they fit
  'bwd_onepass': st.sampled_from([0, 1]),           # one-launch grouping or histogram/scan/scatter
  'bwd_group_cols': st.sampled_from([1, 3, 64]),    # columns per launch group
  'bwd_bucket_pairs': st.sampled_from([200, 448]),
  'fwd_hot_rows': st.sampled_from([0, 1]),
  'bwd_wide': st.sampled_from([0, 1, 2]),           # wide sorted walk: never / one id per sample / all
  'bwd_xcd': st.sampled_from([0, 1, 2, 4]),         # reduce jobs round robin / by rule / equal slot ranges / equal work ranges
  'fwd_xcd': st.sampled_from([0, 2]),               # lookup tiles round robin / contiguous per XCD
  'fwd_interleave': st.sampled_from([0, 3]),        # lookup tiles column first / row tile first wherever tile counts agree
  'bwd_streams': st.sampled_from([0, 2, 4]),        # launch groups on the caller's stream / over library streams
  'bwd_large_first': st.sampled_from([0, 1]),       # order of the launch groups
  'bwd_rowsort_ratio': st.sampled_from([0, 8, 64]), # row-sorted buckets never / by the shipped ratio / for sparse columns too
  'bwd_lds_pad': st.sampled_from([0, 14]),          # fewer resident tiles in the grouping launches
  'bwd_simple': st.sampled_from([0, 1]),            # the grouping kernels' general instantiation / the one for plain columns
  'fwd_d16': st.sampled_from([0, 1]),               # the gather's general instantiation / the one for rows of 16 floats
})


@_cfg(20)
@given(cols=st.lists(st.one_of(big_column, column), min_size=1, max_size=5), opts=plan_options,
       seed=st.integers(0, 2**31 - 1))
def test_group_lookup_random_plans(cols, opts, seed):
  """The same check over columns large enough for the multi-tile grouping, the split buckets and
  the row-range (dense) buckets, with the plan switches drawn too: every plan the host can choose
  gives the oracle's result."""
  from hybridbackend_amd import _lib
  old = {k: _lib.set_option(k, v) for k, v in opts.items()}
  try:
    _check_group_lookup(cols, seed)
  finally:
    for k, v in old.items():
      _lib.set_option(k, v)


@_cfg(40)
@given(cols=st.lists(st.one_of(big_column, column), min_size=1, max_size=6),
       seed=st.integers(0, 2**31 - 1))
def test_group_lookup_deterministic_random(cols, seed):
  """Option bwd_deterministic over the same random columns (round 6): the emitted rows are the
  distinct valid rows ascending, their sums BIT-EQUAL to the sequential fp32 sum in id order, the
  fused step (SGD or Adagrad by the seed, on contiguous tables or -- every third draw -- on weights and
  accumulator interleaved per row) bit-equal to the oracle's apply of those sums."""
  import oracle
  import hybridbackend_amd as hb
  from hybridbackend_amd import _lib
  rng = np.random.RandomState(seed)
  tables, ids, splits, buckets, combs, grads = [], [], [], [], [], []
  for c in cols:
    if c['dim'] % 4 != 0 and c['dim'] > 64:
      c = dict(c, dim=64)
    tables.append(rng.uniform(-1, 1, size=(c['rows'], c['dim'])).astype(np.float32))
    if c['ragged']:
      lens = rng.randint(0, c['max_len'] + 1, size=c['n_seg'])
      sp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
      n = int(sp[-1])
    else:
      sp, n = None, c['n_seg']
    splits.append(sp)
    ids.append(np.asarray(_ids(rng, n, c['rows'], c['skew']), np.int64))
    buckets.append(c['rows'])
    combs.append(c['combiner'])
    grads.append(rng.randn(c['n_seg'], c['dim']).astype(np.float32))
  opt = 'adagrad' if seed % 2 else 'sgd'
  interleaved = seed % 3 == 0
  # (mostly the row-sorted jobs' in-order form, with the grouping form and the bucket count drawn too --
  # one bucket per column makes jobs of many chunks; every fifth draw the sort for every column)
  opts = {'bwd_deterministic': 2 if seed % 5 == 4 else 1, 'bwd_onepass': (seed // 5) % 2,
          'bwd_buckets_log2': (-1, -1, 0, 3)[(seed // 10) % 4], 'bwd_pairs_packed': (seed // 40) % 2,
          'bwd_seg_inline': (seed // 80) % 2, 'bwd_scatter_staged': (seed // 160) % 2,
          'bwd_scale_fused': (seed // 320) % 2}
  old = {k: _lib.set_option(k, v) for k, v in opts.items()}
  try:
    t_dev = [dev(t.copy()) for t in tables]
    a_dev = [torch.full_like(t, 0.1) for t in t_dev]
    inter = [dev(np.concatenate([t, np.full_like(t, 0.1)], axis=1)) for t in tables]
    lookup = hb.embedding.GroupLookup(t_dev, buckets, combs)
    grad = (hb.embedding.GroupLookupGrad(lookup, interleaved=inter) if interleaved
            else hb.embedding.GroupLookupGrad(lookup, accums=a_dev))
    res = grad([dev(i) for i in ids], [dev(g) for g in grads],
               [None if s is None else dev(s) for s in splits], apply_lr=0.03, optimizer=opt)
    for k in range(len(cols)):
      rows = ids[k] % buckets[k]
      sp = splits[k] if splits[k] is not None else np.arange(rows.size + 1, dtype=np.int32)
      g_id = oracle.segment_combine_grad(grads[k], sp, combs[k])
      uniq = np.unique(rows)
      want = oracle.unsorted_segment_sum(g_id, np.searchsorted(uniq, rows).astype(np.int32),
                                         uniq.size)
      u, g, nu = res[k]
      n = int(nu.item())
      assert n == uniq.size
      np.testing.assert_equal(u.cpu().numpy()[:n], uniq)
      np.testing.assert_equal(g.cpu().numpy()[:n], want)
      want_t, want_a = tables[k].copy(), np.full(tables[k].shape, 0.1, np.float32)
      if opt == 'adagrad':
        oracle.sparse_adagrad_apply(want_t, want_a, uniq, want, 0.03)
      else:
        oracle.sparse_sgd_apply(want_t, uniq, want, 0.03)
      d = tables[k].shape[1]
      got_t = inter[k].cpu().numpy()[:, :d] if interleaved else t_dev[k].cpu().numpy()
      got_a = inter[k].cpu().numpy()[:, d:] if interleaved else a_dev[k].cpu().numpy()
      np.testing.assert_equal(got_t, want_t)
      np.testing.assert_equal(got_a, want_a)
  finally:
    for k, v in old.items():
      _lib.set_option(k, v)


def _check_group_lookup(cols, seed):
  import oracle
  import hybridbackend_amd as hb
  rng = np.random.RandomState(seed)
  tables, ids, splits, buckets, combs, grads = [], [], [], [], [], []
  for c in cols:
    if c['dim'] % 4 != 0 and c['dim'] > 64:
      c = dict(c, dim=64)
    tables.append(rng.uniform(-1, 1, size=(c['rows'], c['dim'])).astype(np.float32))
    if c['ragged']:
      lens = rng.randint(0, c['max_len'] + 1, size=c['n_seg'])
      sp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
      n = int(sp[-1])
    else:
      sp, n = None, c['n_seg']
    splits.append(sp)
    ids.append(_ids(rng, n, c['rows'], c['skew']))
    buckets.append(c['rows'])
    combs.append(c['combiner'])
    grads.append(rng.randn(c['n_seg'], c['dim']).astype(np.float32))
  # hbk_group_lookup_* take one ids dtype per column; the Python wrapper one per call
  if any(i.dtype == np.int32 for i in ids):
    ids = [np.asarray(i, np.int64) for i in ids]
  lookup = hb.embedding.GroupLookup([dev(t) for t in tables], buckets, combs)
  d_ids = [dev(i) for i in ids]
  d_sp = [None if s is None else dev(s) for s in splits]
  outs = lookup(d_ids, d_sp)
  want = oracle.group_lookup_fwd(tables, ids, splits, buckets, combs)
  for o, w in zip(outs, want):
    np.testing.assert_equal(o.cpu().numpy(), w)
  res = hb.embedding.GroupLookupGrad(lookup)(d_ids, [dev(g) for g in grads], d_sp)
  for k in range(len(cols)):
    rows = np.asarray(ids[k], np.int64) % buckets[k]
    sp = splits[k] if splits[k] is not None else np.arange(rows.size + 1, dtype=np.int32)
    g_id = oracle.segment_combine_grad(grads[k], sp, combs[k]).astype(np.float64)
    dense, mag = dense_sums(tables[k].shape, rows, g_id)
    u, g, nu = res[k]
    n = int(nu.item())
    got_rows = u.cpu().numpy()[:n]
    assert len(set(got_rows.tolist())) == n and n == np.unique(rows).size
    got = np.zeros_like(dense)
    got[got_rows] = g.cpu().numpy()[:n]
    # fp32 sums in an order that is not fixed: 1e-5 relative to the magnitude of the summed terms
    assert_sums_close(got, dense, mag, floor=1e-12, err_msg=f'column {k}')
  # the fused optimizer step equals the oracle's step applied to the emitted slices, bit for bit
  opt = 'adagrad' if seed % 2 else 'sgd'
  t2 = [dev(t.copy()) for t in tables]
  a2 = [torch.full_like(t, 0.1) for t in t2]
  lookup2 = hb.embedding.GroupLookup(t2, buckets, combs)
  res2 = hb.embedding.GroupLookupGrad(lookup2, accums=a2)(
    d_ids, [dev(g) for g in grads], d_sp, apply_lr=0.03, optimizer=opt)
  for k in range(len(cols)):
    u, g, nu = res2[k]
    n = int(nu.item())
    want_t, want_a = tables[k].copy(), np.full(tables[k].shape, 0.1, np.float32)
    if opt == 'adagrad':
      oracle.sparse_adagrad_apply(want_t, want_a, u.cpu().numpy()[:n], g.cpu().numpy()[:n], 0.03)
      np.testing.assert_equal(a2[k].cpu().numpy(), want_a)
    else:
      oracle.sparse_sgd_apply(want_t, u.cpu().numpy()[:n], g.cpu().numpy()[:n], 0.03)
    np.testing.assert_equal(t2[k].cpu().numpy(), want_t)


@_cfg(80)
@given(lens=st.lists(st.integers(0, 5000), min_size=1, max_size=5),
       P=st.sampled_from([1, 2, 3, 7, 8, 9, 16, 33, 64, 65, 300]),
       dtype=st.sampled_from([np.int32, np.int64, np.uint32, np.uint64]),
       seed=st.integers(0, 2**31 - 1))
def test_partition_random(lens, P, dtype, seed):
  import oracle
  import hybridbackend_amd as hb
  if dtype in (np.uint32, np.uint64) and not hasattr(torch, 'uint64'):
    dtype = np.int64
  rng = np.random.RandomState(seed)
  info = np.iinfo(dtype)
  xs = [rng.randint(info.min, info.max, size=n, dtype=dtype) for n in lens]
  outs, sizes, idxs = hb.distribute.partition_by_modulo_n([dev(x) for x in xs], P)
  for x, o, s, i in zip(xs, outs, sizes, idxs):
    wo, ws, wi = oracle.partition_by_modulo(x, P)
    np.testing.assert_equal(o.cpu().numpy(), wo)
    np.testing.assert_equal(s.cpu().numpy(), ws)
    np.testing.assert_equal(i.cpu().numpy(), wi)


sharded_column = st.fixed_dictionaries({
  'dim': st.sampled_from([4, 6, 8, 16, 20, 64, 128]),
  'rows': st.sampled_from([3, 64, 1000, 50021]),
  'ragged': st.booleans(),
  'combiner': st.sampled_from(['sum', 'mean', 'sqrtn']),
})


@_cfg(24)
@given(world=st.sampled_from([2, 3, 5]), cols=st.lists(sharded_column, min_size=1, max_size=5),
       wire16=st.booleans(), hot=st.sampled_from([False, True, 'auto']),
       dedup=st.sampled_from(['none', 'all', 'mixed']), seed=st.integers(0, 2**31 - 1))
def test_sharded_driver_random_in_process_world(world, cols, wire16, hot, dedup, seed):
  """hbk_sharded_lookup_fwd/_bwd with W in-process ranks (W not a power of two included), random
  columns, some ranks / columns empty, requester-side dedup on no / every / every other column,
  ids drawn from few or many values: forward == unsharded oracle, backward == dense
  scatter-add."""
  import threading
  import oracle
  import hybridbackend_amd as hb
  from hybridbackend_amd.embedding.sharded import ShardedGroupLookup
  rng = np.random.RandomState(seed)
  n = len(cols)
  dims = [c['dim'] for c in cols]
  rows = [max(c['rows'], world) for c in cols]   # every rank owns at least one row
  combs = [c['combiner'] for c in cols]
  tables = [rng.uniform(-1, 1, size=(rows[k], dims[k])).astype(np.float32) for k in range(n)]
  ids, splits, grads = [], [], []
  for r in range(world):
    ri, rs, rg = [], [], []
    for k, c in enumerate(cols):
      n_seg = int(rng.choice([0, 1, 77, 600]))
      if c['ragged']:
        lens = rng.randint(0, 6, size=n_seg)
        sp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        cnt = int(sp[-1])
      else:
        sp, cnt = None, n_seg
      rs.append(sp)
      # (every other column repeats its ids heavily: what dedup is for)
      hi = 2**40 if k % 2 == 0 else 40
      ri.append(rng.randint(0, hi, size=cnt).astype(np.int64))
      rg.append(rng.randn(n_seg, dims[k]).astype(np.float32))
    ids.append(ri)
    splits.append(rs)
    grads.append(rg)
  comms = hb.distribute.Collective.local_world(world)
  shards = [[dev(t[r::world].copy()) for t in tables] for r in range(world)]
  results, errors = [None] * world, []

  def run(r):
    try:
      with torch.cuda.stream(torch.cuda.Stream()):
        drv = ShardedGroupLookup(shards[r], comms[r], buckets=rows, combiners=combs,
                                 wire_dtype=torch.float16 if wire16 else None, hot_rows=hot,
                                 dedup=[dedup == 'all' or (dedup == 'mixed' and k % 2 == 1)
                                        for k in range(n)])
        outs = drv([dev(i) for i in ids[r]], [None if s is None else dev(s) for s in splits[r]])
        sl = drv.backward([dev(g) for g in grads[r]])
        if hot == 'auto':   # a second step, after the counts of the first have (maybe) landed
          torch.cuda.current_stream().synchronize()
          outs = drv([dev(i) for i in ids[r]], [None if s is None else dev(s) for s in splits[r]])
          sl = drv.backward([dev(g) for g in grads[r]])
        torch.cuda.current_stream().synchronize()
        results[r] = ([o.cpu().numpy() for o in outs],
                      [(u.cpu().numpy()[:int(k.item())], g.cpu().numpy()[:int(k.item())])
                       for u, g, k in sl])
        drv.close()
    except Exception as e:  # pylint: disable=broad-except
      errors.append((r, repr(e)))

  threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
  for t in threads:
    t.start()
  for t in threads:
    t.join(timeout=45)
  for cm in comms:
    cm.close()
  assert not errors, errors
  eff, rel, floor = tables, 1e-5, 1e-6
  if wire16:
    eff = [oracle.cast_f16_to_f32(oracle.cast_f32_to_f16(t)) for t in tables]
    rel, floor = WIRE16_REL, WIRE16_FLOOR
  for r in range(world):
    want = oracle.group_lookup_fwd(eff, ids[r], splits[r], rows, combs)
    for k in range(n):
      np.testing.assert_equal(results[r][0][k], want[k])
  for k in range(n):
    dense, mag = world_grad_sums(rows[k], dims[k], [(ids[r][k], grads[r][k], splits[r][k], combs[k])
                                                    for r in range(world)])
    got = np.zeros_like(dense)
    for r in range(world):
      lr_, g_ = results[r][1][k]
      assert len(set(lr_.tolist())) == len(lr_)
      got[lr_ * world + r] += g_
    assert_sums_close(got, dense, mag, rel=rel, floor=floor, err_msg=f'column {k}')
