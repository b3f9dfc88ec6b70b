"""Pins the CPU oracle (oracle/) to every known-answer vector the reference holds for
the path before anything is allowed to trust it."""
import json
import os

import numpy as np
import pytest

import oracle


def _load(golden_dir, name):
  with open(os.path.join(golden_dir, name)) as f:
    return json.load(f)


def test_murmur3_vectors(golden_dir):
  g = _load(golden_dir, 'murmur3.json')
  got = oracle.murmur3_hash32(g['keys'])
  assert got.tolist() == g['hash32']


def test_murmur3_against_reference_header_build():
  ref = oracle.ref_lib()
  if ref is None:
    pytest.skip('oracle/_ref not built (reference tree not mounted)')
  rng = np.random.RandomState(1)
  keys = rng.randint(-2**63, 2**63 - 1, size=20000, dtype=np.int64)
  got = oracle.murmur3_hash32(keys)
  want = np.array([ref.ref_murmur3_hash32_i64(int(k)) for k in keys[:2000]], np.uint32)
  assert (got[:2000] == want).all()


def test_partition_by_modulo_kat(golden_dir):
  for k in _load(golden_dir, 'partition.json')['modulo']:
    for dt in (np.int32, np.int64):
      o, s, i = oracle.partition_by_modulo(np.array(k['input'], dt), k['num_partitions'])
      assert o.tolist() == k['output']
      assert s.tolist() == k['sizes']
      assert i.tolist() == k['indices']


def test_partition_by_dual_modulo_kat(golden_dir):
  for k in _load(golden_dir, 'partition.json')['dual']:
    o, s, i = oracle.partition_by_dual_modulo(
      np.array(k['input'], np.int64), k['num_partitions'], k['modulus'], k['stage'])
    assert o.tolist() == k['output']
    assert s.tolist() == k['sizes']
    assert i.tolist() == k['indices']


def test_partition_reference_property_cases(golden_dir):
  # partition_test.py:40-65 and :83-114: x == take(y, idx), len(sizes) == P
  for case in _load(golden_dir, 'partition.json')['property']:
    np.random.seed(case['seed'])
    for _ in range(case['columns']):
      x = np.random.randint(low=case['low'], high=case['high'], size=case['size'],
                            dtype=case['dtype'])
      y, sizes, idx = oracle.partition_by_modulo(x, case['num_partitions'])
      assert len(y) == len(idx)
      assert len(sizes) == case['num_partitions']
      np.testing.assert_equal(x, np.take(y, idx))
      # stability + grouping (what "bit-exact vs the CPU functor" adds)
      shard = np.mod(x.astype(np.int64), case['num_partitions'])
      order = np.argsort(shard, kind='stable')
      np.testing.assert_equal(y, x[order])
      np.testing.assert_equal(sizes, np.bincount(shard, minlength=case['num_partitions']))


def test_partition_empty_input():
  # partition_test.py:67-81
  y, sizes, idx = oracle.partition_by_modulo(np.array([], np.int64), 7)
  assert len(y) == 0 and len(idx) == 0
  assert sizes.tolist() == [0] * 7


def test_partition_unsigned_and_dual_vs_numpy():
  rng = np.random.RandomState(3)
  for dt in (np.uint32, np.uint64, np.int32, np.int64):
    info = np.iinfo(dt)
    x = rng.randint(info.min, info.max, size=5000, dtype=dt)
    for P, M in ((4, 2), (3, 5), (1, 1), (8, 3)):
      y, sizes, idx = oracle.partition_by_modulo(x, P)
      shard = np.array([int(v) % P for v in x])
      np.testing.assert_equal(y, x[np.argsort(shard, kind='stable')])
      pre = np.array([int(v) % (P * M) for v in x])
      for stage, sh in ((1, pre % P), (2, pre // M)):
        y, sizes, idx = oracle.partition_by_dual_modulo(x, P, M, stage)
        np.testing.assert_equal(y, x[np.argsort(sh, kind='stable')])
        np.testing.assert_equal(sizes, np.bincount(sh, minlength=P))
        np.testing.assert_equal(x, y[idx])


def test_alltoallv_kat(golden_dir):
  g = _load(golden_dir, 'alltoallv.json')
  s = g['single']
  outs, sizes = oracle.alltoallv_sim([np.array(v, np.int64) for v in s['inputs']], s['sizes'])
  assert [o.tolist() for o in outs] == s['outputs']
  assert [z.tolist() for z in sizes] == s['out_sizes']
  n = g['n']
  for col in range(2):
    outs, sizes = oracle.alltoallv_sim(
      [np.array(n['inputs'][r][col], np.float32) for r in range(2)],
      [n['sizes'][r][col] for r in range(2)])
    for r in range(2):
      assert outs[r].tolist() == n['outputs'][r][col]
      assert sizes[r].tolist() == n['out_sizes'][r][col]


def test_alltoallv_grad_kat(golden_dir):
  # alltoall_test.py:228-243: loss = sum of per-rank mean(exchanged); d/dx = alltoallv of
  # the upstream grad with the exchanged sizes (collective.py:334-347)
  g = _load(golden_dir, 'alltoallv.json')['grad']
  sizes, gv = g['sizes'], g['g']
  world = 2
  xs = [np.full(sum(sizes[r]), 1.0, np.float32) for r in range(world)]
  outs, out_sizes = oracle.alltoallv_sim(xs, sizes)
  ups = [np.full(outs[r].shape, gv / outs[r].size, np.float32) for r in range(world)]
  grads, _ = oracle.alltoallv_sim(ups, out_sizes)
  g0 = gv / (sizes[0][0] + sizes[1][0])
  g1 = gv / (sizes[0][1] + sizes[1][1])
  np.testing.assert_allclose(grads[0], sizes[0][0] * [g0] + sizes[0][1] * [g1], rtol=1e-6)
  np.testing.assert_allclose(grads[1], sizes[1][0] * [g0] + sizes[1][1] * [g1], rtol=1e-6)


def test_active_ranks():
  # collective.h:80-112
  assert oracle.compute_active_ranks(0, 8, 4, 5) == list(range(8))
  assert oracle.compute_active_ranks(1, 8, 4, 5) == [4, 5, 6, 7]
  assert oracle.compute_active_ranks(2, 8, 4, 5) == [1, 5]
  assert oracle.compute_active_ranks(2, 4, 1, 2) == [0, 1, 2, 3]
  assert oracle.compute_active_size(1, 8, 4) == 4
  assert oracle.compute_active_size(2, 8, 4) == 2


def test_shard_rows_rule():
  # variables.py:93-123
  assert oracle.shard_rows(10, 4, 0) == (True, 3, 0)
  assert oracle.shard_rows(10, 4, 1) == (True, 3, 3)
  assert oracle.shard_rows(10, 4, 2) == (True, 2, 6)
  assert oracle.shard_rows(10, 4, 3) == (True, 2, 8)
  assert oracle.shard_rows(4, 4, 0)[0] is False          # bucket <= W: replicated
  assert oracle.shard_rows(1000, 8, 0, batch_size=1000)[0] is False
  for W in (1, 2, 3, 8):
    rows = [oracle.shard_rows(1000003, W, r)[1] for r in range(W)]
    assert rows == [len(range(r, 1000003, W)) for r in range(W)]  # == owner = id mod W


def test_unique_first_occurrence():
  u, idx = oracle.unique([5, 3, 5, 7, 3, 3, 9])
  assert u.tolist() == [5, 3, 7, 9]
  assert idx.tolist() == [0, 1, 0, 2, 1, 1, 3]
  rng = np.random.RandomState(0)
  x = rng.randint(0, 500, size=5000).astype(np.int64)
  u, idx = oracle.unique(x)
  _, first = np.unique(x, return_index=True)
  np.testing.assert_equal(u, x[np.sort(first)])
  np.testing.assert_equal(u[idx], x)


def test_floormod_matches_python():
  rng = np.random.RandomState(0)
  x = rng.randint(-2**62, 2**62, size=1000).astype(np.int64)
  for m in (1, 7, 1000000, 2**31 + 11):
    np.testing.assert_equal(oracle.floormod(x, m), np.array([int(v) % m for v in x]))


def test_fp16_wire_cast_is_round_to_nearest_even():
  rng = np.random.RandomState(0)
  x = np.concatenate([rng.randn(10000) * s for s in (1e-8, 1e-5, 1e-3, 1, 100, 7e4)])
  x = x.astype(np.float32)
  got = oracle.cast_f32_to_f16(x).view(np.uint16)
  want = x.astype(np.float16).view(np.uint16)
  np.testing.assert_equal(got, want)
  h = np.arange(65536, dtype=np.uint16).view(np.float16)
  back = oracle.cast_f16_to_f32(h)
  ok = ~np.isnan(h.astype(np.float32))
  np.testing.assert_equal(back.view(np.uint32)[ok], h.astype(np.float32).view(np.uint32)[ok])


def test_combiner_in_order_vs_float64():
  rng = np.random.RandomState(5)
  emb = rng.uniform(-1e-3, 1e-3, size=(4000, 16)).astype(np.float32)
  lens = rng.poisson(8, size=300).clip(0, 32)
  splits = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
  idx = rng.randint(0, 4000, size=int(splits[-1])).astype(np.int32)
  for comb in ('sum', 'mean', 'sqrtn'):
    f32 = oracle.segment_combine(emb, idx, splits, comb)
    f64 = oracle.segment_combine(emb, idx, splits, comb, f64=True)
    np.testing.assert_allclose(f32, f64, rtol=1e-5, atol=1e-9)
    assert (f32[lens == 0] == 0).all()      # empty segments are zero rows


def test_config1_ragged_lookup_fixture(golden_dir):
  g = _load(golden_dir, 'config1_ragged_lookup.json')
  table = np.frombuffer(bytes.fromhex(g['table_f32_hex']), np.float32).reshape(-1, g['dim'])
  want = np.frombuffer(bytes.fromhex(g['expected_f32_hex']), np.float32).reshape(-1, g['dim'])
  values = np.array(g['values'], np.int64)
  splits = np.array(g['row_splits'], np.int32)
  got = oracle.group_lookup_fwd([table], [values], [splits], [g['bucket']], [g['combiner']])[0]
  # the fixture's expectation is numpy float64 rounded to fp32 (make_golden.py), not the oracle
  np.testing.assert_allclose(got, want, rtol=g['rtol'], atol=1e-10)
  assert (got[np.diff(splits) == 0] == 0).all()
  # independent numpy restatement of embedding_lookup_sparse(mean)
  rows = values % g['bucket']
  for s in range(len(splits) - 1):
    seg = rows[splits[s]:splits[s + 1]]
    ref = table[seg].astype(np.float64).mean(axis=0) if len(seg) else np.zeros(g['dim'])
    np.testing.assert_allclose(got[s], ref, rtol=1e-5, atol=1e-10)


def test_sharded_pipeline_equals_unsharded_lookup():
  # sharding.py:171-205 composed over W simulated ranks == plain table[ids]
  rng = np.random.RandomState(11)
  table = rng.uniform(-1, 1, size=(1003, 8)).astype(np.float32)
  for W in (1, 2, 3, 8):
    shards = oracle.make_shards(table, W)
    ids = [rng.randint(0, 1003, size=rng.randint(0, 200)).astype(np.int64) for _ in range(W)]
    outs = oracle.sharded_lookup_fwd(shards, ids)
    for r in range(W):
      np.testing.assert_equal(outs[r], table[ids[r]])
    outs16 = oracle.sharded_lookup_fwd(shards, ids, wire_f16=True)
    for r in range(W):
      np.testing.assert_equal(outs16[r], table[ids[r]].astype(np.float16).astype(np.float32))


def test_sharded_backward_equals_dense_scatter():
  rng = np.random.RandomState(12)
  R, D, W = 211, 4, 4
  table = rng.uniform(-1, 1, size=(R, D)).astype(np.float32)
  shards = oracle.make_shards(table, W)
  ids = [rng.randint(0, R, size=150).astype(np.int64) for _ in range(W)]
  _, kept = oracle.sharded_lookup_fwd(shards, ids, keep=True)
  grads = [rng.randn(150, D).astype(np.float32) for _ in range(W)]
  slices = oracle.sharded_lookup_bwd(kept, grads, W)
  dense = np.zeros((R, D), np.float64)
  for r in range(W):
    np.add.at(dense, ids[r], grads[r].astype(np.float64))
  got = np.zeros((R, D), np.float64)
  for r in range(W):
    rows, g_u = slices[r]
    assert len(set(rows.tolist())) == len(rows)      # deduplicated
    got[rows * W + r] += g_u
  np.testing.assert_allclose(got, dense, rtol=1e-5, atol=1e-6)


def test_sparse_adagrad_apply_matches_float64():
  """accum += g^2; var -= lr * g / sqrt(accum), fp32, entries in order (TF AdagradOptimizer's
  sparse apply; third-party TF: restated from its documented update rule, entries marked 'derived')."""
  rng = np.random.RandomState(3)
  table = rng.uniform(-1, 1, size=(50, 8)).astype(np.float32)
  accum = np.full((50, 8), 0.1, np.float32)
  rows = rng.permutation(50)[:20].astype(np.int64)
  g = rng.randn(20, 8).astype(np.float32)
  t, a = table.copy(), accum.copy()
  oracle.sparse_adagrad_apply(t, a, rows, g, 0.05)
  a64 = accum.astype(np.float64)
  t64 = table.astype(np.float64)
  a64[rows] += g.astype(np.float64) ** 2
  t64[rows] -= 0.05 * g / np.sqrt(a64[rows])
  np.testing.assert_allclose(a, a64, rtol=1e-6)
  np.testing.assert_allclose(t, t64, rtol=1e-6, atol=1e-7)
  untouched = np.setdiff1d(np.arange(50), rows)
  np.testing.assert_equal(t[untouched], table[untouched])


@pytest.mark.parametrize('local_size,nodes', [(2, 2), (4, 2), (3, 2), (2, 3), (1, 4), (4, 1)])
def test_hierarchical_lookup_equals_unsharded(local_size, nodes):
  """The two-staged lookup of multi-node jobs (sharding.py:210-276: dual-modulo stage 1 +
  intra-node alltoallv, stage 2 + inter-node alltoallv, and back) returns exactly the rows of the
  unsharded table, whatever the node shape."""
  world = local_size * nodes
  rng = np.random.RandomState(100 + world)
  table = rng.uniform(-1, 1, size=(1009, 8)).astype(np.float32)
  shards = oracle.make_shards(table, world)
  ids = [rng.randint(0, 1009, size=rng.randint(0, 400)).astype(np.int64) for _ in range(world)]
  ids[0] = np.zeros(0, np.int64) if world > 1 else ids[0]
  got = oracle.hierarchical_lookup_fwd(shards, ids, local_size)
  flat = oracle.sharded_lookup_fwd(shards, ids)
  for r in range(world):
    np.testing.assert_equal(got[r], table[ids[r]])
    np.testing.assert_equal(got[r], flat[r])


# ----------------------------------------------------------------------------------------------
# External pin of the fp32 rows R1, R7-R10: the examples PUBLISHED in the TensorFlow 1.15 API
# documentation (the third-party dependency those rows live in; tests/golden/tf115_semantics.json)
# and an independent second implementation (torch's embedding_bag on the CPU).
def csr_of(segment_ids, num_segments=None):
  """sorted segment ids -> row_splits (rows = num_segments or last id + 1; missing ids = empty)"""
  seg = np.asarray(segment_ids, np.int64)
  n_seg = int(num_segments) if num_segments is not None else (int(seg[-1]) + 1 if seg.size else 0)
  return np.concatenate([[0], np.cumsum(np.bincount(seg, minlength=n_seg))]).astype(np.int32)


def test_tf115_unique_example(golden_dir):
  for k in _load(golden_dir, 'tf115_semantics.json')['unique']:
    y, idx = oracle.unique(np.array(k['x'], np.int64))
    assert y.tolist() == k['y'] and idx.tolist() == k['idx']


def test_tf115_floormod_examples(golden_dir):
  for k in _load(golden_dir, 'tf115_semantics.json')['floormod']:
    assert oracle.floormod(np.array([k['x']], np.int64), k['y']).tolist() == [k['out']]


def test_tf115_sparse_segment_examples(golden_dir):
  g = _load(golden_dir, 'tf115_semantics.json')
  for k in g['sparse_segment_sum']:
    splits = csr_of(k['segment_ids'], k['num_segments'])
    got = oracle.segment_combine(np.array(k['data'], np.float32), np.array(k['indices'], np.int32),
                                 splits, 'sum')
    np.testing.assert_equal(got, np.array(k['out'], np.float32))
  for k in g['segment_sum']:
    got = oracle.segment_combine(np.array(k['data'], np.float32), np.array(k['indices'], np.int32),
                                 csr_of(k['segment_ids']), 'sum')
    np.testing.assert_equal(got, np.array(k['out'], np.float32))
  for k in g['segment_mean']:
    got = oracle.segment_combine(np.array(k['data'], np.float32), np.array(k['indices'], np.int32),
                                 csr_of(k['segment_ids']), 'mean')
    np.testing.assert_equal(got, np.array(k['out'], np.float32))
  for k in g['segment_sqrt_n']:
    got = oracle.segment_combine(np.array(k['data'], np.float32), np.array(k['indices'], np.int32),
                                 csr_of(k['segment_ids']), 'sqrtn')
    want = np.array(k['out_times_sqrt_n'], np.float32) / np.sqrt(np.array(k['n'], np.float32))[:, None]
    np.testing.assert_equal(got, want)
  for k in g['unsorted_segment_sum']:
    got = oracle.unsorted_segment_sum(np.array(k['data'], np.float32),
                                      np.array(k['segment_ids'], np.int32), k['num_segments'])
    np.testing.assert_equal(got, np.array(k['out'], np.float32))


def test_tf115_embedding_lookup_sparse_example(golden_dir):
  for k in _load(golden_dir, 'tf115_semantics.json')['embedding_lookup_sparse']:
    rng = np.random.RandomState(4)
    params = rng.randn(5, 20).astype(np.float32)
    ids = np.array(k['sp_ids'], np.int64)
    splits = np.array(k['row_splits'], np.int32)
    assert csr_of([i[0] for i in k['sp_indices']], k['dense_shape'][0]).tolist() == splits.tolist()
    for comb in ('sum', 'mean', 'sqrtn'):
      got = oracle.group_lookup_fwd([params], [ids], [splits], [0], [comb])[0]
      for s, rows in enumerate(k['rows_of_output']):
        if not rows:
          assert (got[s] == 0).all()          # an empty row is a zero row
          continue
        ref = params[rows].astype(np.float64).sum(axis=0)
        ref = ref / len(rows) if comb == 'mean' else ref / np.sqrt(len(rows)) if comb == 'sqrtn' else ref
        np.testing.assert_allclose(got[s], ref, rtol=1e-6)


def test_tf115_sparse_apply_rules(golden_dir):
  for k in _load(golden_dir, 'tf115_semantics.json')['sparse_apply']:
    var, accum = np.array(k['var'], np.float32), np.array(k['accum'], np.float32)
    rows, g, lr = np.array(k['indices'], np.int64), np.array(k['grad'], np.float32), k['lr']
    v = oracle.sparse_sgd_apply(var.copy(), rows, g, lr)
    want = var.copy()
    want[rows] -= np.float32(lr) * g          # "var -= alpha * delta"
    np.testing.assert_equal(v, want)
    v, a = oracle.sparse_adagrad_apply(var.copy(), accum.copy(), rows, g, lr)
    wa = accum.astype(np.float64)
    wv = var.astype(np.float64)
    wa[rows] += g.astype(np.float64) ** 2     # "accum += grad * grad"
    wv[rows] -= lr * g * (1 / np.sqrt(wa[rows]))   # "var -= lr * grad * (1 / sqrt(accum))"
    np.testing.assert_allclose(a, wa, rtol=1e-6)
    np.testing.assert_allclose(v, wv, rtol=1e-6)
    np.testing.assert_equal(v[1], var[1])     # rows without a gradient are untouched


def test_lookup_against_torch_embedding_bag():
  """A second, independent implementation of gather + segment combiner and of its gradient
  (duplicate-row reduction): torch.nn.functional.embedding_bag on the CPU."""
  import torch
  import torch.nn.functional as F
  rng = np.random.RandomState(21)
  for dim, rows, n_seg in ((16, 1000, 300), (5, 37, 64), (128, 4000, 50)):
    table = rng.uniform(-1, 1, size=(rows, dim)).astype(np.float32)
    lens = rng.poisson(4, size=n_seg).clip(0, 12)
    lens[0] = 0
    splits = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ids = rng.randint(0, rows, size=int(splits[-1])).astype(np.int64)
    w = torch.tensor(table, requires_grad=True)
    for comb, mode in (('sum', 'sum'), ('mean', 'mean')):
      got = oracle.group_lookup_fwd([table], [ids], [splits], [0], [comb])[0]
      ref = F.embedding_bag(torch.from_numpy(ids), w, torch.from_numpy(splits[:-1].astype(np.int64)),
                            mode=mode)
      np.testing.assert_allclose(got, ref.detach().numpy(), rtol=1e-5, atol=1e-6)
      # backward: IndexedSlices of the oracle scattered densely == autograd's dense gradient
      g_out = rng.randn(n_seg, dim).astype(np.float32)
      w.grad = None
      ref.backward(torch.from_numpy(g_out))
      g_rows = oracle.segment_combine_grad(g_out, splits, comb)     # d(combiner): one row per id
      uniq, inv = oracle.unique(ids)
      g_u = oracle.unsorted_segment_sum(g_rows, inv, uniq.size)     # duplicate-row reduction
      dense = np.zeros_like(table)
      dense[uniq] = g_u
      np.testing.assert_allclose(dense, w.grad.numpy(), rtol=1e-5, atol=1e-6)
