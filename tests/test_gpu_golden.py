"""The HIP path (through the C ABI) against EXTERNAL vectors, not only against the oracle:

  * the worked examples published in the TensorFlow 1.15 API documentation for the ops the fp32
    rows R1, R7-R10 are made of (tests/golden/tf115_semantics.json) -- TF 1.15 is the third-party
    dependency those rows live in;
  * the reference's own alltoallv known-answer tests
    (hybridbackend/tensorflow/distribute/tests/alltoall_test.py:219-269) replayed through
    hbk_alltoallv_n / hbk_alltoall_n with two in-process ranks on the GPU.
"""
import json
import os
import threading

import numpy as np
import pytest
import torch

import oracle
import hybridbackend_amd as hb
from tests.support.tolerance import assert_sums_close

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def dev(a):
  return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def host(t):
  return t.detach().cpu().numpy()


def _golden(golden_dir, name):
  with open(os.path.join(golden_dir, name)) as f:
    return json.load(f)


def csr_of(segment_ids, num_segments=None):
  seg = np.asarray(segment_ids, np.int64)
  n_seg = int(num_segments) if num_segments is not None else (int(seg[-1]) + 1 if seg.size else 0)
  return np.concatenate([[0], np.cumsum(np.bincount(seg, minlength=n_seg))]).astype(np.int32)


# ----------------------------------------------------------------------------------------------
# TF 1.15 published examples
def test_tf115_unique_example(golden_dir):
  for k in _golden(golden_dir, 'tf115_semantics.json')['unique']:
    y, idx = hb.embedding.unique(dev(np.array(k['x'], np.int64)))
    assert host(y).tolist() == k['y'] and host(idx).tolist() == k['idx']


def test_tf115_floormod_examples(golden_dir):
  lib = hb._lib.lib()
  g = _golden(golden_dir, 'tf115_semantics.json')['floormod']
  ins = [dev(np.array([k['x']], np.int64)) for k in g]
  outs = [torch.empty_like(t) for t in ins]
  hb._lib.check(lib.hbk_floormod_n(
    len(g), hb._lib.INT64, hb._lib.ptr_array([t.data_ptr() for t in ins]),
    hb._lib.i64_array([1] * len(g)), hb._lib.i64_array([k['y'] for k in g]),
    hb._lib.ptr_array([t.data_ptr() for t in outs]), hb._lib.current_stream()))
  assert [int(o.item()) for o in outs] == [k['out'] for k in g]
  # the same arithmetic fused into the lookup (bucket > 0): row = floormod(id, bucket)
  for k in g:
    if k['y'] > 4096:
      continue
    table = torch.arange(k['y'], device=DEV, dtype=torch.float32).reshape(-1, 1).contiguous()
    out = hb.embedding.group_lookup([table], [dev(np.array([k['x']], np.int64))],
                                    buckets=[k['y']])[0]
    assert int(out.item()) == k['out']


def test_tf115_sparse_segment_examples(golden_dir):
  """sparse_segment_sum / mean / sqrt_n(data, indices, segment_ids) = the fused lookup with
  table = data, ids = indices, row_splits = CSR of segment_ids."""
  g = _golden(golden_dir, 'tf115_semantics.json')
  for k in g['sparse_segment_sum']:
    splits = csr_of(k['segment_ids'], k['num_segments'])
    out = hb.embedding.group_lookup([dev(np.array(k['data'], np.float32))],
                                    [dev(np.array(k['indices'], np.int64))], [dev(splits)],
                                    combiners='sum')[0]
    np.testing.assert_equal(host(out), np.array(k['out'], np.float32))
  for k in g['segment_sum']:
    out = hb.embedding.group_lookup([dev(np.array(k['data'], np.float32))],
                                    [dev(np.array(k['indices'], np.int64))],
                                    [dev(csr_of(k['segment_ids']))], combiners='sum')[0]
    np.testing.assert_equal(host(out), np.array(k['out'], np.float32))
  for k in g['segment_mean']:
    out = hb.embedding.group_lookup([dev(np.array(k['data'], np.float32))],
                                    [dev(np.array(k['indices'], np.int32))],
                                    [dev(csr_of(k['segment_ids']))], combiners=None)[0]  # None = mean
    np.testing.assert_equal(host(out), np.array(k['out'], np.float32))
  for k in g['segment_sqrt_n']:
    out = hb.embedding.group_lookup([dev(np.array(k['data'], np.float32))],
                                    [dev(np.array(k['indices'], np.int64))],
                                    [dev(csr_of(k['segment_ids']))], combiners='sqrtn')[0]
    want = np.array(k['out_times_sqrt_n'], np.float32) / np.sqrt(np.array(k['n'], np.float32))[:, None]
    np.testing.assert_equal(host(out), want)


def test_tf115_unsorted_segment_sum_example(golden_dir):
  """UnsortedSegmentSum(data, segment_ids) = the backward's duplicate-row reduction with one id
  per segment: grad_rows[u] = sum of the gradient rows whose id is unique_rows[u]."""
  for k in _golden(golden_dir, 'tf115_semantics.json')['unsorted_segment_sum']:
    data = np.array(k['data'], np.float32)
    table = torch.zeros(k['num_segments'], data.shape[1], device=DEV)
    lookup = hb.embedding.GroupLookup([table], None, 'sum')
    urows, grows, nu = hb.embedding.GroupLookupGrad(lookup)(
      [dev(np.array(k['segment_ids'], np.int64))], [dev(data)])[0]
    n = int(nu.item())
    got = np.zeros((k['num_segments'], data.shape[1]), np.float32)
    got[host(urows)[:n]] = host(grows)[:n]
    # (negative segment ids are dropped, as the docstring says)
    assert sorted(host(urows)[:n].tolist()) == sorted(set(i for i in k['segment_ids'] if i >= 0))
    np.testing.assert_equal(got, np.array(k['out'], np.float32))


def test_tf115_embedding_lookup_sparse_example(golden_dir):
  for k in _golden(golden_dir, 'tf115_semantics.json')['embedding_lookup_sparse']:
    rng = np.random.RandomState(4)
    params = rng.randn(5, 20).astype(np.float32)
    ids, splits = np.array(k['sp_ids'], np.int64), np.array(k['row_splits'], np.int32)
    for comb in ('sum', 'mean', 'sqrtn', None):
      out = host(hb.embedding.group_lookup([dev(params)], [dev(ids)], [dev(splits)],
                                           combiners=comb)[0])
      eff = comb or k['default_combiner']
      for s, rows in enumerate(k['rows_of_output']):
        if not rows:
          assert (out[s] == 0).all()          # an empty row is a zero row
          continue
        ref = params[rows].astype(np.float64).sum(axis=0)
        ref = ref / len(rows) if eff == 'mean' else ref / np.sqrt(len(rows)) if eff == 'sqrtn' else ref
        np.testing.assert_allclose(out[s], ref, rtol=1e-5)
      np.testing.assert_equal(out, oracle.group_lookup_fwd([params], [ids], [splits], [0], [eff])[0])


def test_tf115_sparse_apply_rules(golden_dir):
  for k in _golden(golden_dir, 'tf115_semantics.json')['sparse_apply']:
    var, accum = np.array(k['var'], np.float32), np.array(k['accum'], np.float32)
    rows, g, lr = np.array(k['indices'], np.int64), np.array(k['grad'], np.float32), k['lr']
    # SGD: "var -= alpha * delta"
    t = dev(var.copy())
    hb.embedding.GroupLookupGrad(hb.embedding.GroupLookup([t], None, 'sum'))(
      [dev(rows)], [dev(g)], apply_lr=lr)
    want = var.copy()
    want[rows] -= np.float32(lr) * g
    np.testing.assert_equal(host(t), want)
    # Adagrad: "accum += grad * grad; var -= lr * grad * (1 / sqrt(accum))"
    t, a = dev(var.copy()), dev(accum.copy())
    hb.embedding.GroupLookupGrad(hb.embedding.GroupLookup([t], None, 'sum'), accums=[a])(
      [dev(rows)], [dev(g)], apply_lr=lr, optimizer='adagrad')
    wa, wv = accum.astype(np.float64), var.astype(np.float64)
    wa[rows] += g.astype(np.float64) ** 2
    wv[rows] -= lr * g * (1 / np.sqrt(wa[rows]))
    np.testing.assert_allclose(host(a), wa, rtol=1e-6)
    np.testing.assert_allclose(host(t), wv, rtol=1e-6)
    np.testing.assert_equal(host(t)[1], var[1])


def test_lookup_against_torch_embedding_bag_on_device():
  """Second implementation: torch's own embedding_bag + autograd on the GPU."""
  import torch.nn.functional as F
  rng = np.random.RandomState(23)
  for dim, rows, n_seg in ((16, 100000, 20000), (128, 5000, 3000), (5, 37, 500)):
    table = torch.empty(rows, dim, device=DEV).uniform_(-1, 1)
    lens = rng.poisson(4, size=n_seg).clip(0, 12)
    splits = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ids = dev(rng.randint(0, rows, size=int(splits[-1])).astype(np.int64))
    for comb in ('sum', 'mean'):
      w = table.clone().requires_grad_(True)
      ref = F.embedding_bag(ids, w, dev(splits[:-1].astype(np.int64)), mode=comb)
      lookup = hb.embedding.GroupLookup([table], None, comb)
      out = lookup([ids], [dev(splits)])[0]
      # two fp32 implementations, each within 1e-5 of the magnitude of a segment's terms
      lens_d = dev(np.maximum(lens, 1).astype(np.float32))[:, None]
      seg_mag = F.embedding_bag(ids, table.abs(), dev(splits[:-1].astype(np.int64)), mode='sum')
      if comb == 'mean':
        seg_mag = seg_mag / lens_d
      assert_sums_close(host(out), host(ref.detach()), host(seg_mag), rel=2e-5)
      g_out = torch.randn(n_seg, dim, device=DEV)
      ref.backward(g_out)
      urows, grows, nu = hb.embedding.GroupLookupGrad(lookup)([ids], [g_out], [dev(splits)])[0]
      n = int(nu.item())
      dense = torch.zeros_like(table)
      dense[urows[:n]] = grows[:n]
      seg_of = np.repeat(np.arange(n_seg), lens)
      g_id = g_out.abs() / lens_d if comb == 'mean' else g_out.abs()
      mag = torch.zeros_like(table).index_add_(0, ids, g_id[dev(seg_of)])
      assert_sums_close(host(dense), host(w.grad), host(mag), rel=2e-5)


# ----------------------------------------------------------------------------------------------
# the reference's alltoallv KATs through hbk_alltoallv_n, two in-process ranks
def _run_ranks(world, fn):
  comms = hb.distribute.Collective.local_world(world)
  results, errors = [None] * world, []

  def run(r):
    try:
      with torch.cuda.stream(torch.cuda.Stream()):
        results[r] = fn(r, comms[r])
        torch.cuda.current_stream().synchronize()
    except Exception as e:  # pylint: disable=broad-except
      errors.append((r, repr(e)))

  threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
  for t in threads:
    t.start()
  for t in threads:
    t.join(timeout=45)
  for c in comms:
    c.close()
  assert not errors, errors
  return results


def test_reference_alltoallv_kat_single(golden_dir):
  # alltoall_test.py:219-226: hb.distribute.alltoall(value, sizes) on two devices
  s = _golden(golden_dir, 'alltoallv.json')['single']

  def fn(r, coll):
    out, out_sizes = coll.alltoall(dev(np.array(s['inputs'][r], np.int64)),
                                   sizes=dev(np.array(s['sizes'][r], np.int32)))
    return host(out).tolist(), host(out_sizes).tolist()

  res = _run_ranks(2, fn)
  for r in range(2):
    assert res[r][0] == s['outputs'][r]
    assert res[r][1] == s['out_sizes'][r]


@pytest.mark.parametrize('wire16', [False, True])
def test_reference_alltoallv_kat_n_columns(golden_dir, wire16):
  # alltoall_test.py:254-269: HbNcclAlltoallvN, two columns (values exactly representable in fp16)
  n = _golden(golden_dir, 'alltoallv.json')['n']

  def fn(r, coll):
    vals = [dev(np.array(n['inputs'][r][c], np.float32)) for c in range(2)]
    sizes = [dev(np.array(n['sizes'][r][c], np.int32)) for c in range(2)]
    recv = coll.alltoall_n(sizes)                       # the op's own size exchange
    torch.cuda.current_stream().synchronize()
    outs = coll.alltoallv_n(vals, [n['sizes'][r][c] for c in range(2)],
                            [host(x).tolist() for x in recv],
                            wire_dtype=torch.float16 if wire16 else None)
    return [host(o).tolist() for o in outs], [host(x).tolist() for x in recv]

  res = _run_ranks(2, fn)
  for r in range(2):
    for c in range(2):
      assert res[r][0][c] == n['outputs'][r][c]
      assert res[r][1][c] == n['out_sizes'][r][c]


def test_reference_alltoallv_grad_kat(golden_dir):
  # alltoall_test.py:228-243: loss = sum over ranks of mean(exchanged); the gradient is the
  # alltoallv of the upstream gradient with the exchanged sizes (collective.py:334-347)
  g = _golden(golden_dir, 'alltoallv.json')['grad']
  sizes, gv = g['sizes'], g['g']

  def fn(r, coll):
    x = torch.ones(sum(sizes[r]), device=DEV)
    out, out_sizes = coll.alltoall(x, sizes=dev(np.array(sizes[r], np.int32)))
    up = torch.full_like(out, gv / out.numel())
    back = coll.alltoallv_n([up], [host(out_sizes).tolist()], [sizes[r]])[0]
    return host(back)

  res = _run_ranks(2, fn)
  g0 = gv / (sizes[0][0] + sizes[1][0])
  g1 = gv / (sizes[0][1] + sizes[1][1])
  np.testing.assert_allclose(res[0], sizes[0][0] * [g0] + sizes[0][1] * [g1], rtol=1e-6)
  np.testing.assert_allclose(res[1], sizes[1][0] * [g0] + sizes[1][1] * [g1], rtol=1e-6)
