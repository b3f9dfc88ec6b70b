"""BASELINE.json configs 3, 4 and 5 at their stated workload shape on one MI355X.

Config 3 (26 columns, 1M x 16 tables row-sharded 8 ways, batch 65536 per rank) and config 5
(200 columns, dims 4..128, lookup + grad apply, 8 ranks) run the code that runs at 8 GPUs --
hbk_sharded_lookup_fwd/_bwd -- with the ranks as host threads on one GPU (device copies stand in
for RCCL; 8-GPU hardware is not available to a test).  Config 4 (25 x 1M + 1 x 100M rows, dim
128, Zipf(1.2) ids) runs at full size: the big table is 51.2 GB, so row offsets pass 2^32 floats.
Checks at these sizes: on-device properties against torch's own indexing ops, and the oracle on
sub-samples it finishes in seconds.
"""
import threading

import numpy as np
import pytest
import torch

import oracle
import hybridbackend_amd as hb
from hybridbackend_amd.embedding.sharded import ShardedGroupLookup
from tests.support.tolerance import assert_sums_close, world_grad_sums

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def dev(a):
  return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def host(t):
  return t.detach().cpu().numpy()


def _zipf_ranks(n, n_rows, alpha, gen):
  """Zipf(alpha) ranks in [1, n_rows] by the inverse CDF of the continuous power law."""
  u = torch.rand(n, device=DEV, dtype=torch.float64, generator=gen)
  a = 1.0 - alpha
  x = ((float(n_rows) ** a - 1.0) * u + 1.0) ** (1.0 / a)
  return x.floor().clamp_(1, n_rows).to(torch.int64)


def _run_ranks(world, fn):
  comms = hb.distribute.Collective.local_world(world)
  results, errors = [None] * world, []

  def run(r):
    try:
      with torch.cuda.stream(torch.cuda.Stream()):
        results[r] = fn(r, comms[r])
        torch.cuda.current_stream().synchronize()
    except Exception as e:  # pylint: disable=broad-except
      import traceback
      errors.append((r, repr(e), traceback.format_exc()))

  threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
  for t in threads:
    t.start()
  for t in threads:
    t.join(timeout=90)
  for c in comms:
    c.close()
  assert not errors, errors
  return results


# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize('hot_rows', [0, 1])
def test_config4_full_size_zipf_forward_backward(hbk_option, hot_rows):
  """25 x 1M + 1 x 100M rows, dim 128, Zipf(1.2), batch 65536: forward == table[ids % R] and
  backward == index_add_ on device; oracle on the sub-sample of ids whose rows lie beyond float
  offset 2^32 of the big table.  hot_rows: the forward through the per-wave gather or through the
  256-segment tiles that stage repeated rows in LDS."""
  hbk_option('fwd_hot_rows', hot_rows)
  torch.cuda.empty_cache()          # (blocks cached by earlier tests count as used otherwise)
  free, _ = torch.cuda.mem_get_info()
  if free < 100 * 2**30:
    pytest.skip('needs ~75 GB of HBM')
  dim, batch, n_cols = 128, 65536, 26
  rows = [1_000_000] * 25 + [100_000_000]
  gen = torch.Generator(device=DEV)
  gen.manual_seed(7)
  tables = []
  for c in range(n_cols):
    t = torch.empty(rows[c], dim, device=DEV)
    t.uniform_(-1e-3, 1e-3, generator=gen)
    tables.append(t)
  far0 = (1 << 32) // dim + 1000           # first row whose float offset is >= 2^32 (+ margin)
  ids = []
  for c in range(n_cols):
    rank = _zipf_ranks(batch, rows[c], 1.2, gen)
    if rows[c] > far0:
      # fixed permutation of the big table: a rotation that puts the Zipf head beyond float
      # offset 2^32 (rank 1 -> row far0), the tail wraps around the whole table
      row = (rank - 1 + far0) % rows[c]
    else:
      row = (rank * 2654435761 + 12345) % rows[c]   # multiplicative permutation (odd, coprime)
    # ids arrive un-bucketized: add a multiple of the bucket so the fused floor-mod has work to do
    k = torch.randint(0, 1 << 12, (batch,), device=DEV, dtype=torch.int64, generator=gen)
    ids.append(row + k * rows[c])
  lookup = hb.embedding.GroupLookup(tables, buckets=rows, combiners='sum')
  outs = lookup(ids)
  grads = [torch.randn(batch, dim, device=DEV, generator=gen) for _ in range(n_cols)]
  res = hb.embedding.GroupLookupGrad(lookup)(ids, grads)
  torch.cuda.synchronize()
  for c in range(n_cols):
    r = torch.remainder(ids[c], rows[c])
    assert torch.equal(outs[c], tables[c][r]), f'forward column {c}'
    urows, grows, nu = res[c]
    k = int(nu.item())
    uniq, inv = torch.unique(r, return_inverse=True)
    assert k == uniq.numel(), f'column {c}: {k} rows emitted, {uniq.numel()} distinct'
    order = torch.argsort(urows[:k])
    assert torch.equal(urows[:k][order], uniq), f'backward rows of column {c}'
    dense = torch.zeros(uniq.numel(), dim, device=DEV, dtype=torch.float64)
    dense.index_add_(0, inv, grads[c].double())
    got = grows[:k][order].double()
    # hot rows sum thousands of N(0,1) terms: fp32 error ~ 1e-5 relative to the summed magnitude
    mag = torch.zeros(uniq.numel(), dim, device=DEV, dtype=torch.float64)
    mag.index_add_(0, inv, grads[c].double().abs())
    assert bool(((got - dense).abs() <= 1e-5 * mag + 1e-6).all()), f'backward values of column {c}'
  # the oracle on the sub-sample beyond float offset 2^32 (big table)
  c = n_cols - 1
  r = torch.remainder(ids[c], rows[c])
  slab_rows = 1 << 21
  sel = torch.nonzero((r >= far0) & (r < far0 + slab_rows)).flatten()
  assert sel.numel() > 20000, 'the Zipf head must lie beyond offset 2^32'
  assert int(r.max().item()) * dim >= (1 << 32)
  sel = sel[:20000]
  h_ids = host(ids[c][sel])
  slab = host(tables[c][far0:far0 + slab_rows])
  o_rows = oracle.floormod(h_ids, rows[c])                # integer part: bit-exact
  np.testing.assert_equal(o_rows, host(r[sel]))
  want = oracle.gather(slab, o_rows - far0)
  np.testing.assert_equal(host(outs[c][sel]), want)
  # backward + fused SGD apply on the sub-sample alone, against the oracle's reduction and apply
  sub_ids = ids[c][sel].contiguous()
  sub_g = grads[c][sel].contiguous()
  one = hb.embedding.GroupLookup([tables[c]], buckets=[rows[c]], combiners='sum')
  urows, grows, nu = hb.embedding.GroupLookupGrad(one)([sub_ids], [sub_g], apply_lr=0.5)[0]
  torch.cuda.synchronize()
  k = int(nu.item())
  ou, oinv = oracle.unique(o_rows)
  assert k == ou.size
  want64 = oracle.unsorted_segment_sum(host(sub_g), oinv, ou.size, f64=True)
  pos = {int(v): i for i, v in enumerate(ou.tolist())}
  perm = [pos[int(v)] for v in host(urows)[:k]]
  absum = oracle.unsorted_segment_sum(np.abs(host(sub_g)), oinv, ou.size, f64=True)
  assert (np.abs(host(grows)[:k] - want64[perm]) <= 1e-5 * absum[perm] + 1e-6).all()
  # the step touched exactly the looked-up rows of the 51 GB table, each once
  after = host(tables[c][far0:far0 + slab_rows])
  ref = slab.copy()
  oracle.sparse_sgd_apply(ref, host(urows)[:k] - far0, host(grows)[:k], 0.5)
  np.testing.assert_equal(after, ref)


# ----------------------------------------------------------------------------------------------
def test_config3_workload_eight_ranks_in_process():
  """26 columns, 1M x 16 tables sharded 8 ways, batch 65536 per rank, through
  hbk_sharded_lookup_fwd/_bwd with 8 in-process ranks: forward == the unsharded oracle lookup
  (bit-exact), backward + fused SGD == dense scatter-add of all ranks' gradients."""
  world, n_cols, n_rows, dim, batch = 8, 26, 1_000_000, 16, 65536
  gen = torch.Generator(device=DEV)
  tables = []
  for c in range(n_cols):
    gen.manual_seed(1234 + c)
    tables.append(torch.empty(n_rows, dim, device=DEV).uniform_(-1e-3, 1e-3, generator=gen))
  h_tables = [host(t) for t in tables]
  ids = [[None] * n_cols for _ in range(world)]
  grads = [[None] * n_cols for _ in range(world)]
  for r in range(world):
    for c in range(n_cols):
      gen.manual_seed(42 + c + 100000 * r)
      ids[r][c] = torch.randint(0, 1 << 40, (batch,), device=DEV, dtype=torch.int64, generator=gen)
      grads[r][c] = torch.randn(batch, dim, device=DEV, generator=gen)
  shards = [[t[r::world].contiguous() for t in tables] for r in range(world)]
  lr = 0.25

  def fn(r, coll):
    drv = ShardedGroupLookup(shards[r], coll, buckets=[n_rows] * n_cols, combiners='sum')
    outs = drv(ids[r])
    drv.backward(grads[r], apply_lr=lr)
    torch.cuda.current_stream().synchronize()
    outs = [o.clone() for o in outs]
    drv.close()
    return outs

  res = _run_ranks(world, fn)
  for r in range(world):
    want = oracle.group_lookup_fwd(h_tables, [host(i) for i in ids[r]], [None] * n_cols,
                                   [n_rows] * n_cols, ['sum'] * n_cols, n_threads=8)
    for c in range(n_cols):
      np.testing.assert_equal(host(res[r][c]), want[c])
  # the shards after the fused SGD step == table - lr * (dense scatter-add over all ranks)
  for c in range(n_cols):
    dense = torch.zeros(n_rows, dim, device=DEV, dtype=torch.float64)
    mag = torch.zeros(n_rows, dim, device=DEV, dtype=torch.float64)
    for r in range(world):
      dense.index_add_(0, torch.remainder(ids[r][c], n_rows), grads[r][c].double())
      mag.index_add_(0, torch.remainder(ids[r][c], n_rows), grads[r][c].double().abs())
    ref = tables[c].double() - lr * dense
    mag = tables[c].double().abs() + lr * mag
    for r in range(world):
      assert_sums_close(host(shards[r][c]), host(ref[r::world]), host(mag[r::world]),
                        err_msg=f'column {c}, shard {r}')
      untouched = dense[r::world].abs().sum(1) == 0
      assert torch.equal(shards[r][c][untouched], tables[c][r::world][untouched])


# ----------------------------------------------------------------------------------------------
_DIMS = [4, 8, 12, 16, 24, 32, 36, 48, 64, 80, 128]


def _config5_columns(rng, n_cols):
  dims = [_DIMS[c % len(_DIMS)] for c in range(n_cols)]
  rows = [int(10 ** rng.uniform(3, 5)) for _ in range(n_cols)]   # log-uniform
  ragged = [c % 3 == 2 for c in range(n_cols)]
  combiners = [['sum', 'mean', 'sqrtn'][c % 3] for c in range(n_cols)]
  return dims, rows, ragged, combiners


def test_config5_200_columns_lookup_and_grad_apply_single_gpu():
  """200 columns, dims cycling 4..128, one third ragged: forward bit-equal to the oracle,
  backward IndexedSlices within 1e-5, fused SGD and Adagrad apply bit-equal to the oracle's apply
  of the emitted slices."""
  rng = np.random.RandomState(5)
  n_cols, batch = 200, 2048
  dims, rows, ragged, combiners = _config5_columns(rng, n_cols)
  tables = [rng.uniform(-1, 1, size=(rows[c], dims[c])).astype(np.float32) for c in range(n_cols)]
  ids, splits, grads = [], [], []
  for c in range(n_cols):
    if ragged[c]:
      lens = rng.poisson(4, size=batch).clip(0, 16)
      sp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    else:
      sp = None
    n = batch if sp is None else int(sp[-1])
    splits.append(sp)
    ids.append(rng.randint(0, 2**40, size=n).astype(np.int64))
    grads.append(rng.randn(batch, dims[c]).astype(np.float32))
  d_ids = [dev(i) for i in ids]
  d_sp = [None if s is None else dev(s) for s in splits]
  d_g = [dev(g) for g in grads]
  for opt in ('sgd', 'adagrad'):
    d_tab = [dev(t.copy()) for t in tables]
    d_acc = [torch.full_like(t, 0.1) for t in d_tab]
    lookup = hb.embedding.GroupLookup(d_tab, rows, combiners)
    outs = lookup(d_ids, d_sp)
    want = oracle.group_lookup_fwd(tables, ids, splits, rows, combiners, n_threads=8)
    for c in range(n_cols):
      np.testing.assert_equal(host(outs[c]), want[c])
    res = hb.embedding.GroupLookupGrad(lookup, accums=d_acc)(d_ids, d_g, d_sp, apply_lr=0.05,
                                                               optimizer=opt)
    torch.cuda.synchronize()
    for c in range(n_cols):
      urows, grows, nu = res[c]
      k = int(nu.item())
      r = ids[c] % rows[c]
      ou, oinv = oracle.unique(r)
      assert k == ou.size and set(host(urows)[:k].tolist()) == set(ou.tolist())
      sp = splits[c] if splits[c] is not None else np.arange(r.size + 1, dtype=np.int32)
      g_id = oracle.segment_combine_grad(grads[c], sp, combiners[c])
      want64 = oracle.unsorted_segment_sum(g_id, oinv, ou.size, f64=True)
      abs64 = oracle.unsorted_segment_sum(np.abs(g_id), oinv, ou.size, f64=True)
      pos = {int(v): i for i, v in enumerate(ou.tolist())}
      perm = [pos[int(v)] for v in host(urows)[:k]]
      assert_sums_close(host(grows)[:k], want64[perm], abs64[perm], err_msg=f'{opt}, column {c}')
      t_ref, a_ref = tables[c].copy(), np.full_like(tables[c], 0.1)
      if opt == 'sgd':
        oracle.sparse_sgd_apply(t_ref, host(urows)[:k], host(grows)[:k], 0.05)
      else:
        oracle.sparse_adagrad_apply(t_ref, a_ref, host(urows)[:k], host(grows)[:k], 0.05)
        np.testing.assert_equal(host(d_acc[c]), a_ref)
      np.testing.assert_equal(host(d_tab[c]), t_ref)


def test_config5_stated_size_single_gpu():
  """Config 5 at the size SURVEY 8(d) states: 200 columns, dims cycling 4..128, table rows
  log-uniform in [1e3, 1e7], batch 65536, one third of the columns ragged (Poisson(8) ids per
  sample, clipped to [0, 32]) -- ~35 GB of tables, ~44 M ids per step.  ONE forward call and ONE
  backward call with the fused optimizer step (SGD, then Adagrad) for all 200 columns; checked on
  the device against torch's own indexing: forward == table[ids % R] (bit-equal for one id per
  sample; segment sums within 1e-5 of float64 index_add_), backward rows distinct and ==
  index_add_ of the combiner's gradient, tables / accumulators after the step == the optimizer's
  formula from float64; the oracle on the first 4096 samples of every seventh column
  (bit-equal)."""
  torch.cuda.empty_cache()          # (blocks cached by earlier tests count as used otherwise)
  free, _ = torch.cuda.mem_get_info()
  if free < 160 * 2**30:
    pytest.skip('needs ~130 GB of HBM')
  n_cols, batch, lr = 200, 65536, 0.05
  rng = np.random.RandomState(5)
  gen = torch.Generator(device=DEV)
  gen.manual_seed(5)
  dims = [_DIMS[c % len(_DIMS)] for c in range(n_cols)]
  rows = [int(10 ** rng.uniform(3, 7)) for _ in range(n_cols)]
  combiners = [['sum', 'mean', 'sqrtn'][c % 3] for c in range(n_cols)]
  before = [torch.empty(rows[c], dims[c], device=DEV).uniform_(-1, 1, generator=gen)
            for c in range(n_cols)]
  ids, splits, lens_l, grads = [], [], [], []
  for c in range(n_cols):
    if c % 3 == 2:
      lens = torch.poisson(torch.full((batch,), 8.0, device=DEV), generator=gen).clamp_(0, 32).long()
      sp = torch.zeros(batch + 1, dtype=torch.int32, device=DEV)
      sp[1:] = lens.cumsum(0).int()
      n = int(sp[-1].item())
    else:
      lens, sp, n = None, None, batch
    lens_l.append(lens)
    splits.append(sp)
    ids.append(torch.randint(0, 1 << 40, (n,), device=DEV, dtype=torch.int64, generator=gen))
    grads.append(torch.randn(batch, dims[c], device=DEV, generator=gen))
  assert sum(int(i.numel()) for i in ids) > 40_000_000

  def scaled(x, lens, combiner):   # per-segment scale of the combiner (float64)
    if combiner == 'sum':
      return x
    d = lens.clamp(min=1).double()
    return x / (d if combiner == 'mean' else d.sqrt()).unsqueeze(1)

  for opt in ('sgd', 'adagrad'):
    tabs = [t.clone() for t in before]
    accs = [torch.full_like(t, 0.1) for t in tabs] if opt == 'adagrad' else None
    lookup = hb.embedding.GroupLookup(tabs, rows, combiners)
    outs = lookup(ids, splits)
    res = hb.embedding.GroupLookupGrad(lookup, accums=accs)(ids, grads, splits, apply_lr=lr,
                                                             optimizer=opt)
    torch.cuda.synchronize()
    for c in range(n_cols):
      r = torch.remainder(ids[c], rows[c])
      emb = before[c][r]
      if splits[c] is None:
        if opt == 'sgd':
          assert torch.equal(outs[c], emb), f'forward column {c}'
        g_id = grads[c].double()
      else:
        seg = torch.repeat_interleave(torch.arange(batch, device=DEV), lens_l[c])
        if opt == 'sgd':
          want = torch.zeros(batch, dims[c], device=DEV, dtype=torch.float64)
          want.index_add_(0, seg, emb.double())
          mag = torch.zeros_like(want).index_add_(0, seg, emb.double().abs())
          want, mag = scaled(want, lens_l[c], combiners[c]), scaled(mag, lens_l[c], combiners[c])
          assert bool(((outs[c].double() - want).abs() <= 1e-5 * mag + 1e-7).all()), f'forward column {c}'
        g_id = scaled(grads[c].double(), lens_l[c], combiners[c])[seg]
      urows, grows, nu = res[c]
      k = int(nu.item())
      uniq, inv = torch.unique(r, return_inverse=True)
      assert k == uniq.numel(), f'column {c}: {k} rows emitted, {uniq.numel()} distinct'
      order = torch.argsort(urows[:k])
      assert torch.equal(urows[:k][order], uniq), f'backward rows of column {c}'
      dense = torch.zeros(k, dims[c], device=DEV, dtype=torch.float64).index_add_(0, inv, g_id)
      mag = torch.zeros(k, dims[c], device=DEV, dtype=torch.float64).index_add_(0, inv, g_id.abs())
      assert bool(((grows[:k][order].double() - dense).abs() <= 1e-5 * mag + 1e-6).all()), \
          f'backward values of column {c}'
      # the step, from float64: what the fp32 sum may be off (1e-5 of the summed magnitudes)
      # carries through the optimizer's formula
      got_t = tabs[c][uniq].double()
      if opt == 'sgd':
        want_t = before[c][uniq].double() - lr * dense
        bound = 2e-5 * lr * mag + 2e-7 * want_t.abs() + 1e-7
      else:
        acc = 0.1 + dense * dense
        want_t = before[c][uniq].double() - lr * dense / acc.sqrt()
        bound = 1e-4 * lr * mag + 2e-7 * want_t.abs() + 1e-7
        got_a = accs[c][uniq].double()
        assert bool(((got_a - acc).abs() <= 3e-5 * mag * dense.abs() + 2e-6 * acc + 1e-7).all()), \
            f'accumulators of column {c}'
      assert bool(((got_t - want_t).abs() <= bound).all()), f'stepped rows of column {c}'
      untouched = torch.ones(rows[c], dtype=torch.bool, device=DEV)
      untouched[uniq] = False
      assert torch.equal(tabs[c][untouched], before[c][untouched]), f'untouched rows of column {c}'
      if opt == 'adagrad':
        assert bool((accs[c][untouched] == 0.1).all()), f'untouched accumulators of column {c}'
      if opt == 'sgd' and c % 7 == 0:
        # the oracle on the first 4096 samples: rows compacted so that the host sees a small table
        m = 4096 if splits[c] is None else int(splits[c][4096].item())
        u4, i4 = torch.unique(r[:m], return_inverse=True)
        sp4 = None if splits[c] is None else host(splits[c][:4097])
        want = oracle.group_lookup_fwd([host(before[c][u4])], [host(i4)], [sp4], [0], [combiners[c]])
        np.testing.assert_equal(host(outs[c][:4096]), want[0])
    del tabs, accs, lookup, outs, res


def test_config5_200_columns_eight_ranks_end_to_end_step():
  """The 8-GPU step of config 5 with in-process ranks: 200 columns, mixed dims, one third ragged,
  forward + backward + fused SGD through hbk_sharded_lookup_fwd/_bwd."""
  world, n_cols, batch = 8, 200, 8192
  rng = np.random.RandomState(55)
  dims, rows, ragged, combiners = _config5_columns(rng, n_cols)
  tables = [rng.uniform(-1, 1, size=(rows[c], dims[c])).astype(np.float32) for c in range(n_cols)]
  ids, splits, grads = [], [], []
  for r in range(world):
    rid, rsp, rg = [], [], []
    for c in range(n_cols):
      if ragged[c]:
        lens = rng.poisson(3, size=batch).clip(0, 12)
        sp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
      else:
        sp = None
      n = batch if sp is None else int(sp[-1])
      rsp.append(sp)
      rid.append(rng.randint(0, 2**40, size=n).astype(np.int64))
      rg.append(rng.randn(batch, dims[c]).astype(np.float32))
    ids.append(rid)
    splits.append(rsp)
    grads.append(rg)
  shards = [[dev(t[r::world].copy()) for t in tables] for r in range(world)]
  lr = 0.1

  def fn(r, coll):
    drv = ShardedGroupLookup(shards[r], coll, buckets=rows, combiners=combiners)
    outs = drv([dev(i) for i in ids[r]], [None if s is None else dev(s) for s in splits[r]])
    drv.backward([dev(g) for g in grads[r]], apply_lr=lr)
    torch.cuda.current_stream().synchronize()
    outs = [host(o) for o in outs]
    drv.close()
    return outs

  res = _run_ranks(world, fn)
  for r in range(world):
    want = oracle.group_lookup_fwd(tables, ids[r], splits[r], rows, combiners, n_threads=8)
    for c in range(n_cols):
      np.testing.assert_equal(res[r][c], want[c])
  for c in range(n_cols):
    dense, mag = world_grad_sums(rows[c], dims[c], [(ids[r][c], grads[r][c], splits[r][c], combiners[c])
                                                    for r in range(world)])
    ref = tables[c].astype(np.float64) - lr * dense
    for r in range(world):
      assert_sums_close(host(shards[r][c]), ref[r::world], (np.abs(tables[c]) + lr * mag)[r::world],
                        err_msg=f'column {c}, shard {r}')
