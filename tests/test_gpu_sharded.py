"""The sharded pipeline driver (R12, sharding.py:171-205) on one GPU: W virtual ranks run the
real HIP phases in one process; only the transport is simulated (device copies with the
Alltoallv offset arithmetic of nccl_collective.cc:250-288).  The RCCL transport itself is
covered at world size 1 in test_gpu_parity.py and by the gloo tests' offset arithmetic."""
import numpy as np
import pytest
import torch

import oracle
import hybridbackend_amd as hb
from hybridbackend_amd import _lib
from hybridbackend_amd.embedding.sharded import ShardedGroupLookup
from tests.support.tolerance import WIRE16_FLOOR, WIRE16_REL, assert_sums_close, dense_sums, world_grad_sums

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def dev(a):
  return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def fake_alltoallv(send_vals, send_sizes):
  """send_vals[r]: tensor [sum(send_sizes[r]), ...]; send_sizes[r][i] rows r -> i."""
  W = len(send_vals)
  outs = []
  for r in range(W):
    chunks = []
    for i in range(W):
      off = sum(send_sizes[i][:r])
      chunks.append(send_vals[i][off:off + send_sizes[i][r]])
    outs.append(torch.cat(chunks, 0).contiguous())
  return outs


@pytest.mark.parametrize('world', [1, 2, 4, 8])
@pytest.mark.parametrize('wire16', [False, True])
def test_sharded_forward_equals_unsharded(world, wire16):
  rng = np.random.RandomState(100 + world)
  dims = [16, 4, 32, 16, 128]
  rows = [1000003, 977, 5000, 64, 20011]
  combiners = ['sum', 'mean', 'sqrtn', 'sum', 'mean']
  n = len(dims)
  tables = [rng.uniform(-1e-3, 1e-3, size=(rows[c], dims[c])).astype(np.float32)
            for c in range(n)]
  drivers, ids, splits = [], [], []
  for r in range(world):
    shards = [dev(t[r::world]) for t in tables]
    drivers.append(ShardedGroupLookup(shards, None, buckets=rows, combiners=combiners,
                                      world_size=world))
    rid, rsp = [], []
    for c in range(n):
      if c % 2 == 0:
        rsp.append(None)
        rid.append(rng.randint(0, 2**40, size=rng.randint(0, 3000)).astype(np.int64))
      else:
        lens = rng.poisson(3, size=rng.randint(1, 500)).clip(0, 12)
        sp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        rsp.append(sp)
        rid.append(rng.randint(0, 2**40, size=int(sp[-1])).astype(np.int64))
    ids.append(rid)
    splits.append(rsp)
  # phase 1 on every rank
  sts = [drivers[r].partition([dev(i) for i in ids[r]],
                              [None if s is None else dev(s) for s in splits[r]])
         for r in range(world)]
  for r in range(world):   # integer step: bit-exact vs the CPU functor
    for c in range(n):
      oy, osz, oi = oracle.partition_by_modulo(oracle.floormod(ids[r][c], rows[c]), world)
      np.testing.assert_equal(sts[r].send_ids[c].cpu().numpy(), oy)
      np.testing.assert_equal(sts[r].send_sizes[c].cpu().numpy(), osz)
      np.testing.assert_equal(sts[r].shard_index[c].cpu().numpy(), oi)
  sizes = [[sts[r].send_sizes[c].cpu().tolist() for r in range(world)] for c in range(n)]
  # exchange 1: ids
  recv_ids = [[None] * n for _ in range(world)]
  for c in range(n):
    got = fake_alltoallv([sts[r].send_ids[c] for r in range(world)], sizes[c])
    for r in range(world):
      recv_ids[r][c] = got[r]
  send_rows = [drivers[r].owner_gather(sts[r], recv_ids[r]) for r in range(world)]
  # exchange 2: rows travel back with the transposed sizes
  recv_rows = [[None] * n for _ in range(world)]
  for c in range(n):
    back = [[sizes[c][i][r] for i in range(world)] for r in range(world)]
    vals = [send_rows[r][c] for r in range(world)]
    if wire16:
      vals = hb.distribute.cast_n(vals, torch.float16)
    got = fake_alltoallv(vals, back)
    if wire16:
      got = hb.distribute.cast_n(got, torch.float32)
    for r in range(world):
      recv_rows[r][c] = got[r]
  for r in range(world):
    outs = drivers[r].stitch(sts[r], recv_rows[r])
    eff = tables
    if wire16:
      eff = [oracle.cast_f16_to_f32(oracle.cast_f32_to_f16(t)) for t in tables]
    want = oracle.group_lookup_fwd(eff, ids[r], splits[r], rows, combiners)
    for c in range(n):
      np.testing.assert_equal(outs[c].cpu().numpy(), want[c])


@pytest.mark.parametrize('groups,inline', [(0, 0), (2, 0), (3, 0), (0, 1), (2, 1)])
def test_sharded_call_through_rccl_world1(hbk_option, groups, inline):
  # one rank pipelines ONE column group by default (nothing on the wire to hide); the option
  # forces the multi-group pipeline of W > 1 through the same calls; inline: the RCCL group is
  # enqueued on the compute stream (sharded_copy_self: the own slice really goes through it)
  hbk_option('sharded_groups', groups)
  hbk_option('sharded_inline', inline)
  if inline:
    hbk_option('sharded_copy_self', 1)
  rng = np.random.RandomState(7)
  coll = hb.distribute.Collective(world_size=1, rank=0)
  try:
    tables = [rng.uniform(-1, 1, size=(5000, 16)).astype(np.float32) for _ in range(3)]
    ids = [rng.randint(0, 2**40, size=4000).astype(np.int64) for _ in range(3)]
    drv = ShardedGroupLookup([dev(t) for t in tables], coll, buckets=[5000] * 3)
    outs = drv([dev(i) for i in ids])
    torch.cuda.synchronize()
    want = oracle.group_lookup_fwd(tables, ids, [None] * 3, [5000] * 3, ['sum'] * 3)
    for o, w in zip(outs, want):
      np.testing.assert_equal(o.cpu().numpy(), w)
    phases = drv.last_host_us()   # enqueue 1-2 / wait for the sizes / enqueue the rest
    assert len(phases) == 3 and all(0.0 <= v < 5e6 for v in phases) and sum(phases) > 0.0
  finally:
    coll.close()


def test_sharded_hot_rows_follow_the_data_world1():
  """ShardedGroupLookup(hot_rows='auto') through RCCL at world size 1: the owner gather's hot-row
  staging follows the distinct rows / ids of the last backward; outputs stay bit-exact."""
  rng = np.random.RandomState(17)
  coll = hb.distribute.Collective(world_size=1, rank=0)
  try:
    rows, batch = 4000, 5000
    tables = [rng.uniform(-1, 1, size=(rows, d)).astype(np.float32) for d in (128, 16)]
    drv = ShardedGroupLookup([dev(t) for t in tables], coll, buckets=[rows] * 2, hot_rows='auto')
    skew = [(rng.zipf(1.3, size=batch) % rows).astype(np.int64) for _ in tables]
    flat = [rng.randint(0, rows, size=batch).astype(np.int64) for _ in tables]
    g = [dev(rng.randn(batch, t.shape[1]).astype(np.float32)) for t in tables]

    def forward(ids):
      outs = drv([dev(i) for i in ids])
      torch.cuda.synchronize()
      for o, t, i in zip(outs, tables, ids):
        np.testing.assert_equal(o.cpu().numpy(), t[i])
    forward(skew)
    assert drv.hot_rows == [False, False]
    drv.backward(g)
    torch.cuda.synchronize()
    forward(skew)
    assert drv.hot_rows == [True, True]
    drv.backward(g)
    torch.cuda.synchronize()
    forward(flat)
    drv.backward(g)
    torch.cuda.synchronize()
    forward(flat)
    assert drv.hot_rows == [False, False]
    drv.close()
  finally:
    coll.close()


@pytest.mark.parametrize('world', [1, 2, 8])
def test_sharded_backward_equals_dense_scatter(world):
  rng = np.random.RandomState(200 + world)
  dims, rows = [16, 8, 32], [5003, 300, 64]
  combiners = ['sum', 'mean', 'sqrtn']
  n = len(dims)
  tables = [rng.uniform(-1, 1, size=(rows[c], dims[c])).astype(np.float32) for c in range(n)]
  drivers, ids, splits, grads = [], [], [], []
  for r in range(world):
    drivers.append(ShardedGroupLookup([dev(t[r::world]) for t in tables], None, buckets=rows,
                                      combiners=combiners, world_size=world))
    rid, rsp, rg = [], [], []
    for c in range(n):
      if c == 0:
        sp = None
        k = int(rng.randint(1, 2000))
      else:
        lens = rng.poisson(3, size=rng.randint(1, 300)).clip(0, 12)
        sp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        k = int(sp[-1])
      rsp.append(sp)
      rid.append(rng.randint(0, 2**40, size=k).astype(np.int64))
      rg.append(rng.randn(k if sp is None else sp.size - 1, dims[c]).astype(np.float32))
    ids.append(rid)
    splits.append(rsp)
    grads.append(rg)
  sts = [drivers[r].partition([dev(i) for i in ids[r]],
                              [None if s is None else dev(s) for s in splits[r]])
         for r in range(world)]
  sizes = [[sts[r].send_sizes[c].cpu().tolist() for r in range(world)] for c in range(n)]
  for r in range(world):
    got = [fake_alltoallv([sts[q].send_ids[c] for q in range(world)], sizes[c])[r]
           for c in range(n)]
    drivers[r].owner_gather(sts[r], got)
  # B1 on every rank, then the reverse exchange: requester r sends sizes[c][r] back
  send = [drivers[r].stitch_bwd(sts[r], [dev(g) for g in grads[r]]) for r in range(world)]
  slices = []
  for r in range(world):
    recv = [fake_alltoallv([send[q][c] for q in range(world)], sizes[c])[r] for c in range(n)]
    slices.append(drivers[r].owner_bwd(sts[r], recv))
  for c in range(n):
    dense, mag = world_grad_sums(rows[c], dims[c], [(ids[r][c], grads[r][c], splits[r][c], combiners[c])
                                                    for r in range(world)])
    got = np.zeros_like(dense)
    for r in range(world):
      urows, grows, nu = slices[r][c]
      k = int(nu.item())
      lr = urows.cpu().numpy()[:k]
      assert len(set(lr.tolist())) == k                 # deduplicated on the owner
      got[lr * world + r] += grows.cpu().numpy()[:k]
    assert_sums_close(got, dense, mag)


def test_sharded_backward_through_rccl_world1_with_apply():
  rng = np.random.RandomState(9)
  coll = hb.distribute.Collective(world_size=1, rank=0)
  try:
    table = rng.uniform(-1, 1, size=(3000, 16)).astype(np.float32)
    ids = rng.randint(0, 2**40, size=5000).astype(np.int64)
    g = rng.randn(5000, 16).astype(np.float32)
    t_dev = dev(table.copy())
    drv = ShardedGroupLookup([t_dev], coll, buckets=[3000])
    drv([dev(ids)])
    drv.backward([dev(g)], apply_lr=0.1)
    torch.cuda.synchronize()
    want, mag = dense_sums(table.shape, ids % 3000, g)
    assert_sums_close(t_dev.cpu().numpy(), table.astype(np.float64) - 0.1 * want,
                      np.abs(table) + 0.1 * mag)
  finally:
    coll.close()


@pytest.mark.parametrize('world,wire16,id64,inline,pack_early', [
    (2, False, False, 0, 1), (4, True, False, 0, 1), (8, False, False, 0, 1), (4, False, True, 0, 1),
    (4, False, False, 1, 1), (8, True, False, 1, 1), (2, False, True, 1, 1),
    (4, 'unfused', False, 0, 1), (1, True, False, 0, 1),
    (2, False, False, 0, 0), (8, True, False, 1, 0), (4, False, True, 0, 0)])
def test_cxx_driver_multi_rank_in_process_world(hbk_option, world, wire16, id64, inline,
                                                pack_early):
  """hbk_sharded_lookup_fwd/_bwd (the code that runs at 8 GPUs) with W ranks as host threads
  of one process on one GPU; only the transport differs from production (device copies
  instead of RCCL).  Forward == unsharded oracle lookup, backward == dense scatter-add.
  inline: the exchanges are enqueued on the compute stream itself (option sharded_inline: one
  column group, no hops to the communicator's stream) instead of pipelined beside it."""
  import threading
  if id64:   # ids travel as int32 by default (all buckets < 2^31); this keeps int64 on the wire
    hbk_option('sharded_id64', 1)
  hbk_option('sharded_inline', inline)
  # the ids are packed peer-major behind the partition from the sizes on the device (default), or
  # by the forward once the host has the sizes
  hbk_option('sharded_pack_early', pack_early)
  # fp16 wire: by default the owner gather writes fp16 rows and the stitch reads them (no cast
  # passes, the own slice stays in place); 'unfused' keeps the two casts through a wire workspace
  hbk_option('sharded_wire_fused', 0 if wire16 == 'unfused' else 1)
  rng = np.random.RandomState(300 + world)
  # world 4 includes a dim that is not a multiple of 4 floats: unpack path instead of the
  # in-place segmented stitch
  dims = [16, 6, 128, 4] if world == 4 else [16, 8, 128, 4]
  rows = [50021, 211, 3000, 64]
  combiners = ['sum', 'mean', 'sqrtn', 'sum']
  n = len(dims)
  tables = [rng.uniform(-1, 1, size=(rows[c], dims[c])).astype(np.float32) for c in range(n)]
  ids, splits, grads = [], [], []
  for r in range(world):
    rid, rsp, rg = [], [], []
    for c in range(n):
      if c % 2 == 0:
        sp, k = None, int(rng.randint(0, 3000))
      else:
        lens = rng.poisson(3, size=rng.randint(1, 400)).clip(0, 12)
        sp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        k = int(sp[-1])
      rsp.append(sp)
      rid.append(rng.randint(0, 2**40, size=k).astype(np.int64))
      rg.append(rng.randn(k if sp is None else sp.size - 1, dims[c]).astype(np.float32))
    ids.append(rid)
    splits.append(rsp)
    grads.append(rg)
  comms = hb.distribute.Collective.local_world(world)
  shards = [[dev(t[r::world].copy()) for t in tables] for r in range(world)]
  results, errors = [None] * world, []

  def run(r):
    try:
      with torch.cuda.stream(torch.cuda.Stream()):
        drv = ShardedGroupLookup(shards[r], comms[r], buckets=rows, combiners=combiners,
                                 wire_dtype=torch.float16 if wire16 else None)
        for _ in range(2):   # second step reuses the grown buffers
          outs = drv([dev(i) for i in ids[r]], [None if s is None else dev(s) for s in splits[r]])
          slices = drv.backward([dev(g) for g in grads[r]], apply_lr=0.0)
        torch.cuda.current_stream().synchronize()
        results[r] = ([o.cpu().numpy() for o in outs],
                      [(u.cpu().numpy()[:int(k.item())], g.cpu().numpy()[:int(k.item())])
                       for u, g, k in slices])
        drv.close()
    except Exception as e:  # pylint: disable=broad-except
      errors.append((r, repr(e)))

  threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
  for t in threads:
    t.start()
  for t in threads:
    t.join(timeout=45)
  assert not errors, errors
  assert all(x is not None for x in results)
  eff = tables
  rel, floor = 1e-5, 1e-6
  if wire16:
    eff = [oracle.cast_f16_to_f32(oracle.cast_f32_to_f16(t)) for t in tables]
    # gradients travel as fp16 too: every rank's contribution to a row is rounded on its own
    rel, floor = WIRE16_REL, WIRE16_FLOOR
  for r in range(world):
    want = oracle.group_lookup_fwd(eff, ids[r], splits[r], rows, combiners)
    for c in range(n):
      np.testing.assert_equal(results[r][0][c], want[c])
  for c in range(n):
    dense, mag = world_grad_sums(rows[c], dims[c], [(ids[r][c], grads[r][c], splits[r][c], combiners[c])
                                                    for r in range(world)])
    got = np.zeros_like(dense)
    for r in range(world):
      lr_, g_ = results[r][1][c]
      assert len(set(lr_.tolist())) == len(lr_)
      got[lr_ * world + r] += g_
    assert_sums_close(got, dense, mag, rel=rel, floor=floor)
  for cm in comms:
    cm.close()


@pytest.mark.parametrize('world', [1, 2, 4])
def test_cxx_driver_deterministic_backward_equals_the_unsharded_in_order_sum(hbk_option, world):
  """bwd_deterministic through the sharded step (round 6): the owner sums the gradient rows it
  receives in the order they arrive in the exchange buffer -- requester 0's ids in id order, then
  requester 1's, ... (the partition is stable) -- so the IndexedSlices of the W shards are BIT-EQUAL
  to the sequential fp32 sum over the ranks' batches concatenated in rank order, i.e. to what ONE
  unsharded table would accumulate; the fused SGD step lands bit-equal to the oracle's apply of
  those sums.  Ragged mean / sqrtn and scalar columns, fp32 wire, no requester-side dedup (which
  sums a requester's duplicates first: another association, by design)."""
  import threading
  hbk_option('bwd_deterministic', 1)
  rng = np.random.RandomState(1700 + world)
  dims, rows = [16, 8, 32, 4], [5003, 300, 64, 1000]
  combiners = ['sum', 'mean', 'sqrtn', 'sum']
  n = len(dims)
  tables = [rng.uniform(-1, 1, size=(rows[c], dims[c])).astype(np.float32) for c in range(n)]
  ids, splits, grads = [], [], []
  for r in range(world):
    rid, rsp, rg = [], [], []
    for c in range(n):
      if c in (0, 3):
        sp, k = None, 1500
      else:
        lens = rng.poisson(3, size=400).clip(0, 12)
        sp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        k = int(sp[-1])
      rsp.append(sp)
      rid.append((rng.zipf(1.3, size=k) % rows[c]).astype(np.int64) if c == 3
                 else rng.randint(0, 2**40, size=k).astype(np.int64))
      rg.append(rng.randn(k if sp is None else sp.size - 1, dims[c]).astype(np.float32))
    ids.append(rid)
    splits.append(rsp)
    grads.append(rg)
  comms = hb.distribute.Collective.local_world(world)
  shards = [[dev(t[r::world].copy()) for t in tables] for r in range(world)]
  results, errors = [None] * world, []
  lr = 0.05

  def run(r):
    try:
      with torch.cuda.stream(torch.cuda.Stream()):
        drv = ShardedGroupLookup(shards[r], comms[r], buckets=rows, combiners=combiners)
        d_ids = [dev(i) for i in ids[r]]
        d_sp = [None if s is None else dev(s) for s in splits[r]]
        drv(d_ids, d_sp)
        sl = drv.backward([dev(g) for g in grads[r]], apply_lr=lr)
        torch.cuda.current_stream().synchronize()
        results[r] = [(u.cpu().numpy()[:int(k.item())], g.cpu().numpy()[:int(k.item())])
                      for u, g, k in sl]
        drv.close()
    except Exception:  # pylint: disable=broad-except
      import traceback
      errors.append((r, traceback.format_exc()))

  threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
  for t in threads:
    t.start()
  for t in threads:
    t.join(timeout=60)
  for cm in comms:
    cm.close()
  assert not errors, errors
  for c in range(n):
    # the ranks' per-id gradients, concatenated in rank order
    g_id = np.concatenate([oracle.segment_combine_grad(
        grads[r][c], splits[r][c] if splits[r][c] is not None
        else np.arange(ids[r][c].size + 1, dtype=np.int32), combiners[c]) for r in range(world)])
    row = np.concatenate([ids[r][c] % rows[c] for r in range(world)])
    for owner in range(world):
      mine = row % world == owner
      local = row[mine] // world
      uniq = np.unique(local)
      want = oracle.unsorted_segment_sum(g_id[mine], np.searchsorted(uniq, local).astype(np.int32),
                                         uniq.size)
      got_rows, got_sums = results[owner][c]
      np.testing.assert_equal(got_rows, uniq, err_msg=f'column {c}, owner {owner}')
      np.testing.assert_equal(got_sums, want, err_msg=f'column {c}, owner {owner}')
      ref = tables[c][owner::world].copy()
      oracle.sparse_sgd_apply(ref, uniq, want, lr)
      np.testing.assert_equal(shards[owner][c].cpu().numpy(), ref)


@pytest.mark.parametrize('world,inline,id64,pack_early,block,refuse', [
    (1, 1, False, 1, False, -1), (2, 1, False, 1, False, -1), (4, 1, False, 1, True, -1),
    (8, 1, False, 1, False, -1), (4, 0, False, 1, False, -1), (8, 0, True, 1, True, -1),
    (2, 1, True, 0, False, -1), (4, 0, False, 0, False, -1), (3, 1, False, 1, True, -1),
    (2, 1, False, 1, False, 1), (4, 0, False, 1, True, 2)])
def test_cxx_driver_p2p_form_in_process_world(hbk_option, world, inline, id64, pack_early, block,
                                              refuse):
  """The p2p form of the sharded forward (round 5, hbk_sharded_p2p_bind): every rank registers its
  output tensors once, a step sends (id, output row) pairs and the owner gather stores each row
  straight into the requester's output -- no reply buffer, no rows exchange, no stitch.  W ranks as
  host threads of one process (same address space: the peer pointers are the tensors' own; across
  processes they come from hipIpcOpenMemHandle, tests/test_gpu_multi.py).  Forward bit-equal to the
  unsharded oracle over several steps with other ids, outputs as separate tensors or as the column
  blocks of ONE [batch, sum dims] tensor (block), dims with 16-byte and 4-byte chunks; the backward
  of such a step == dense scatter-add and the fused SGD step lands where the dense gradient says;
  inline and communicator-stream exchanges, int32 / int64 ids on the wire, early / late id pack.
  refuse >= 0 (round 6): that rank's bind finds a peer it cannot map (the answer of a driver without
  hipIpc* support, injected by option sharded_p2p_test_refuse) -- EVERY rank's bind then returns
  False (they agree on the minimum), the plans keep the exchange form and the same checks hold."""
  import threading
  hbk_option('sharded_inline', inline)
  hbk_option('sharded_p2p_test_refuse', refuse)
  hbk_option('sharded_pack_early', pack_early)
  if id64:
    hbk_option('sharded_id64', 1)
  rng = np.random.RandomState(900 + world)
  # (a dim-6 column -- 4-byte chunks -- only with separate outputs: inside one wide tensor it would
  # take the rows of the wide columns off the 16-byte boundaries their chunks need)
  dims = [16, 128, 4, 32, 8 if block else 6]
  rows = [50021, 3000, 64, 100003, 211]
  n = len(dims)
  batch = 1500
  tables = [rng.uniform(-1, 1, size=(rows[c], dims[c])).astype(np.float32) for c in range(n)]
  steps = 3
  ids = [[[rng.randint(0, 2**40, size=batch).astype(np.int64) for _ in range(n)]
          for _ in range(world)] for _ in range(steps)]
  grads = [[rng.randn(batch, dims[c]).astype(np.float32) for c in range(n)] for _ in range(world)]
  comms = hb.distribute.Collective.local_world(world)
  shards = [[dev(t[r::world].copy()) for t in tables] for r in range(world)]
  results, errors = [None] * world, []
  lr = 0.05

  def run(r):
    try:
      with torch.cuda.stream(torch.cuda.Stream()):
        drv = ShardedGroupLookup(shards[r], comms[r], buckets=rows)
        if block:
          wide = torch.full((batch, sum(dims)), float('nan'), device=DEV)
          outs, at = [], 0
          for d in dims:
            outs.append(wide[:, at:at + d])
            at += d
        else:
          outs = [torch.full((batch, d), float('nan'), device=DEV) for d in dims]
        assert drv.p2p_bind(outs) is (refuse < 0)
        fwd = []
        for st in range(steps):
          got = drv([dev(i) for i in ids[st][r]], None, outs)
          assert all(g.data_ptr() == o.data_ptr() for g, o in zip(got, outs))
          torch.cuda.current_stream().synchronize()
          # (peers may still be writing THEIR outputs; this rank's are complete behind its stream)
          fwd.append([o.cpu().numpy().copy() for o in outs])
        slices = drv.backward([dev(g) for g in grads[r]], apply_lr=0.0)
        torch.cuda.current_stream().synchronize()
        sl = [(u.cpu().numpy()[:int(k.item())], g.cpu().numpy()[:int(k.item())])
              for u, g, k in slices]
        drv([dev(i) for i in ids[steps - 1][r]], None, outs)
        drv.backward([dev(g) for g in grads[r]], apply_lr=lr, emit=False)
        torch.cuda.current_stream().synchronize()
        # a ragged step on a plan with registered outputs is refused, as are other outputs
        # (every rank fails before its first exchange: nobody is left waiting)
        if refuse < 0:
          with pytest.raises(_lib.InvalidArgumentError):
            drv([dev(i) for i in ids[0][r]], None, [torch.empty_like(o) for o in outs])
        results[r] = (fwd, sl)
        drv.close()
    except Exception as e:  # pylint: disable=broad-except
      import traceback
      errors.append((r, traceback.format_exc()))

  threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
  for t in threads:
    t.start()
  for t in threads:
    t.join(timeout=60)
  assert not errors, errors
  assert all(x is not None for x in results)
  for st in range(steps):
    for r in range(world):
      want = oracle.group_lookup_fwd(tables, ids[st][r], [None] * n, rows, ['sum'] * n)
      for c in range(n):
        np.testing.assert_equal(results[r][0][st][c], want[c])
  for c in range(n):
    dense, mag = world_grad_sums(rows[c], dims[c], [(ids[steps - 1][r][c], grads[r][c], None, 'sum')
                                                    for r in range(world)])
    got = np.zeros_like(dense)
    for r in range(world):
      lr_, g_ = results[r][1][c]
      assert len(set(lr_.tolist())) == len(lr_)
      got[lr_ * world + r] += g_
    assert_sums_close(got, dense, mag)
    for r in range(world):
      assert_sums_close(shards[r][c].cpu().numpy(),
                        (tables[c].astype(np.float64) - lr * dense)[r::world],
                        (np.abs(tables[c]) + lr * mag)[r::world])
  for cm in comms:
    cm.close()


@pytest.mark.parametrize('world,inline,p2p', [(1, 0, False), (2, 0, False), (4, 1, False),
                                              (4, 0, True), (8, 0, False)])
def test_pipelined_lookup_two_plans_in_process_world(hbk_option, world, inline, p2p):
  """hb.embedding.PipelinedLookup (round 5): two plans over the same shards and the same
  communicator, begin(step i + 1) enqueued before end(step i) (hbk_sharded_lookup_fwd_begin /
  _end), each plan on its own compute stream -- the ids of step i + 1 travel ahead of the rows of
  step i.  Every step's outputs equal the unsharded oracle, ragged and scalar columns, W ranks as
  host threads; also with the p2p form (scalar columns) and with inline exchanges."""
  import threading
  hbk_option('sharded_inline', inline)
  hbk_option('sharded_groups', 1)
  rng = np.random.RandomState(1200 + world)
  dims = [16, 8, 128, 4]
  rows = [50021, 211, 3000, 64]
  combiners = ['sum'] * 4 if p2p else ['sum', 'mean', 'sqrtn', 'sum']
  n = len(dims)
  tables = [rng.uniform(-1, 1, size=(rows[c], dims[c])).astype(np.float32) for c in range(n)]
  steps = 5
  ids, splits = [], []
  for st in range(steps):
    sids, ssp = [], []
    for r in range(world):
      rid, rsp = [], []
      for c in range(n):
        if c % 2 == 0 or p2p:
          sp, k = None, 1200
        else:
          lens = rng.poisson(3, size=300).clip(0, 12)
          sp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
          k = int(sp[-1])
        rsp.append(sp)
        rid.append(rng.randint(0, 2**40, size=k).astype(np.int64))
      sids.append(rid)
      ssp.append(rsp)
    ids.append(sids)
    splits.append(ssp)
  comms = hb.distribute.Collective.local_world(world)
  shards = [[dev(t[r::world].copy()) for t in tables] for r in range(world)]
  results, errors = [None] * world, []

  def run(r):
    try:
      with torch.cuda.stream(torch.cuda.Stream()):
        plans = [ShardedGroupLookup(shards[r], comms[r], buckets=rows, combiners=combiners)
                 for _ in range(2)]
        # (inline = 1: the plans keep their inline exchanges -- correct, but nothing overlaps)
        pipe = hb.embedding.PipelinedLookup(plans, stream_exchanges=not inline)
        outs = [[torch.full((1200 if (c % 2 == 0 or p2p) else 300, dims[c]), float('nan'), device=DEV)
                 for c in range(n)] for _ in range(2)]
        if p2p:
          for k in range(2):
            with torch.cuda.stream(pipe.streams[k]):
              assert plans[k].p2p_bind(outs[k]) is True
        got = []
        for st in range(steps):
          k = pipe.next_plan()
          b = pipe.bind(k, [dev(i) for i in ids[st][r]],
                        [None if s is None else dev(s) for s in splits[st][r]], outs[k])
          done = pipe.step(b)
          if done is not None:
            torch.cuda.current_stream().synchronize()
            got.append([o.cpu().numpy().copy() for o in done])
        done = pipe.flush()
        torch.cuda.current_stream().synchronize()
        got.append([o.cpu().numpy().copy() for o in done])
        results[r] = got
        pipe.close()
    except Exception:  # pylint: disable=broad-except
      import traceback
      errors.append((r, traceback.format_exc()))

  threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
  for t in threads:
    t.start()
  for t in threads:
    t.join(timeout=90)
  assert not errors, errors
  assert all(x is not None and len(x) == steps for x in results)
  for st in range(steps):
    for r in range(world):
      want = oracle.group_lookup_fwd(tables, ids[st][r], splits[st][r], rows, combiners)
      for c in range(n):
        np.testing.assert_equal(results[r][st][c], want[c], err_msg=f'step {st} rank {r} column {c}')
  for cm in comms:
    cm.close()


def test_sharded_p2p_through_rccl_world1_and_refusals(hbk_option):
  """The p2p form over a real RCCL communicator of one rank (the own slice stays in place: the owner
  gather reads the ids and slots where the pack left them), both step forms; what the form does
  not take is refused: requester-side dedup and the fp16 wire at bind, ragged ids at the step."""
  rng = np.random.RandomState(77)
  coll = hb.distribute.Collective(world_size=1, rank=0)
  try:
    tables = [rng.uniform(-1, 1, size=(5000, d)).astype(np.float32) for d in (16, 128, 5)]
    t_dev = [dev(t) for t in tables]
    for inline, copy_self in ((1, 0), (0, 0), (1, 1), (0, 1)):
      hbk_option('sharded_inline', inline)
      hbk_option('sharded_copy_self', copy_self)
      drv = ShardedGroupLookup(t_dev, coll, buckets=[5000] * 3)
      store = [torch.empty(4001, t.shape[1], device=DEV) for t in tables]
      outs = [x[:4000] for x in store]
      assert drv.p2p_bind(outs) is True
      for step in range(3):
        ids = [rng.randint(0, 2**40, size=4000).astype(np.int64) for _ in range(3)]
        drv([dev(i) for i in ids], None, outs)
        torch.cuda.synchronize()
        want = oracle.group_lookup_fwd(tables, ids, [None] * 3, [5000] * 3, ['sum'] * 3)
        for o, w in zip(outs, want):
          np.testing.assert_equal(o.cpu().numpy(), w)
      # a batch larger than the registered outputs is refused at the step (remote owners would
      # store outside the tensor): the bind carries the rows of every output
      # (the same memory seen as one row more: the host layer's shape check passes, the library's
      # own bound is what answers)
      big = [dev(rng.randint(0, 2**40, size=4001).astype(np.int64)) for _ in range(3)]
      with pytest.raises(_lib.InvalidArgumentError, match='registered output of 4000 rows'):
        drv(big, None, [store[c][:4001] for c in range(3)])
      sp = dev(np.arange(0, 4001, 2, dtype=np.int32))
      with pytest.raises(_lib.InvalidArgumentError, match='ragged'):
        drv([dev(i) for i in ids], [sp, None, None],
            [torch.empty(2000, 16, device=DEV), outs[1], outs[2]])
      drv.p2p_unbind()                       # back to the exchange form: ragged ids are fine again
      got = drv([dev(i) for i in ids], [sp, None, None])
      torch.cuda.synchronize()
      want = oracle.group_lookup_fwd(tables, ids, [np.arange(0, 4001, 2, dtype=np.int32), None, None],
                                     [5000] * 3, ['sum'] * 3)
      np.testing.assert_equal(got[0].cpu().numpy(), want[0])
      drv.close()
    # a driver that cannot map a peer's memory (hipIpcGetMemHandle / hipIpcOpenMemHandle refused),
    # injected: the bind answers False, the plan keeps the exchange form and its results stay right
    hbk_option('sharded_p2p_test_refuse', 0)
    drv = ShardedGroupLookup(t_dev, coll, buckets=[5000] * 3)
    outs = [torch.empty(4000, t.shape[1], device=DEV) for t in tables]
    assert drv.p2p_bind(outs) is False
    ids = [rng.randint(0, 2**40, size=4000).astype(np.int64) for _ in range(3)]
    drv([dev(i) for i in ids], None, outs)
    torch.cuda.synchronize()
    for o, w in zip(outs, oracle.group_lookup_fwd(tables, ids, [None] * 3, [5000] * 3, ['sum'] * 3)):
      np.testing.assert_equal(o.cpu().numpy(), w)
    drv.close()
    hbk_option('sharded_p2p_test_refuse', -1)
    drv = ShardedGroupLookup(t_dev, coll, buckets=[5000] * 3, dedup=True)
    with pytest.raises(_lib.InvalidArgumentError, match='dedup'):
      drv.p2p_bind([torch.empty(10, t.shape[1], device=DEV) for t in tables])
    drv.close()
    drv = ShardedGroupLookup(t_dev, coll, buckets=[5000] * 3, wire_dtype=torch.float16)
    with pytest.raises(_lib.InvalidArgumentError, match='fp16'):
      drv.p2p_bind([torch.empty(10, t.shape[1], device=DEV) for t in tables])
    drv.close()
  finally:
    coll.close()


@pytest.mark.parametrize('world,kind,groups,pack_early,wire16', [
    (1, 'zipf', 0, 1, False), (2, 'zipf', 0, 1, False), (4, 'zipf', 2, 1, False),
    (8, 'zipf', 0, 1, False), (2, 'uniform', 0, 1, False), (8, 'uniform', 3, 1, False),
    (4, 'zipf', 0, 0, False), (4, 'zipf', 2, 1, True), (2, 'few', 0, 1, False)])
def test_cxx_driver_requester_dedup_in_process_world(hbk_option, world, kind, groups, pack_early,
                                                     wire16):
  """Requester-side dedup (hbk_sharded_column_t.dedup; the reference's tutorials: tf.unique ->
  lookup -> tf.gather, docs/tutorial/ranking/data.py:180-182): every distinct id of a flagged
  column goes on the wire once.  W ranks as host threads on one GPU: forward bit-equal to the
  unsharded oracle (Zipf(1.2), uniform and three-distinct-ids batches, ragged and scalar columns,
  deduplicated and plain columns side by side), the rows that travel are the DISTINCT ids',
  backward == dense scatter-add with the duplicates summed on the requester, and the fused SGD
  step leaves the shards where the dense gradient says."""
  import threading
  hbk_option('sharded_groups', groups)
  hbk_option('sharded_pack_early', pack_early)
  # (column groups pipelined beside the exchanges on the communicator's stream when groups are
  # asked for; the shipped default since round 5 is inline)
  hbk_option('sharded_inline', 0 if groups else 1)
  rng = np.random.RandomState(500 + world)
  dims = [16, 8, 128, 4, 32]
  rows = [50021, 211, 3000, 64, 100003]
  combiners = ['sum', 'mean', 'sqrtn', 'sum', 'mean']
  dedup = [True, True, True, False, True]
  n = len(dims)
  tables = [rng.uniform(-1, 1, size=(rows[c], dims[c])).astype(np.float32) for c in range(n)]

  def draw(k, r_):
    if kind == 'zipf':
      return ((rng.zipf(1.2, size=k) * 7919) % (1 << 40)).astype(np.int64)
    if kind == 'few':
      return rng.choice(np.array([5, 12345678901, 77], np.int64), size=k)
    return rng.randint(0, 2**40, size=k).astype(np.int64)
  ids, splits, grads = [], [], []
  for r in range(world):
    rid, rsp, rg = [], [], []
    for c in range(n):
      if c % 2 == 0:
        sp, k = None, int(rng.randint(0, 3000)) if c else 2500
      else:
        lens = rng.poisson(3, size=rng.randint(1, 400)).clip(0, 12)
        sp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        k = int(sp[-1])
      rsp.append(sp)
      rid.append(draw(k, rows[c]))
      rg.append(rng.randn(k if sp is None else sp.size - 1, dims[c]).astype(np.float32))
    ids.append(rid)
    splits.append(rsp)
    grads.append(rg)
  comms = hb.distribute.Collective.local_world(world)
  shards = [[dev(t[r::world].copy()) for t in tables] for r in range(world)]
  results, errors = [None] * world, []
  lr = 0.25

  def run(r):
    try:
      with torch.cuda.stream(torch.cuda.Stream()):
        drv = ShardedGroupLookup(shards[r], comms[r], buckets=rows, combiners=combiners,
                                 dedup=dedup, wire_dtype=torch.float16 if wire16 else None)
        d_ids = [dev(i) for i in ids[r]]
        d_sp = [None if s is None else dev(s) for s in splits[r]]
        outs = drv(d_ids, d_sp)
        slices = drv.backward([dev(g) for g in grads[r]], apply_lr=0.0)
        torch.cuda.current_stream().synchronize()
        first = [o.cpu().numpy().copy() for o in outs]
        owned = [int(drv._lib.hbk_sharded_owned_ids(drv._plan(), c)) for c in range(n)]
        # second step: grown buffers reused, this time with the fused step
        outs = drv(d_ids, d_sp)
        slices = drv.backward([dev(g) for g in grads[r]], apply_lr=lr)
        torch.cuda.current_stream().synchronize()
        results[r] = ([o.cpu().numpy() for o in outs], first,
                      [(u.cpu().numpy()[:int(k.item())], g.cpu().numpy()[:int(k.item())])
                       for u, g, k in slices], owned)
        drv.close()
    except Exception as e:  # pylint: disable=broad-except
      errors.append((r, repr(e)))

  threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
  for t in threads:
    t.start()
  for t in threads:
    t.join(timeout=60)
  assert not errors, errors
  assert all(x is not None for x in results)
  eff = tables
  rel, floor = 1e-5, 1e-6
  if wire16:
    eff = [oracle.cast_f16_to_f32(oracle.cast_f32_to_f16(t)) for t in tables]
    rel, floor = WIRE16_REL, WIRE16_FLOOR
  for r in range(world):
    want = oracle.group_lookup_fwd(eff, ids[r], splits[r], rows, combiners)
    for c in range(n):
      np.testing.assert_equal(results[r][1][c], want[c])   # step 1
      np.testing.assert_equal(results[r][0][c], want[c])   # step 2 (same tables: lr only in its backward)
  # what the owners were asked for: distinct ids per requester where deduplicated
  for c in range(n):
    for owner in range(world):
      asked = 0
      for r in range(world):
        b = ids[r][c] % rows[c]
        mine = b[b % world == owner]
        asked += np.unique(mine).size if dedup[c] else mine.size
      assert results[owner][3][c] == asked, (c, owner)
  for c in range(n):
    dense, mag = world_grad_sums(rows[c], dims[c], [(ids[r][c], grads[r][c], splits[r][c], combiners[c])
                                                    for r in range(world)])
    got = np.zeros_like(dense)
    for r in range(world):
      lr_, g_ = results[r][2][c]
      assert len(set(lr_.tolist())) == len(lr_)
      got[lr_ * world + r] += g_
    assert_sums_close(got, dense, mag, rel=rel, floor=floor)
    # the fused step of the second backward
    for r in range(world):
      assert_sums_close(shards[r][c].cpu().numpy(),
                        tables[c][r::world].astype(np.float64) - lr * dense[r::world],
                        (np.abs(tables[c]) + lr * mag)[r::world], rel=rel, floor=floor)
  for cm in comms:
    cm.close()


# ----------------------------------------------------------------------------------
# feature columns: one dense [batch, sum of dims] block written / differentiated in place
def _dense_case(rng, world):
  cols = [hb.feature_column.EmbeddingColumn('a', 50021, 16, 'sum'),
          hb.feature_column.EmbeddingColumn('b', 37, 4, 'mean'),       # small: stays replicated
          # wide, one id per sample, hinted as skewed: the hot-row tiles (3000 rows / 300 samples:
          # repeats inside a tile), replicated or through the sharded plan's owner gather
          hb.feature_column.EmbeddingColumn('c', 3000, 128, 'sqrtn', hot_rows=True),
          hb.feature_column.EmbeddingColumn('d', 977, 6, 'mean')]
  batch = 300
  tables = [rng.uniform(-1, 1, size=(c.num_buckets, c.dimension)).astype(np.float32)
            for c in cols]
  feats, grads = [], []
  for _ in range(world):
    f = {}
    for k, c in enumerate(cols):
      if k % 2 == 0:
        f[c.key] = rng.randint(0, 2**40, size=batch).astype(np.int64)
      else:
        lens = rng.poisson(3, size=batch).clip(0, 9)
        sp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        f[c.key] = (rng.randint(0, 2**40, size=int(sp[-1])).astype(np.int64), sp)
    feats.append(f)
    grads.append(rng.randn(batch, sum(c.dimension for c in cols)).astype(np.float32))
  return cols, tables, feats, grads, batch


def _dev_feats(f):
  return {k: (tuple(dev(x) for x in v) if isinstance(v, tuple) else dev(v)) for k, v in f.items()}


def _want_dense(cols, tables, f):
  ids = [f[c.key][0] if isinstance(f[c.key], tuple) else f[c.key] for c in cols]
  sps = [f[c.key][1] if isinstance(f[c.key], tuple) else None for c in cols]
  outs = oracle.group_lookup_fwd(tables, ids, sps, [c.num_buckets for c in cols],
                                 [c.combiner for c in cols])
  return np.concatenate(outs, axis=1), ids, sps


def test_dense_features_single_gpu():
  rng = np.random.RandomState(41)
  cols, tables, feats, grads, batch = _dense_case(rng, 1)
  layer = hb.feature_column.DenseFeatures(
    cols, DEV, init=lambda c, rows, d: dev(tables[cols.index(c)].copy()))
  out = layer(_dev_feats(feats[0]))
  want, ids, sps = _want_dense(cols, tables, feats[0])
  np.testing.assert_equal(out.cpu().numpy(), want)
  per_col = hb.feature_column.dense_features(_dev_feats(feats[0]), layer)
  assert [tuple(t.shape) for t in per_col] == [(batch, c.dimension) for c in cols]
  res = layer.backward(dev(grads[0]))
  off = 0
  for k, c in enumerate(cols):
    g = grads[0][:, off:off + c.dimension]
    off += c.dimension
    dense, mag = world_grad_sums(c.num_buckets, c.dimension,
                                 [(ids[k], np.ascontiguousarray(g), sps[k], c.combiner)])
    u, gr, nu = res[k]
    n = int(nu.item())
    got = np.zeros_like(dense)
    rows = u.cpu().numpy()[:n]
    assert len(set(rows.tolist())) == n
    got[rows] = gr.cpu().numpy()[:n]
    assert_sums_close(got, dense, mag)


def test_dense_features_sharded_and_replicated_columns_in_process_world():
  import threading
  world = 2
  rng = np.random.RandomState(42)
  cols, tables, feats, grads, batch = _dense_case(rng, world)
  comms = hb.distribute.Collective.local_world(world)
  results, errors = [None] * world, []

  def run(r):
    try:
      with torch.cuda.stream(torch.cuda.Stream()):
        def init(c, rows, d):
          t = tables[cols.index(c)]
          return dev((t[r::world] if rows != c.num_buckets else t).copy())
        layer = hb.feature_column.DenseFeatures(cols, DEV, coll=comms[r], batch_size=batch,
                                                init=init)
        assert layer.sharded == [True, False, True, True]
        f = _dev_feats(feats[r])
        layer.prefetch(f)          # (the loader's hint: partition + size exchange ahead of the step)
        out = layer(f)
        res = layer.backward(dev(grads[r]))
        torch.cuda.current_stream().synchronize()
        results[r] = (out.cpu().numpy(),
                      [(u.cpu().numpy()[:int(k.item())], g.cpu().numpy()[:int(k.item())])
                       for u, g, k in res])
        layer.close()
    except Exception as e:  # pylint: disable=broad-except
      errors.append((r, repr(e)))

  threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
  for t in threads:
    t.start()
  for t in threads:
    t.join(timeout=45)
  assert not errors, errors
  wants = [_want_dense(cols, tables, feats[r]) for r in range(world)]
  for r in range(world):
    np.testing.assert_equal(results[r][0], wants[r][0])
  off = 0
  for k, c in enumerate(cols):
    dense, mag = world_grad_sums(
      c.num_buckets, c.dimension,
      [(wants[r][1][k], np.ascontiguousarray(grads[r][:, off:off + c.dimension]), wants[r][2][k],
        c.combiner) for r in range(world)])
    got = np.zeros_like(dense)
    for r in range(world):
      rows, vals = results[r][1][k]
      # sharded tables report local rows (global = local * W + rank); replicated ones global
      # rows, each rank its own share (summed here = the cross-rank aggregation)
      glob = rows * world + r if k != 1 else rows
      np.add.at(got, glob, vals.astype(np.float64))
    assert_sums_close(got, dense, mag)
    off += c.dimension
  for cm in comms:
    cm.close()


# ----------------------------------------------------------------------------------
# aggregation of replicated gradients (allreduce bucket, allgatherv)
@pytest.mark.parametrize('world', [2, 4])
def test_gradient_aggregation_in_process_world(world):
  import threading
  rng = np.random.RandomState(50 + world)
  shapes = [(1000, 16), (7,), (0,), (333, 5)]
  dense = [[rng.randn(*s).astype(np.float32) for s in shapes] for _ in range(world)]
  ints = [rng.randint(-1000, 1000, size=257).astype(np.int64) for _ in range(world)]
  sp_vals = [rng.randn(10 + 3 * r, 8).astype(np.float32) for r in range(world)]
  sp_idx = [rng.randint(0, 100, size=10 + 3 * r).astype(np.int64) for r in range(world)]
  shard_g = [rng.randn(5, 4).astype(np.float32) for _ in range(world)]
  comms = hb.distribute.Collective.local_world(world)
  results, errors = [None] * world, []

  def run(r):
    try:
      with torch.cuda.stream(torch.cuda.Stream()):
        c = comms[r]
        mx = c.allreduce(dev(ints[r]), reduce_op=c.MAX)
        grads = [dev(x) for x in dense[r]] + [(dev(sp_vals[r]), dev(sp_idx[r])), dev(shard_g[r])]
        agg = hb.distribute.aggregate_gradients(
          grads, c, sharded=[False] * len(shapes) + [False, True])
        torch.cuda.current_stream().synchronize()
        results[r] = (mx.cpu().numpy(), [a.cpu().numpy() for a in agg[:len(shapes)]],
                      (agg[len(shapes)][0].cpu().numpy(), agg[len(shapes)][1].cpu().numpy()),
                      agg[-1].cpu().numpy())
    except Exception as e:  # pylint: disable=broad-except
      errors.append((r, repr(e)))

  threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
  for t in threads:
    t.start()
  for t in threads:
    t.join(timeout=45)
  assert not errors, errors
  want_max = np.max(np.stack(ints), axis=0)
  for r in range(world):
    np.testing.assert_equal(results[r][0], want_max)
    for k in range(len(shapes)):
      acc = dense[0][k].copy()          # the transport sums in rank order, then scales
      for q in range(1, world):
        acc = acc + dense[q][k]
      np.testing.assert_equal(results[r][1][k], acc * np.float32(1.0 / world))
    np.testing.assert_equal(results[r][2][0],
                            np.concatenate(sp_vals) * np.float32(1.0 / world))
    np.testing.assert_equal(results[r][2][1], np.concatenate(sp_idx))
    np.testing.assert_equal(results[r][3], shard_g[r])       # sharded: untouched
  for cm in comms:
    cm.close()


@pytest.mark.parametrize('world', [1, 3, 8])
def test_broadcast_in_process_world_and_rccl(world):
  """HbNcclBroadcast (nccl_broadcast.cc:31-92) = hbk_broadcast: every rank ends with the root's
  tensor, whatever it held; in-process ranks, any root, several dtypes; and through RCCL at world 1."""
  import threading
  rng = np.random.RandomState(61)
  vals = [[rng.randn(1000, 4).astype(np.float32), rng.randint(-5, 5, size=7).astype(np.int64),
           rng.randn(1).astype(np.float32)] for _ in range(world)]
  comms = hb.distribute.Collective.local_world(world)
  results, errors = [None] * world, []

  def run(r):
    try:
      with torch.cuda.stream(torch.cuda.Stream()):
        got = []
        for k, v in enumerate(vals[r]):
          got.append(hb.distribute.broadcast(dev(v), comms[r], root_rank=(k + 1) % world))
        x = dev(vals[r][0])
        hb.distribute.broadcast(x, comms[r], root_rank=0, out=x)     # in place
        torch.cuda.current_stream().synchronize()
        results[r] = [g.cpu().numpy() for g in got] + [x.cpu().numpy()]
    except Exception as e:  # pylint: disable=broad-except
      errors.append((r, repr(e)))

  threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
  for t in threads:
    t.start()
  for t in threads:
    t.join(timeout=45)
  assert not errors, errors
  for r in range(world):
    for k in range(3):
      np.testing.assert_equal(results[r][k], vals[(k + 1) % world][k])
    np.testing.assert_equal(results[r][3], vals[0][0])
  for cm in comms:
    cm.close()
  if world == 1:
    coll = hb.distribute.Collective(world_size=1, rank=0)
    try:
      x = rng.randn(513, 3).astype(np.float32)
      np.testing.assert_equal(hb.distribute.broadcast(dev(x), coll).cpu().numpy(), x)
    finally:
      coll.close()


def test_allreduce_allgather_through_rccl_world1():
  coll = hb.distribute.Collective(world_size=1, rank=0)
  try:
    rng = np.random.RandomState(60)
    xs = [rng.randn(1000, 16).astype(np.float32), rng.randn(3).astype(np.float32)]
    outs = coll.allreduce_n([dev(x) for x in xs], scale=0.5)
    for x, o in zip(xs, outs):
      np.testing.assert_equal(o.cpu().numpy(), x * np.float32(0.5))
    one = coll.allreduce(dev(xs[0]))
    np.testing.assert_equal(one.cpu().numpy(), xs[0])
    g = coll.allgather(dev(xs[0]))
    np.testing.assert_equal(g.cpu().numpy(), xs[0])
    assert hb.distribute.aggregate_gradients([dev(xs[1])], coll)[0] is not None
  finally:
    coll.close()


def test_cxx_driver_empty_ranks_and_columns_in_process_world():
  """Edge cases of the sharded step: a rank with no ids at all, a column nobody looks up, a
  column whose ids all belong to one owner (the other owner gets empty messages), empty ragged
  segments -- forward and backward must still agree with the unsharded oracle."""
  import threading
  world = 2
  rng = np.random.RandomState(77)
  dims, rows = [16, 8, 4], [1001, 64, 10]
  combiners = ['sum', 'mean', 'sum']
  tables = [rng.uniform(-1, 1, size=(rows[c], dims[c])).astype(np.float32) for c in range(3)]
  # rank 0: nothing at all; rank 1: column 0 only even ids (owner 0), column 1 ragged with empty
  # segments, column 2 empty
  sp1 = np.array([0, 0, 3, 3, 3, 7, 7], np.int32)
  ids = [[np.zeros(0, np.int64)] * 3,
         [(rng.randint(0, 500, size=777) * 2).astype(np.int64),
          rng.randint(0, 2**40, size=7).astype(np.int64), np.zeros(0, np.int64)]]
  splits = [[None, np.zeros(1, np.int32), None], [None, sp1, None]]
  grads = [[np.zeros((0, 16), np.float32), np.zeros((0, 8), np.float32),
            np.zeros((0, 4), np.float32)],
           [rng.randn(777, 16).astype(np.float32), rng.randn(6, 8).astype(np.float32),
            np.zeros((0, 4), np.float32)]]
  comms = hb.distribute.Collective.local_world(world)
  results, errors = [None] * world, []

  def run(r):
    try:
      with torch.cuda.stream(torch.cuda.Stream()):
        drv = ShardedGroupLookup([dev(t[r::world].copy()) for t in tables], comms[r],
                                 buckets=rows, combiners=combiners)
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
  assert not errors, errors
  for r in range(world):
    want = oracle.group_lookup_fwd(tables, ids[r], splits[r], rows, combiners)
    for c in range(3):
      np.testing.assert_equal(results[r][0][c], want[c])
  for c in range(3):
    dense, mag = world_grad_sums(rows[c], dims[c], [(ids[r][c], grads[r][c], splits[r][c], combiners[c])
                                                    for r in range(world)])
    got = np.zeros_like(dense)
    for r in range(world):
      lr_, g_ = results[r][1][c]
      got[lr_ * world + r] += g_
    assert_sums_close(got, dense, mag)
  assert results[1][1][0][0].size == 0          # owner 1 got no row of column 0
  for cm in comms:
    cm.close()


# ----------------------------------------------------------------------------------
# config 1 made real: ParquetDataset -> values + row_splits in HBM -> fused lookup
def test_parquet_to_dense_features(tmp_path):
  pa = pytest.importorskip('pyarrow')
  pq = pytest.importorskip('pyarrow.parquet')
  rng = np.random.RandomState(90)
  n = 5000
  scalar = rng.randint(0, 2**40, size=n)
  lists = [rng.randint(0, 2**40, size=rng.randint(0, 9)).tolist() for _ in range(n)]
  path = str(tmp_path / 'day-0.parquet')
  pq.write_table(pa.table({'uid': pa.array(scalar, pa.int64()),
                           'clicks': pa.array(lists, pa.list_(pa.int64()))}),
                 path, row_group_size=1024)
  cols = [hb.feature_column.EmbeddingColumn('uid', 100003, 16, 'sum'),
          hb.feature_column.EmbeddingColumn('clicks', 1000000, 16, 'mean')]   # config 1's table
  tables = [rng.uniform(-1e-3, 1e-3, size=(c.num_buckets, c.dimension)).astype(np.float32)
            for c in cols]
  layer = hb.feature_column.DenseFeatures(
    cols, DEV, init=lambda c, rows, d: dev(tables[cols.index(c)]))
  seen = 0
  for batch in hb.data.ParquetDataset(path, 777, device=DEV):
    out = layer(batch).cpu().numpy()
    k = out.shape[0]
    ids = [scalar[seen:seen + k], np.array(sum(lists[seen:seen + k], []), np.int64)]
    lens = [len(x) for x in lists[seen:seen + k]]
    sps = [None, np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)]
    want = oracle.group_lookup_fwd(tables, ids, sps, [c.num_buckets for c in cols],
                                   [c.combiner for c in cols])
    np.testing.assert_equal(out, np.concatenate(want, axis=1))
    seen += k
  assert seen == n


def test_dense_features_adagrad_sharded_in_process_world():
  """A training step's embedding side at W = 2: forward, backward and the fused Adagrad apply on
  sharded + replicated tables equal a single-process float64 Adagrad step on the full tables
  (replicated tables are NOT stepped at W > 1: their gradients need the cross-rank aggregation of
  training/gradient.py:119-177 first, so backward() only returns their IndexedSlices)."""
  import threading
  world = 2
  rng = np.random.RandomState(93)
  cols, tables, feats, grads, batch = _dense_case(rng, world)
  comms = hb.distribute.Collective.local_world(world)
  results, errors = [None] * world, []

  def run(r):
    try:
      with torch.cuda.stream(torch.cuda.Stream()):
        def init(c, rows, d):
          t = tables[cols.index(c)]
          return dev((t[r::world] if rows != c.num_buckets else t).copy())
        layer = hb.feature_column.DenseFeatures(cols, DEV, coll=comms[r], batch_size=batch,
                                                init=init, initial_accumulator_value=0.1)
        layer(_dev_feats(feats[r]))
        res = layer.backward(dev(grads[r]), apply_lr=0.05, optimizer='adagrad')
        torch.cuda.current_stream().synchronize()
        n1 = int(res[1][2].item())
        results[r] = ([w.cpu().numpy() for w in layer.weights],
                      [a.cpu().numpy() for a in layer.accums],
                      (res[1][0].cpu().numpy()[:n1], res[1][1].cpu().numpy()[:n1]))
        layer.close()
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
  off = 0
  for k, c in enumerate(cols):
    per_rank, per_rank_mag = [], []
    for r in range(world):
      _, ids, sps = _want_dense(cols, tables, feats[r])
      g = np.ascontiguousarray(grads[r][:, off:off + c.dimension])
      dense, mag = world_grad_sums(c.num_buckets, c.dimension, [(ids[k], g, sps[k], c.combiner)])
      per_rank.append(dense)
      per_rank_mag.append(mag)
    t64 = tables[k].astype(np.float64)
    if k != 1:        # sharded: the owner applies the sum of both ranks' gradients once
      g = per_rank[0] + per_rank[1]
      mag = per_rank_mag[0] + per_rank_mag[1]
      a = 0.1 + g * g
      want = t64 - 0.05 * g / np.sqrt(a)
      # magnitudes as in test_gpu_parity.py's Adagrad bounds: d(g^2) = 2 |g| dg, and
      # |d/dg (g / sqrt(a0 + g^2))| <= 1 / sqrt(a0)
      var_mag = np.abs(t64) + 0.05 * (mag / np.sqrt(0.1) + 1.0)
      acc_mag = 0.1 + g * g + 2.0 * np.abs(g) * mag
      for r in range(world):
        assert_sums_close(results[r][0][k], want[r::world], var_mag[r::world])
        assert_sums_close(results[r][1][k], a[r::world], acc_mag[r::world])
    else:             # replicated: untouched (the replicas would diverge), slices handed back
      for r in range(world):
        np.testing.assert_equal(results[r][0][k], tables[k])
        np.testing.assert_equal(results[r][1][k], np.full_like(tables[k], 0.1))
        rows_r, g_r = results[r][2]
        got = np.zeros_like(per_rank[r])
        got[rows_r] = g_r
        assert_sums_close(got, per_rank[r], per_rank_mag[r])
    off += c.dimension


@pytest.mark.parametrize('form', ['bound', 'functional'])
@pytest.mark.parametrize('dedup', [False, True])
@pytest.mark.parametrize('world', [1, 3])
def test_sharded_prefetch_next_step(world, dedup, form):
  """hbk_sharded_prefetch: step i + 1 is partitioned on the plan's own stream while step i is in
  flight; a matching forward consumes it, a non-matching one drops it; the backward of step i
  still sees step i's shard index.  Results equal the unprefetched driver (= the oracle).
  form 'functional': ``drv(ids)`` with new tensors every step and ``prefetch(next id tensors)``."""
  import threading
  rng = np.random.RandomState(95)
  dims, rows = [16, 8], [50021, 300]
  tables = [rng.uniform(-1, 1, size=(rows[c], dims[c])).astype(np.float32) for c in range(2)]
  steps = 4
  # (dedup: the prefetched partition also carries the distinct-id stage; ids repeat heavily then)
  hi = 500 if dedup else 2**40
  ids = [[[rng.randint(0, hi, size=rng.randint(1, 3000)).astype(np.int64) for _ in range(2)]
          for _ in range(steps)] for _ in range(world)]
  grads = [[[rng.randn(ids[r][s][c].size, dims[c]).astype(np.float32) for c in range(2)]
            for s in range(steps)] for r in range(world)]
  if world == 1:
    comms = [hb.distribute.Collective(world_size=1, rank=0)]       # real RCCL communicator
  else:
    comms = hb.distribute.Collective.local_world(world)
  results, errors = [None] * world, []

  def run(r):
    try:
      with torch.cuda.stream(torch.cuda.Stream()):
        drv = ShardedGroupLookup([dev(t[r::world].copy()) for t in tables], comms[r],
                                 buckets=rows, combiners='sum', dedup=dedup)
        dev_ids = [[dev(i) for i in ids[r][s]] for s in range(steps)]
        if form == 'bound':
          bound = [drv.bind(dev_ids[s]) for s in range(steps)]
        else:
          bound = dev_ids     # the functional form: prefetch() is handed the next step's id tensors
        got = []
        for s in range(steps):
          outs = drv.launch(bound[s]) if form == 'bound' else drv(dev_ids[s])
          if s + 1 < steps:
            # step 2's prefetch names the wrong batch (step 0 again): it must be dropped
            drv.prefetch(bound[0] if s == 1 else bound[s + 1])
          sl = drv.backward([dev(g) for g in grads[r][s]])
          torch.cuda.current_stream().synchronize()
          got.append(([o.cpu().numpy() for o in outs],
                      [(u.cpu().numpy()[:int(k.item())], g.cpu().numpy()[:int(k.item())])
                       for u, g, k in sl]))
        results[r] = got
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
  for s in range(steps):
    for r in range(world):
      want = oracle.group_lookup_fwd(tables, ids[r][s], [None, None], rows, ['sum', 'sum'])
      for c in range(2):
        np.testing.assert_equal(results[r][s][0][c], want[c])
    for c in range(2):
      dense, mag = world_grad_sums(rows[c], dims[c], [(ids[r][s][c], grads[r][s][c], None, 'sum')
                                                      for r in range(world)])
      got = np.zeros_like(dense)
      for r in range(world):
        lr_, g_ = results[r][s][1][c]
        got[lr_ * world + r] += g_
      assert_sums_close(got, dense, mag)


# ----------------------------------------------------------------------------------
# multi-node shape: the two-staged lookup (sharding.py:210-276) with virtual ranks
def _group_exchange(values, sizes, groups):
  """values[r][c]: tensor whose rows are chunked by sizes[r][c] (host lists) over r's group, in
  group order.  Returns (recv_values[r][c], recv_sizes[r][c])."""
  world, n = len(values), len(values[0])
  rv = [[None] * n for _ in range(world)]
  rs = [[None] * n for _ in range(world)]
  for g in groups:
    for c in range(n):
      chunks = {}
      for q in g:
        offs = np.concatenate([[0], np.cumsum(sizes[q][c])]).astype(int)
        for k, r in enumerate(g):
          chunks[(q, r)] = values[q][c][offs[k]:offs[k + 1]]
      for r in g:
        rv[r][c] = torch.cat([chunks[(q, r)] for q in g])
        rs[r][c] = [int(chunks[(q, r)].shape[0]) for q in g]
  return rv, rs


@pytest.mark.parametrize('local_size,nodes', [(2, 2), (3, 2), (2, 3)])
def test_hierarchical_lookup_virtual_ranks(local_size, nodes):
  """Every compute phase of the two-staged lookup on the GPU (dual-modulo partitions, unique,
  owner gather with `// W`, restores and stitches as fused N-column launches), the exchanges
  simulated by slicing: the result equals the oracle's restatement of sharding.py:210-276 and the
  unsharded lookup."""
  from hybridbackend_amd.embedding import HierarchicalGroupLookup
  world = local_size * nodes
  rng = np.random.RandomState(200 + world)
  dims, rows = [16, 6, 128], [5003, 64, 977]
  n = len(dims)
  tables = [rng.uniform(-1, 1, size=(rows[c], dims[c])).astype(np.float32) for c in range(n)]
  ids = [[rng.randint(0, 2**40, size=rng.randint(0, 900)).astype(np.int64) for _ in range(n)]
         for _ in range(world)]
  drv = [HierarchicalGroupLookup([dev(t[r::world].copy()) for t in tables], world, local_size,
                                 buckets=rows) for r in range(world)]
  intra = [list(range(m * local_size, (m + 1) * local_size)) for m in range(nodes)]
  inter = [[m * local_size + l for m in range(nodes)] for l in range(local_size)]
  st1 = [drv[r].stage_one([dev(i) for i in ids[r]]) for r in range(world)]
  h = lambda sizes: [[s.tolist() for s in sizes[r]] for r in range(world)]   # noqa: E731
  s0_sizes = h([st1[r][1] for r in range(world)])
  r0, r0_sizes = _group_exchange([st1[r][0] for r in range(world)], s0_sizes, intra)
  st2 = [drv[r].stage_two(r0[r]) for r in range(world)]
  s1_sizes = h([st2[r][1] for r in range(world)])
  r1, r1_sizes = _group_exchange([st2[r][0] for r in range(world)], s1_sizes, inter)
  emb = [drv[r].owner_gather(r1[r]) for r in range(world)]
  b1, _ = _group_exchange(emb, r1_sizes, inter)
  rows0 = [drv[r].unstage_two(b1[r], st2[r][2], st2[r][3]) for r in range(world)]
  b0, _ = _group_exchange(rows0, r0_sizes, intra)
  outs = [drv[r].unstage_one(b0[r], st1[r][2]) for r in range(world)]
  for c in range(n):
    want = oracle.hierarchical_lookup_fwd(oracle.make_shards(tables[c], world),
                                          [ids[r][c] % rows[c] for r in range(world)],
                                          local_size)
    for r in range(world):
      np.testing.assert_equal(outs[r][c].cpu().numpy(), want[r])
      np.testing.assert_equal(want[r], tables[c][ids[r][c] % rows[c]])


def test_hierarchical_call_through_rccl_world1():
  """The whole two-staged forward through a real communicator (world 1: both topologies have
  one active rank), i.e. the plumbing of the four topology-aware exchanges."""
  from hybridbackend_amd.embedding import HierarchicalGroupLookup
  rng = np.random.RandomState(210)
  tables = [rng.uniform(-1, 1, size=(r, d)).astype(np.float32) for r, d in ((3001, 16), (50, 8))]
  ids = [rng.randint(0, 2**40, size=k).astype(np.int64) for k in (2000, 0)]
  coll = hb.distribute.Collective(world_size=1, rank=0)
  try:
    drv = HierarchicalGroupLookup([dev(t) for t in tables], 1, 1, buckets=[3001, 50], coll=coll)
    outs = drv([dev(i) for i in ids])
    for c in range(2):
      np.testing.assert_equal(outs[c].cpu().numpy(), tables[c][ids[c] % tables[c].shape[0]])
  finally:
    coll.close()


@pytest.mark.parametrize('local_size,nodes', [(2, 2), (3, 2)])
def test_hierarchical_call_in_process_world(local_size, nodes):
  """The two-staged forward through the communicator API with INTRA_NODE / INTER_NODE topologies:
  in-process ranks arranged as `nodes` x `local_size` (active ranks and offsets of
  hbtf/distribute/collective.h:80-112 inside hbk_alltoall_n / hbk_alltoallv_n)."""
  import threading
  from hybridbackend_amd.embedding import HierarchicalGroupLookup
  world = local_size * nodes
  rng = np.random.RandomState(220 + world)
  dims, rows = [16, 4], [5003, 97]
  tables = [rng.uniform(-1, 1, size=(rows[c], dims[c])).astype(np.float32) for c in range(2)]
  ids = [[rng.randint(0, 2**40, size=rng.randint(0, 1200)).astype(np.int64) for _ in range(2)]
         for _ in range(world)]
  comms = hb.distribute.Collective.local_world(world, local_size=local_size)
  results, errors = [None] * world, []

  def run(r):
    try:
      with torch.cuda.stream(torch.cuda.Stream()):
        drv = HierarchicalGroupLookup([dev(t[r::world].copy()) for t in tables], world,
                                      local_size, buckets=rows, coll=comms[r])
        outs = drv([dev(i) for i in ids[r]])
        torch.cuda.current_stream().synchronize()
        results[r] = [o.cpu().numpy() for o in outs]
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
  for r in range(world):
    for c in range(2):
      np.testing.assert_equal(results[r][c], tables[c][ids[r][c] % rows[c]])


# ----------------------------------------------------------------------------------
# sharded checkpoints of the feature layer (training/saver.py), re-sharding included
@pytest.mark.parametrize('through', ['ours', 'reference_bundle'])
def test_dense_features_checkpoint_reshard_in_process_world(tmp_path, through):
  """Two ranks train one step (forward, backward, fused Adagrad), save; four ranks restore the
  checkpoint (every rank gathers the rows it now owns) and their forward equals the oracle lookup
  on the updated tables -- weights AND optimizer slots survive the change of world size.
  through='reference_bundle': the checkpoint is first rewritten as the TensorFlow tensor bundle
  the reference would have saved at W = 2 (training/tf_bundle.py) and restored from THAT."""
  import threading
  from hybridbackend_amd.training import export_reference
  from hybridbackend_amd.training import load_full
  rng = np.random.RandomState(97)
  cols, tables, feats2, grads2, batch = _dense_case(rng, 2)
  prefix = str(tmp_path / 'model.ckpt-1')

  def world_run(world, fn):
    comms = hb.distribute.Collective.local_world(world)
    barrier = threading.Barrier(world)
    results, errors = [None] * world, []

    def run(r):
      try:
        with torch.cuda.stream(torch.cuda.Stream()):
          results[r] = fn(r, comms[r], barrier.wait)
          torch.cuda.current_stream().synchronize()
      except Exception as e:  # pylint: disable=broad-except
        import traceback
        errors.append((r, repr(e), traceback.format_exc()))
        barrier.abort()
    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
      t.start()
    for t in threads:
      t.join(timeout=45)
    for cm in comms:
      cm.close()
    assert not errors, errors
    return results

  def train_and_save(r, coll, barrier):
    def init(c, rows, d):
      t = tables[cols.index(c)]
      return dev((t[r::2] if rows != c.num_buckets else t).copy())
    layer = hb.feature_column.DenseFeatures(cols, DEV, coll=coll, batch_size=batch, init=init,
                                            initial_accumulator_value=0.1)
    layer(_dev_feats(feats2[r]))
    layer.backward(dev(grads2[r]), apply_lr=0.05, optimizer='adagrad')
    layer.save(prefix, barrier=barrier)
    layer.close()
    return True

  world_run(2, train_and_save)
  # the logical tables after the step, read back from the checkpoint
  updated = [load_full(prefix, f'{c.key}_embedding/embedding_weights') for c in cols]
  slots = [load_full(prefix, f'{c.key}_embedding/embedding_weights/Adagrad') for c in cols]
  for k, c in enumerate(cols):
    assert updated[k].shape == tables[k].shape
    if k != 1:
      assert not np.array_equal(updated[k], tables[k])       # the step happened
      assert (slots[k] >= 0.1).all() and (slots[k] > 0.1).any()

  if through == 'reference_bundle':
    export_reference(prefix, prefix + '.tf')
  _, _, feats4, _, _ = _dense_case(np.random.RandomState(98), 4)

  def restore_and_lookup(r, coll, barrier):
    layer = hb.feature_column.DenseFeatures(cols, DEV, coll=coll, batch_size=batch,
                                            initial_accumulator_value=0.1)
    if through == 'ours':
      layer.restore(prefix, barrier=barrier)
    else:
      layer.restore_reference(prefix + '.tf', barrier=barrier)
    out = layer(_dev_feats(feats4[r]))
    torch.cuda.current_stream().synchronize()
    res = (out.cpu().numpy(), [w.cpu().numpy() for w in layer.weights],
           [a.cpu().numpy() for a in layer.accums], list(layer.sharded))
    layer.close()
    return res

  res = world_run(4, restore_and_lookup)
  for r in range(4):
    want, _, _ = _want_dense(cols, updated, feats4[r])
    np.testing.assert_equal(res[r][0], want)
    for k in range(len(cols)):
      sharded = res[r][3][k]
      np.testing.assert_equal(res[r][1][k], updated[k][r::4] if sharded else updated[k])
      np.testing.assert_equal(res[r][2][k], slots[k][r::4] if sharded else slots[k])
