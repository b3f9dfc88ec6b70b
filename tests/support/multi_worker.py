"""One rank of tests/test_gpu_multi.py: a real process on its own GPU with a real RCCL communicator
(hbk_comm_create over ncclCommInitRank) -- the transport every other test replaces by in-process
device copies.  Started W times by the test; every rank runs the same sequence of cases, checks ITS
OWN results against the CPU oracle (inputs of all ranks are regenerated from seeds, so nothing but the
128-byte RCCL id travels outside RCCL) and writes `<dir>/result_<rank>.json`.

  python tests/support/multi_worker.py --rank R --world W --dir D [--cases kat,alltoallv,...]

Reference behaviour replayed: the 2-rank known-answer vectors of
hybridbackend/tensorflow/distribute/tests/alltoall_test.py:219-269 (tests/golden/alltoallv.json),
the Alltoallv offset arithmetic of nccl_collective.cc:250-288 (oracle.alltoallv_sim), the sharded
lookup composition of embedding/sharding.py:171-205 against the UNSHARDED oracle lookup, its
gradient (collective.py:334-347) against the dense scatter-add of all ranks' gradients, and the
gradient aggregation of training/gradient.py:119-217.
"""
import argparse
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def file_broadcast(directory, rank, tag):
  """The 128-byte RCCL id from rank 0 to everybody through the file system (the reference uses a TF
  gRPC broadcast, rpc.py:88-124): written to a temporary name, renamed when complete."""
  def bcast(data):
    path = os.path.join(directory, f'id_{tag}.bin')
    if rank == 0:
      with open(path + '.tmp', 'wb') as f:
        f.write(data)
      os.replace(path + '.tmp', path)
      return data
    deadline = time.time() + 120
    while not os.path.exists(path):
      if time.time() > deadline:
        raise RuntimeError('rank 0 never published the communicator id')
      time.sleep(0.01)
    with open(path, 'rb') as f:
      return f.read()
  return bcast


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--rank', type=int, required=True)
  ap.add_argument('--world', type=int, required=True)
  ap.add_argument('--dir', required=True)
  ap.add_argument('--cases', default='kat,alltoallv,alltoall,sharded,dedup,p2p,reduce')
  ap.add_argument('--local-size', type=int, default=0)
  a = ap.parse_args()
  rank, W = a.rank, a.world
  result = dict(rank=rank, world=W, ok=False, passed=[], errors=[])
  try:
    torch.cuda.set_device(rank % torch.cuda.device_count())
    run(a, result)
    result['ok'] = not result['errors']
  except Exception:  # pylint: disable=broad-except
    result['errors'].append(traceback.format_exc())
  with open(os.path.join(a.dir, f'result_{rank}.json.tmp'), 'w') as f:
    json.dump(result, f)
  os.replace(os.path.join(a.dir, f'result_{rank}.json.tmp'),
             os.path.join(a.dir, f'result_{rank}.json'))
  # (a rank that failed leaves at once: its peers' next collective would otherwise wait for it
  # until the test's timeout -- they are killed by the test instead)
  os._exit(0 if result['ok'] else 1)  # pylint: disable=protected-access


def run(a, result):
  import oracle
  import hybridbackend_amd as hb
  from hybridbackend_amd import _lib
  from hybridbackend_amd.embedding.sharded import ShardedGroupLookup
  from tests.support.tolerance import (WIRE16_FLOOR, WIRE16_REL, assert_sums_close,
                                       world_grad_sums)
  rank, W = a.rank, a.world
  DEV = torch.device('cuda', torch.cuda.current_device())
  cases = a.cases.split(',')

  def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)

  def host(t):
    return t.detach().cpu().numpy()

  def case(name):
    def deco(fn):
      if name.split(':')[0] not in cases:
        return fn
      try:
        fn()
        torch.cuda.synchronize()
        result['passed'].append(name)
      except Exception:  # pylint: disable=broad-except
        result['errors'].append(f'{name}: ' + traceback.format_exc())
        raise
      return fn
    return deco

  coll = hb.distribute.Collective(W, rank, local_size=a.local_size or W,
                                  broadcast_fn=file_broadcast(a.dir, rank, 'main'))
  result['rccl_ranks_seen'] = int(_lib.lib().hbk_comm_rccl_ranks(coll._handle))
  assert result['rccl_ranks_seen'] == W

  # ---- R5: the reference's own known-answer vectors (two ranks) ------------------------------
  @case('kat')
  def _kat():
    if W != 2:
      return
    g = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'alltoallv.json')))
    s = g['single']
    out, out_sizes = coll.alltoall(dev(np.array(s['inputs'][rank], np.int64)),
                                   sizes=dev(np.array(s['sizes'][rank], np.int32)))
    np.testing.assert_equal(host(out), np.array(s['outputs'][rank], np.int64))
    np.testing.assert_equal(host(out_sizes), np.array(s['out_sizes'][rank], np.int32))
    n = g['n']
    vals = [dev(np.array(n['inputs'][c][rank], np.float32)) for c in range(2)]
    send = [n['sizes'][c][rank] for c in range(2)]
    recv = [n['out_sizes'][c][rank] for c in range(2)]
    outs = coll.alltoallv_n(vals, send, recv)
    for c in range(2):
      np.testing.assert_equal(host(outs[c]), np.array(n['outputs'][c][rank], np.float32))
    # gradient KAT (alltoall_test.py:228-243): d/dx of sum(alltoallv(x)) * g is g everywhere, i.e.
    # the reverse exchange with the received sizes returns every element to its sender
    gs = g['grad']['sizes']
    x = dev(np.full(sum(gs[rank]), g['grad']['g'], np.float32))
    recv_sizes = [gs[q][rank] for q in range(2)]
    y = coll.alltoallv_n([x], [gs[rank]], [recv_sizes])[0]
    back = coll.alltoallv_n([y], [recv_sizes], [gs[rank]])[0]
    np.testing.assert_equal(host(back), host(x))

  # ---- R5 / R6: Alltoallv[N] against the offset arithmetic of nccl_collective.cc:250-288 -----
  @case('alltoallv')
  def _alltoallv():
    rng = np.random.RandomState(1000)        # the same stream on every rank
    shapes = [(16,), (), (4,), (128,)]
    for dtype, wire in ((np.float32, None), (np.float32, torch.float16), (np.int64, None),
                        (np.int32, None)):
      n_cols = 4
      sizes = rng.randint(0, 300, size=(n_cols, W, W))        # [c][sender][receiver]
      sizes[1] = 0                                              # a column nobody sends
      sizes[2, :, W - 1] = 0                                    # a rank that receives nothing
      vals = []                                                 # [c][sender] -> array
      for c in range(n_cols):
        per = []
        for q in range(W):
          k = int(sizes[c, q].sum())
          if np.issubdtype(dtype, np.floating):
            per.append(rng.randn(k, *shapes[c]).astype(dtype))
          else:
            per.append(rng.randint(-2**30, 2**30, size=(k,) + shapes[c]).astype(dtype))
        vals.append(per)
      send = [sizes[c, rank].tolist() for c in range(n_cols)]
      recv = [sizes[c, :, rank].tolist() for c in range(n_cols)]
      outs = coll.alltoallv_n([dev(vals[c][rank]) for c in range(n_cols)], send, recv,
                              wire_dtype=wire)
      torch.cuda.synchronize()
      for c in range(n_cols):
        want, want_sizes = oracle.alltoallv_sim(vals[c], [sizes[c, q].tolist() for q in range(W)])
        assert list(want_sizes[rank]) == recv[c]
        w = want[rank]
        if wire is not None:
          w = oracle.cast_f16_to_f32(oracle.cast_f32_to_f16(w))
        np.testing.assert_equal(host(outs[c]), w.reshape(host(outs[c]).shape))
    coll.check_async_errors()

  # ---- equal split (HbNcclAlltoall[N]; the size exchange in front of every Alltoallv) ---------
  @case('alltoall')
  def _alltoall():
    rng = np.random.RandomState(2000)
    per = [3, 64, 1]
    full = [rng.randint(0, 2**40, size=(W, W * p)).astype(np.int64) for p in per]   # [sender][..]
    outs = coll.alltoall_n([dev(f[rank]) for f in full])
    for p, f, o in zip(per, full, outs):
      want = np.concatenate([f[q][rank * p:(rank + 1) * p] for q in range(W)])
      np.testing.assert_equal(host(o), want)

  # ---- R12: the sharded step against the UNSHARDED oracle ------------------------------------
  def sharded_inputs(seed, zipf=False):
    rng = np.random.RandomState(seed)
    dims = [16, 8, 128, 4]
    rows = [50021, 211, 3000, 64]
    combiners = ['sum', 'mean', 'sqrtn', 'sum']
    n = len(dims)
    tables = [rng.uniform(-1, 1, size=(rows[c], dims[c])).astype(np.float32) for c in range(n)]
    ids, splits, grads = [], [], []
    for _ in range(W):
      rid, rsp, rg = [], [], []
      for c in range(n):
        if c % 2 == 0:
          sp, k = None, int(rng.randint(0, 3000))
        else:
          lens = rng.poisson(3, size=rng.randint(1, 400)).clip(0, 12)
          sp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
          k = int(sp[-1])
        rsp.append(sp)
        if zipf:
          rid.append((rng.zipf(1.2, size=k) % rows[c]).astype(np.int64))
        else:
          rid.append(rng.randint(0, 2**40, size=k).astype(np.int64))
        rg.append(rng.randn(k if sp is None else sp.size - 1, dims[c]).astype(np.float32))
      ids.append(rid)
      splits.append(rsp)
      grads.append(rg)
    return dims, rows, combiners, tables, ids, splits, grads

  def sharded_step(label, seed, wire16=False, dedup=False, zipf=False, options=()):
    dims, rows, combiners, tables, ids, splits, grads = sharded_inputs(seed, zipf)
    n = len(dims)
    saved = [(k, _lib.set_option(k, v)) for k, v in options]
    try:
      lr = 0.05
      shards = [dev(t[rank::W].copy()) for t in tables]
      drv = ShardedGroupLookup(shards, coll, buckets=rows, combiners=combiners,
                               wire_dtype=torch.float16 if wire16 else None, dedup=dedup)
      my_ids = [dev(i) for i in ids[rank]]
      my_sp = [None if s is None else dev(s) for s in splits[rank]]
      my_g = [dev(g) for g in grads[rank]]
      for _ in range(2):        # the second step reuses the grown buffers
        outs = drv(my_ids, my_sp)
        slices = drv.backward(my_g, apply_lr=0.0)
      torch.cuda.synchronize()
      eff = tables
      # fp32 sums in a run-dependent order: bounded by the magnitude of their terms
      # (tests/support/tolerance.py; a fixed atol here cost round 5 its GPU record)
      rel, floor = 1e-5, 1e-6
      if wire16:
        eff = [oracle.cast_f16_to_f32(oracle.cast_f32_to_f16(t)) for t in tables]
        rel, floor = WIRE16_REL, WIRE16_FLOOR          # gradients are rounded per sender
      want = oracle.group_lookup_fwd(eff, ids[rank], splits[rank], rows, combiners)
      for c in range(n):
        np.testing.assert_equal(host(outs[c]), want[c], err_msg=f'{label}: forward, column {c}')
      dense, mags = [], []
      for c in range(n):
        d, mag = world_grad_sums(rows[c], dims[c], [(ids[q][c], grads[q][c], splits[q][c], combiners[c])
                                                   for q in range(W)])
        dense.append(d)
        mags.append(mag)
        u, g, k = slices[c]
        k = int(k.item())
        lr_, g_ = host(u)[:k], host(g)[:k]
        assert len(set(lr_.tolist())) == k, f'{label}: column {c}: rows emitted twice'
        mine = d[rank::W]
        got = np.zeros_like(mine)
        got[lr_] = g_
        assert_sums_close(got, mine, mag[rank::W], rel=rel, floor=floor,
                          err_msg=f'{label}: backward, column {c}')
      # one more step with the fused SGD apply: the shard ends at table - lr * dense gradient
      drv(my_ids, my_sp)
      drv.backward(my_g, apply_lr=lr, emit=False)
      torch.cuda.synchronize()
      for c in range(n):
        want_shard = (tables[c].astype(np.float64) - lr * dense[c])[rank::W]
        assert_sums_close(host(shards[c]), want_shard,
                          (np.abs(tables[c]) + lr * mags[c])[rank::W], rel=rel, floor=floor,
                          err_msg=f'{label}: SGD step, column {c}')
      drv.close()
    finally:
      for k, v in reversed(saved):
        _lib.set_option(k, v)

  @case('sharded')
  def _sharded():
    sharded_step('shipped default (inline exchanges), fp32 wire', 300)
    sharded_step('pipelined, two groups', 309, options=(('sharded_inline', 0), ('sharded_groups', 2)))
    sharded_step('one group on the communicator stream', 301,
                 options=(('sharded_inline', 0), ('sharded_groups', 1)))
    sharded_step('three groups', 302, options=(('sharded_inline', 0), ('sharded_groups', 3)))
    sharded_step('inline exchanges, two groups', 303, options=(('sharded_inline', 1), ('sharded_groups', 2)))
    sharded_step('fp16 wire fused', 304, wire16=True)
    sharded_step('fp16 wire through casts', 305, wire16=True, options=(('sharded_wire_fused', 0),))
    sharded_step('int64 ids on the wire', 306, options=(('sharded_id64', 1),))
    sharded_step('late id pack', 307, options=(('sharded_pack_early', 0),))
    sharded_step('own slice through a copy', 308, options=(('sharded_copy_self', 1),))

  @case('dedup')
  def _dedup():
    sharded_step('requester-side dedup, Zipf ids', 400, dedup=True, zipf=True)
    sharded_step('requester-side dedup, pipelined, fp16', 401, dedup=True, zipf=True, wire16=True,
                 options=(('sharded_inline', 0), ('sharded_groups', 2)))
    sharded_step('dedup on uniform ids', 402, dedup=True)

  # ---- the p2p form: owners store rows straight into the requester's output (IPC mappings) -----
  @case('p2p')
  def _p2p():
    for label, seed, options in (
        ('p2p, inline', 600, (('sharded_inline', 1),)),
        ('p2p, communicator stream', 601, (('sharded_inline', 0),)),
        ('p2p, int64 ids, late pack', 602, (('sharded_id64', 1), ('sharded_pack_early', 0))),
        # the last rank's bind "cannot map a peer" (injected): every rank falls back together
        ('p2p refused by one rank', 603, (('sharded_p2p_test_refuse', W - 1),))):
      saved = [(k, _lib.set_option(k, v)) for k, v in options]
      try:
        rng = np.random.RandomState(seed)
        dims = [16, 6, 128, 4]
        rows = [50021, 211, 3000, 64]
        n = len(dims)
        batch = 2000
        tables = [rng.uniform(-1, 1, size=(rows[c], dims[c])).astype(np.float32) for c in range(n)]
        ids = [[[rng.randint(0, 2**40, size=batch).astype(np.int64) for _ in range(n)]
                for _ in range(W)] for _ in range(3)]
        grads = [[rng.randn(batch, dims[c]).astype(np.float32) for c in range(n)] for _ in range(W)]
        shards = [dev(t[rank::W].copy()) for t in tables]
        drv = ShardedGroupLookup(shards, coll, buckets=rows)
        outs = [torch.full((batch, d), float('nan'), device=DEV) for d in dims]
        bound = drv.p2p_bind(outs)
        if any(k == 'sharded_p2p_test_refuse' for k, _ in options):
          assert bound is False, 'a refused mapping must unbind every rank'
        else:
          result.setdefault('p2p_bound', []).append(bool(bound))
        # a box whose driver cannot map a peer's memory (hipIpcGetMemHandle / OpenMemHandle) is not a
        # parity failure: every rank gets the same False (the bind is a collective that agrees on the
        # minimum), the plan keeps the exchange form -- and THAT is what the steps below then check
        for st in range(3):
          drv([dev(i) for i in ids[st][rank]], None, outs)
          torch.cuda.synchronize()
          want = oracle.group_lookup_fwd(tables, ids[st][rank], [None] * n, rows, ['sum'] * n)
          for c in range(n):
            np.testing.assert_equal(host(outs[c]), want[c], err_msg=f'{label}: step {st}, column {c}')
        drv.backward([dev(g) for g in grads[rank]], apply_lr=0.05, emit=False)
        torch.cuda.synchronize()
        for c in range(n):
          dense, mag = world_grad_sums(rows[c], dims[c], [(ids[2][q][c], grads[q][c], None, 'sum')
                                                          for q in range(W)])
          assert_sums_close(host(shards[c]), (tables[c].astype(np.float64) - 0.05 * dense)[rank::W],
                            (np.abs(tables[c]) + 0.05 * mag)[rank::W],
                            err_msg=f'{label}: SGD step, column {c}')
        drv.close()
      finally:
        for k, v in reversed(saved):
          _lib.set_option(k, v)

  # ---- (f1) gradient aggregation: Allreduce / Allgatherv -------------------------------------
  @case('reduce')
  def _reduce():
    rng = np.random.RandomState(3000)
    shapes = [(1000,), (17, 3), (1,), (4096, 16)]
    full = [[rng.randn(*s).astype(np.float32) for s in shapes] for _ in range(W)]
    outs = coll.allreduce_n([dev(x) for x in full[rank]], scale=1.0 / W)
    for c, o in enumerate(outs):
      want = np.sum([full[q][c].astype(np.float64) for q in range(W)], axis=0) / W
      mag = np.sum([np.abs(full[q][c]).astype(np.float64) for q in range(W)], axis=0) / W
      assert_sums_close(host(o), want, mag, err_msg=f'allreduce, tensor {c}')
    counts = rng.randint(0, 50, size=W)
    parts = [rng.randn(int(k), 8).astype(np.float32) for k in counts]
    got = coll.allgather(dev(parts[rank]))
    np.testing.assert_equal(host(got), np.concatenate(parts, 0))
    idx_parts = [rng.randint(0, 1000, size=int(k)).astype(np.int64) for k in counts]
    agg = hb.distribute.aggregate_gradients(
        [dev(full[rank][0]), (dev(parts[rank]), dev(idx_parts[rank])), dev(full[rank][1])], coll,
        sharded=[False, False, True])
    assert_sums_close(
        host(agg[0]), np.sum([full[q][0].astype(np.float64) for q in range(W)], axis=0) / W,
        np.sum([np.abs(full[q][0]).astype(np.float64) for q in range(W)], axis=0) / W,
        err_msg='aggregate_gradients, dense')
    np.testing.assert_allclose(host(agg[1][0]), np.concatenate(parts, 0) / W, rtol=1e-6)
    np.testing.assert_equal(host(agg[1][1]), np.concatenate(idx_parts))
    np.testing.assert_equal(host(agg[2]), full[rank][1])        # sharded: stays local
    # HbNcclBroadcast: every root in turn
    for root in range(W):
      got = hb.distribute.broadcast(dev(full[rank][3]), coll, root_rank=root)
      np.testing.assert_equal(host(got), full[root][3])

  # ---- R4: sub-group topologies (Collective::compute_active_ranks, collective.h:80-112) ---------
  @case('topology')
  def _topology():
    from hybridbackend_amd.distribute.collective import Topology, compute_active_ranks
    L = a.local_size or W
    rng = np.random.RandomState(4000)
    S = rng.randint(0, 200, size=(W, W))                 # rows sender q would send to rank p
    for topo in (Topology.INTRA_NODE, Topology.INTER_NODE, Topology.ALL):
      def send_buffer(q):
        act = compute_active_ranks(topo, W, L, q)
        chunks = [np.arange(S[q][p], dtype=np.float32)[:, None] * np.ones((1, 4), np.float32)
                  + q * 1000.0 + p for p in act]
        return act, (np.concatenate(chunks, 0) if chunks else np.zeros((0, 4), np.float32))
      act, mine = send_buffer(rank)
      assert act == oracle.compute_active_ranks(topo, W, L, rank)
      send = [int(S[rank][p]) for p in act]
      recv = [int(S[q][rank]) for q in act]
      out = coll.alltoallv_n([dev(mine)], [send], [recv], topology=topo)[0]
      want = []
      for q in act:
        q_act, q_buf = send_buffer(q)
        off = sum(int(S[q][p]) for p in q_act[:q_act.index(rank)])
        want.append(q_buf[off:off + int(S[q][rank])])
      np.testing.assert_equal(host(out), np.concatenate(want, 0))

  coll.check_async_errors()
  coll.close()


if __name__ == '__main__':
  main()
