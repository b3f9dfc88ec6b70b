"""The N > 1 host path on CPU: two real processes over torch.distributed (gloo).

What runs here is the PRODUCT's pipeline driver (`ShardedGroupLookup.__call__`: the size
exchange layout, exchange order and reversed sizes of sharding.py:171-205) and its offset
helpers; the HIP compute phases cannot run without a GPU, so a test-only subclass computes
them with the CPU oracle, and a test-only transport carries the exchanges over gloo using the
Alltoallv offset arithmetic (nccl_collective.cc:250-288).  The reference's own 2-rank KATs
(alltoall_test.py:219-226, :254-269) are replayed through the same transport."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


class GlooTransport:
  """alltoall_n / alltoallv_n with the Collective's signatures, over gloo (CPU tensors)."""

  def __init__(self, world_size, rank):
    self.world_size, self.rank = world_size, rank

  def active_size(self, topology=0):
    return self.world_size

  def alltoall_n(self, values, topology=0):
    outs = []
    for v in values:
      gathered = [None] * self.world_size
      dist.all_gather_object(gathered, v.numpy())
      part = v.numel() // self.world_size
      outs.append(torch.from_numpy(np.concatenate(
        [g[self.rank * part:(self.rank + 1) * part] for g in gathered])))
    return outs

  def alltoallv_n(self, values, send_sizes, recv_sizes, common_sizes=None, wire_dtype=None,
                  topology=0, outs=None):
    from hybridbackend_amd.distribute import alltoallv_offsets
    import oracle
    res = []
    for c, v in enumerate(values):
      arr = v.numpy()
      if wire_dtype == torch.float16:
        arr = oracle.cast_f32_to_f16(arr)
      gathered = [None] * self.world_size
      dist.all_gather_object(gathered, (arr, list(send_sizes[c])))
      chunks = []
      for i, (peer_arr, peer_sizes) in enumerate(gathered):
        offs, _ = alltoallv_offsets(peer_sizes)
        n = peer_sizes[self.rank]
        assert n == recv_sizes[c][i]          # what the size exchange announced
        chunks.append(peer_arr[offs[self.rank]:offs[self.rank] + n])
      out = np.concatenate(chunks, 0)
      if wire_dtype == torch.float16:
        out = oracle.cast_f16_to_f32(out)
      res.append(torch.from_numpy(np.ascontiguousarray(out)))
    return res


def _worker(rank, world, port, wire16, result_dir):
  sys.path.insert(0, ROOT)
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    import oracle
    from hybridbackend_amd.embedding.sharded import ShardedGroupLookup, _Step

    class OracleSharded(ShardedGroupLookup):
      """Compute phases restated with the oracle (CPU); host logic inherited."""

      def _setup(self):
        pass

      def partition(self, ids, row_splits=None):
        st = _Step()
        st.ids, st.row_splits = list(ids), row_splits or [None] * len(ids)
        outs, sizes, idxs = [], [], []
        for c, t in enumerate(ids):
          x = t.numpy()
          if self.buckets[c]:
            x = oracle.floormod(x, self.buckets[c])
          o, s, i = oracle.partition_by_modulo(x, self.world_size)
          outs.append(torch.from_numpy(o))
          sizes.append(s)
          idxs.append(torch.from_numpy(i))
        st.send_ids, st.shard_index = outs, idxs
        st.send_sizes = torch.from_numpy(np.stack(sizes).astype(np.int32))
        return st

      def owner_gather(self, st, recv_ids):
        st.recv_ids = recv_ids
        st.send_rows = [torch.from_numpy(oracle.gather(self.shards[c].numpy(),
                                                       r.numpy() // self.world_size))
                        for c, r in enumerate(recv_ids)]
        return st.send_rows

      def stitch(self, st, recv_rows):
        outs = []
        for c, e in enumerate(recv_rows):
          idx = st.shard_index[c].numpy()
          sp = st.row_splits[c]
          if sp is None:
            outs.append(torch.from_numpy(e.numpy()[idx]))
          else:
            outs.append(torch.from_numpy(oracle.segment_combine(
              e.numpy(), idx, sp.numpy(), self.combiners[c])))
        return outs

    transport = GlooTransport(world, rank)
    # --- the reference's known-answer vectors through the transport ---
    g = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'alltoallv.json')))
    s = g['single']
    recv_sizes = transport.alltoall_n([torch.tensor(s['sizes'][rank], dtype=torch.int32)])[0]
    assert recv_sizes.tolist() == s['out_sizes'][rank]
    out = transport.alltoallv_n([torch.tensor(s['inputs'][rank], dtype=torch.int64)],
                                [s['sizes'][rank]], [recv_sizes.tolist()])[0]
    assert out.tolist() == s['outputs'][rank]
    n = g['n']
    outs = transport.alltoallv_n(
      [torch.tensor(n['inputs'][rank][c], dtype=torch.float32) for c in range(2)],
      n['sizes'][rank], n['out_sizes'][rank])
    for c in range(2):
      assert outs[c].tolist() == n['outputs'][rank][c]

    # --- the product's pipeline driver over 2 ranks ---
    rng = np.random.RandomState(1234)            # same stream on both ranks
    dims, rows = [16, 4, 8], [1003, 50, 777]
    combiners = ['sum', 'mean', 'sqrtn']
    tables = [rng.uniform(-1, 1, size=(rows[c], dims[c])).astype(np.float32) for c in range(3)]
    all_ids, all_splits = [], []
    for r in range(world):
      rid, rsp = [], []
      for c in range(3):
        if c == 0:
          rsp.append(None)
          rid.append(rng.randint(0, 2**40, size=200 + 17 * r).astype(np.int64))
        else:
          lens = rng.poisson(3, size=40 + r).clip(0, 9)
          sp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
          rsp.append(sp)
          rid.append(rng.randint(0, 2**40, size=int(sp[-1])).astype(np.int64))
      all_ids.append(rid)
      all_splits.append(rsp)
    shards = [torch.from_numpy(np.ascontiguousarray(t[rank::world])) for t in tables]
    drv = OracleSharded(shards, transport, buckets=rows, combiners=combiners,
                        wire_dtype=torch.float16 if wire16 else None, world_size=world)
    outs = drv([torch.from_numpy(i) for i in all_ids[rank]],
               [None if s_ is None else torch.from_numpy(s_) for s_ in all_splits[rank]])
    eff = tables
    if wire16:
      eff = [oracle.cast_f16_to_f32(oracle.cast_f32_to_f16(t)) for t in tables]
    want = oracle.group_lookup_fwd(eff, all_ids[rank], all_splits[rank], rows, combiners)
    for c in range(3):
      np.testing.assert_equal(outs[c].numpy(), want[c])
    open(os.path.join(result_dir, f'ok{rank}'), 'w').write('ok')
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize('wire16', [False, True])
def test_two_rank_pipeline_over_gloo(tmp_path, wire16):
  world = 2
  port = _free_port()
  mp.spawn(_worker, args=(world, port, wire16, str(tmp_path)), nprocs=world, join=True)
  for r in range(world):
    assert (tmp_path / f'ok{r}').exists()


def test_offsets_and_active_ranks_match_oracle():
  sys.path.insert(0, ROOT)
  import oracle
  from hybridbackend_amd.distribute import alltoallv_offsets, compute_active_ranks
  assert alltoallv_offsets([1, 2, 0, 5], 16) == ([0, 16, 48, 48], 128)
  big, total = alltoallv_offsets([2**30, 2**30, 2**30], 4)   # > 2 GiB: 64-bit offsets
  assert big == [0, 2**32, 2**33] and total == 3 * 2**32
  for topo in (0, 1, 2):
    for world, local in ((8, 8), (8, 4), (16, 8), (4, 1)):
      for rank in range(world):
        assert compute_active_ranks(topo, world, local, rank) == \
          oracle.compute_active_ranks(topo, world, local, rank)
