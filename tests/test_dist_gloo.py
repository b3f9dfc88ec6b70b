"""The N > 1 host path on CPU: two real processes over torch.distributed (gloo).

The HIP compute phases cannot run without a GPU, so the per-rank compute is done with the CPU
oracle; what IS the product here is the host arithmetic that decides what travels where:
`hbk_sharded_layout` (csrc/sharded.hip: peer-major message sizes and run offsets of both
exchanges, exactly what `hbk_sharded_lookup_fwd/_bwd` use at 8 GPUs), `alltoallv_offsets`
and `compute_active_ranks`.  Messages are built from that layout, one per peer, exchanged
between two processes, unpacked with the same layout, and the result must equal the unsharded
oracle lookup; the backward runs through the same layout reversed and must equal the dense
scatter-add of both ranks' gradients on every shard.  The reference's own 2-rank KATs (alltoall_test.py:219-226, :254-269) are
replayed through the same transport."""
import ctypes as C
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


def gloo_alltoallv(rank, world, send, send_sizes):
  """send: 1-D array; send_sizes[i] elements go to rank i (Alltoallv offset arithmetic of
  nccl_collective.cc:250-288 through hb.distribute.alltoallv_offsets).  Returns (recv, sizes)."""
  from hybridbackend_amd.distribute import alltoallv_offsets
  gathered = [None] * world
  dist.all_gather_object(gathered, (np.ascontiguousarray(send), [int(x) for x in send_sizes]))
  chunks, sizes = [], []
  for peer_arr, peer_sizes in gathered:
    offs, _ = alltoallv_offsets(peer_sizes)
    chunks.append(peer_arr[offs[rank]:offs[rank] + peer_sizes[rank]])
    sizes.append(peer_sizes[rank])
  return np.concatenate(chunks), sizes


def product_layout(dims, S, R):
  """hbk_sharded_layout through the C ABI (host-only code path of libhbk_core.so)."""
  from hybridbackend_amd import _lib
  lib = _lib.lib()
  N, W = S.shape
  i32 = lambda n: np.zeros(n, np.int32)   # noqa: E731
  i64 = lambda n: np.zeros(n, np.int64)   # noqa: E731
  out = dict(ids_send_peer=i32(W), ids_recv_peer=i32(W), rows_send_peer=i32(W),
             rows_recv_peer=i32(W), req_id_off=i64(N * W), req_row_off=i64(N * W),
             own_id_off=i64(N * W), own_row_off=i64(N * W), col_shard_off=i64(N * W))
  d = np.ascontiguousarray(dims, np.int32)
  S = np.ascontiguousarray(S, np.int32)
  R = np.ascontiguousarray(R, np.int32)
  p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
  _lib.check(lib.hbk_sharded_layout(N, W, p(d), p(S), p(R), *[p(v) for v in out.values()]))
  for k in ('req_id_off', 'req_row_off', 'own_id_off', 'own_row_off'):
    out[k] = out[k].reshape(W, N)
  out['col_shard_off'] = out['col_shard_off'].reshape(N, W)
  return out


def _worker(rank, world, port, wire16, result_dir):
  sys.path.insert(0, ROOT)
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    import oracle
    # --- the reference's known-answer vectors through the transport ---
    g = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'alltoallv.json')))
    s = g['single']
    out, sizes = gloo_alltoallv(rank, world, np.array(s['inputs'][rank], np.int64),
                                s['sizes'][rank])
    assert out.tolist() == s['outputs'][rank] and sizes == s['out_sizes'][rank]
    n = g['n']
    for c in range(2):
      out, sizes = gloo_alltoallv(rank, world, np.array(n['inputs'][rank][c], np.float32),
                                  n['sizes'][rank][c])
      assert out.tolist() == n['outputs'][rank][c] and sizes == n['out_sizes'][rank][c]

    # --- the sharded pipeline, messages laid out by the product's hbk_sharded_layout ---
    rng = np.random.RandomState(1234)            # same stream on both ranks
    dims, rows = [16, 4, 8], [1003, 50, 777]
    combiners = ['sum', 'mean', 'sqrtn']
    N = 3
    tables = [rng.uniform(-1, 1, size=(rows[c], dims[c])).astype(np.float32) for c in range(N)]
    all_ids, all_splits = [], []
    for r in range(world):
      rid, rsp = [], []
      for c in range(N):
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
    ids, splits = all_ids[rank], all_splits[rank]
    shards = [np.ascontiguousarray(t[rank::world]) for t in tables]
    # 1 bucketize + partition (oracle = the reference CPU functor)
    part = [oracle.partition_by_modulo(oracle.floormod(ids[c], rows[c]), world) for c in range(N)]
    S = np.stack([p[1] for p in part]).astype(np.int32)                      # [N][W]
    # 2 ONE size exchange for all columns: chunk for peer q = S[:, q]
    recv, _ = gloo_alltoallv(rank, world, np.ascontiguousarray(S.T).reshape(-1), [N] * world)
    R = recv.reshape(world, N).astype(np.int32)                              # [W][N]
    lay = product_layout(dims, S, R)
    # 3 ids, one message per peer
    send_ids = np.zeros(int(lay['ids_send_peer'].sum()), np.int64)
    for q in range(world):
      for c in range(N):
        o, k = int(lay['col_shard_off'][c][q]), int(S[c][q])
        d0 = int(lay['req_id_off'][q][c])
        send_ids[d0:d0 + k] = part[c][0][o:o + k]
    recv_ids, got_sizes = gloo_alltoallv(rank, world, send_ids, lay['ids_send_peer'])
    assert got_sizes == lay['ids_recv_peer'].tolist()
    # 4 owner gather into the peer-major reply
    send_rows = np.zeros(int(lay['rows_send_peer'].sum()), np.float32)
    for q in range(world):
      for c in range(N):
        k = int(R[q][c])
        i0, f0 = int(lay['own_id_off'][q][c]), int(lay['own_row_off'][q][c])
        emb = oracle.gather(shards[c], recv_ids[i0:i0 + k] // world)
        send_rows[f0:f0 + k * dims[c]] = emb.reshape(-1)
    wire = send_rows
    if wire16:
      wire = oracle.cast_f32_to_f16(send_rows)
    recv_rows, got_sizes = gloo_alltoallv(rank, world, wire, lay['rows_send_peer'])
    assert got_sizes == lay['rows_recv_peer'].tolist()
    if wire16:
      recv_rows = oracle.cast_f16_to_f32(recv_rows)
    # 5 unpack column-major, stitch + combiner
    eff = tables
    if wire16:
      eff = [oracle.cast_f16_to_f32(oracle.cast_f32_to_f16(t)) for t in tables]
    want = oracle.group_lookup_fwd(eff, ids, splits, rows, combiners)
    for c in range(N):
      col = np.zeros((ids[c].size, dims[c]), np.float32)
      for q in range(world):
        o, k = int(lay['col_shard_off'][c][q]), int(S[c][q])
        f0 = int(lay['req_row_off'][q][c])
        col[o:o + k] = recv_rows[f0:f0 + k * dims[c]].reshape(k, dims[c])
      idx = part[c][2]
      if splits[c] is None:
        got = col[idx]
      else:
        got = oracle.segment_combine(col, idx, splits[c], combiners[c])
      np.testing.assert_equal(got, want[c])

    # --- backward through the same layout, reversed (collective.py:334-347): d(stitch) rows are
    # written where the forward's rows arrived, travel back with send/recv sizes swapped, and
    # the owner reduces duplicates over the ids it still holds from the forward ---
    if not wire16:
      all_grads = [[rng.randn(all_ids[r][c].size if all_splits[r][c] is None
                              else all_splits[r][c].size - 1, dims[c]).astype(np.float32)
                    for c in range(N)] for r in range(world)]
      grads = all_grads[rank]
      back = np.zeros(int(lay['rows_recv_peer'].sum()), np.float32)
      for c in range(N):
        sp = splits[c] if splits[c] is not None else np.arange(ids[c].size + 1, dtype=np.int32)
        g_id = oracle.segment_combine_grad(grads[c], sp, combiners[c])          # [n_ids, dim]
        col = np.zeros((ids[c].size, dims[c]), np.float32)
        col[part[c][2]] = g_id                                               # d(gather by index)
        for q in range(world):
          o, k = int(lay['col_shard_off'][c][q]), int(S[c][q])
          f0 = int(lay['req_row_off'][q][c])
          back[f0:f0 + k * dims[c]] = col[o:o + k].reshape(-1)
      got_back, got_sizes = gloo_alltoallv(rank, world, back, lay['rows_recv_peer'])
      assert got_sizes == lay['rows_send_peer'].tolist()
      for c in range(N):
        rows_c, g_c = [], []
        for q in range(world):
          k = int(R[q][c])
          i0, f0 = int(lay['own_id_off'][q][c]), int(lay['own_row_off'][q][c])
          rows_c.append(recv_ids[i0:i0 + k] // world)
          g_c.append(got_back[f0:f0 + k * dims[c]].reshape(k, dims[c]))
        rows_c, g_c = np.concatenate(rows_c), np.concatenate(g_c)
        shard_grad = np.zeros(shards[c].shape, np.float64)
        np.add.at(shard_grad, rows_c, g_c.astype(np.float64))
        dense = np.zeros(tables[c].shape, np.float64)                        # all ranks' ids
        for r in range(world):
          sp = all_splits[r][c] if all_splits[r][c] is not None else \
            np.arange(all_ids[r][c].size + 1, dtype=np.int32)
          np.add.at(dense, all_ids[r][c] % rows[c],
                    oracle.segment_combine_grad(all_grads[r][c], sp, combiners[c])
                    .astype(np.float64))
        np.testing.assert_allclose(shard_grad, dense[rank::world], rtol=1e-6, atol=1e-9)
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


def test_layout_is_consistent_between_ranks():
  """What rank a plans to send to rank b is what rank b plans to receive from rank a, for every
  pair, and the run offsets tile the buffers without gaps (4 ranks, mixed dims)."""
  sys.path.insert(0, ROOT)
  rng = np.random.RandomState(5)
  W, N = 4, 5
  dims = [4, 16, 6, 8, 36]       # 6: runs are padded to 16 bytes
  pad4 = lambda x: (x + 3) // 4 * 4   # noqa: E731
  S = [rng.randint(0, 50, size=(N, W)).astype(np.int32) for _ in range(W)]   # per rank
  R = [np.stack([S[q][:, r] for q in range(W)]).astype(np.int32) for r in range(W)]
  lays = [product_layout(dims, S[r], R[r]) for r in range(W)]
  for a in range(W):
    for b in range(W):
      assert lays[a]['ids_send_peer'][b] == lays[b]['ids_recv_peer'][a]
      assert lays[a]['rows_recv_peer'][b] == lays[b]['rows_send_peer'][a]
    L = lays[a]
    run = 0
    for q in range(W):
      for c in range(N):
        assert L['req_id_off'][q][c] == run
        run += S[a][c][q]
    assert run == L['ids_send_peer'].sum()
    frun = 0
    for q in range(W):
      for c in range(N):
        assert L['own_row_off'][q][c] == frun
        frun += pad4(R[a][q][c] * dims[c])
    assert frun == L['rows_send_peer'].sum()
    for c in range(N):
      assert L['col_shard_off'][c].tolist() == np.concatenate([[0], np.cumsum(S[a][c])[:-1]]).tolist()


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
