"""Host-side marshalling helpers of the Python call forms (hybridbackend_amd/_marshal.py): lazy
per-column views, per-thread argument blocks, the one-pass tensor check.  No GPU: the device ops
that use them are covered by the GPU tests -- except for the RETURN TYPES of the functional N-ary
ops: whatever path a call takes the result is a plain list of tensors
(``torch.cat(outs)``, ``outs + [...]``, ``isinstance(outs, list)`` work); the lazy sequences are an
explicit opt-in (``lazy=True``)."""
import ctypes
import threading

import numpy as np
import torch

import pytest

from hybridbackend_amd import _marshal as m


def test_runs_are_lazy_views_of_one_allocation():
  flat = torch.arange(10)
  r = m.Runs(flat, [3, 0, 7])
  assert r._views is None and len(r) == 3          # nothing materialised by len()
  assert r[0].tolist() == [0, 1, 2] and r[1].numel() == 0 and r[2].tolist() == list(range(3, 10))
  assert [x.numel() for x in r] == [3, 0, 7]
  r[2][0] = 99                                        # views, not copies
  assert flat[3].item() == 99
  assert [v.tolist() for v in r[0:2]] == [[0, 1, 2], []]
  rows = m.Rows(torch.arange(6).view(3, 2))
  assert len(rows) == 3 and rows[2].tolist() == [4, 5] and [x.tolist() for x in rows][0] == [0, 1]
  z = m.Zipped(r, rows)
  assert len(z) == 3 and z[1][1].tolist() == [2, 3] and len(list(z)) == 3
  assert isinstance(z[0:2], list) and len(z[0:2]) == 2


def test_arg_block_is_per_thread_and_addressable():
  blk, addr = m.arg_block(4, 3)
  blk[0] = [1, 2, 3, 2**47]
  blk[2] = np.arange(4, dtype=np.uint64) * np.uint64(8) + np.uint64(1000)
  raw = ctypes.cast(addr, ctypes.POINTER(ctypes.c_uint64))
  assert raw[3] == 2**47 and raw[2 * 4 + 1] == 1008
  again, addr2 = m.arg_block(4, 3)
  assert addr2 == addr and again is blk               # reused by the same thread
  other = []
  t = threading.Thread(target=lambda: other.append(m.arg_block(4, 3)[1]))
  t.start()
  t.join()
  assert other[0] != addr                              # another thread, another block


def test_vector_pass_rejects_what_needs_the_detailed_checks():
  ok = [torch.zeros(3, dtype=torch.int64), torch.zeros(0, dtype=torch.int64)]
  assert m.vector_pass(ok, (torch.int64,)) is None    # host tensors: the slow path raises properly
  assert m.vector_pass([torch.zeros(3)], (torch.int64,)) is None
  assert m.vector_pass([torch.zeros(2, 2, dtype=torch.int64)], (torch.int64,)) is None


def test_plain_turns_lazy_sequences_into_lists_of_tensors():
  flat = torch.arange(10)
  runs = m.Runs(flat, [3, 0, 7])
  assert len(runs) == 3 and runs[2].tolist() == list(range(3, 10))
  got = m.plain(runs)
  assert isinstance(got, list) and [t.tolist() for t in got] == [[0, 1, 2], [], list(range(3, 10))]
  assert torch.cat(got).tolist() == list(range(10))           # what the lazy object refuses
  with pytest.raises(TypeError):
    torch.cat(m.Runs(flat, [3, 0, 7]))
  rows = m.plain(m.Rows(torch.arange(6).view(3, 2)))
  assert isinstance(rows, list) and torch.stack(rows).tolist() == [[0, 1], [2, 3], [4, 5]]
  z = m.plain(m.Zipped(m.Runs(flat, [4, 6]), m.Rows(torch.zeros(2, 1))))
  assert isinstance(z, list) and isinstance(z[0], tuple) and z[1][0].tolist() == list(range(4, 10))
  assert m.plain([flat]) == [flat]


def test_cached_workspace_sizes_follow_the_options_generation():
  g = m.options_generation()
  m.options_changed()
  assert m.options_generation() == g + 1


@pytest.mark.gpu
def test_functional_ops_return_plain_lists(hbk_option):
  import hybridbackend_amd as hb
  dev = torch.device('cuda:0')
  ids = [torch.randint(0, 1 << 40, (1000 + 17 * c,), device=dev) for c in range(5)]
  outs, sizes, idxs = hb.distribute.partition_by_modulo_n(ids, 4)
  for seq in (outs, sizes, idxs):
    assert isinstance(seq, list) and all(isinstance(t, torch.Tensor) for t in seq)
  assert torch.cat(outs).numel() == sum(t.numel() for t in ids)
  assert torch.stack(sizes).shape == (5, 4)
  assert len(outs + [ids[0]]) == 6
  # the slow path (non-contiguous input) returns the same kind of thing
  strided = [torch.randint(0, 1 << 40, (2000,), device=dev)[::2] for _ in range(3)]
  o2, s2, i2 = hb.distribute.partition_by_modulo_n([t.contiguous() for t in strided], 4)
  assert type(o2) is type(outs) and type(s2) is type(sizes)
  o3, s3, i3 = hb.distribute.partition_by_dual_modulo_n(ids, 2, 2, 1)
  assert isinstance(o3, list) and torch.cat(i3).dtype == torch.int32
  lazy = hb.distribute.partition_by_modulo_n(ids, 4, lazy=True)
  assert not isinstance(lazy[0], list)
  assert all(torch.equal(a, b) for a, b in zip(lazy[0], outs))
  assert all(torch.equal(a, b) for a, b in zip(lazy[1], sizes))

  u = hb.embedding.unique_n(ids)
  assert isinstance(u, list) and isinstance(u[0], tuple) and len(u[0]) == 3
  assert torch.cat([t[2] for t in u]).numel() == 5
  ul = hb.embedding.unique_n(ids, lazy=True)
  assert all(torch.equal(a[1], b[1]) for a, b in zip(ul, u))

  tables = [torch.randn(100, 8, device=dev) for _ in range(5)]
  small = [t % 100 for t in ids]
  g = hb.embedding.GroupLookup(tables)
  res = g(small)
  assert isinstance(res, list) and torch.cat(res, 0).shape == (sum(t.numel() for t in ids), 8)
  assert torch.equal(res[3], tables[3][small[3]])
  res_l = g(small, lazy=True)
  assert not isinstance(res_l, list) and torch.equal(res_l[3], res[3])
  f = hb.embedding.group_lookup(tables, small)
  assert isinstance(f, list) and torch.stack([t[:1000] for t in f]).shape == (5, 1000, 8)

  # a cached workspace size must not survive an option that changes it
  big = [torch.randint(0, 1 << 40, (300000,), device=dev) for _ in range(8)]
  a = hb.distribute.partition_by_modulo_n(big, 8)
  hbk_option('partition_onepass', 0)
  b = hb.distribute.partition_by_modulo_n(big, 8)
  assert all(torch.equal(x, y) for x, y in zip(a[0], b[0]))
  ua = hb.embedding.unique_n(big)
  hbk_option('unique_buckets_log2', 12)
  ub = hb.embedding.unique_n(big)
  assert all(torch.equal(x[1], y[1]) for x, y in zip(ua, ub))
