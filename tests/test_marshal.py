"""Return types of the functional N-ary ops: whatever path a call takes, the
result is a plain list of tensors (``torch.cat(outs)``, ``outs + [...]``, ``isinstance(outs, list)``
work); the lazy sequences of ``_marshal`` are an explicit opt-in (``lazy=True``)."""
import pytest
import torch

from hybridbackend_amd import _marshal


def test_plain_turns_lazy_sequences_into_lists_of_tensors():
  flat = torch.arange(10)
  runs = _marshal.Runs(flat, [3, 0, 7])
  assert len(runs) == 3 and runs[2].tolist() == list(range(3, 10))
  got = _marshal.plain(runs)
  assert isinstance(got, list) and [t.tolist() for t in got] == [[0, 1, 2], [], list(range(3, 10))]
  assert torch.cat(got).tolist() == list(range(10))           # what the lazy object refuses
  with pytest.raises(TypeError):
    torch.cat(_marshal.Runs(flat, [3, 0, 7]))
  rows = _marshal.plain(_marshal.Rows(torch.arange(6).view(3, 2)))
  assert isinstance(rows, list) and torch.stack(rows).tolist() == [[0, 1], [2, 3], [4, 5]]
  z = _marshal.plain(_marshal.Zipped(_marshal.Runs(flat, [4, 6]), _marshal.Rows(torch.zeros(2, 1))))
  assert isinstance(z, list) and isinstance(z[0], tuple) and z[1][0].tolist() == list(range(4, 10))
  assert _marshal.plain([flat]) == [flat]


def test_cached_workspace_sizes_follow_the_options_generation():
  g = _marshal.options_generation()
  _marshal.options_changed()
  assert _marshal.options_generation() == g + 1


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
