"""Input side: Parquet -> ids / (values, row_splits), the layout the fused lookup
takes (hybridbackend/tensorflow/data/dataframe.py:283-377; reader semantics of
data/tabular/dataset_v1.py:46-97: batch_size, row-group partitions, drop_remainder).  Host-only
tests here; the GPU leg feeds the batches into the lookup."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pa = pytest.importorskip('pyarrow')
pq = pytest.importorskip('pyarrow.parquet')


def _write(tmp_path, n=1000, row_group=300, seed=0):
  rng = np.random.RandomState(seed)
  ids = rng.randint(-2**40, 2**40, size=n)
  small = rng.randint(0, 1000, size=n).astype(np.int32)
  lists = [rng.randint(0, 2**40, size=rng.randint(0, 7)).tolist() for _ in range(n)]
  lists[5] = None                      # a null list reads as an empty list
  lists[n - 1] = []
  table = pa.table({'a': pa.array(ids, pa.int64()), 's': pa.array(small, pa.int32()),
                    'b': pa.array(lists, pa.list_(pa.int64()))})
  path = str(tmp_path / 'part-0.parquet')
  pq.write_table(table, path, row_group_size=row_group)
  return path, ids, small, [x if x is not None else [] for x in lists]


def _collect(ds):
  a, s, b = [], [], []
  sizes = []
  for batch in ds:
    ids = batch['a'].numpy()
    values, splits = (x.numpy() for x in batch['b'])
    assert splits.dtype == np.int32 and splits[0] == 0 and splits[-1] == values.size
    assert ids.size == splits.size - 1 == batch['s'].numpy().size
    sizes.append(ids.size)
    a.append(ids)
    s.append(batch['s'].numpy())
    b += [values[splits[i]:splits[i + 1]].tolist() for i in range(ids.size)]
  return np.concatenate(a), np.concatenate(s), b, sizes


@pytest.mark.parametrize('batch_size', [128, 300, 1000, 4096])
def test_parquet_batches_values_and_row_splits(tmp_path, batch_size):
  import hybridbackend_amd as hb
  path, ids, small, lists = _write(tmp_path)
  a, s, b, sizes = _collect(hb.data.ParquetDataset(path, batch_size))
  np.testing.assert_equal(a, ids)
  np.testing.assert_equal(s, small.astype(np.int64))     # ids are widened to int64
  assert b == lists
  assert all(x == batch_size for x in sizes[:-1]) and sum(sizes) == ids.size


def test_parquet_partitions_drop_remainder_and_field_selection(tmp_path):
  import hybridbackend_amd as hb
  path, ids, _, lists = _write(tmp_path, n=1000, row_group=100)
  seen = []
  for part in range(3):            # row groups part, part + 3, ..: one partition per rank
    ds = hb.data.ParquetDataset([path], 64, fields=['a', 'b', 's'], partition_count=3,
                                partition_index=part, drop_remainder=True)
    a, _, _, sizes = _collect(ds)
    assert all(x == 64 for x in sizes)
    groups = list(range(part, 10, 3))
    want = np.concatenate([ids[g * 100:(g + 1) * 100] for g in groups])
    np.testing.assert_equal(a, want[:a.size])
    assert want.size - a.size < 64
    seen.append(a.size)
  assert sum(seen) <= 1000
  only = next(iter(hb.data.ParquetDataset(path, 10, fields=['b'])))
  assert list(only.keys()) == ['b']
  with pytest.raises(ValueError):
    hb.data.ParquetDataset(path, 0)
  with pytest.raises(TypeError):
    table = pa.table({'f': pa.array([0.5, 1.5])})
    bad = str(tmp_path / 'bad.parquet')
    pq.write_table(table, bad)
    next(iter(hb.data.ParquetDataset(bad, 2)))
