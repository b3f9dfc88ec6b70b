"""Sharded checkpoints (hybridbackend/tensorflow/training/saver.py:97-185,
embedding/variables.py:114-141) on CPU tensors: the host logic is device agnostic."""
import json
import os
import threading

import numpy as np
import pytest
import torch

from hybridbackend_amd.embedding.variables import sharded_bucket_size
from hybridbackend_amd.training import Saver
from hybridbackend_amd.training import ShardedSlice
from hybridbackend_amd.training import load_full


def _shards(table, world):
  return [torch.from_numpy(np.ascontiguousarray(table[r::world])) for r in range(world)]


def _run(world, fn):
  barrier = threading.Barrier(world)
  errors = []

  def run(r):
    try:
      fn(r, Saver(r, world, barrier.wait if world > 1 else None))
    except Exception as e:  # pylint: disable=broad-except
      errors.append((r, repr(e)))
      barrier.abort()
  threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
  for t in threads:
    t.start()
  for t in threads:
    t.join(timeout=60)
  assert not errors, errors


def _save(tmp_path, world, table, accum, small):
  prefix = str(tmp_path / 'model.ckpt-100')
  w_sh, a_sh = _shards(table, world), _shards(accum, world)

  def fn(r, saver):
    saver.save(prefix, {
      'cat_embedding/embedding_weights': ShardedSlice(w_sh[r], table.shape[0], world, r),
      'cat_embedding/embedding_weights/Adagrad': ShardedSlice(a_sh[r], table.shape[0], world, r),
      'small_embedding/embedding_weights': torch.from_numpy(small.copy()),   # replicated
    })
  _run(world, fn)
  return prefix


def test_layout_of_the_files_and_slice_info(tmp_path):
  rng = np.random.RandomState(0)
  R, D, W = 1003, 8, 2
  table = rng.randn(R, D).astype(np.float32)
  prefix = _save(tmp_path, W, table, np.full((R, D), 0.1, np.float32), rng.randn(5, 4).astype(np.float32))
  files = sorted(os.listdir(tmp_path))
  # (g0: the generation tag -- a later save under the same prefix writes g1 files and switches the
  # index over with one rename)
  assert files == ['model.ckpt-100.data-g0-00000-of-00002', 'model.ckpt-100.data-g0-00001-of-00002',
                   'model.ckpt-100.index']                 # the temporary directory is gone
  index = json.load(open(prefix + '.index'))
  var = index['variables']['cat_embedding/embedding_weights']
  assert var['full_shape'] == [R, D] and len(var['slices']) == W
  for r, s in enumerate(sorted(var['slices'], key=lambda s: s['phase'])):
    _, rows, offset = sharded_bucket_size(R, W, r)           # variables.py:107-123
    assert s['var_shape'] == [rows, D] and s['var_offset'] == [offset, 0]
    assert (s['stride'], s['phase']) == (W, r)
  small = index['variables']['small_embedding/embedding_weights']
  assert len(small['slices']) == 1 and small['slices'][0]['file'].endswith('data-g0-00000-of-00002')
  np.testing.assert_equal(load_full(prefix, 'cat_embedding/embedding_weights'), table)
  # the reference's "full tensor" is the concatenation of the shards: a permutation of the table
  ref_view = load_full(prefix, 'cat_embedding/embedding_weights', layout='reference')
  np.testing.assert_equal(ref_view, np.concatenate([table[r::W] for r in range(W)]))


@pytest.mark.parametrize('w_save,w_load', [(2, 4), (4, 2), (3, 1), (1, 3), (2, 2), (8, 5)])
def test_round_trip_across_world_sizes(tmp_path, w_save, w_load):
  rng = np.random.RandomState(w_save * 10 + w_load)
  R, D = 1009, 16
  table = rng.randn(R, D).astype(np.float32)
  accum = rng.rand(R, D).astype(np.float32)
  small = rng.randn(7, 4).astype(np.float32)
  prefix = _save(tmp_path, w_save, table, accum, small)
  got_w = [torch.zeros(len(range(r, R, w_load)), D) for r in range(w_load)]
  got_a = [torch.zeros(len(range(r, R, w_load)), D) for r in range(w_load)]
  got_s = [torch.zeros(7, 4) for _ in range(w_load)]
  untouched = [torch.full((3,), 5.0) for _ in range(w_load)]

  def fn(r, saver):
    saver.restore(prefix, {
      'cat_embedding/embedding_weights': ShardedSlice(got_w[r], R, w_load, r),
      'cat_embedding/embedding_weights/Adagrad': ShardedSlice(got_a[r], R, w_load, r),
      'small_embedding/embedding_weights': got_s[r],
      'not_in_the_checkpoint': untouched[r]})
  _run(w_load, fn)
  for r in range(w_load):
    np.testing.assert_equal(got_w[r].numpy(), table[r::w_load])     # the rows this rank owns
    np.testing.assert_equal(got_a[r].numpy(), accum[r::w_load])
    np.testing.assert_equal(got_s[r].numpy(), small)
    assert bool((untouched[r] == 5.0).all())


def test_reference_layout_is_the_contiguous_quirk(tmp_path):
  """layout='reference': what a restore by the reference at another world size reads -- contiguous
  slices of the concatenated shards (variables.py:118-123), i.e. rows of OTHER ids; identical to
  the logical layout only at the world size the checkpoint was written with."""
  rng = np.random.RandomState(9)
  R, D = 101, 4
  table = rng.randn(R, D).astype(np.float32)
  prefix = _save(tmp_path, 2, table, np.zeros((R, D), np.float32), np.zeros((2, 2), np.float32))
  concat = np.concatenate([table[0::2], table[1::2]])
  for w_load in (2, 4):
    for r in range(w_load):
      _, rows, offset = sharded_bucket_size(R, w_load, r)
      t = torch.zeros(rows, D)
      Saver(0, 1).restore(prefix, {'cat_embedding/embedding_weights': ShardedSlice(t, R, w_load, r)},
                          layout='reference')
      np.testing.assert_equal(t.numpy(), concat[offset:offset + rows])
      if w_load == 2:
        np.testing.assert_equal(t.numpy(), table[r::2])
      else:
        assert not np.array_equal(t.numpy(), table[r::4])


def test_errors(tmp_path):
  with pytest.raises(ValueError):
    Saver(0, 2)                                            # multi-rank needs a barrier
  with pytest.raises(ValueError):
    ShardedSlice(torch.zeros(10, 4), 100, 4, 0)            # 25 rows expected
  rng = np.random.RandomState(1)
  prefix = _save(tmp_path, 1, rng.randn(50, 4).astype(np.float32), np.zeros((50, 4), np.float32),
                 np.zeros((2, 2), np.float32))
  with pytest.raises(ValueError):
    Saver(0, 1).restore(prefix, {'cat_embedding/embedding_weights':
                                 ShardedSlice(torch.zeros(30, 4), 60, 2, 0)})   # other bucket size
  with pytest.raises(ValueError):
    Saver(0, 1).restore(prefix, {'small_embedding/embedding_weights': torch.zeros(3, 3)})


def test_fewer_rows_than_ranks_bfloat16_and_overwrite(tmp_path):
  """A 3-row table on 4 ranks (rank 3 holds nothing: an empty data file must load), bfloat16 and
  uint8 tensors, and saving twice under one prefix: the second save replaces the first through
  the index (old data files gone, no file of the live index ever overwritten)."""
  import threading
  from hybridbackend_amd.training.saver import Saver, ShardedSlice, load_full
  prefix = str(tmp_path / 'ckpt')
  W, B = 4, 3
  for gen in range(2):   # (a third save would reuse generation 0's names: never more than two on disk)
    table = torch.arange(B * 2, dtype=torch.float32).reshape(B, 2) + 100 * gen
    bar = threading.Barrier(W)
    errors = []

    def run(r):
      try:
        shard = table[r::W].clone()
        extra = {'bf': torch.tensor([1.5, -2.25, 3.0], dtype=torch.bfloat16) + gen,
                 'u8': torch.tensor([1, 2, 250], dtype=torch.uint8)}
        Saver(r, W, bar.wait).save(prefix, {'t': ShardedSlice(shard, B, W, r), **extra})
      except Exception as e:  # pylint: disable=broad-except
        errors.append(repr(e))
    ts = [threading.Thread(target=run, args=(r,)) for r in range(W)]
    for t in ts:
      t.start()
    for t in ts:
      t.join()
    assert not errors, errors
    np.testing.assert_equal(load_full(prefix, 't'), table.numpy())
    np.testing.assert_equal(load_full(prefix, 't', layout='reference'), table.numpy())
    got = {'t': ShardedSlice(torch.zeros(2, 2), B, 2, 0), 'bf': torch.zeros(3, dtype=torch.bfloat16),
           'u8': torch.zeros(3, dtype=torch.uint8)}
    Saver().restore(prefix, got)
    assert torch.equal(got['t'].tensor, table[0::2])
    assert torch.equal(got['bf'], torch.tensor([1.5, -2.25, 3.0], dtype=torch.bfloat16) + gen)
    assert got['u8'].tolist() == [1, 2, 250]
    # the generation just replaced stays until the NEXT save (a reader holding the old index still
    # finds its data); older ones are gone
    files = sorted(f for f in os.listdir(tmp_path) if '.data-' in f)
    assert len(files) == W * min(gen + 1, 2), files
    assert sum(f'-g{gen}-' in f for f in files) == W, files
