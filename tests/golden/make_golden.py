"""Writes tests/golden/*.json -- the known-answer vectors the oracle is pinned to.

Sources (data only, no reference source text is stored):
  * KATs asserted by the reference's own tests (values transcribed):
      hybridbackend/tensorflow/distribute/tests/alltoall_test.py:219-226 (alltoallv, W=2),
      :254-269 (alltoallv N=2 columns), :228-243 (alltoallv gradient)
  * vectors derived by hand from the reference CPU functors
      hybridbackend/tensorflow/distribute/partition/partition_by_modulo_functors.cc:39-70
      hybridbackend/tensorflow/distribute/partition/partition_by_dual_modulo_functors.cc:37-91
    
  * murmur3 vectors produced by compiling the reference's own header
      hybridbackend/common/murmur3.cu.h:32-77 (oracle/_ref, `make -C oracle ref`);
    when /root/reference is mounted this script re-generates them from that build.
  * a small values+row_splits parquet stand-in for config 1 (generated here).

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402  pylint: disable=wrong-import-position


def dump(name, obj):
  with open(os.path.join(HERE, name), 'w') as f:
    json.dump(obj, f, indent=1, sort_keys=True)
    f.write('\n')


def main():
  # --- murmur3 (from the reference header compiled as-is) ---
  keys = [0, 1, 2, -1, 1234567890123, -9223372036854775807, 42, -42,
          9223372036854775807, 1 << 32, (1 << 32) - 1, 1000003, 65536 * 26]
  ref = oracle.ref_lib()
  if ref is not None:
    hashes = [int(ref.ref_murmur3_hash32_i64(k)) for k in keys]
    src = 'oracle/_ref/libref_murmur3.so (reference header compiled as-is)'
  else:
    hashes = [int(h) for h in oracle.murmur3_hash32(keys)]
    src = 'oracle restatement (reference tree not mounted)'
  dump('murmur3.json', {'source': src, 'seed': 0, 'keys': keys, 'hash32': hashes})

  # --- partition KATs derived from the CPU functor ---
  dump('partition.json', {
    'modulo': [{
      'input': [3, -1, 8, 5, -6, 0, 7], 'num_partitions': 3,
      'output': [3, -6, 0, 7, -1, 8, 5], 'sizes': [3, 1, 3],
      'indices': [0, 4, 5, 6, 1, 2, 3]}],
    'dual': [
      {'stage': 1, 'input': [10, 3, 7, 4, -1, 9, 6, 5], 'num_partitions': 2,
       'modulus': 2, 'output': [10, 4, 6, 3, 7, -1, 9, 5], 'sizes': [3, 5],
       'indices': [0, 3, 4, 1, 5, 6, 2, 7]},
      {'stage': 2, 'input': [10, 4, 6], 'num_partitions': 2, 'modulus': 2,
       'output': [4, 10, 6], 'sizes': [1, 2], 'indices': [1, 0, 2]}],
    # property tests of the reference (partition_test.py:40-65, :83-114): seeds only
    'property': [
      {'seed': 0, 'low': -1000000000, 'high': 1000000000, 'size': 10000,
       'dtype': 'int32', 'num_partitions': 5, 'columns': 1},
      {'seed': 0, 'low': -1000000000, 'high': 1000000000, 'size': 100000,
       'dtype': 'int64', 'num_partitions': 3, 'columns': 10}]})

  # --- alltoallv KATs asserted by the reference tests ---
  dump('alltoallv.json', {
    'single': {  # alltoall_test.py:219-226
      'inputs': [[1, 2, 3], [4, 5, 6]], 'sizes': [[1, 2], [1, 2]],
      'outputs': [[1, 4], [2, 3, 5, 6]], 'out_sizes': [[1, 1], [2, 2]]},
    'n': {  # alltoall_test.py:254-269  rank -> column -> {ids, sizes}
      'inputs': [[[1., 2., 3.], [4., 5., 6.]], [[7., 8., 9.], [10., 11., 12.]]],
      'sizes': [[[1, 2], [2, 1]], [[2, 1], [1, 2]]],
      'outputs': [[[1., 7., 8.], [4., 5., 10.]], [[2., 3., 9.], [6., 11., 12.]]],
      'out_sizes': [[[1, 2], [2, 1]], [[2, 1], [1, 2]]]},
    'grad': {  # alltoall_test.py:228-243
      'sizes': [[5, 1], [3, 4]], 'g': 2.0}})

  # --- config 1 stand-in: a ragged list<int64> column as values + row_splits
  #     (layout of hybridbackend/tensorflow/data/dataframe.py:366-376) with the
  #     oracle's embedding_lookup_sparse(mean) result over a 1000x16 table ---
  rng = np.random.RandomState(20210901)
  lens = rng.poisson(3, size=64).clip(0, 9)
  splits = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
  values = rng.randint(0, 1 << 40, size=int(splits[-1])).astype(np.int64)
  table = rng.uniform(-1e-3, 1e-3, size=(1000, 16)).astype(np.float32)
  outs = oracle.group_lookup_fwd([table], [values], [splits], [1000], ['mean'])
  dump('config1_ragged_lookup.json', {
    'table_seed': 20210901, 'bucket': 1000, 'dim': 16, 'combiner': 'mean',
    'row_splits': splits.tolist(), 'values': values.tolist(),
    'table_f32_hex': table.tobytes().hex(),
    'expected_f32_hex': outs[0].tobytes().hex()})


if __name__ == '__main__':
  main()
