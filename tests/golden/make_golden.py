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
  * tf115_semantics.json: TensorFlow 1.15 is the third-party dependency the fp32 rows R1, R7-R10
    of the path live in (pin: reference README.md:44-48, Makefile:19); its sources are absent from
    /root/reference and nothing can run it here.  Every entry says what it is ("provenance"):
      "published"  numbers printed in the TF 1.15 API documentation, transcribed (docstrings of
                   tf.unique, tf.sparse.segment_sum, tf.math.segment_sum / segment_mean,
                   tf.math.unsorted_segment_sum; the symbolic example of
                   tf.nn.embedding_lookup_sparse);
      "derived"    numbers WE computed from a rule the documentation states in words (segment
                   sum / sqrt(N); floor-mod "follows Python semantics"; "if the segment id is
                   negative the value is dropped"; empty segments give zero rows; the
                   SparseApplyAdagrad / ApplyGradientDescent update rules) -- they pin the
                   restatement to the documented rule, not to TF's binary.
  * config1_ragged_lookup.json: the expectation is numpy float64 arithmetic rounded to fp32 --
    NOT the oracle's output -- so oracle and HIP path are both checked against a third party.

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
  dump('tf115_semantics.json', tf115_semantics())

  # tf.nn.embedding_lookup_sparse(..., combiner=None -> "mean") in float64, rounded once to fp32;
  # empty rows are zero.  (Python's % is TF's FloorMod.)
  rows = values % 1000
  want = np.zeros((64, 16), np.float64)
  for s in range(64):
    seg = rows[splits[s]:splits[s + 1]]
    if len(seg):
      want[s] = table[seg].astype(np.float64).sum(axis=0) / len(seg)
  dump('config1_ragged_lookup.json', {
    'table_seed': 20210901, 'bucket': 1000, 'dim': 16, 'combiner': 'mean',
    'provenance': 'numpy float64 mean over table[values % bucket], rounded to fp32 (no oracle)',
    'rtol': 1e-5,
    'row_splits': splits.tolist(), 'values': values.tolist(),
    'table_f32_hex': table.tobytes().hex(),
    'expected_f32_hex': want.astype(np.float32).tobytes().hex()})


def tf115_semantics():
  """Worked examples published in the TF 1.15 API docs, transcribed as data."""
  c_int = [[1, 2, 3, 4], [-1, -2, -3, -4], [5, 6, 7, 8]]
  return {
    'source': 'TensorFlow 1.15 API documentation (third-party dependency of the reference; '
              'pinned by README.md:44-48 / Makefile:19 of the reference)',
    # tf.unique docstring: "tensor 'x' is [1, 1, 2, 4, 4, 4, 7, 8, 8] ..."
    'unique': [{'provenance': 'published', 'doc': 'tf.unique', 'x': [1, 1, 2, 4, 4, 4, 7, 8, 8],
                'y': [1, 2, 4, 7, 8], 'idx': [0, 0, 1, 2, 2, 2, 3, 4, 4]}],
    # tf.sparse.segment_sum docstring, the four cases on c = [[1,2,3,4],[-1,-2,-3,-4],[5,6,7,8]]
    'sparse_segment_sum': [
      {'provenance': 'published', 'doc': 'tf.sparse.segment_sum: select two rows, one segment', 'data': c_int,
       'indices': [0, 1], 'segment_ids': [0, 0], 'num_segments': None,
       'out': [[0, 0, 0, 0]]},
      {'provenance': 'published', 'doc': 'tf.sparse.segment_sum: select two rows, two segments', 'data': c_int,
       'indices': [0, 1], 'segment_ids': [0, 1], 'num_segments': None,
       'out': [[1, 2, 3, 4], [-1, -2, -3, -4]]},
      {'provenance': 'published', 'doc': 'tf.sparse.segment_sum: with missing segment ids', 'data': c_int,
       'indices': [0, 1], 'segment_ids': [0, 2], 'num_segments': 4,
       'out': [[1, 2, 3, 4], [0, 0, 0, 0], [-1, -2, -3, -4], [0, 0, 0, 0]]},
      {'provenance': 'published', 'doc': 'tf.sparse.segment_sum: select all rows, two segments', 'data': c_int,
       'indices': [0, 1, 2], 'segment_ids': [0, 0, 1], 'num_segments': None,
       'out': [[0, 0, 0, 0], [5, 6, 7, 8]]}],
    # tf.math.segment_mean docstring (sparse_segment_mean = "like SegmentMean, but segment_ids
    # can have rank less than data's first dimension, selecting a subset ... by indices")
    'segment_mean': [
      {'provenance': 'published', 'doc': 'tf.math.segment_mean', 'data': [[1.0, 2, 3, 4], [4, 3, 2, 1], [5, 6, 7, 8]],
       'indices': [0, 1, 2], 'segment_ids': [0, 0, 1],
       'out': [[2.5, 2.5, 2.5, 2.5], [5, 6, 7, 8]]}],
    # tf.sparse.segment_sqrt_n: "the sum along sparse segments divided by the sqrt of N, N the
    # size of the segment" -- evaluated on the segment_mean example's data
    'segment_sqrt_n': [
      {'provenance': 'derived', 'doc': 'tf.sparse.segment_sqrt_n (definition: sum / sqrt(N))',
       'data': [[1.0, 2, 3, 4], [4, 3, 2, 1], [5, 6, 7, 8]],
       'indices': [0, 1, 2], 'segment_ids': [0, 0, 1],
       'out_times_sqrt_n': [[5.0, 5, 5, 5], [5, 6, 7, 8]], 'n': [2, 1]}],
    # tf.math.unsorted_segment_sum docstring
    'unsorted_segment_sum': [
      {'provenance': 'published', 'doc': 'tf.math.unsorted_segment_sum',
       'data': [[1, 2, 3, 4], [5, 6, 7, 8], [4, 3, 2, 1]],
       'segment_ids': [0, 1, 0], 'num_segments': 2, 'out': [[5, 5, 5, 5], [5, 6, 7, 8]]},
      # "If the given segment ID i is negative, the value is dropped"; "If the sum is empty for a
      # given segment ID i, output[i] = 0" -- the sentences of the same docstring, applied
      {'provenance': 'derived', 'doc': 'tf.math.unsorted_segment_sum: negative ids dropped, absent ids zero',
       'data': [[1, 2, 3, 4], [5, 6, 7, 8], [4, 3, 2, 1], [9, 9, 9, 9]],
       'segment_ids': [3, -1, 3, 0], 'num_segments': 5,
       'out': [[9, 9, 9, 9], [0, 0, 0, 0], [0, 0, 0, 0], [5, 5, 5, 5], [0, 0, 0, 0]]}],
    # tf.math.segment_sum docstring: c = [[1,2,3,4],[4,3,2,1],[5,6,7,8]], segment_ids [0,0,1]
    'segment_sum': [
      {'provenance': 'published', 'doc': 'tf.math.segment_sum',
       'data': [[1, 2, 3, 4], [4, 3, 2, 1], [5, 6, 7, 8]], 'indices': [0, 1, 2],
       'segment_ids': [0, 0, 1], 'out': [[5, 5, 5, 5], [5, 6, 7, 8]]}],
    # tf.nn.embedding_lookup_sparse docstring: ids at [0,0]:1 [0,1]:3 [1,0]:0 [2,3]:1 ->
    # output[0] = combine(params[1], params[3]), output[1] = params[0], output[2] = params[1];
    # sp_weights=None is "all weights 1"; combiner=None defaults to "mean".
    'embedding_lookup_sparse': [
      {'provenance': 'published', 'doc': 'tf.nn.embedding_lookup_sparse (unit weights)',
       'sp_indices': [[0, 0], [0, 1], [1, 0], [2, 3]], 'sp_ids': [1, 3, 0, 1],
       'dense_shape': [3, 4], 'row_splits': [0, 2, 3, 4],
       'rows_of_output': [[1, 3], [0], [1]], 'default_combiner': 'mean'},
      # the same call with an EMPTY row in the middle: its segment is empty, and an empty segment
      # of sparse_segment_* is a zero row (the "with missing segment ids" case above)
      {'provenance': 'derived', 'doc': 'tf.nn.embedding_lookup_sparse with an empty row',
       'sp_indices': [[0, 0], [0, 1], [2, 0], [3, 3]], 'sp_ids': [1, 3, 0, 1],
       'dense_shape': [4, 4], 'row_splits': [0, 2, 2, 3, 4],
       'rows_of_output': [[1, 3], [], [0], [1]], 'default_combiner': 'mean'}],
    # tf.math.floormod: "the result is consistent with a flooring divide ... follows Python
    # semantics": floor(x / y) * y + mod(x, y) = x
    'floormod': [{'provenance': 'derived', 'doc': 'tf.math.floormod (Python semantics)', 'x': x, 'y': y, 'out': x % y}
                 for x, y in ((7, 3), (-7, 3), (-1, 1000000), (0, 5), (-1000000, 1000000),
                              (-(1 << 40) - 1, 1000000), ((1 << 40) + 123, 1000000),
                              (-9223372036854775807, 1000003), (9223372036854775807, 1000003))],
    # op docs: ApplyGradientDescent "var -= alpha * delta"; SparseApplyAdagrad "for rows we have
    # grad for: accum += grad * grad; var -= lr * grad * (1 / sqrt(accum))"
    'sparse_apply': [
      {'provenance': 'derived', 'doc': 'SparseApplyAdagrad / ApplyGradientDescent update rules', 'var': [[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]],
       'accum': [[0.1, 0.1], [0.1, 0.1], [0.1, 0.1]], 'indices': [2, 0],
       'grad': [[0.5, -1.0], [2.0, 0.25]], 'lr': 0.5}],
  }


if __name__ == '__main__':
  main()
