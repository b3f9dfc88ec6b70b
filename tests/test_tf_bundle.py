"""TensorFlow tensor-bundle checkpoints read / written on the host (training/tf_bundle.py; the
files the reference saves and restores, hybridbackend/tensorflow/training/saver.py:97-246).

No TensorFlow here and no checkpoint fixture in the reference tree: the format is pinned by its
published vectors (RFC 3720 B.4 for CRC-32C, the OrderedCode examples and its ordering property,
leveldb's table layout spelled out byte by byte below) and by round trips."""
import os
import struct
import threading

import numpy as np
import pytest
import torch

from hybridbackend_amd.training import Saver
from hybridbackend_amd.training import ShardedSlice
from hybridbackend_amd.training import export_reference
from hybridbackend_amd.training import tf_bundle as tb


def test_crc32c_published_vectors():
  # RFC 3720 (iSCSI) appendix B.4
  assert tb.crc32c(bytes(32)) == 0x8a9136aa
  assert tb.crc32c(b'\xff' * 32) == 0x62a8ab43
  assert tb.crc32c(bytes(range(32))) == 0x46dd794e
  assert tb.crc32c(bytes(range(31, -1, -1))) == 0x113fdb5c
  assert tb.crc32c(b'123456789') == 0xe3069283
  # continuing a message, arrays, unaligned starts
  assert tb.crc32c(b'6789', tb.crc32c(b'12345')) == 0xe3069283
  a = np.frombuffer(bytes(range(64)), np.uint8)
  assert tb.crc32c(a[1:33].copy()) == tb.crc32c(bytes(range(1, 33)))
  assert tb.crc32c(np.arange(32, dtype=np.uint8)) == 0x46dd794e
  assert tb.crc32c(b'') == 0
  # the mask (crc32c.h): rotate right by 15 bits, add a constant; invertible
  for v in (0, 1, 0xe3069283, 0xffffffff):
    m = tb.mask_crc(v)
    assert m == (((v >> 15) | (v << 17)) + 0xa282ead8) & 0xffffffff
    assert tb.unmask_crc(m) == v
    assert m != v


def test_ordered_code_examples_and_ordering():
  # ordered_code.cc: one byte for [-64, 64), the header bits give the length
  cases = {0: '80', 1: '81', -1: '7f', 63: 'bf', -64: '40', 64: 'c040', -65: '3fbf',
           8191: 'dfff', 8192: 'e02000', -8192: '2000', -8193: '1fdfff'}
  for v, hx in cases.items():
    assert tb._oc_write_signed(v).hex() == hx, v
    assert tb._oc_read_signed(bytes.fromhex(hx), 0) == (v, len(hx) // 2)
  rng = np.random.RandomState(3)
  vals = [int(x) for x in rng.randint(-2**62, 2**62, size=400)]
  vals += [int(s * (1 << b) + d) for b in range(0, 63) for s in (-1, 1) for d in (-1, 0, 1)]
  vals += [-2**63, 2**63 - 1]
  enc = {v: tb._oc_write_signed(v) for v in set(vals)}
  for v, e in enc.items():
    assert tb._oc_read_signed(e + b'\x55', 0) == (v, len(e))
  # THE property of the code: bytewise order of the encodings == numeric order
  assert sorted(enc, key=lambda v: enc[v]) == sorted(enc)
  # unsigned numbers: length byte + big-endian digits; strings: escapes + terminator
  assert tb._oc_write_num(0) == b'\x00' and tb._oc_write_num(2) == b'\x01\x02'
  assert tb._oc_write_num(0x1234) == b'\x02\x12\x34'
  assert tb._oc_write_string(b'a\x00b\xffc') == b'a\x00\xffb\xff\x00c\x00\x01'
  key = tb.encode_slice_key('emb/w\x00\xff', [(0, -1), (5, 7), (1000000, 12345678901)])
  assert tb.decode_slice_key(key) == ('emb/w\x00\xff', [(0, -1), (5, 7), (1000000, 12345678901)])
  # slice keys sort after the header key "" and before every tensor name
  assert b'' < key < b'\x01' <= b'a'


def _trailer(block):
  return block + b'\x00' + struct.pack('<I', tb.mask_crc(tb.crc32c(block + b'\x00')))


def test_reads_an_index_assembled_by_hand(tmp_path):
  """One float tensor ``a = [1.5, -2]``: every byte of the index written out here from the format
  descriptions (table_format.txt, tensor_bundle.proto), none of it by BundleWriter."""
  data = np.array([1.5, -2.0], np.float32)
  prefix = str(tmp_path / 'ck')
  with open(prefix + '.data-00000-of-00001', 'wb') as f:
    f.write(b'\xaa' * 3 + data.tobytes())                   # the tensor starts at offset 3
  header = bytes.fromhex('0801' '1a020801')                 # num_shards: 1, version { producer: 1 }
  entry = (bytes.fromhex('0801'                             # dtype: DT_FLOAT
                         '1204' '1202' '0802'               # shape { dim { size: 2 } }
                         '2003'                             # offset: 3
                         '2808'                             # size: 8
                         '35') +                            # crc32c (fixed32), masked
           struct.pack('<I', tb.mask_crc(tb.crc32c(data.tobytes()))))
  # data block: (shared, non_shared, value_len, key delta, value) x 2, restarts [0], 1 restart
  block = (bytes([0, 0, len(header)]) + header +
           bytes([0, 1, len(entry)]) + b'a' + entry +
           struct.pack('<II', 0, 1))
  meta = struct.pack('<II', 0, 1)                           # empty metaindex block
  blob = _trailer(block)
  h_data = bytes([0, len(block)])                           # BlockHandle: varint offset, size
  off_meta = len(blob)
  blob += _trailer(meta)
  index = bytes([0, 1, len(h_data)]) + b'a' + h_data + struct.pack('<II', 0, 1)
  off_index = len(blob)
  blob += _trailer(index)
  footer = bytes([off_meta, len(meta)]) + bytes([off_index, len(index)])
  footer += b'\x00' * (40 - len(footer)) + bytes.fromhex('57fb808b247547db')
  with open(prefix + '.index', 'wb') as f:
    f.write(blob + footer)
  r = tb.BundleReader(prefix)
  assert r.names() == ['a'] and r.header == {'num_shards': 1, 'endianness': 0, 'producer': 1}
  assert r.dtype_and_shape('a') == (np.dtype(np.float32), (2,))
  np.testing.assert_equal(r.read('a'), data)
  # what BundleWriter makes of the same tensor is read the same way
  w = tb.BundleWriter(str(tmp_path / 'ck2'))
  w.add('a', data)
  w.finish()
  np.testing.assert_equal(tb.BundleReader(str(tmp_path / 'ck2')).read('a'), data)


def test_snappy_blocks(tmp_path):
  # format_description.txt by hand: length 13; literal "abcd"; copy (1-byte offset form) of 8
  # bytes from 4 back -- overlapping; literal "X"
  comp = bytes([13, (4 - 1) << 2]) + b'abcd' + bytes([((8 - 4) << 2) | 1, 4, 0]) + b'X'
  assert tb.snappy_uncompress(comp) == b'abcdabcdabcdX'
  # 2-byte-offset copy and a long literal (length in one extra byte)
  lit = bytes(range(200)) * 2                                   # 400 bytes
  comp = (tb._put_varint(400 + 70) + bytes([61 << 2]) + (400 - 1).to_bytes(2, 'little') + lit +
          bytes([((64 - 1) << 2) | 2]) + (400).to_bytes(2, 'little') +
          bytes([((6 - 1) << 2) | 2]) + (200).to_bytes(2, 'little'))
  assert tb.snappy_uncompress(comp) == lit + lit[:64] + (lit + lit[:64])[-200:][:6]
  with pytest.raises(ValueError):
    tb.snappy_uncompress(bytes([5, 0]) + b'a')                  # 1 byte where 5 are promised
  with pytest.raises(ValueError):
    tb.snappy_uncompress(bytes([4, (4 << 2) | 1, 9]))           # reference before the start
  # an index whose data block is stored compressed (all literals: a valid snappy stream)
  prefix = str(tmp_path / 'ck')
  w = tb.BundleWriter(prefix)
  w.add('x', np.arange(6, dtype=np.float32))
  w.finish()
  raw = open(prefix + '.index', 'rb').read()
  entries = tb._read_table(prefix + '.index', True)
  block = tb._block_bytes(entries)
  assert raw.startswith(block)
  comp = tb._put_varint(len(block)) + bytes([60 << 2, len(block) - 1]) + block
  blob = comp + b'\x01' + struct.pack('<I', tb.mask_crc(tb.crc32c(comp + b'\x01')))
  off_meta = len(blob)
  meta = struct.pack('<II', 0, 1)
  blob += _trailer(meta)
  index = (bytes([0, 1]) + tb._put_varint(len(tb._put_varint(0) + tb._put_varint(len(comp)))) + b'x' +
           tb._put_varint(0) + tb._put_varint(len(comp)) + struct.pack('<II', 0, 1))
  off_index = len(blob)
  blob += _trailer(index)
  footer = (tb._put_varint(off_meta) + tb._put_varint(len(meta)) + tb._put_varint(off_index) +
            tb._put_varint(len(index)))
  footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', tb.TABLE_MAGIC)
  open(prefix + '.index', 'wb').write(blob + footer)
  np.testing.assert_equal(tb.BundleReader(prefix).read('x'), np.arange(6, dtype=np.float32))


def test_round_trip_many_tensors_dtypes_and_slices(tmp_path):
  rng = np.random.RandomState(11)
  prefix = str(tmp_path / 'model.ckpt-7')
  w = tb.BundleWriter(prefix)
  want = {}
  for i in range(300):                       # several data blocks, long shared key prefixes
    name = f'tower/layer_{i // 7}/dense_{i}/kernel'
    want[name] = rng.randn(rng.randint(1, 5), rng.randint(1, 4)).astype(np.float32)
  want['global_step'] = np.array(123456789012, np.int64)            # a scalar: shape []
  want['empty'] = np.zeros((0, 16), np.float32)
  want['f64'] = rng.randn(3)
  want['i32'] = rng.randint(-9, 9, size=(2, 2, 2)).astype(np.int32)
  want['u8'] = rng.randint(0, 255, size=33).astype(np.uint8)
  want['half'] = rng.randn(5).astype(np.float16)
  want['flag'] = np.array([True, False])
  for name, arr in want.items():
    w.add(name, arr)
  bf = (rng.randn(4, 2).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
  w.add('bf16', bf, bfloat16=True)
  # an embedding table saved as 3 row slices + its Adagrad slot; a column-sliced matrix
  table = rng.randn(10, 4).astype(np.float32)
  for lo, hi in ((0, 4), (4, 7), (7, 10)):
    w.add_slice('emb/embedding_weights', table.shape, (lo, 0), table[lo:hi])
    w.add_slice('emb/embedding_weights/Adagrad', table.shape, (lo, 0), table[lo:hi] * 2)
  mat = rng.randn(3, 8)
  w.add_slice('cols', mat.shape, (0, 5), mat[:, 5:])
  w.add_slice('cols', mat.shape, (0, 0), mat[:, :5])
  with pytest.raises(ValueError):
    w.add('global_step', np.zeros(1))
  with pytest.raises(ValueError):
    w.add_slice('emb/embedding_weights', table.shape, (8, 0), table[:4])
  assert w.finish() == prefix
  assert sorted(os.listdir(tmp_path)) == ['model.ckpt-7.data-00000-of-00001', 'model.ckpt-7.index']
  r = tb.BundleReader(prefix)
  assert set(r.names()) == set(want) | {'bf16', 'emb/embedding_weights',
                                        'emb/embedding_weights/Adagrad', 'cols'}
  for name, arr in want.items():
    got = r.read(name)
    assert got.dtype == arr.dtype and got.shape == arr.shape, name
    np.testing.assert_equal(got, arr)
  assert r.is_bfloat16('bf16') and not r.is_bfloat16('half')
  np.testing.assert_equal(r.read('bf16'), bf)
  np.testing.assert_equal(r.read('emb/embedding_weights'), table)
  np.testing.assert_equal(r.read('emb/embedding_weights/Adagrad'), table * 2)
  np.testing.assert_equal(r.read('cols'), mat)
  parts = r.slices('emb/embedding_weights')
  assert [ext for ext, _ in parts] == [[(0, 4), (0, -1)], [(4, 3), (0, -1)], [(7, 3), (0, -1)]]
  assert r.slices('f64') == []


def test_corruption_is_detected(tmp_path):
  prefix = str(tmp_path / 'ck')
  w = tb.BundleWriter(prefix)
  w.add('x', np.arange(100, dtype=np.float32))
  w.finish()
  data = prefix + '.data-00000-of-00001'
  raw = bytearray(open(data, 'rb').read())
  raw[17] ^= 1
  open(data, 'wb').write(raw)
  with pytest.raises(ValueError, match='checksum'):
    tb.BundleReader(prefix).read('x')
  assert tb.BundleReader(prefix, verify=False).read('x').shape == (100,)
  raw[17] ^= 1
  open(data, 'wb').write(raw[:-4])
  with pytest.raises(ValueError, match='truncated'):
    tb.BundleReader(prefix).read('x')
  idx = bytearray(open(prefix + '.index', 'rb').read())
  bad = bytearray(idx)
  bad[5] ^= 0x40
  open(prefix + '.index', 'wb').write(bad)
  with pytest.raises(ValueError, match='checksum'):
    tb.BundleReader(prefix)
  bad = bytearray(idx)
  bad[-1] ^= 1
  open(prefix + '.index', 'wb').write(bad)
  with pytest.raises(ValueError, match='magic'):
    tb.BundleReader(prefix)
  open(prefix + '.index', 'wb').write(idx[:20])
  with pytest.raises(ValueError):
    tb.BundleReader(prefix)


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


@pytest.mark.parametrize('w_save,w_load', [(3, 2), (1, 4), (4, 1), (2, 2)])
def test_reference_checkpoint_bridge(tmp_path, w_save, w_load):
  """A checkpoint of ours -> the bundle the reference would have saved at the same world size
  (slices at the reference's contiguous offsets, variables.py:118-123) -> restored here at
  another world size: every rank ends up with ITS ids' rows."""
  rng = np.random.RandomState(w_save * 10 + w_load)
  R, D = 1003, 8
  table = rng.randn(R, D).astype(np.float32)
  accum = rng.rand(R, D).astype(np.float32)
  small = rng.randn(5, 4).astype(np.float32)
  ours = str(tmp_path / 'ours.ckpt')
  name = 'cat_embedding/embedding_weights'

  def save(r, saver):
    saver.save(ours, {
      name: ShardedSlice(torch.from_numpy(table[r::w_save].copy()), R, w_save, r),
      name + '/Adagrad': ShardedSlice(torch.from_numpy(accum[r::w_save].copy()), R, w_save, r),
      'small/embedding_weights': torch.from_numpy(small.copy())})
  _run(w_save, save)
  theirs = export_reference(ours, str(tmp_path / 'theirs.ckpt'))
  # one data file per rank that saved (what MergeV2Checkpoints leaves), every shard in its own
  assert sorted(f for f in os.listdir(tmp_path) if f.startswith('theirs')) == (
      [f'theirs.ckpt.data-{k:05d}-of-{w_save:05d}' for k in range(w_save)] + ['theirs.ckpt.index'])
  reader = tb.BundleReader(theirs)
  assert reader.header['num_shards'] == w_save
  if w_save > 1:
    assert [e['shard_id'] for _, e in sorted(reader._slice_data[name],
                                             key=lambda x: x[0][0][0])] == list(range(w_save))
  # TF's full tensor = the shards back to back (the reference's view); by id = de-interleaved
  concat = np.concatenate([table[r::w_save] for r in range(w_save)])
  np.testing.assert_equal(reader.read(name), concat)
  np.testing.assert_equal(tb.read_reference_table(reader, name, 'reference'), concat)
  np.testing.assert_equal(tb.read_reference_table(reader, name, 'logical'), table)
  offs = [ext[0][0] for ext, _ in reader.slices(name)] if w_save > 1 else []
  q, rem = divmod(R, w_save)
  assert offs == [q * r + min(r, rem) for r in range(w_save)][:len(offs)]
  got = [None] * w_load

  def load(r, saver):
    rows = R // w_load + (r < R % w_load)
    vs = {name: ShardedSlice(torch.zeros(rows, D), R, w_load, r),
          'renamed/Adagrad': ShardedSlice(torch.zeros(rows, D), R, w_load, r),
          'small/embedding_weights': torch.zeros(5, 4),
          'not_in_the_checkpoint': torch.full((2,), 7.0)}
    saver.restore_reference(theirs, vs, names={'renamed/Adagrad': name + '/Adagrad'})
    got[r] = vs
  _run(w_load, load)
  for r in range(w_load):
    np.testing.assert_equal(got[r][name].tensor.numpy(), table[r::w_load])
    np.testing.assert_equal(got[r]['renamed/Adagrad'].tensor.numpy(), accum[r::w_load])
    np.testing.assert_equal(got[r]['small/embedding_weights'].numpy(), small)
    np.testing.assert_equal(got[r]['not_in_the_checkpoint'].numpy(), [7.0, 7.0])


def test_reference_table_that_was_not_row_sharded_by_id(tmp_path):
  prefix = str(tmp_path / 'ck')
  w = tb.BundleWriter(prefix)
  t = np.arange(40, dtype=np.float32).reshape(10, 4)
  w.add_slice('t', t.shape, (0, 0), t[:7])        # 7 + 3 rows: not what id mod 2 gives (5 + 5)
  w.add_slice('t', t.shape, (7, 0), t[7:])
  w.finish()
  r = tb.BundleReader(prefix)
  np.testing.assert_equal(tb.read_reference_table(r, 't', 'reference'), t)
  with pytest.raises(ValueError, match='row-sharded'):
    tb.read_reference_table(r, 't', 'logical')
