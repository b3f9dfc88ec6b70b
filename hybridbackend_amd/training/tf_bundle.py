"""TensorFlow tensor-bundle checkpoints (``<prefix>.index`` + ``<prefix>.data-SSSSS-of-NNNNN``) read
and written on the host -- the files the reference saves (``hybridbackend/tensorflow/training/
saver.py:97-185`` drives TF's ``SaveV2`` / ``MergeV2Checkpoints``) and restores
(``saver.py:187-246``), so that tables trained with the reference can be loaded here and tables
trained here handed back.

PARITY UNPINNED against TensorFlow's binary: TensorFlow is in neither this image nor the GPU box
and the reference tree holds no checkpoint fixture, so nothing TF wrote was ever read by this
module.  It follows the published formats, each pinned as far as a published vector goes:

* the index is a leveldb table (``tensorflow/core/lib/io/table_format.txt`` = leveldb's
  ``doc/table_format.md``): data blocks of prefix-compressed entries (varint32 shared /
  non-shared / value length, key delta, value) + restart array, a 5-byte trailer per block
  (compression type, masked CRC-32C), metaindex + index block, 48-byte footer ending in the
  magic ``0xdb4775248b80fb57``.  TF writes the bundle index uncompressed; a snappy block (the
  default of leveldb tables in general) is decompressed all the same;
* keys and values (``tensorflow/core/protobuf/tensor_bundle.proto``): key ``""`` ->
  ``BundleHeaderProto`` (num_shards, endianness, version), tensor name -> ``BundleEntryProto``
  (dtype, shape, shard_id, offset, size, masked crc32c, slices).  A partitioned variable -- what
  ``SaveSliceInfo`` makes of an embedding shard, ``embedding/variables.py:126-141`` -- has a full
  entry that only lists its slices; each slice's bytes live under the key
  ``EncodeTensorNameSlice`` (``tensorflow/core/util/saved_tensor_slice_util.cc``: OrderedCode
  ``0, name, rank, (start, length) per dimension``) -- decoded here rather than re-encoded, so
  the convention for a full extent does not matter to the reader;
* CRC-32C and its mask (``tensorflow/core/lib/hash/crc32c.h``: rotate right by 15, add
  ``0xa282ead8``): RFC 3720 B.4 vectors, ``tests/test_tf_bundle.py``.

What HybridBackend's row sharding means for the rows is the saver's business
(``training/saver.py``: slices are contiguous in the file, strided in ownership);
``read_reference_table`` applies it.
"""
import os
import struct

import numpy as np

from hybridbackend_amd import _lib

TABLE_MAGIC = 0xdb4775248b80fb57
_MASK_DELTA = 0xa282ead8

# tensorflow/core/framework/types.proto (DataType)
_DT = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8,
       9: np.int64, 10: np.bool_, 17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64,
       14: np.uint16}          # 14 = DT_BFLOAT16: numpy has none, the bit patterns travel
DT_BFLOAT16 = 14
_DT_OF = {np.dtype(v): k for k, v in _DT.items() if k != DT_BFLOAT16}


def crc32c(data, crc=0):
  """CRC-32C of bytes / a contiguous numpy array, continuing from ``crc``."""
  if isinstance(data, np.ndarray):
    if data.nbytes == 0:
      return crc
    return _lib.lib().hbk_host_crc32c(crc, data.ctypes.data, data.nbytes)
  data = bytes(data)
  return _lib.lib().hbk_host_crc32c(crc, data, len(data))


def mask_crc(crc):
  return (((crc >> 15) | (crc << 17)) + _MASK_DELTA) & 0xffffffff


def unmask_crc(masked):
  rot = (masked - _MASK_DELTA) & 0xffffffff
  return ((rot >> 17) | (rot << 15)) & 0xffffffff


# ---------------------------------------------------------------------------------------------
# varints and the few protobuf messages of tensor_bundle.proto / tensor_shape.proto /
# tensor_slice.proto / versions.proto
def _get_varint(buf, pos):
  out, shift = 0, 0
  while True:
    if pos >= len(buf):
      raise ValueError('truncated varint')
    b = buf[pos]
    pos += 1
    out |= (b & 0x7f) << shift
    if b < 0x80:
      return out, pos
    shift += 7
    if shift > 63:
      raise ValueError('varint longer than 10 bytes')


def _put_varint(v):
  v &= (1 << 64) - 1          # negative int64: two's complement, 10 bytes
  out = bytearray()
  while v >= 0x80:
    out.append((v & 0x7f) | 0x80)
    v >>= 7
  out.append(v)
  return bytes(out)


def _signed64(v):
  return v - (1 << 64) if v >= (1 << 63) else v


def _fields(buf):
  """(field number, wire type, value) of one message; value: int (varint / fixed) or bytes."""
  pos, out = 0, []
  while pos < len(buf):
    tag, pos = _get_varint(buf, pos)
    num, wire = tag >> 3, tag & 7
    if wire == 0:
      val, pos = _get_varint(buf, pos)
    elif wire == 1:
      val = struct.unpack_from('<Q', buf, pos)[0]
      pos += 8
    elif wire == 2:
      n, pos = _get_varint(buf, pos)
      if pos + n > len(buf):
        raise ValueError('truncated length-delimited field')
      val = bytes(buf[pos:pos + n])
      pos += n
    elif wire == 5:
      val = struct.unpack_from('<I', buf, pos)[0]
      pos += 4
    else:
      raise ValueError(f'unsupported protobuf wire type {wire}')
    out.append((num, wire, val))
  return out


def _tag(num, wire):
  return _put_varint(num << 3 | wire)


def _bytes_field(num, payload):
  return _tag(num, 2) + _put_varint(len(payload)) + payload


def _parse_shape(buf):
  dims = []
  for num, _, val in _fields(buf):
    if num == 2:              # repeated Dim { int64 size = 1; string name = 2; }
      size = 0
      for n2, _, v2 in _fields(val):
        if n2 == 1:
          size = _signed64(v2)
      dims.append(size)
    elif num == 3 and val:    # unknown_rank
      raise ValueError('tensor of unknown rank in a checkpoint')
  return dims


def _encode_shape(dims):
  return b''.join(_bytes_field(2, (_tag(1, 0) + _put_varint(d)) if d else b'') for d in dims)


def _parse_slice(buf):
  """TensorSliceProto -> [(start, length)]; a full extent is (0, -1)."""
  ext = []
  for num, _, val in _fields(buf):
    if num == 1:              # repeated Extent { int64 start = 1; oneof { int64 length = 2; } }
      start, length = 0, -1
      for n2, _, v2 in _fields(val):
        if n2 == 1:
          start = _signed64(v2)
        elif n2 == 2:
          length = _signed64(v2)
      ext.append((start, length))
  return ext


def _encode_slice(extents):
  out = b''
  for start, length in extents:
    e = b''
    if length >= 0:
      if start:
        e += _tag(1, 0) + _put_varint(start)
      e += _tag(2, 0) + _put_varint(length)
    out += _bytes_field(1, e)
  return out


def _parse_entry(buf):
  e = {'dtype': 0, 'shape': [], 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': None,
       'slices': []}
  for num, _, val in _fields(buf):
    if num == 1:
      e['dtype'] = val
    elif num == 2:
      e['shape'] = _parse_shape(val)
    elif num == 3:
      e['shard_id'] = val
    elif num == 4:
      e['offset'] = _signed64(val)
    elif num == 5:
      e['size'] = _signed64(val)
    elif num == 6:
      e['crc32c'] = val
    elif num == 7:
      e['slices'].append(_parse_slice(val))
  return e


def _encode_entry(e):
  out = _tag(1, 0) + _put_varint(e['dtype']) + _bytes_field(2, _encode_shape(e['shape']))
  if e.get('shard_id'):
    out += _tag(3, 0) + _put_varint(e['shard_id'])
  if e.get('offset'):
    out += _tag(4, 0) + _put_varint(e['offset'])
  if e.get('size'):
    out += _tag(5, 0) + _put_varint(e['size'])
  if e.get('crc32c') is not None:
    out += _tag(6, 5) + struct.pack('<I', e['crc32c'])
  for sl in e.get('slices', []):
    out += _bytes_field(7, _encode_slice(sl))
  return out


# ---------------------------------------------------------------------------------------------
# OrderedCode (tensorflow/core/lib/strings/ordered_code.cc), as far as slice keys use it
_HEADER_BITS = [(0, 0), (0x80, 0), (0xc0, 0), (0xe0, 0), (0xf0, 0), (0xf8, 0), (0xfc, 0),
                (0xfe, 0), (0xff, 0), (0xff, 0x80), (0xff, 0xc0)]


def _oc_write_num(v):
  body = v.to_bytes(8, 'big').lstrip(b'\x00')
  return bytes([len(body)]) + body


def _oc_read_num(buf, pos):
  n = buf[pos]
  if n > 8 or pos + 1 + n > len(buf):
    raise ValueError('bad OrderedCode number')
  return int.from_bytes(buf[pos + 1:pos + 1 + n], 'big'), pos + 1 + n


def _oc_write_string(s):
  out = bytearray()
  for b in s:
    if b == 0:
      out += b'\x00\xff'
    elif b == 0xff:
      out += b'\xff\x00'
    else:
      out.append(b)
  return bytes(out) + b'\x00\x01'


def _oc_read_string(buf, pos):
  out = bytearray()
  while True:
    if pos >= len(buf):
      raise ValueError('unterminated OrderedCode string')
    b = buf[pos]
    if b == 0:
      nxt = buf[pos + 1]
      if nxt == 1:
        return bytes(out), pos + 2
      if nxt != 0xff:
        raise ValueError('bad OrderedCode escape')
      out.append(0)
      pos += 2
    elif b == 0xff:
      if buf[pos + 1] != 0:
        raise ValueError('bad OrderedCode escape')
      out.append(0xff)
      pos += 2
    else:
      out.append(b)
      pos += 1


def _oc_write_signed(v):
  x = ~v if v < 0 else v
  if x < 64:
    return bytes([(0x80 ^ v) & 0xff])
  n = x.bit_length() // 7 + 1    # n bytes = n header bits + sign + 7 n - 1 magnitude bits
  raw = bytearray((v & ((1 << 80) - 1)).to_bytes(10, 'big')[10 - n:])
  raw[0] ^= _HEADER_BITS[n][0]
  if n > 1:
    raw[1] ^= _HEADER_BITS[n][1]
  return bytes(raw)


def _oc_read_signed(buf, pos):
  first = buf[pos]
  negative = (first & 0x80) == 0
  two = (first << 8) | (buf[pos + 1] if pos + 1 < len(buf) else 0)
  x = (~two & 0xffff) if negative else two
  n = 0
  while n < 11 and x & (0x8000 >> n):
    n += 1
  if n < 1 or n > 10 or pos + n > len(buf):
    raise ValueError('bad OrderedCode signed number')
  raw = bytearray(buf[pos:pos + n])
  raw[0] ^= _HEADER_BITS[n][0]
  if n > 1:
    raw[1] ^= _HEADER_BITS[n][1]
  return int.from_bytes(raw, 'big', signed=True), pos + n


def encode_slice_key(name, extents):
  """``EncodeTensorNameSlice``: the index key of one slice of tensor ``name``."""
  out = _oc_write_num(0) + _oc_write_string(name.encode()) + _oc_write_num(len(extents))
  for start, length in extents:
    out += _oc_write_signed(start) + _oc_write_signed(length)
  return out


def decode_slice_key(key):
  tag, pos = _oc_read_num(key, 0)
  if tag != 0:
    raise ValueError('not a slice key')
  name, pos = _oc_read_string(key, pos)
  rank, pos = _oc_read_num(key, pos)
  ext = []
  for _ in range(rank):
    start, pos = _oc_read_signed(key, pos)
    length, pos = _oc_read_signed(key, pos)
    ext.append((start, length))
  if pos != len(key):
    raise ValueError('trailing bytes in a slice key')
  return name.decode(), ext


# ---------------------------------------------------------------------------------------------
# the leveldb table that is the index
def snappy_uncompress(buf):
  """Raw snappy (format_description.txt): varint length, then literals and back references.
  TensorFlow writes the bundle index uncompressed (``tensor_bundle.cc``: kNoCompression, on
  purpose); leveldb tables in general default to snappy, so a compressed block is read, not
  refused."""
  n, pos = _get_varint(buf, 0)
  out = bytearray()
  while pos < len(buf):
    tag = buf[pos]
    pos += 1
    kind = tag & 3
    if kind == 0:
      ln = tag >> 2
      if ln >= 60:
        extra = ln - 59
        ln = int.from_bytes(buf[pos:pos + extra], 'little')
        pos += extra
      ln += 1
      if pos + ln > len(buf):
        raise ValueError('snappy: literal runs past the input')
      out += buf[pos:pos + ln]
      pos += ln
      continue
    if kind == 1:
      ln = 4 + ((tag >> 2) & 7)
      off = ((tag >> 5) << 8) | buf[pos]
      pos += 1
    elif kind == 2:
      ln = 1 + (tag >> 2)
      off = int.from_bytes(buf[pos:pos + 2], 'little')
      pos += 2
    else:
      ln = 1 + (tag >> 2)
      off = int.from_bytes(buf[pos:pos + 4], 'little')
      pos += 4
    if off == 0 or off > len(out):
      raise ValueError('snappy: bad back reference')
    if off >= ln:
      start = len(out) - off
      out += out[start:start + ln]
    else:
      for _ in range(ln):                # overlapping copy: byte by byte
        out.append(out[-off])
  if len(out) != n:
    raise ValueError(f'snappy: {len(out)} bytes where the header says {n}')
  return bytes(out)


def _read_block(buf, offset, size, verify):
  end = offset + size
  if end + 5 > len(buf):
    raise ValueError('block handle points outside the index file')
  ctype = buf[end]
  if verify:
    stored = struct.unpack_from('<I', buf, end + 1)[0]
    if unmask_crc(stored) != crc32c(bytes(buf[offset:end + 1])):
      raise ValueError('index block checksum mismatch')
  block = buf[offset:end]
  if ctype == 1:
    block = snappy_uncompress(bytes(block))
    size = len(block)
  elif ctype != 0:
    raise ValueError(f'index block of unknown compression type {ctype}')
  if size < 4:
    raise ValueError('index block too small')
  n_restarts = struct.unpack_from('<I', block, size - 4)[0]
  limit = size - 4 - 4 * n_restarts
  if limit < 0:
    raise ValueError('bad restart array')
  pos, key, out = 0, b'', []
  while pos < limit:
    shared, pos = _get_varint(block, pos)
    non_shared, pos = _get_varint(block, pos)
    vlen, pos = _get_varint(block, pos)
    if shared > len(key) or pos + non_shared + vlen > limit:
      raise ValueError('corrupt block entry')
    key = key[:shared] + bytes(block[pos:pos + non_shared])
    pos += non_shared
    out.append((key, bytes(block[pos:pos + vlen])))
    pos += vlen
  return out


def _read_table(path, verify):
  with open(path, 'rb') as f:
    buf = f.read()
  if len(buf) < 48:
    raise ValueError(f'{path}: too short for a table footer')
  footer = buf[-48:]
  if struct.unpack_from('<Q', footer, 40)[0] != TABLE_MAGIC:
    raise ValueError(f'{path}: not a TensorFlow checkpoint index (bad magic)')
  pos = 0
  _, pos = _get_varint(footer, pos)       # metaindex handle (offset, size): nothing in it
  _, pos = _get_varint(footer, pos)
  ioff, pos = _get_varint(footer, pos)
  isize, pos = _get_varint(footer, pos)
  entries = []
  for _, handle in _read_block(buf, ioff, isize, verify):
    boff, p2 = _get_varint(handle, 0)
    bsize, _ = _get_varint(handle, p2)
    entries.extend(_read_block(buf, boff, bsize, verify))
  return entries


def _block_bytes(entries, restart_interval=16):
  out, restarts, prev = bytearray(), [], b''
  for i, (key, val) in enumerate(entries):
    shared = 0
    if i % restart_interval == 0:
      restarts.append(len(out))
    else:
      while shared < min(len(prev), len(key)) and prev[shared] == key[shared]:
        shared += 1
    out += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(val))
    out += key[shared:] + val
    prev = key
  if not restarts:
    restarts = [0]
  for r in restarts:
    out += struct.pack('<I', r)
  out += struct.pack('<I', len(restarts))
  return bytes(out)


def _write_table(path, entries, block_bytes=4096):
  """entries: (key bytes, value bytes), sorted by key, keys distinct."""
  blob, index = bytearray(), []

  def emit(block):
    off = len(blob)
    blob.extend(block)
    blob.append(0)                                  # kNoCompression
    blob.extend(struct.pack('<I', mask_crc(crc32c(block + b'\x00'))))
    return _put_varint(off) + _put_varint(len(block))

  cur, cur_size = [], 0
  for key, val in entries:
    cur.append((key, val))
    cur_size += len(key) + len(val) + 3
    if cur_size >= block_bytes:
      index.append((cur[-1][0], emit(_block_bytes(cur))))
      cur, cur_size = [], 0
  if cur:
    index.append((cur[-1][0], emit(_block_bytes(cur))))
  meta = emit(_block_bytes([]))
  idx = emit(_block_bytes(index, restart_interval=1))
  footer = meta + idx
  footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC)
  blob.extend(footer)
  with open(path, 'wb') as f:
    f.write(blob)
    f.flush()
    os.fsync(f.fileno())


# ---------------------------------------------------------------------------------------------
class BundleReader:
  """``BundleReader(prefix)``: the tensors of a TF checkpoint.  ``verify`` checks the CRC-32C of
  every index block at open and of every tensor when it is read."""

  def __init__(self, prefix, verify=True):
    self.prefix, self.verify = prefix, verify
    self.entries, self._slice_data = {}, {}
    self.header = {'num_shards': 1, 'endianness': 0, 'producer': 0}
    seen_header = False
    for key, val in _read_table(prefix + '.index', verify):
      if key == b'':
        seen_header = True
        for num, _, v in _fields(val):
          if num == 1:
            self.header['num_shards'] = v
          elif num == 2:
            self.header['endianness'] = v
          elif num == 3:
            for n2, _, v2 in _fields(v):
              if n2 == 1:
                self.header['producer'] = v2
      elif key[:1] == b'\x00':
        name, ext = decode_slice_key(key)
        self._slice_data.setdefault(name, []).append((ext, _parse_entry(val)))
      else:
        self.entries[key.decode()] = _parse_entry(val)
    if not seen_header:
      raise ValueError(f'{prefix}.index: no bundle header')
    if self.header['endianness'] != 0:
      raise ValueError('big-endian checkpoint: unsupported')

  def names(self):
    return sorted(self.entries)

  def dtype_and_shape(self, name):
    e = self.entries[name]
    if e['dtype'] not in _DT:
      raise ValueError(f'{name}: unsupported DataType {e["dtype"]} (strings / variants / '
                       'resources are not tensors of numbers)')
    return np.dtype(_DT[e['dtype']]), tuple(e['shape'])

  def is_bfloat16(self, name):
    return self.entries[name]['dtype'] == DT_BFLOAT16

  def _load(self, name, e):
    dt, shape = np.dtype(_DT[e['dtype']]), tuple(e['shape'])
    want = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
    if e['size'] != want:
      raise ValueError(f'{name}: entry says {e["size"]} bytes, shape {shape} needs {want}')
    path = f'{self.prefix}.data-{e["shard_id"]:05d}-of-{self.header["num_shards"]:05d}'
    if want == 0:
      return np.empty(shape, dt)
    arr = np.fromfile(path, dtype=dt, count=want // dt.itemsize, offset=e['offset'])
    if arr.nbytes != want:
      raise ValueError(f'{path}: truncated ({name})')
    if self.verify and e['crc32c'] is not None:
      # stored masked (tensor_bundle.cc masks what it writes); with no TensorFlow-written file to
      # check that against, the plain value is accepted as well -- a corrupt tensor matches neither
      actual = crc32c(arr)
      if e['crc32c'] not in (mask_crc(actual), actual):
        raise ValueError(f'{name}: tensor checksum mismatch in {path}')
    return arr.reshape(shape)

  def slices(self, name):
    """[(extents, array)] of a partitioned variable in the order of their starts ([] for a
    tensor saved whole).  extents: (start, length) per dimension, (0, -1) = everything."""
    e = self.entries[name]
    if not e['slices']:
      return []
    have = {tuple(ext): ent for ext, ent in self._slice_data.get(name, [])}
    out = []
    for ext in e['slices']:
      if tuple(ext) not in have:
        raise ValueError(f'{name}: slice {ext} is listed but not stored')
      out.append((ext, self._load(name, have[tuple(ext)])))
    out.sort(key=lambda s: [st for st, _ in s[0]])
    return out

  def read(self, name):
    """The full tensor; a partitioned variable is assembled from its slices (every element must
    be covered exactly by what is stored)."""
    dt, shape = self.dtype_and_shape(name)
    e = self.entries[name]
    if not e['slices']:
      return self._load(name, e)
    full = np.empty(shape, dt)
    covered = 0
    for ext, arr in self.slices(name):
      idx = tuple(slice(None) if ln < 0 else slice(st, st + ln) for st, ln in ext)
      if full[idx].shape != arr.shape:
        raise ValueError(f'{name}: slice {ext} has shape {arr.shape}')
      full[idx] = arr
      covered += arr.size
    if covered != full.size:
      raise ValueError(f'{name}: the stored slices cover {covered} of {full.size} elements')
    return full


def _c_order(array):
  arr = np.asarray(array)       # (np.ascontiguousarray would turn a scalar into a vector)
  return arr if arr.flags.c_contiguous else arr.copy(order='C')


class BundleWriter:
  """Writes one checkpoint.  ``add`` a whole tensor, ``add_slice`` one piece of a partitioned
  variable (its ``SaveSliceInfo``: full name and shape, offsets).  ``num_shards`` data files, as
  ``MergeV2Checkpoints`` leaves one per rank that saved (saver.py:154-180); ``shard=`` says which
  file a tensor / slice goes to (default 0)."""

  def __init__(self, prefix, num_shards=1):
    self.prefix, self.num_shards = prefix, int(num_shards)
    if self.num_shards < 1:
      raise ValueError('num_shards must be >= 1')
    self._whole, self._parts = {}, {}

  def _shard(self, shard):
    if not 0 <= shard < self.num_shards:
      raise ValueError(f'shard {shard} of {self.num_shards}')
    return int(shard)

  @staticmethod
  def _code(arr, bfloat16):
    if bfloat16:
      if arr.dtype not in (np.uint16, np.int16):
        raise ValueError('bfloat16 tensors are passed as their 16-bit patterns')
      return DT_BFLOAT16
    if arr.dtype not in _DT_OF:
      raise ValueError(f'unsupported dtype {arr.dtype}')
    return _DT_OF[arr.dtype]

  def add(self, name, array, bfloat16=False, shard=0):
    arr = _c_order(array)
    if name in self._whole or name in self._parts:
      raise ValueError(f'{name}: added twice')
    self._whole[name] = (arr, self._code(arr, bfloat16), self._shard(shard))

  def add_slice(self, full_name, full_shape, offsets, array, bfloat16=False, shard=0):
    arr = _c_order(array)
    if full_name in self._whole:
      raise ValueError(f'{full_name}: added twice')
    if len(full_shape) != arr.ndim or len(offsets) != arr.ndim:
      raise ValueError('full_shape / offsets / slice rank differ')
    ext = []
    for d in range(arr.ndim):
      if offsets[d] < 0 or offsets[d] + arr.shape[d] > full_shape[d]:
        raise ValueError(f'{full_name}: slice outside the full shape')
      whole = offsets[d] == 0 and arr.shape[d] == full_shape[d]
      ext.append((0, -1) if whole else (int(offsets[d]), int(arr.shape[d])))
    part = self._parts.setdefault(full_name, {'shape': [int(x) for x in full_shape],
                                              'code': self._code(arr, bfloat16), 'slices': []})
    if part['shape'] != [int(x) for x in full_shape] or part['code'] != self._code(arr, bfloat16):
      raise ValueError(f'{full_name}: slices disagree about the full tensor')
    part['slices'].append((ext, arr, self._shard(shard)))

  def finish(self):
    n = self.num_shards
    paths = [f'{self.prefix}.data-{k:05d}-of-{n:05d}' for k in range(n)]
    files = [open(p, 'wb') for p in paths]
    offs = [0] * n
    table = {}
    try:
      def put(arr, code, shard):
        files[shard].write(arr.tobytes())
        e = {'dtype': code, 'shape': list(arr.shape), 'shard_id': shard, 'offset': offs[shard],
             'size': arr.nbytes, 'crc32c': mask_crc(crc32c(arr))}
        offs[shard] += arr.nbytes
        return e
      for name in sorted(self._whole):
        arr, code, shard = self._whole[name]
        table[name.encode()] = _encode_entry(put(arr, code, shard))
      for name in sorted(self._parts):
        part = self._parts[name]
        for ext, arr, shard in part['slices']:
          table[encode_slice_key(name, ext)] = _encode_entry(put(arr, part['code'], shard))
        table[name.encode()] = _encode_entry({
          'dtype': part['code'], 'shape': part['shape'],
          'slices': [ext for ext, _, _ in part['slices']]})
      for f in files:
        f.flush()
        os.fsync(f.fileno())
    finally:
      for f in files:
        f.close()
    header = (_tag(1, 0) + _put_varint(n) +                       # num_shards
              _bytes_field(3, _tag(1, 0) + _put_varint(1)))       # version { producer: 1 }
    table[b''] = header
    _write_table(self.prefix + '.index', sorted(table.items()))
    return self.prefix


def read_reference_table(reader, name, layout='logical', world=None):
  """An embedding table saved by the reference as ``W`` row slices (``embedding/variables.py:
  114-141``): ``layout='reference'`` = what TF calls the full tensor (the slices back to back);
  ``'logical'`` = the table by id -- slice ``r`` (in offset order) holds the rows ``r, r + W, ..``
  because the owner of an id is ``id mod W`` (``sharding.py:182,189``; see training/saver.py).
  ``world``: the world size the checkpoint was written at.  Default: the number of slices -- right
  for the plain variables of the reference; a checkpoint whose tables were ALSO partitioned
  (``variables.py:131-140``: W x P slices per table) must pass it, the slice count cannot tell the
  two apart -- such tables are refused here rather than de-interleaved wrongly."""
  parts = reader.slices(name)
  if layout == 'reference' or len(parts) <= 1:
    return reader.read(name)
  if layout != 'logical':
    raise ValueError("layout must be 'logical' or 'reference'")
  dt, shape = reader.dtype_and_shape(name)
  if world is not None and int(world) != len(parts):
    raise ValueError(f'{name}: {len(parts)} slices for a world of {world}: a partitioned variable '
                     f'(W x P slices, variables.py:131-140) is not supported')
  world = len(parts)
  if any(any(ln >= 0 for _, ln in ext[1:]) for ext, _ in parts):
    raise ValueError(f'{name}: sliced along more than the rows: not a row-sharded table')
  out = np.empty(shape, dt)
  for r, (_, arr) in enumerate(parts):
    want = (shape[0] - r + world - 1) // world
    if arr.shape[0] != want:
      raise ValueError(f'{name}: slice {r} of {world} has {arr.shape[0]} rows, id-mod-{world} '
                       f'ownership gives it {want}: not a table the reference row-sharded')
    out[r::world] = arr
  return out
