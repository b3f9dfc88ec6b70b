"""Host-side marshalling helpers of the Python call forms (hybridbackend_amd/_marshal.py): lazy
per-column views, per-thread argument blocks, the one-pass tensor check.  No GPU: the device ops
that use them are covered by the GPU tests."""
import ctypes
import threading

import numpy as np
import torch

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
