"""Collective communication over RCCL/xGMI -- host mirror of
``hybridbackend/tensorflow/distribute/collective.py`` (``Collective.get().alltoall``,
``:271-350``) and ``ops.py:34-39`` (``Topology``) over the C ABI's communicator
(``hbk_comm_*``, ``hbk_alltoall_n``, ``hbk_alltoallv_n``).

One process per GPU.  The 128-byte RCCL id is produced on rank 0 by ``hbk_comm_get_id``
(op ``HbGetNcclId``) and broadcast with whatever host channel exists -- the reference uses
a TF gRPC broadcast (rpc.py:88-124); here ``torch.distributed`` (any backend) or a
caller-supplied function.
"""
import ctypes as C

import torch

from hybridbackend_amd import _lib


class Topology(object):  # pylint: disable=useless-object-inheritance
  r'''Communication topology (ops.py:34-39).'''
  ALL = 0  # Communication across all GPUs
  INTRA_NODE = 1  # Communication across all GPUs in current node
  INTER_NODE = 2  # Communication across all GPUS with same rank in every nodes


def compute_active_ranks(topology, world_size, local_size, rank):
  """Collective::compute_active_ranks, hybridbackend/tensorflow/distribute/collective.h:80-99."""
  if topology == Topology.INTRA_NODE:
    node = rank // local_size
    return list(range(node * local_size, (node + 1) * local_size))
  if topology == Topology.INTER_NODE:
    return [r for r in range(world_size)
            if local_size == 1 or r % local_size == rank % local_size]
  return list(range(world_size))


def alltoallv_offsets(sizes, common_size=1):
  """Running element offsets of the chunks inside an Alltoallv buffer
  (nccl_collective.cc:261-284), 64-bit."""
  off, out = 0, []
  for s in sizes:
    out.append(off)
    off += int(s) * int(common_size)
  return out, off


class Collective:
  """One RCCL communicator + its private stream (nccl/collective.h:41-126)."""

  _default = None

  def __init__(self, world_size, rank, local_size=None, unique_id=None, broadcast_fn=None):
    self._lib = _lib.lib()
    self.world_size = int(world_size)
    self.rank = int(rank)
    self.local_size = int(local_size or world_size)
    if unique_id is None:
      buf = (C.c_uint8 * _lib.COMM_ID_BYTES)()
      if self.rank == 0:
        _lib.check(self._lib.hbk_comm_get_id(buf))
      if self.world_size > 1:
        if broadcast_fn is None:
          broadcast_fn = _torch_broadcast_bytes
        data = broadcast_fn(bytes(buf))
        buf = (C.c_uint8 * _lib.COMM_ID_BYTES).from_buffer_copy(data)
      unique_id = buf
    self._handle = C.c_void_p()
    _lib.check(self._lib.hbk_comm_create(
      C.byref(self._handle), unique_id, self.world_size, self.local_size, self.rank))
    self._wire_ws = None

  @classmethod
  def local_world(cls, world_size, local_size=None):
    """TEST transport (tests/support/libhbk_testing.so, not part of the product library):
    ``world_size`` communicators of one in-process world -- one host thread and one stream per
    rank, all on the current GPU -- plugged in through the public custom-transport hook
    ``hbk_comm_create_custom``; ``local_size`` ranks per "node" for the INTRA_NODE / INTER_NODE
    topologies.  Returns the list of communicators."""
    lib = _lib.lib()
    tlib = _lib.testing_lib()
    world = C.c_void_p()
    if tlib.hbk_testing_local_world_create(C.byref(world), world_size) != 0:
      raise _lib.HbkError(_lib.INTERNAL, 'could not create the in-process world')
    comms = []
    for r in range(world_size):
      c = cls.__new__(cls)
      c._lib = lib
      c.world_size, c.rank = world_size, r
      c.local_size = int(local_size) if local_size is not None else world_size
      c._handle = C.c_void_p()
      c._wire_ws = None
      c._world = world
      _lib.check(tlib.hbk_testing_comm_create(C.byref(c._handle), world, r, c.local_size))
      comms.append(c)
    return comms

  @classmethod
  def get(cls):
    if cls._default is None:
      raise _lib.HbkError(_lib.INTERNAL, 'Collective is not initialized')
    return cls._default

  @classmethod
  def set_default(cls, coll):
    cls._default = coll

  def close(self):
    if self._handle:
      self._lib.hbk_comm_destroy(self._handle)
      self._handle = C.c_void_p()

  def active_size(self, topology=Topology.ALL):
    return len(compute_active_ranks(topology, self.world_size, self.local_size, self.rank))

  def check_async_errors(self):
    _lib.check(self._lib.hbk_comm_check_async(self._handle))

  # -- equal split (HbNcclAlltoallN; also the sizes exchange before every Alltoallv) --
  def alltoall_n(self, values, topology=Topology.ALL):
    n = len(values)
    if n == 0:
      return []
    dev = values[0].device
    code = _lib.torch_dtype_code(values[0].dtype)
    for v in values:
      _lib.require_device_tensor(v, 'value')
    outs = [torch.empty_like(v) for v in values]
    _lib.check(self._lib.hbk_alltoall_n(
      self._handle, n, code, topology,
      _lib.ptr_array([v.data_ptr() for v in values]),
      _lib.i64_array([v.numel() for v in values]),
      _lib.ptr_array([o.data_ptr() for o in outs]), _lib.current_stream(dev)))
    return outs

  # -- HbNcclAlltoallvN with host-known sizes --
  def alltoallv_n(self, values, send_sizes, recv_sizes, common_sizes=None,
                  wire_dtype=None, topology=Topology.ALL, outs=None):
    """values[c]: [sum(send_sizes[c]), *common]; send/recv_sizes: host int lists
    [n][active].  Returns the received tensors [sum(recv_sizes[c]), *common]."""
    n = len(values)
    if n == 0:
      return []
    dev = values[0].device
    active = self.active_size(topology)
    dtype = values[0].dtype
    code = _lib.torch_dtype_code(dtype)
    wire = code if wire_dtype is None else _lib.torch_dtype_code(wire_dtype)
    if common_sizes is None:
      common_sizes = [int(v[0].numel()) if v.dim() > 1 and v.shape[0] > 0
                      else int(torch.Size(v.shape[1:]).numel()) for v in values]
    flat_s, flat_r = [], []
    for c in range(n):
      if len(send_sizes[c]) != active or len(recv_sizes[c]) != active:
        raise _lib.InvalidArgumentError(
          _lib.INVALID_ARGUMENT,
          f'Sizes of input {c} must have {active} elements')  # nccl_alltoallv.cc:468-474
      flat_s += [int(x) for x in send_sizes[c]]
      flat_r += [int(x) for x in recv_sizes[c]]
    if outs is None:
      outs = []
      for c, v in enumerate(values):
        rows = sum(int(x) for x in recv_sizes[c])
        outs.append(torch.empty((rows,) + tuple(v.shape[1:]), dtype=dtype, device=dev))
    for v in values:
      _lib.require_device_tensor(v, 'value')
    ss, rs = _lib.i32_array(flat_s), _lib.i32_array(flat_r)
    cs = _lib.i64_array(common_sizes)
    ws_ptr, ws_bytes = None, 0
    if wire != code:
      need = self._lib.hbk_alltoallv_wire_workspace_bytes(n, cs, ss, rs, active)
      if self._wire_ws is None or self._wire_ws.numel() < need:
        self._wire_ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
      ws_ptr, ws_bytes = self._wire_ws.data_ptr(), self._wire_ws.numel()
    _lib.check(self._lib.hbk_alltoallv_n(
      self._handle, n, code, wire, topology, cs,
      _lib.ptr_array([v.data_ptr() for v in values]), ss,
      _lib.ptr_array([o.data_ptr() for o in outs]), rs,
      C.c_void_p(ws_ptr), C.c_size_t(ws_bytes), _lib.current_stream(dev)))
    return outs

  def alltoall(self, value, sizes=None, common_shape=None, topology=Topology.ALL,
               wire_dtype=None, name=None):
    r'''Shuffle value partitions across devices (collective.py:271-350).

    With ``sizes`` (an int32 device vector of rows per peer) this is ``HbNcclAlltoallv``:
    returns ``(exchanged_value, exchanged_sizes)``.  Like the reference op it has to bring
    the exchanged sizes to the host to size the output (nccl_alltoallv.cc:306-329): one
    stream sync.  Use ``alltoallv_n`` with host-known sizes to avoid it.
    '''
    del name, common_shape
    if sizes is None:
      return self.alltoall_n([value], topology)[0]
    recv_sizes = self.alltoall_n([sizes], topology)[0]
    host_s = sizes.tolist()
    host_r = recv_sizes.tolist()   # the one host sync
    out = self.alltoallv_n([value], [host_s], [host_r], wire_dtype=wire_dtype,
                           topology=topology)[0]
    return out, recv_sizes


  # -- HbNcclAllreduce[N] / HbNcclAllreduceMergedN: the N tensors travel as one bucket --
  SUM, PROD, MAX, MIN = 0, 1, 2, 3   # CollectiveOps (hbtf/distribute/ops.py)

  def allreduce_n(self, values, reduce_op=0, scale=1.0, outs=None):
    """Reduce every tensor across ranks; ``scale`` (fp32) multiplies the result in the same
    pass (``1 / world_size`` = the mean of training/gradient.py:77-99)."""
    n = len(values)
    if n == 0:
      return []
    dev = values[0].device
    code = _lib.torch_dtype_code(values[0].dtype)
    for v in values:
      _lib.require_device_tensor(v, 'value')
      if v.dtype != values[0].dtype:
        raise _lib.InvalidArgumentError(
          _lib.INVALID_ARGUMENT, 'all tensors of a packed allreduce share one dtype')
    if outs is None:
      outs = [torch.empty_like(v) for v in values]
    counts = _lib.i64_array([v.numel() for v in values])
    need = self._lib.hbk_allreduce_workspace_bytes(n, counts, code)
    if need and (getattr(self, '_red_ws', None) is None or self._red_ws.numel() < need):
      self._red_ws = torch.empty(need, dtype=torch.uint8, device=dev)
    ws = self._red_ws if need else None
    _lib.check(self._lib.hbk_allreduce_n(
      self._handle, n, code, int(reduce_op), _lib.ptr_array([v.data_ptr() for v in values]),
      counts, _lib.ptr_array([o.data_ptr() for o in outs]), C.c_float(scale),
      C.c_void_p(ws.data_ptr() if ws is not None else None),
      C.c_size_t(ws.numel() if ws is not None else 0), _lib.current_stream(dev)))
    return outs

  def allreduce(self, value, reduce_op=None, name=None):
    r'''Reduce values across devices (collective.py:176-209; op HbNcclAllreduce).'''
    del name
    return self.allreduce_n([value], 0 if reduce_op is None else reduce_op)[0]

  def allgather(self, value, name=None):
    r'''Gather ``value`` (rows along dim 0, any count per rank) from all devices, rank order
    (collective.py ``allgather``; op HbNcclAllgatherv).  Like the reference op it brings the
    per-rank sizes to the host to size the output: one stream sync.'''
    del name
    _lib.require_device_tensor(value, 'value')
    W = self.world_size
    inner = int(torch.Size(value.shape[1:]).numel()) if value.dim() > 1 else 1
    mine = torch.full((W,), value.numel(), dtype=torch.int64, device=value.device)
    counts = self.alltoall_n([mine])[0].tolist() if W > 1 else [value.numel()]
    rows = sum(counts) // max(inner, 1)
    out = torch.empty((rows,) + tuple(value.shape[1:]), dtype=value.dtype, device=value.device)
    _lib.check(self._lib.hbk_allgatherv(
      self._handle, _lib.torch_dtype_code(value.dtype), C.c_void_p(value.data_ptr()),
      _lib.i64_array(counts), C.c_void_p(out.data_ptr()), _lib.current_stream(value.device)))
    return out


def broadcast(value, coll, root_rank=0, out=None):
  r'''Broadcast ``value`` of rank ``root_rank`` to every rank (collective.py ``broadcast``; op
  HbNcclBroadcast, nccl_broadcast.cc:31-92).  Every rank passes a tensor of the same shape and
  dtype; returns the root's values.'''
  _lib.require_device_tensor(value, 'value')
  if out is None:
    out = torch.empty_like(value)
  _lib.check(coll._lib.hbk_broadcast(
    coll._handle, _lib.torch_dtype_code(value.dtype), C.c_void_p(value.data_ptr()),
    C.c_void_p(out.data_ptr()), value.numel(), int(root_rank), _lib.current_stream(value.device)))
  return out


def aggregate_gradients(grads, coll, sharded=None):
  """Cross-rank aggregation of one step's gradients -- mirror of
  hybridbackend/tensorflow/training/gradient.py:119-217.

  grads[i] is a dense tensor or an ``IndexedSlices`` tuple ``(values [k, dim], indices [k])``;
  ``sharded[i]`` marks gradients of sharded variables, which are returned untouched (each rank
  owns its rows, gradient.py:193-217).  Dense gradients of replicated variables are summed in ONE
  packed allreduce and divided by the world size in the same pass; sparse ones are allgathered
  (values and indices) and their values divided by the world size (gradient.py:77-99,160-177).
  """
  n = len(grads)
  sharded = list(sharded) if sharded is not None else [False] * n
  W = coll.world_size
  out = list(grads)
  if W <= 1:
    return out
  dense = [i for i in range(n) if not sharded[i] and grads[i] is not None
           and not isinstance(grads[i], (tuple, list))]
  if dense:
    red = coll.allreduce_n([grads[i] for i in dense], scale=1.0 / W)
    for i, r in zip(dense, red):
      out[i] = r
  for i in range(n):
    if sharded[i] or grads[i] is None or not isinstance(grads[i], (tuple, list)):
      continue
    values, indices = grads[i]
    out[i] = (coll.allgather(values) * (1.0 / W), coll.allgather(indices))
  return out


def _torch_broadcast_bytes(data):
  import torch.distributed as dist  # pylint: disable=import-outside-toplevel
  if not dist.is_initialized():
    raise _lib.HbkError(
      _lib.INTERNAL, 'torch.distributed is not initialized: pass broadcast_fn or unique_id')
  obj = [data]
  dist.broadcast_object_list(obj, src=0)
  return obj[0]
