"""Two-staged sharded lookup for multi-node jobs -- host mirror of
``hybridbackend/tensorflow/embedding/sharding.py:210-276`` (used when ``num_nodes > 1``,
sharding.py:335-337) for N columns at once.

ids first travel inside the node to the GPU whose LOCAL index owns them (dual-modulo stage one,
intra-node alltoallv over xGMI), are deduplicated, then cross nodes to the owning rank (stage two,
inter-node alltoallv), are deduplicated again and gathered; rows come back the same way.  Every
compute phase is a fused N-column launch of the C ABI (``hbk_partition_by_dual_modulo_n``,
``hbk_unique_n``, ``hbk_group_lookup_fwd`` as gather / restore / stitch); the four exchanges go
through ``Collective.alltoallv_n`` with ``Topology.INTRA_NODE`` / ``INTER_NODE``.  The phases are
separate methods so that a test can drive W virtual ranks with its own transport on one GPU; a
single node uses the one-call C++ driver (``ShardedGroupLookup``) instead.
"""
import torch

from hybridbackend_amd import _lib
from hybridbackend_amd.distribute import partition as _partition
from hybridbackend_amd.distribute.collective import Topology
from hybridbackend_amd.embedding.lookup import GroupLookup
from hybridbackend_amd.embedding.unique import unique_n


class HierarchicalGroupLookup:
  """Args: shards (this rank's rows per column, owner = id mod W), world_size, local_size,
  buckets (ids are taken modulo them first), coll (None when only the phases are used)."""

  def __init__(self, shards, world_size, local_size, buckets=None, coll=None):
    self.shards = list(shards)
    self.world, self.local = int(world_size), int(local_size)
    if self.world % self.local != 0:
      raise _lib.InvalidArgumentError(
        _lib.INVALID_ARGUMENT, 'local_size must divide world_size')
    self.nodes = self.world // self.local
    self.buckets = [int(b or 0) for b in (buckets or [0] * len(self.shards))]
    self.coll = coll
    self._lib = _lib.lib()
    self._owner = GroupLookup(self.shards, None, 'sum', divisor=self.world)   # row = id // W
    self.device = self.shards[0].device

  # -- gather(rows_of_a_column, index): restore duplicates / stitch, all columns in one launch --
  @staticmethod
  def _take(rows, index):
    return GroupLookup(rows, None, 'sum')(index)

  def _trim(self, uniq, nu):
    counts = torch.stack([k.reshape(()) for k in nu]).tolist()     # one host sync for N columns
    return [u[:int(k)] for u, k in zip(uniq, counts)]

  # -- requester side, stage one (sharding.py:224-228) --
  def stage_one(self, ids):
    work = ids
    if any(self.buckets):
      work = [torch.empty_like(t) for t in ids]
      _lib.check(self._lib.hbk_floormod_n(
        len(ids), _lib.torch_dtype_code(ids[0].dtype),
        _lib.ptr_array([t.data_ptr() for t in ids]), _lib.i64_array([t.numel() for t in ids]),
        _lib.i64_array([b or (1 << 62) for b in self.buckets]),
        _lib.ptr_array([t.data_ptr() for t in work]), _lib.current_stream(self.device)))
    return _partition.partition_by_dual_modulo_n(work, self.local, self.nodes, 1)

  # -- intra-node peer, after the first exchange: unique + stage two (:234-240) --
  def stage_two(self, s0_ids):
    res = unique_n(s0_ids)
    uniq = self._trim([r[0] for r in res], [r[2] for r in res])
    inv = [r[1] for r in res]
    outs, sizes, idx = _partition.partition_by_dual_modulo_n(uniq, self.nodes, self.local, 2)
    return outs, sizes, idx, inv

  # -- owner, after the second exchange: unique, // W, gather, restore (:245-258) --
  def owner_gather(self, s1_ids):
    res = unique_n(s1_ids)
    uniq = self._trim([r[0] for r in res], [r[2] for r in res])
    emb = self._owner(uniq)
    return self._take(emb, [r[1] for r in res])

  # -- back on the intra-node peer: stitch stage two, restore stage one's duplicates (:265-270) --
  def unstage_two(self, rows, s1_index, s0_inverse):
    return self._take(self._take(rows, s1_index), s0_inverse)

  # -- back on the requester: stitch stage one (:272-276) --
  def unstage_one(self, rows, s0_index):
    return self._take(rows, s0_index)

  def __call__(self, ids):
    """The whole forward over ``self.coll`` (four topology-aware alltoallv, each with the one
    host sync the reference op also pays for its sizes)."""
    c = self.coll
    n = len(ids)
    dims = [int(t.shape[1]) for t in self.shards]

    def xchg(values, sizes, topo, common=None):
      recv_sizes = c.alltoall_n(sizes, topo)
      hs = [s.tolist() for s in sizes]
      hr = [s.tolist() for s in recv_sizes]
      return c.alltoallv_n(values, hs, hr, common_sizes=common, topology=topo), hr, hs

    o0, s0, i0 = self.stage_one(ids)
    r0, hr0, hs0 = xchg(o0, s0, Topology.INTRA_NODE)
    o1, s1, i1, inv0 = self.stage_two(r0)
    r1, hr1, hs1 = xchg(o1, s1, Topology.INTER_NODE)
    rows = self.owner_gather(r1)
    back1 = c.alltoallv_n(rows, hr1, hs1, common_sizes=dims, topology=Topology.INTER_NODE)
    rows0 = self.unstage_two(back1, i1, inv0)
    back0 = c.alltoallv_n(rows0, hr0, hs0, common_sizes=dims, topology=Topology.INTRA_NODE)
    del n
    return self.unstage_one(back0, i0)
