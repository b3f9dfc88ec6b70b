// RCCL-over-xGMI communicator behind HbGetNcclId / HbCreateNcclCollective /
// HbNcclAlltoall[N] / HbNcclAlltoallv[N] (R4, R5, R6).
//   lifecycle      hbtf/distribute/nccl/nccl_get_id.cc:35-62, nccl_create.cc:45-132,
//                  nccl_collective.cc:40-65,434-465
//   active ranks   hbtf/distribute/collective.h:80-112 (one communicator; sub-topologies by
//                  choosing peers)
//   all-to-all     nccl_collective.cc:112-151 (equal split), :250-288 (v), :153-248/:290-384 (N)
//   stream fences  hbtf/common/stream.cc:83-142: the comm stream waits for the compute stream's
//                  tail before the exchange, the compute stream waits for the exchange after it
// One process per GPU; on one 8 x MI355X node every peer is one xGMI hop, so the grouped
// ncclSend/ncclRecv pattern maps one message per link.  Unlike the reference nothing here
// syncs the host: receive sizes come in as host arrays the caller obtained ONCE for all N
// columns (one [N x W] hbk_alltoall_n), not once per op (nccl_alltoallv.cc:316,533).
#include <rccl/rccl.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace hbk {
int cast_n_impl(int32_t n, int32_t src_dtype, int32_t dst_dtype, const void* const* inputs,
                const int64_t* lens, void* const* outputs, hipStream_t stream);
}

struct hbk_comm {
  bool custom = false;         // exchanges go through `transport` instead of RCCL
  hbk_transport_t transport;
  ncclComm_t comm;
  hipStream_t stream;
  hipEvent_t compute_done;
  hipEvent_t comm_done;
  int world_size;
  int local_size;
  int rank;
  int device;
  bool aborted;
  std::mutex mu;  // NCCL calls on one communicator are serialised (nccl/collective.h:113)
};

namespace hbk {
namespace {

#define HBK_NCCL_OK(expr)                                                            \
  do {                                                                               \
    ncclResult_t r__ = (expr);                                                       \
    if (r__ != ncclSuccess) {                                                        \
      return ::hbk::fail(HBK_INTERNAL, "%s failed: %s (%s:%d)", #expr,               \
                         ncclGetErrorString(r__), __FILE__, __LINE__);               \
    }                                                                                \
  } while (0)

// inside ncclGroupStart .. ncclGroupEnd: a failing call must not leave the group open (the
// communicator would be unusable): remember the first error, keep going to the ncclGroupEnd
#define HBK_NCCL_IN_GROUP(first_err, expr)                                           \
  do {                                                                               \
    ncclResult_t r__ = (expr);                                                       \
    if (r__ != ncclSuccess && (first_err) == ncclSuccess) (first_err) = r__;         \
  } while (0)

bool to_nccl(int32_t dtype, ncclDataType_t* out) {
  switch (dtype) {
    case HBK_INT8: *out = ncclInt8; return true;
    case HBK_UINT8: *out = ncclUint8; return true;
    case HBK_INT32: *out = ncclInt32; return true;
    case HBK_UINT32: *out = ncclUint32; return true;
    case HBK_INT64: *out = ncclInt64; return true;
    case HBK_UINT64: *out = ncclUint64; return true;
    case HBK_HALF: *out = ncclFloat16; return true;
    case HBK_FLOAT: *out = ncclFloat32; return true;
    case HBK_DOUBLE: *out = ncclFloat64; return true;
    default: return false;
  }
}

// Collective::compute_active_ranks, hbtf/distribute/collective.h:80-99
void active_ranks(const hbk_comm* c, int32_t topology, std::vector<int>* out) {
  out->clear();
  if (topology == HBK_TOPOLOGY_INTRA_NODE) {
    const int node = c->rank / c->local_size;
    for (int r = node * c->local_size; r < (node + 1) * c->local_size; ++r) out->push_back(r);
  } else if (topology == HBK_TOPOLOGY_INTER_NODE) {
    for (int r = 0; r < c->world_size; ++r) {
      if (c->local_size == 1 || (r % c->local_size) == (c->rank % c->local_size)) {
        out->push_back(r);
      }
    }
  } else {
    for (int r = 0; r < c->world_size; ++r) out->push_back(r);
  }
}

int fence_in(hbk_comm* c, hipStream_t compute) {
  HBK_HIP_OK(hipEventRecord(c->compute_done, compute));
  HBK_HIP_OK(hipStreamWaitEvent(c->stream, c->compute_done, 0));
  return HBK_OK;
}

int fence_out(hbk_comm* c, hipStream_t compute) {
  HBK_HIP_OK(hipEventRecord(c->comm_done, c->stream));
  HBK_HIP_OK(hipStreamWaitEvent(compute, c->comm_done, 0));
  return HBK_OK;
}

// One exchange through a custom transport inside `ranks` (the active ranks of the topology):
// chunk k (send_off[k], send_len[k] elements) goes to ranks[k], the chunk from ranks[k] lands at
// recv_off[k].
int custom_exchange(hbk_comm* c, const std::vector<int>& ranks, const void* sendbuf,
                    const std::vector<int64_t>& send_off, const std::vector<int64_t>& send_len,
                    void* recvbuf, const std::vector<int64_t>& recv_off, size_t esize,
                    hipStream_t stream, bool skip_self = false) {
  std::vector<int32_t> r32(ranks.begin(), ranks.end());
  const int rc = c->transport.exchange(c->transport.ctx, c->rank, r32.data(), (int32_t)r32.size(),
                                       sendbuf, send_off.data(), send_len.data(), recvbuf,
                                       recv_off.data(), esize, skip_self ? 1 : 0,
                                       reinterpret_cast<hbk_stream_t>(stream));
  if (rc != HBK_OK) return fail(rc, "custom transport: exchange failed (%d)", rc);
  return HBK_OK;
}

}  // namespace
}  // namespace hbk

// A communicator over a caller-provided transport (the reference's Collective is an abstract
// class with NCCL as one implementation, hbtf/distribute/collective.h:70-201).
extern "C" int hbk_comm_create_custom(hbk_comm_t* comm, const hbk_transport_t* transport,
                                      int32_t world_size, int32_t local_size, int32_t rank) {
  using namespace hbk;
  HBK_REQUIRE(comm != nullptr && transport != nullptr && transport->exchange != nullptr,
              "comm_create_custom: NULL argument");
  HBK_REQUIRE(world_size >= 1 && local_size >= 1 && world_size % local_size == 0,
              "comm_create_custom: local_size (%d) must divide world_size (%d)", local_size,
              world_size);
  HBK_REQUIRE(rank >= 0 && rank < world_size, "comm_create_custom: rank %d out of [0, %d)", rank,
              world_size);
  hbk_comm* c = new hbk_comm();
  c->custom = true;
  c->transport = *transport;
  c->comm = nullptr;
  c->world_size = world_size;
  c->local_size = local_size;
  c->rank = rank;
  c->aborted = false;
  c->stream = nullptr;
  c->compute_done = nullptr;
  c->comm_done = nullptr;
  (void)hipGetDevice(&c->device);
  // Like an RCCL communicator, the handle has a private stream for its Alltoallv exchanges and the
  // events that fence it against the compute stream (round 5): the pipelined sharded step then runs
  // its exchanges BESIDE the gathers over a custom transport too -- the in-process test world
  // exercises the same stream / event structure as production (hbtf/common/stream.cc:83-142).
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->compute_done, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->comm_done, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    delete c;
    return fail(HBK_INTERNAL, "comm_create_custom: could not create the comm stream / events");
  }
  *comm = c;
  return HBK_OK;
}

extern "C" int hbk_comm_get_id(uint8_t id[HBK_COMM_ID_BYTES]) {
  using namespace hbk;
  static_assert(sizeof(ncclUniqueId) == HBK_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  HBK_REQUIRE(id != nullptr, "comm_get_id: id is NULL");
  ncclUniqueId nid;
  HBK_NCCL_OK(ncclGetUniqueId(&nid));
  memcpy(id, &nid, HBK_COMM_ID_BYTES);
  return HBK_OK;
}

// "built <major.minor.patch> / runs <version code>" of RCCL, for logs and the bench line
extern "C" int hbk_comm_rccl_versions(int32_t* built, int32_t* runtime) {
  using namespace hbk;
  HBK_REQUIRE(built != nullptr && runtime != nullptr, "comm_rccl_versions: NULL argument");
  *built = NCCL_VERSION_CODE;
  int rt = 0;
  if (ncclGetVersion(&rt) != ncclSuccess) rt = 0;
  *runtime = rt;
  return HBK_OK;
}

extern "C" int hbk_comm_create(hbk_comm_t* comm, const uint8_t id[HBK_COMM_ID_BYTES],
                               int32_t world_size, int32_t local_size, int32_t rank) {
  using namespace hbk;
  HBK_REQUIRE(comm != nullptr && id != nullptr, "comm_create: NULL argument");
  HBK_REQUIRE(world_size >= 1, "comm_create: world_size must be >= 1, got %d", world_size);
  HBK_REQUIRE(local_size >= 1 && world_size % local_size == 0,
              "comm_create: local_size (%d) must divide world_size (%d)", local_size,
              world_size);
  HBK_REQUIRE(rank >= 0 && rank < world_size, "comm_create: rank %d out of [0, %d)", rank,
              world_size);
  hbk_comm* c = new hbk_comm();
  c->world_size = world_size;
  c->local_size = local_size;
  c->rank = rank;
  c->aborted = false;
  c->comm = nullptr;
  c->stream = nullptr;
  hipError_t he = hipGetDevice(&c->device);
  if (he != hipSuccess) {
    delete c;
    return fail(HBK_INTERNAL, "comm_create: hipGetDevice failed: %s", hipGetErrorString(he));
  }
  // The library is compiled against one RCCL (its headers) and may run on another (the process
  // may have loaded the RCCL a framework bundles first: torch ships 2.26, /opt/rocm 2.27).  The
  // entry points used here -- unique id, comm init / destroy / abort / async error, group start /
  // end, send, recv, allreduce, allgather -- have kept their signatures throughout RCCL 2.x;
  // anything else is refused here instead of misbehaving later.
  {
    int rt = 0;
    if (ncclGetVersion(&rt) != ncclSuccess || rt / 10000 != NCCL_MAJOR) {
      delete c;
      return fail(HBK_INTERNAL, "comm_create: built against RCCL %d.%d.%d, the process runs RCCL "
                                "version code %d: another major version is not supported",
                  NCCL_MAJOR, NCCL_MINOR, NCCL_PATCH, rt);
    }
  }
  ncclUniqueId nid;
  memcpy(&nid, id, HBK_COMM_ID_BYTES);
  ncclResult_t nr = ncclCommInitRank(&c->comm, world_size, nid, rank);
  if (nr != ncclSuccess) {
    delete c;
    return fail(HBK_INTERNAL, "comm_create: ncclCommInitRank failed: %s",
                ncclGetErrorString(nr));
  }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->compute_done, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->comm_done, hipEventDisableTiming) != hipSuccess) {
    ncclCommDestroy(c->comm);
    delete c;
    return fail(HBK_INTERNAL, "comm_create: could not create the comm stream / events");
  }
  *comm = c;
  return HBK_OK;
}

extern "C" int hbk_comm_destroy(hbk_comm_t comm) {
  using namespace hbk;
  if (comm == nullptr) return HBK_OK;
  if (comm->custom) {
    if (comm->stream != nullptr) (void)hipStreamSynchronize(comm->stream);
    if (comm->transport.destroy != nullptr) comm->transport.destroy(comm->transport.ctx);
    if (comm->compute_done != nullptr) (void)hipEventDestroy(comm->compute_done);
    if (comm->comm_done != nullptr) (void)hipEventDestroy(comm->comm_done);
    if (comm->stream != nullptr) (void)hipStreamDestroy(comm->stream);
    delete comm;
    return HBK_OK;
  }
  (void)hipStreamSynchronize(comm->stream);
  if (!comm->aborted) ncclCommDestroy(comm->comm);
  (void)hipEventDestroy(comm->compute_done);
  (void)hipEventDestroy(comm->comm_done);
  (void)hipStreamDestroy(comm->stream);
  delete comm;
  return HBK_OK;
}

// NcclCollective::CheckAsyncErrors, nccl_collective.cc:449-465
extern "C" int hbk_comm_check_async(hbk_comm_t comm) {
  using namespace hbk;
  HBK_REQUIRE(comm != nullptr, "comm_check_async: comm is NULL");
  if (comm->custom) return HBK_OK;
  std::unique_lock<std::mutex> lock(comm->mu);
  if (comm->aborted) return fail(HBK_INTERNAL, "communicator was aborted");
  ncclResult_t async = ncclSuccess;
  HBK_NCCL_OK(ncclCommGetAsyncError(comm->comm, &async));
  if (async != ncclSuccess && async != ncclInProgress) {
    ncclCommAbort(comm->comm);
    comm->aborted = true;
    return fail(HBK_INTERNAL, "RCCL async error: %s; communicator aborted",
                ncclGetErrorString(async));
  }
  return HBK_OK;
}

extern "C" int hbk_comm_world_size(hbk_comm_t comm) { return comm ? comm->world_size : 0; }
extern "C" int hbk_comm_rank(hbk_comm_t comm) { return comm ? comm->rank : -1; }
// what RCCL itself says about the communicator (ncclCommCount): the number of ranks it connected --
// 0 for a custom transport (no RCCL communicator behind it), -1 on error.  bench.py prints it as
// `rccl_ranks_seen`, so a multi-GPU number comes with proof that RCCL spanned the ranks.
extern "C" int hbk_comm_rccl_ranks(hbk_comm_t comm) {
  if (comm == nullptr) return -1;
  if (comm->custom) return 0;
  int n = -1;
  if (ncclCommCount(comm->comm, &n) != ncclSuccess) return -1;
  return n;
}
extern "C" hbk_stream_t hbk_comm_stream(hbk_comm_t comm) {
  return comm ? reinterpret_cast<hbk_stream_t>(comm->stream) : nullptr;
}

extern "C" int hbk_comm_active_ranks(hbk_comm_t comm, int32_t topology, int32_t* ranks_out) {
  if (comm == nullptr) return 0;
  std::vector<int> ranks;
  hbk::active_ranks(comm, topology, &ranks);
  if (ranks_out != nullptr) {
    for (size_t i = 0; i < ranks.size(); ++i) ranks_out[i] = ranks[i];
  }
  return (int)ranks.size();
}

extern "C" int hbk_alltoall_n(hbk_comm_t comm, int32_t n, int32_t dtype, int32_t topology,
                              const void* const* inputs, const int64_t* counts,
                              void* const* outputs, hbk_stream_t compute_stream) {
  using namespace hbk;
  HBK_REQUIRE(comm != nullptr, "alltoall_n: comm is NULL");
  HBK_REQUIRE(n >= 0, "alltoall_n: n must be >= 0");
  if (n == 0) return HBK_OK;
  HBK_REQUIRE(inputs && counts && outputs, "alltoall_n: NULL argument array");
  ncclDataType_t nt;
  HBK_REQUIRE(to_nccl(dtype, &nt), "alltoall_n: unsupported dtype %d", dtype);
  const size_t esize = (size_t)dtype_size(dtype);
  std::vector<int> ranks;
  active_ranks(comm, topology, &ranks);
  const int64_t active = (int64_t)ranks.size();
  for (int32_t c = 0; c < n; ++c) {
    HBK_REQUIRE(counts[c] >= 0 && counts[c] % active == 0,
                "Number of elements in input (%lld) must can be divided into %lld partitions",
                (long long)counts[c], (long long)active);  // nccl_collective.cc:119-123
    HBK_REQUIRE(counts[c] == 0 || (inputs[c] && outputs[c]), "alltoall_n: NULL buffer %d", c);
  }
  if (comm->custom) {
    for (int32_t c = 0; c < n; ++c) {
      const int64_t part = counts[c] / active;
      std::vector<int64_t> off(active), len(active);
      for (int64_t i = 0; i < active; ++i) {
        off[i] = i * part;
        len[i] = part;
      }
      int lrc = custom_exchange(comm, ranks, inputs[c], off, len, outputs[c], off, esize,
                                as_stream(compute_stream));
      if (lrc != HBK_OK) return lrc;
    }
    return HBK_OK;
  }
  std::unique_lock<std::mutex> lock(comm->mu);
  HBK_REQUIRE(!comm->aborted, "alltoall_n: communicator was aborted");
  int rc = fence_in(comm, as_stream(compute_stream));
  if (rc != HBK_OK) return rc;
  ncclResult_t in_group = ncclSuccess;
  HBK_NCCL_OK(ncclGroupStart());
  for (int32_t c = 0; c < n; ++c) {
    if (counts[c] == 0) continue;
    const size_t part = (size_t)(counts[c] / active);
    const char* sendbuf = reinterpret_cast<const char*>(inputs[c]);
    char* recvbuf = reinterpret_cast<char*>(outputs[c]);
    for (int64_t i = 0; i < active; ++i) {
      const size_t off = (size_t)i * part * esize;
      if (ranks[i] == comm->rank) continue;  // own slice: copied below, never through RCCL
      HBK_NCCL_IN_GROUP(in_group, ncclSend(sendbuf + off, part, nt, ranks[i], comm->comm, comm->stream));
      HBK_NCCL_IN_GROUP(in_group, ncclRecv(recvbuf + off, part, nt, ranks[i], comm->comm, comm->stream));
    }
  }
  HBK_NCCL_OK(ncclGroupEnd());
  HBK_NCCL_OK(in_group);
  for (int32_t c = 0; c < n; ++c) {
    const size_t part = (size_t)(counts[c] / active);
    for (int64_t i = 0; i < active && part > 0; ++i) {
      if (ranks[i] != comm->rank) continue;
      const size_t off = (size_t)i * part * esize;
      HBK_HIP_OK(hipMemcpyAsync(reinterpret_cast<char*>(outputs[c]) + off,
                                reinterpret_cast<const char*>(inputs[c]) + off, part * esize,
                                hipMemcpyDeviceToDevice, comm->stream));
    }
  }
  return fence_out(comm, as_stream(compute_stream));
}

extern "C" size_t hbk_alltoallv_wire_workspace_bytes(int32_t n, const int64_t* common_sizes,
                                                     const int32_t* send_sizes,
                                                     const int32_t* recv_sizes,
                                                     int32_t active) {
  if (n <= 0 || !common_sizes || !send_sizes || !recv_sizes || active <= 0) return 0;
  size_t total = 0;
  for (int32_t c = 0; c < n; ++c) {
    int64_t s = 0, r = 0;
    for (int32_t i = 0; i < active; ++i) {
      s += send_sizes[(size_t)c * active + i];
      r += recv_sizes[(size_t)c * active + i];
    }
    total += (((size_t)(s * common_sizes[c]) * 2 + 15) & ~(size_t)15) +
             (((size_t)(r * common_sizes[c]) * 2 + 15) & ~(size_t)15);
  }
  return total;
}

namespace hbk {
// hbk_alltoallv_n with explicit ordering: when `before` / `after` are given the exchange is
// enqueued on the communicator's own stream behind `before` and `after` is recorded when it
// is done -- the compute stream is NOT fenced, so the caller can overlap other work with the
// exchange (sharded.hip pipelines column groups this way).  With both NULL it fences against
// `compute_stream` on both sides like the reference (hbtf/common/stream.cc:83-142).
// `skip_self`: the caller has placed this rank's own chunk itself (the sharded driver's owner
// gather writes it where the exchange would have copied it); not with the fp16 wire, whose casts
// also round the own chunk (as the reference's do).
// `inline_x`: the whole exchange -- casts, the grouped send / receive, the own-slice copy -- is
// enqueued on `compute_stream` itself: no fence, no event, no hop to the communicator's stream
// and back (11 us each way on this chip, profiles/r02_hop_probe.txt); nothing overlaps it either.
int alltoallv_events(hbk_comm_t comm, int32_t n, int32_t dtype, int32_t wire_dtype,
                     int32_t topology, const int64_t* common_sizes, const void* const* inputs,
                     const int32_t* send_sizes, void* const* outputs,
                     const int32_t* recv_sizes, void* wire_ws, size_t wire_ws_bytes,
                     hbk_stream_t compute_stream, hipEvent_t before, hipEvent_t after,
                     bool skip_self, bool inline_x) {
  HBK_REQUIRE(comm != nullptr, "alltoallv_n: comm is NULL");
  HBK_REQUIRE(n >= 0, "alltoallv_n: n must be >= 0");
  if (n == 0) return HBK_OK;
  HBK_REQUIRE(common_sizes && inputs && send_sizes && outputs && recv_sizes,
              "alltoallv_n: NULL argument array");
  const bool half_wire = (dtype == HBK_FLOAT && wire_dtype == HBK_HALF);
  HBK_REQUIRE(half_wire || wire_dtype == dtype || wire_dtype == HBK_FLOAT,
              "alltoallv_n: wire_dtype must be float or half (half only for float data)");
  HBK_REQUIRE(!(skip_self && half_wire), "alltoallv_n: skip_self is not available with the fp16 wire");
  ncclDataType_t nt;
  HBK_REQUIRE(to_nccl(half_wire ? HBK_HALF : dtype, &nt), "alltoallv_n: unsupported dtype %d",
              dtype);
  const size_t esize = half_wire ? 2 : (size_t)dtype_size(dtype);
  std::vector<int> ranks;
  active_ranks(comm, topology, &ranks);
  const int32_t active = (int32_t)ranks.size();

  std::vector<int64_t> send_rows(n), recv_rows(n);
  for (int32_t c = 0; c < n; ++c) {
    HBK_REQUIRE(common_sizes[c] >= 1, "alltoallv_n: common_size[%d] must be >= 1", c);
    int64_t s = 0, r = 0;
    for (int32_t i = 0; i < active; ++i) {
      HBK_REQUIRE(send_sizes[(size_t)c * active + i] >= 0 &&
                      recv_sizes[(size_t)c * active + i] >= 0,
                  "alltoallv_n: negative size for input %d", c);
      s += send_sizes[(size_t)c * active + i];
      r += recv_sizes[(size_t)c * active + i];
    }
    send_rows[c] = s;
    recv_rows[c] = r;
    HBK_REQUIRE((s == 0 || inputs[c]) && (r == 0 || outputs[c]),
                "alltoallv_n: NULL buffer for input %d", c);
  }

  // fp16 wire: carve send / receive staging out of the workspace
  std::vector<const void*> wire_in(n);
  std::vector<void*> wire_out(n);
  if (half_wire) {
    const size_t need =
        hbk_alltoallv_wire_workspace_bytes(n, common_sizes, send_sizes, recv_sizes, active);
    HBK_REQUIRE(need == 0 || (wire_ws != nullptr && wire_ws_bytes >= need),
                "alltoallv_n: wire workspace too small: need %zu bytes, got %zu", need,
                wire_ws_bytes);
    char* p = reinterpret_cast<char*>(wire_ws);
    for (int32_t c = 0; c < n; ++c) {
      wire_in[c] = p;
      p += ((size_t)(send_rows[c] * common_sizes[c]) * 2 + 15) & ~(size_t)15;
      wire_out[c] = p;
      p += ((size_t)(recv_rows[c] * common_sizes[c]) * 2 + 15) & ~(size_t)15;
    }
  } else {
    for (int32_t c = 0; c < n; ++c) {
      wire_in[c] = inputs[c];
      wire_out[c] = outputs[c];
    }
  }

  if (comm->custom) {
    // on the handle's private stream, fenced like the RCCL path below (inline: on the caller's)
    hipStream_t cs = inline_x ? as_stream(compute_stream) : comm->stream;
    int lrc;
    if (inline_x) {
      // (nothing to order)
    } else if (before != nullptr) {
      HBK_HIP_OK(hipStreamWaitEvent(cs, before, 0));
    } else if ((lrc = fence_in(comm, as_stream(compute_stream))) != HBK_OK) {
      return lrc;
    }
    if (half_wire) {
      std::vector<int64_t> lens(n);
      std::vector<void*> dst(n);
      for (int32_t c = 0; c < n; ++c) {
        lens[c] = send_rows[c] * common_sizes[c];
        dst[c] = const_cast<void*>(wire_in[c]);
      }
      if ((lrc = cast_n_impl(n, HBK_FLOAT, HBK_HALF, inputs, lens.data(), dst.data(), cs)) !=
          HBK_OK) {
        return lrc;
      }
    }
    for (int32_t c = 0; c < n; ++c) {
      std::vector<int64_t> soff(active), slen(active), roff(active);
      int64_t so = 0, ro = 0;
      for (int32_t i = 0; i < active; ++i) {
        soff[i] = so;
        slen[i] = (int64_t)send_sizes[(size_t)c * active + i] * common_sizes[c];
        roff[i] = ro;
        so += slen[i];
        ro += (int64_t)recv_sizes[(size_t)c * active + i] * common_sizes[c];
      }
      if ((lrc = custom_exchange(comm, ranks, wire_in[c], soff, slen, wire_out[c], roff, esize,
                                 cs, skip_self)) !=
          HBK_OK) {
        return lrc;
      }
    }
    if (half_wire) {
      std::vector<int64_t> lens(n);
      std::vector<const void*> src(n);
      for (int32_t c = 0; c < n; ++c) {
        lens[c] = recv_rows[c] * common_sizes[c];
        src[c] = wire_out[c];
      }
      if ((lrc = cast_n_impl(n, HBK_HALF, HBK_FLOAT, src.data(), lens.data(), outputs, cs)) !=
          HBK_OK) {
        return lrc;
      }
    }
    if (inline_x) return HBK_OK;
    if (after != nullptr) {
      HBK_HIP_OK(hipEventRecord(after, cs));
      return HBK_OK;
    }
    return fence_out(comm, as_stream(compute_stream));
  }
  for (int32_t c = 0; c < n; ++c) {     // checked before any RCCL group is opened
    for (int32_t i = 0; i < active; ++i) {
      HBK_REQUIRE(ranks[i] != comm->rank || send_sizes[(size_t)c * active + i] ==
                                                recv_sizes[(size_t)c * active + i],
                  "alltoallv_n: self send/recv sizes differ for input %d (%d, %d)", c,
                  send_sizes[(size_t)c * active + i], recv_sizes[(size_t)c * active + i]);
    }
  }
  std::unique_lock<std::mutex> lock(comm->mu);
  HBK_REQUIRE(!comm->aborted, "alltoallv_n: communicator was aborted");
  int rc = HBK_OK;
  const hipStream_t xs = inline_x ? as_stream(compute_stream) : comm->stream;
  if (inline_x) {
    // (nothing to order: everything below goes on the caller's stream)
  } else if (before != nullptr) {
    HBK_HIP_OK(hipStreamWaitEvent(comm->stream, before, 0));
  } else {
    rc = fence_in(comm, as_stream(compute_stream));
  }
  if (rc != HBK_OK) return rc;
  if (half_wire) {
    std::vector<int64_t> lens(n);
    std::vector<void*> dst(n);
    for (int32_t c = 0; c < n; ++c) {
      lens[c] = send_rows[c] * common_sizes[c];
      dst[c] = const_cast<void*>(wire_in[c]);
    }
    rc = cast_n_impl(n, HBK_FLOAT, HBK_HALF, inputs, lens.data(), dst.data(), xs);
    if (rc != HBK_OK) return rc;
  }
  ncclResult_t in_group = ncclSuccess;
  hipError_t copy_err = hipSuccess;
  HBK_NCCL_OK(ncclGroupStart());
  for (int32_t c = 0; c < n; ++c) {
    const char* sendbuf = reinterpret_cast<const char*>(wire_in[c]);
    char* recvbuf = reinterpret_cast<char*>(wire_out[c]);
    size_t sendoffset = 0, recvoffset = 0;  // bytes, 64-bit
    for (int32_t i = 0; i < active; ++i) {
      const size_t sendsize = (size_t)send_sizes[(size_t)c * active + i] * (size_t)common_sizes[c];
      const size_t recvsize = (size_t)recv_sizes[(size_t)c * active + i] * (size_t)common_sizes[c];
      if (ranks[i] == comm->rank) {
        // own slice: a device copy on the comm stream (RCCL's self send/recv moves it through
        // one channel's copy loop: 54 MB took 111 us, the blit engine path ~25 us)
        if (sendsize > 0 && copy_err == hipSuccess && !skip_self) {
          copy_err = hipMemcpyAsync(recvbuf + recvoffset, sendbuf + sendoffset, sendsize * esize,
                                    hipMemcpyDeviceToDevice, xs);
        }
      } else {
        if (sendsize > 0) {
          HBK_NCCL_IN_GROUP(in_group, ncclSend(sendbuf + sendoffset, sendsize, nt, ranks[i],
                                               comm->comm, xs));
        }
        if (recvsize > 0) {
          HBK_NCCL_IN_GROUP(in_group, ncclRecv(recvbuf + recvoffset, recvsize, nt, ranks[i],
                                               comm->comm, xs));
        }
      }
      sendoffset += sendsize * esize;
      recvoffset += recvsize * esize;
    }
  }
  HBK_NCCL_OK(ncclGroupEnd());
  HBK_NCCL_OK(in_group);
  HBK_HIP_OK(copy_err);
  if (half_wire) {
    std::vector<int64_t> lens(n);
    std::vector<const void*> src(n);
    for (int32_t c = 0; c < n; ++c) {
      lens[c] = recv_rows[c] * common_sizes[c];
      src[c] = wire_out[c];
    }
    rc = cast_n_impl(n, HBK_HALF, HBK_FLOAT, src.data(), lens.data(), outputs, xs);
    if (rc != HBK_OK) return rc;
  }
  if (inline_x) return HBK_OK;
  if (after != nullptr) {
    HBK_HIP_OK(hipEventRecord(after, xs));
    return HBK_OK;
  }
  return fence_out(comm, as_stream(compute_stream));
}
}  // namespace hbk

extern "C" int hbk_alltoallv_n(hbk_comm_t comm, int32_t n, int32_t dtype, int32_t wire_dtype,
                               int32_t topology, const int64_t* common_sizes,
                               const void* const* inputs, const int32_t* send_sizes,
                               void* const* outputs, const int32_t* recv_sizes, void* wire_ws,
                               size_t wire_ws_bytes, hbk_stream_t compute_stream) {
  return hbk::alltoallv_events(comm, n, dtype, wire_dtype, topology, common_sizes, inputs,
                               send_sizes, outputs, recv_sizes, wire_ws, wire_ws_bytes,
                               compute_stream, nullptr, nullptr, false, false);
}

// ------------------------------------------------------------------------------------------------
// aggregation of replicated gradients (hbtf/training/gradient.py:119-177).
//   dense   HbNcclAllreduce / HbNcclAllreduceN / ..MergedN  (nccl_allreduce.cc:31-260)
//   sparse  HbNcclAllgatherv                                (nccl_allgatherv.cc:31-120)
// The N tensors of a call travel as ONE bucket: packed into a staging buffer, one ncclAllReduce,
// unpacked -- with the `1/W` of gradient.py:77-99 (`_mean`) fused into the unpack as `scale`.
namespace hbk {
namespace {

constexpr int kRedBlock = 256;
constexpr int kMaxRedSegs = 400;

struct RedSeg {
  const void* src;
  void* dst;
  int64_t count;    // elements
  int64_t tile0;
};
struct RedArgs {
  int32_t n_segs;
  int32_t esize;
  float scale;
  int32_t scale_f32;   // 1: dst = src * scale (fp32 only)
  RedSeg seg[kMaxRedSegs];
};
static_assert(sizeof(RedArgs) <= 16384, "kernarg budget");
constexpr int kRedTile = kRedBlock * 8;   // elements per block

// N-segment copy with an optional fp32 scale (pack: scale off; unpack: the mean's 1/W)
__global__ __launch_bounds__(kRedBlock) void bucket_copy_kernel(const RedArgs a) {
  int si = 0, hi = a.n_segs;
  while (hi - si > 1) {
    const int mid = (si + hi) >> 1;
    if (a.seg[mid].tile0 <= (int64_t)blockIdx.x) {
      si = mid;
    } else {
      hi = mid;
    }
  }
  const RedSeg& s = a.seg[si];
  const int64_t base = ((int64_t)blockIdx.x - s.tile0) * kRedTile;
#pragma unroll
  for (int k = 0; k < kRedTile / kRedBlock; ++k) {
    const int64_t i = base + (int64_t)k * kRedBlock + threadIdx.x;
    if (i >= s.count) continue;
    if (a.esize == 4) {
      if (a.scale_f32) {
        reinterpret_cast<float*>(s.dst)[i] = reinterpret_cast<const float*>(s.src)[i] * a.scale;
      } else {
        reinterpret_cast<uint32_t*>(s.dst)[i] = reinterpret_cast<const uint32_t*>(s.src)[i];
      }
    } else if (a.esize == 8) {
      reinterpret_cast<uint64_t*>(s.dst)[i] = reinterpret_cast<const uint64_t*>(s.src)[i];
    } else if (a.esize == 2) {
      reinterpret_cast<uint16_t*>(s.dst)[i] = reinterpret_cast<const uint16_t*>(s.src)[i];
    } else {
      reinterpret_cast<uint8_t*>(s.dst)[i] = reinterpret_cast<const uint8_t*>(s.src)[i];
    }
  }
}

int bucket_copy(int32_t n, const void* const* src, void* const* dst, const int64_t* counts,
                size_t esize, bool scale_f32, float scale, hipStream_t stream) {
  int32_t c0 = 0;
  while (c0 < n) {
    RedArgs args;
    int k = 0;
    int64_t tiles = 0;
    while (c0 < n && k < kMaxRedSegs) {
      const int32_t c = c0++;
      if (counts[c] == 0) continue;
      args.seg[k].src = src[c];
      args.seg[k].dst = dst[c];
      args.seg[k].count = counts[c];
      args.seg[k].tile0 = tiles;
      tiles += (counts[c] + kRedTile - 1) / kRedTile;
      ++k;
    }
    if (k == 0) continue;
    args.n_segs = k;
    args.esize = (int32_t)esize;
    args.scale = scale;
    args.scale_f32 = scale_f32 ? 1 : 0;
    hipLaunchKernelGGL(bucket_copy_kernel, dim3((unsigned)tiles), dim3(kRedBlock), 0, stream, args);
    HBK_HIP_OK(hipGetLastError());
  }
  return HBK_OK;
}

bool to_nccl_op(int32_t op, ncclRedOp_t* out) {
  switch (op) {
    case 0: *out = ncclSum; return true;
    case 1: *out = ncclProd; return true;
    case 2: *out = ncclMax; return true;
    case 3: *out = ncclMin; return true;
    default: return false;
  }
}

}  // namespace
}  // namespace hbk

extern "C" size_t hbk_allreduce_workspace_bytes(int32_t n, const int64_t* counts, int32_t dtype) {
  if (n <= 1 || counts == nullptr) return 0;   // a single tensor is reduced in place, no bucket
  size_t total = 0;
  for (int32_t c = 0; c < n; ++c) total += (size_t)(counts[c] > 0 ? counts[c] : 0);
  return total * (size_t)hbk::dtype_size(dtype) + 16;
}

extern "C" int hbk_allreduce_n(hbk_comm_t comm, int32_t n, int32_t dtype, int32_t reduce_op,
                               const void* const* inputs, const int64_t* counts,
                               void* const* outputs, float scale, void* workspace,
                               size_t workspace_bytes, hbk_stream_t compute_stream) {
  using namespace hbk;
  HBK_REQUIRE(comm != nullptr, "allreduce_n: comm is NULL");
  HBK_REQUIRE(n >= 0, "allreduce_n: n must be >= 0");
  if (n == 0) return HBK_OK;
  HBK_REQUIRE(inputs && counts && outputs, "allreduce_n: NULL argument array");
  ncclDataType_t nt;
  ncclRedOp_t op;
  HBK_REQUIRE(to_nccl(dtype, &nt), "allreduce_n: unsupported dtype %d", dtype);
  HBK_REQUIRE(to_nccl_op(reduce_op, &op),
              "allreduce_n: reduce_op must be 0 (SUM), 1 (PROD), 2 (MAX) or 3 (MIN)");
  HBK_REQUIRE(scale == 1.0f || dtype == HBK_FLOAT, "allreduce_n: scale needs fp32 data");
  const size_t esize = (size_t)dtype_size(dtype);
  int64_t total = 0;
  for (int32_t c = 0; c < n; ++c) {
    HBK_REQUIRE(counts[c] >= 0, "allreduce_n: negative count for input %d", c);
    HBK_REQUIRE(counts[c] == 0 || (inputs[c] && outputs[c]), "allreduce_n: NULL buffer %d", c);
    total += counts[c];
  }
  if (total == 0) return HBK_OK;
  hipStream_t cs = as_stream(compute_stream);
  // the bucket: one tensor is reduced straight from input to output
  const void* red_in = inputs[0];
  void* red_out = outputs[0];
  std::vector<void*> slots(n);
  if (n > 1) {
    const size_t need = hbk_allreduce_workspace_bytes(n, counts, dtype);
    HBK_REQUIRE(workspace != nullptr && workspace_bytes >= need,
                "allreduce_n: workspace too small: need %zu bytes, got %zu", need,
                workspace_bytes);
    char* p = reinterpret_cast<char*>(workspace);
    for (int32_t c = 0; c < n; ++c) {
      slots[c] = p;
      p += (size_t)counts[c] * esize;
    }
    red_in = red_out = workspace;
  }
  hipStream_t rs = comm->custom ? cs : comm->stream;   // stream of pack/reduce/unpack
  // one communicator = one ordered queue: everything enqueued on its stream is under its mutex
  std::unique_lock<std::mutex> lock(comm->mu, std::defer_lock);
  if (!comm->custom) {
    lock.lock();
    HBK_REQUIRE(!comm->aborted, "allreduce_n: communicator was aborted");
    int rc = fence_in(comm, cs);
    if (rc != HBK_OK) return rc;
  }
  if (n > 1) {
    int rc = bucket_copy(n, inputs, slots.data(), counts, esize, false, 1.0f, rs);
    if (rc != HBK_OK) return rc;
  }
  if (comm->custom) {
    HBK_REQUIRE(comm->transport.allreduce != nullptr,
                "allreduce_n: the custom transport has no allreduce");
    const int trc = comm->transport.allreduce(comm->transport.ctx, comm->rank, comm->world_size,
                                              dtype, reduce_op, red_in, red_out, total,
                                              reinterpret_cast<hbk_stream_t>(rs));
    if (trc != HBK_OK) return fail(trc, "custom transport: allreduce failed (%d)", trc);
  } else {
    HBK_NCCL_OK(ncclAllReduce(red_in, red_out, (size_t)total, nt, op, comm->comm, comm->stream));
  }
  if (n > 1) {
    std::vector<const void*> src(slots.begin(), slots.end());
    int rc = bucket_copy(n, src.data(), outputs, counts, esize, scale != 1.0f, scale, rs);
    if (rc != HBK_OK) return rc;
  } else if (scale != 1.0f) {
    const void* src[1] = {red_out};
    int rc = bucket_copy(1, src, outputs, counts, esize, true, scale, rs);
    if (rc != HBK_OK) return rc;
  }
  if (!comm->custom) return fence_out(comm, cs);
  return HBK_OK;
}

// HbNcclAllgatherv: output = inputs of ranks 0..W-1 concatenated; counts[r] (host, elements) is
// what rank r contributes (the op gathers them itself and syncs the host, nccl_allgatherv.cc;
// here the caller obtains them, e.g. with one hbk_alltoall_n of its own count).
extern "C" int hbk_allgatherv(hbk_comm_t comm, int32_t dtype, const void* input,
                              const int64_t* counts, void* output, hbk_stream_t compute_stream) {
  using namespace hbk;
  HBK_REQUIRE(comm != nullptr && counts != nullptr, "allgatherv: NULL argument");
  ncclDataType_t nt;
  HBK_REQUIRE(to_nccl(dtype, &nt), "allgatherv: unsupported dtype %d", dtype);
  const size_t esize = (size_t)dtype_size(dtype);
  const int W = comm->world_size, me = comm->rank;
  std::vector<int64_t> off(W + 1, 0);
  for (int r = 0; r < W; ++r) {
    HBK_REQUIRE(counts[r] >= 0, "allgatherv: negative count for rank %d", r);
    off[r + 1] = off[r] + counts[r];
  }
  if (off[W] == 0) return HBK_OK;
  HBK_REQUIRE(output != nullptr && (counts[me] == 0 || input != nullptr), "allgatherv: NULL buffer");
  hipStream_t cs = as_stream(compute_stream);
  char* out = reinterpret_cast<char*>(output);
  if (comm->custom) {
    std::vector<int64_t> soff(W, 0), slen(W, counts[me]), roff(off.begin(), off.end() - 1);
    std::vector<int> all(W);
    for (int r = 0; r < W; ++r) all[r] = r;
    return custom_exchange(comm, all, input, soff, slen, output, roff, esize, cs);
  }
  std::unique_lock<std::mutex> lock(comm->mu);
  HBK_REQUIRE(!comm->aborted, "allgatherv: communicator was aborted");
  int rc = fence_in(comm, cs);
  if (rc != HBK_OK) return rc;
  ncclResult_t in_group = ncclSuccess;
  HBK_NCCL_OK(ncclGroupStart());
  for (int r = 0; r < W; ++r) {
    if (r == me) continue;
    if (counts[me] > 0) {
      HBK_NCCL_IN_GROUP(in_group, ncclSend(input, (size_t)counts[me], nt, r, comm->comm, comm->stream));
    }
    if (counts[r] > 0) {
      HBK_NCCL_IN_GROUP(in_group, ncclRecv(out + (size_t)off[r] * esize, (size_t)counts[r], nt, r,
                                           comm->comm, comm->stream));
    }
  }
  HBK_NCCL_OK(ncclGroupEnd());
  HBK_NCCL_OK(in_group);
  if (counts[me] > 0) {
    HBK_HIP_OK(hipMemcpyAsync(out + (size_t)off[me] * esize, input, (size_t)counts[me] * esize,
                              hipMemcpyDeviceToDevice, comm->stream));
  }
  return fence_out(comm, cs);
}

// HbNcclBroadcast (hbtf/distribute/nccl/nccl_broadcast.cc:31-92): every rank ends with the root's
// `count` elements in `output` (input is read on the root only; input == output is allowed).
extern "C" int hbk_broadcast(hbk_comm_t comm, int32_t dtype, const void* input, void* output,
                             int64_t count, int32_t root, hbk_stream_t compute_stream) {
  using namespace hbk;
  HBK_REQUIRE(comm != nullptr, "broadcast: comm is NULL");
  HBK_REQUIRE(count >= 0, "broadcast: negative count");
  const int W = comm->world_size, me = comm->rank;
  HBK_REQUIRE(root >= 0 && root < W, "broadcast: root %d out of [0, %d)", root, W);
  ncclDataType_t nt;
  HBK_REQUIRE(to_nccl(dtype, &nt), "broadcast: unsupported dtype %d", dtype);
  if (count == 0) return HBK_OK;
  HBK_REQUIRE(output != nullptr && (me != root || input != nullptr), "broadcast: NULL buffer");
  const size_t esize = (size_t)dtype_size(dtype);
  hipStream_t cs = as_stream(compute_stream);
  if (comm->custom) {
    // the root sends its buffer to every rank, the others send nothing
    std::vector<int64_t> soff(W, 0), slen(W, me == root ? count : 0), roff(W, 0);
    std::vector<int> all(W);
    for (int r = 0; r < W; ++r) all[r] = r;
    return custom_exchange(comm, all, me == root ? input : output, soff, slen, output, roff, esize, cs);
  }
  std::unique_lock<std::mutex> lock(comm->mu);
  HBK_REQUIRE(!comm->aborted, "broadcast: communicator was aborted");
  int rc = fence_in(comm, cs);
  if (rc != HBK_OK) return rc;
  HBK_NCCL_OK(ncclBroadcast(me == root ? input : output, output, (size_t)count, nt, root, comm->comm,
                            comm->stream));
  return fence_out(comm, cs);
}
