// Sharded group lookup: the whole per-step pipeline of hbtf/embedding/sharding.py:171-205
// (R12) for N columns behind ONE C-ABI call per direction, so the host side costs tens of
// microseconds, not a Python loop over columns.
//
//   forward   bucketize -> stable partition by id mod W        (R1, R2: elementwise.hip, partition.hip)
//             pack ids peer-major                               (one launch, N*W segments)
//             sizes [N x W] alltoall + ONE host sync            (R5: comm.hip; the reference syncs
//                                                                once per op, nccl_alltoallv.cc:316,533)
//             ids alltoallv: ONE message per peer              (all columns of a peer travel together:
//                                                                W sends + W receives per exchange, one
//                                                                per xGMI link, instead of N*W)
//             owner gather (N*W virtual columns, `// W`)        (R7/R8: lookup_fwd.hip) straight into
//                                                                the peer-major reply buffer
//             rows alltoallv (fp32 or fp16 wire), one message per peer   (R5, R6)
//             unpack rows column-major, stitch + combiner       (R8, R9: lookup_fwd.hip)
//   backward  d(stitch+combiner) -> pack -> reverse alltoallv (forward's sizes, collective.py:334-347)
//             -> unpack -> duplicate-row reduction (+ fused SGD) (R10: lookup_bwd.hip)
//
// Buffers whose size depends on what the peers send (known only after the size exchange)
// are owned by the plan and grow on demand (hipMalloc, never shrinks); everything else is
// caller-owned as usual.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <chrono>
#include <random>
#include <utility>
#include <vector>

#include "common.h"

namespace hbk {
int alltoallv_events(hbk_comm_t comm, int32_t n, int32_t dtype, int32_t wire_dtype,
                     int32_t topology, const int64_t* common_sizes, const void* const* inputs,
                     const int32_t* send_sizes, void* const* outputs,
                     const int32_t* recv_sizes, void* wire_ws, size_t wire_ws_bytes,
                     hbk_stream_t compute_stream, hipEvent_t before, hipEvent_t after,
                     bool skip_self, bool inline_x);
int partition_by_modulo_fused(int32_t n_cols, int32_t num_partitions, const int64_t* const* inputs,
                              const int64_t* lens, const int64_t* buckets,
                              int64_t* const* outputs, int32_t* const* sizes,
                              int32_t* const* indices, int32_t* sizes_t, void* workspace,
                              size_t workspace_bytes, hipStream_t stream);
namespace {

constexpr int kBlock = 256;
constexpr int kMaxSegs = 480;
constexpr int kMaxPackWorldP2p = 256;   // ranks the p2p slot kernel holds offsets for
constexpr int kCopyTileBytes = kBlock * 16 * 8;  // 32 KB per block

struct Seg {
  const void* src;
  void* dst;
  int64_t bytes;   // of the source
  int32_t tile0;
  int32_t narrow;  // 1: the source is int64, the destination int32 (the id wire format)
};

struct SegArgs {
  int32_t n_segs;
  int32_t pad_;
  Seg seg[kMaxSegs];
};
static_assert(sizeof(SegArgs) <= 16384, "kernarg budget");  // 480 x 32 B + 8

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// N-segment gather copy: packs / unpacks the per-(column, peer) runs of an exchange buffer
__global__ __launch_bounds__(kBlock) void seg_copy_kernel(const SegArgs a) {
  int si = 0, hi = a.n_segs;
  while (hi - si > 1) {
    const int mid = (si + hi) >> 1;
    if (a.seg[mid].tile0 <= (int)blockIdx.x) {
      si = mid;
    } else {
      hi = mid;
    }
  }
  const Seg& s = a.seg[si];
  const int64_t base = (int64_t)((int)blockIdx.x - s.tile0) * kCopyTileBytes;
  const char* src = reinterpret_cast<const char*>(s.src);
  char* dst = reinterpret_cast<char*>(s.dst);
  if (s.narrow) {
    const int64_t n = s.bytes >> 3, e0 = base >> 3;
    const int64_t* src64 = reinterpret_cast<const int64_t*>(src);
    int32_t* dst32 = reinterpret_cast<int32_t*>(dst);
#pragma unroll
    for (int k = 0; k < kCopyTileBytes / 8 / kBlock; ++k) {
      const int64_t e = e0 + (int64_t)k * kBlock + threadIdx.x;
      if (e < n) dst32[e] = (int32_t)__builtin_nontemporal_load(src64 + e);
    }
    return;
  }
  if ((((uintptr_t)src | (uintptr_t)dst | (uintptr_t)s.bytes) & 15) == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int64_t o = base + ((int64_t)k * kBlock + threadIdx.x) * 16;
      if (o < s.bytes) {
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src + o));
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst + o));
      }
    }
  } else {  // 4-byte granularity (every run is a whole number of int32 / fp32 / int64 items)
    for (int k = 0; k < 32; ++k) {
      const int64_t o = base + ((int64_t)k * kBlock + threadIdx.x) * 4;
      if (o < s.bytes) {
        *reinterpret_cast<uint32_t*>(dst + o) = *reinterpret_cast<const uint32_t*>(src + o);
      }
    }
  }
}

int seg_copy(const std::vector<Seg>& segs_in, hipStream_t stream) {
  size_t i = 0;
  while (i < segs_in.size()) {
    SegArgs args;
    int k = 0;
    int64_t tiles = 0;
    while (i < segs_in.size() && k < kMaxSegs) {
      const Seg& s = segs_in[i++];
      if (s.bytes <= 0) continue;
      args.seg[k] = s;
      args.seg[k].tile0 = (int32_t)tiles;
      tiles += (s.bytes + kCopyTileBytes - 1) / kCopyTileBytes;
      ++k;
    }
    if (k == 0) continue;
    args.n_segs = k;
    hipLaunchKernelGGL(seg_copy_kernel, dim3((unsigned)tiles), dim3(kBlock), 0, stream, args);
    HBK_HIP_OK(hipGetLastError());
  }
  return HBK_OK;
}

// sizes [N][W] -> [W][N] (the chunk for peer p of the equal-split size exchange = column p)
__global__ void transpose_sizes_kernel(const int32_t* in, int32_t* out, int n, int w) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n * w) out[(i % w) * n + i / w] = in[i];
}

// p2p form: the OUTPUT SLOT of every id that goes on the wire.  shard_index[j] = k is the position
// of the batch's j-th id among the column's shard-ordered ids; the run of owner q starts at
// shard_off[c][q] there and at dst_off[c][q] in the peer-major slot buffer (same layout as the ids):
//   slots[dst_off[c][q] + k - shard_off[c][q]] = j
// -- the owner stores the row of that id at row j of the requester's output.
constexpr int kSlotCols = 128;
constexpr int kSlotTile = 2048;
struct SlotArgs {
  int32_t n_cols, W, col0, pad_;
  const int64_t* shard_off;   // device [N][W]
  const int64_t* dst_off;     // device [N][W]
  int32_t* slots;
  int32_t tile0[kSlotCols + 1];
  int32_t n[kSlotCols];
  const int32_t* idx[kSlotCols];
};
static_assert(sizeof(SlotArgs) <= 16384, "kernarg budget");

__global__ __launch_bounds__(kBlock) void p2p_slots_kernel(const SlotArgs a) {
  __shared__ int64_t so[kMaxPackWorldP2p], dof[kMaxPackWorldP2p];
  int c = 0, hi = a.n_cols;
  while (hi - c > 1) {
    const int mid = (c + hi) >> 1;
    if (a.tile0[mid] <= (int)blockIdx.x) {
      c = mid;
    } else {
      hi = mid;
    }
  }
  const int W = a.W;
  for (int q = (int)threadIdx.x; q < W; q += kBlock) {
    so[q] = a.shard_off[(size_t)(a.col0 + c) * W + q];
    dof[q] = a.dst_off[(size_t)(a.col0 + c) * W + q];
  }
  __syncthreads();
  const int32_t* idx = a.idx[c];
  const int32_t n = a.n[c];
  const int32_t j0 = ((int)blockIdx.x - a.tile0[c]) * kSlotTile;
#pragma unroll
  for (int t = 0; t < kSlotTile / kBlock; ++t) {
    const int32_t j = j0 + t * kBlock + (int)threadIdx.x;
    if (j >= n) continue;
    const int64_t k = idx[j];
    int q = 0, e = W;   // last q with so[q] <= k
    while (e - q > 1) {
      const int mid = (q + e) >> 1;
      if (so[mid] <= k) {
        q = mid;
      } else {
        e = mid;
      }
    }
    a.slots[dof[q] + k - so[q]] = j;
  }
}

// The ids of a step go peer-major into the outgoing buffer BEFORE the host knows the sizes: the
// offsets of the runs (q, c) are sums over the partition's size matrix S [N][W], which is on the
// device as soon as the partition is -- every workgroup redoes its column's few sums from S
// instead of waiting for the host to send them back.  So the pack runs while the host is still
// asleep on the sizes (it used to be the first thing enqueued after the wake-up: 11 us on the
// step's critical path for 26 x 65536 ids).  Element e of column c's shard-ordered ids belongs to
// the shard q with cum[q] <= e < cum[q + 1] and goes to
//   gbase(group of c) + sum_{q' < q} (ids of the group for q') + sum_{c' < c, same group} S[c'][q]
//   + e - cum[q].
constexpr int kMaxPackCols = 400;
constexpr int kMaxPackWorld = 256;
constexpr int kPackTile = 2048;

struct PackArgs {
  const int32_t* S;        // [N][W], written by the partition on the same stream
  void* dst;               // the set's outgoing ids
  int64_t dst_items;       // bound of every store (S is garbage when the partition gave up)
  int32_t n_cols, W, narrow, pad_;
  const int64_t* src[kMaxPackCols];   // shard-ordered ids of the column
  // p2p form (round 6): the same launch also writes the OUTPUT SLOT of every outgoing id -- the
  // batch position j of the id that the partition put at shard-ordered position idx[j] -- at the id's
  // own place in the peer-major layout (what p2p_slots_kernel did from host-made offsets AFTER the
  // step's host wait: a launch and a table copy off the critical path).  NULL: ids only.
  int32_t* slots;
  const int32_t* idx[kMaxPackCols];   // the partition's `indices` of the column (slots != NULL)
  int64_t gbase[kMaxPackCols];        // first item of the column's group in dst
  int32_t n[kMaxPackCols];
  int32_t tile0[kMaxPackCols];
  int16_t g0[kMaxPackCols], g1[kMaxPackCols];   // columns of its group
};
static_assert(sizeof(PackArgs) <= 16384, "kernarg budget");

__global__ __launch_bounds__(kBlock) void pack_ids_kernel(const PackArgs a) {
  __shared__ int32_t cum[kMaxPackWorld + 1];     // start of shard q inside the column
  __shared__ int64_t base[kMaxPackWorld];        // where the run (q, c) starts in dst
  __shared__ int32_t peer_tot[kMaxPackWorld], before[kMaxPackWorld];
  int c = 0, hi = a.n_cols;
  while (hi - c > 1) {
    const int mid = (c + hi) >> 1;
    if (a.tile0[mid] <= (int)blockIdx.x) {
      c = mid;
    } else {
      hi = mid;
    }
  }
  const int W = a.W, tid = (int)threadIdx.x;
  if (tid < W) {
    int tot = 0, bef = 0;
    for (int cc = a.g0[c]; cc < a.g1[c]; ++cc) {
      const int s = a.S[(size_t)cc * W + tid];
      tot += s;
      if (cc < c) bef += s;
    }
    peer_tot[tid] = tot;
    before[tid] = bef;
  }
  __syncthreads();
  if (tid == 0) {
    int64_t at = a.gbase[c];
    int32_t in_col = 0;
    for (int q = 0; q < W; ++q) {
      cum[q] = in_col;
      base[q] = at + before[q];
      in_col += a.S[(size_t)c * W + q];
      at += peer_tot[q];
    }
    cum[W] = in_col;
  }
  __syncthreads();
  const int32_t n = a.n[c];
  const int64_t* src = a.src[c];
  const int32_t e0 = ((int)blockIdx.x - a.tile0[c]) * kPackTile;
#pragma unroll
  for (int k = 0; k < kPackTile / kBlock; ++k) {
    const int32_t e = e0 + k * kBlock + tid;
    if (e >= n || e >= cum[W]) break;   // (deduplicated column: n is its capacity, cum[W] its ids)
    const int64_t v = __builtin_nontemporal_load(src + e);
    int q = 0, qh = W;       // last q with cum[q] <= e (empty shards share a start: take the last)
    while (qh - q > 1) {
      const int mid = (q + qh) >> 1;
      if (cum[mid] <= e) {
        q = mid;
      } else {
        qh = mid;
      }
    }
    const int64_t at = base[q] + (e - cum[q]);
    if (at < 0 || at >= a.dst_items) continue;
    if (a.narrow) {
      reinterpret_cast<int32_t*>(a.dst)[at] = (int32_t)v;
    } else {
      reinterpret_cast<int64_t*>(a.dst)[at] = v;
    }
  }
  if (a.slots == nullptr) return;   // (uniform)
  // slots[place of the id at shard-ordered position idx[j]] = j
  const int32_t* idx = a.idx[c];
#pragma unroll
  for (int k = 0; k < kPackTile / kBlock; ++k) {
    const int32_t j = e0 + k * kBlock + tid;
    if (j >= n) break;
    const int32_t e = __builtin_nontemporal_load(idx + j);
    if (e < 0 || e >= cum[W]) continue;   // (garbage when the partition gave up)
    int q = 0, qh = W;
    while (qh - q > 1) {
      const int mid = (q + qh) >> 1;
      if (cum[mid] <= e) {
        q = mid;
      } else {
        qh = mid;
      }
    }
    const int64_t at = base[q] + (e - cum[q]);
    if (at < 0 || at >= a.dst_items) continue;
    a.slots[at] = j;
  }
}

// ---- requester-side dedup (round 4) -------------------------------------------------------------
// A column flagged `dedup` sends every distinct id ONCE per step (the reference's tutorials do this
// in user code in front of the patched lookup: docs/tutorial/ranking/data.py:180-182 tf.unique ->
// lookup -> tf.gather; SURVEY 8e lever (b)).  Composition, all on the device, still ONE host sync:
//   partition (stable, by id mod W)  ->  hbk_unique_n over the SHARD-ORDERED ids of the column.
// First-occurrence order of an array in which all ids of shard 0 precede those of shard 1, .. is
// itself grouped by shard, so the unique list IS the column's outgoing id run, shard by shard, and
//   S'[c][q] = distinct ids of shard q  = the distance of two boundaries of that list (binary
//              search on id mod W, below),
//   index'   = inverse o shard_index     (one composed index: the stitch reads the received rows
//              through it exactly as it read the undeduplicated ones through shard_index).
// Backward: the requester reduces the duplicates itself (hbk_group_lookup_bwd over index', the
// received-rows buffer playing the table) before the reverse exchange.
constexpr int kDedupTile = 2048;

struct DedupArgs {
  const int64_t* uniq;     // bases of the set's buffers; column d lives at off[d]
  const int32_t* inv;
  int32_t* idx;
  const int32_t* nu;       // [N] distinct ids of every column (device)
  int32_t* S;              // [N][W]
  int32_t* St;             // [W][N]
  FastDiv wdiv;
  int32_t n_cols, W, N, pad_;
  int64_t off[kMaxPackCols];
  int32_t n[kMaxPackCols];
  int32_t tile0[kMaxPackCols];
  int16_t col[kMaxPackCols];
};
static_assert(sizeof(DedupArgs) <= 16384, "kernarg budget");

// index'[i] = inv[shard_index[i]], in place
__global__ __launch_bounds__(kBlock) void dedup_compose_kernel(const DedupArgs a) {
  int d = 0, hi = a.n_cols;
  while (hi - d > 1) {
    const int mid = (d + hi) >> 1;
    if (a.tile0[mid] <= (int)blockIdx.x) {
      d = mid;
    } else {
      hi = mid;
    }
  }
  const int32_t n = a.n[d];
  const int32_t* inv = a.inv + a.off[d];
  int32_t* idx = a.idx + a.off[d];
  const int32_t e0 = ((int)blockIdx.x - a.tile0[d]) * kDedupTile;
#pragma unroll
  for (int k = 0; k < kDedupTile / kBlock; ++k) {
    const int32_t e = e0 + k * kBlock + (int)threadIdx.x;
    if (e < n) idx[e] = inv[idx[e]];
  }
}

// S'[c][q] and its transpose from the boundaries of the unique list (grouped by shard, ascending)
__global__ __launch_bounds__(kMaxPackWorld) void dedup_sizes_kernel(const DedupArgs a) {
  __shared__ int32_t start[kMaxPackWorld + 1];
  const int d = (int)blockIdx.x, q = (int)threadIdx.x;
  const int c = a.col[d];
  const int32_t u = a.nu[c];
  const int64_t* uniq = a.uniq + a.off[d];
  if (q < a.W) {
    int32_t lo = 0, hi = u;   // first position whose shard is >= q
    while (lo < hi) {
      const int32_t mid = (lo + hi) >> 1;
      if ((int)floormod_i64(uniq[mid], a.wdiv) >= q) {
        hi = mid;
      } else {
        lo = mid + 1;
      }
    }
    start[q] = lo;
  }
  if (q == 0) start[a.W] = u;
  __syncthreads();
  if (q < a.W) {
    const int32_t cnt = start[q + 1] - start[q];
    a.S[(size_t)c * a.W + q] = cnt;
    a.St[(size_t)q * a.N + c] = cnt;
  }
}

struct Buffer {
  void* ptr = nullptr;
  size_t bytes = 0;
  int ensure(size_t need) {
    if (need <= bytes) return HBK_OK;
    if (ptr) HBK_HIP_OK(hipFree(ptr));
    ptr = nullptr;
    bytes = 0;
    const size_t grow = need + need / 4 + (1 << 20);
    HBK_HIP_OK(hipMalloc(&ptr, grow));
    bytes = grow;
    return HBK_OK;
  }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
  }
};

}  // namespace
}  // namespace hbk

// ---- host arithmetic of the peer-major exchange buffers ---------------------------------------
// Pure host code (no device calls): exported so that it can be checked without a GPU, under a
// real multi-process exchange (tests/test_dist_gloo.py).
//   S[c][q] = rows this rank requests from owner q for column c   (partition sizes)
//   R[q][c] = rows requester q asked this rank for, column c       (exchanged sizes)
// Requester-side buffers (ids going out, rows coming back) hold the runs (q, c) in peer-major
// order with S; owner-side buffers (ids coming in, rows going out) in peer-major order with R.
extern "C" int hbk_sharded_layout(int32_t n_cols, int32_t world, const int32_t* dims,
                                  const int32_t* S, const int32_t* R, int32_t* ids_send_peer,
                                  int32_t* ids_recv_peer, int32_t* rows_send_peer,
                                  int32_t* rows_recv_peer, int64_t* req_id_off,
                                  int64_t* req_row_off, int64_t* own_id_off,
                                  int64_t* own_row_off, int64_t* col_shard_off) {
  using namespace hbk;
  HBK_REQUIRE(n_cols >= 1 && world >= 1 && dims && S && R, "sharded_layout: bad argument");
  const int N = n_cols, W = world;
  // every run of rows starts on a 16-byte boundary whatever the mix of dims (<= 3 floats of
  // padding per run), so rows can always be moved as 16-byte chunks in place
  auto pad4 = [](int64_t floats) { return (floats + 3) & ~(int64_t)3; };
  int64_t rid = 0, rrow = 0, oid = 0, orow = 0;
  for (int q = 0; q < W; ++q) {
    int64_t si = 0, ri = 0, sf = 0, rf = 0;
    for (int c = 0; c < N; ++c) {
      const int64_t s = S[(size_t)c * W + q], r = R[(size_t)q * N + c];
      HBK_REQUIRE(s >= 0 && r >= 0 && dims[c] >= 1, "sharded_layout: negative size");
      if (req_id_off) req_id_off[(size_t)q * N + c] = rid;
      if (req_row_off) req_row_off[(size_t)q * N + c] = rrow;
      if (own_id_off) own_id_off[(size_t)q * N + c] = oid;
      if (own_row_off) own_row_off[(size_t)q * N + c] = orow;
      rid += s;
      rrow += pad4(s * dims[c]);
      oid += r;
      orow += pad4(r * dims[c]);
      si += s;
      ri += r;
      rf += pad4(s * dims[c]);   // floats this rank gets back from owner q
      sf += pad4(r * dims[c]);   // floats this rank sends to requester q
    }
    HBK_REQUIRE(sf < (1ll << 31) && rf < (1ll << 31),
                "sharded_layout: more than 2^31 floats for one peer");
    if (ids_send_peer) ids_send_peer[q] = (int32_t)si;
    if (ids_recv_peer) ids_recv_peer[q] = (int32_t)ri;
    if (rows_send_peer) rows_send_peer[q] = (int32_t)sf;
    if (rows_recv_peer) rows_recv_peer[q] = (int32_t)rf;
  }
  if (col_shard_off) {
    for (int c = 0; c < N; ++c) {
      int64_t o = 0;
      for (int q = 0; q < W; ++q) {
        col_shard_off[(size_t)c * W + q] = o;
        o += S[(size_t)c * W + q];
      }
    }
  }
  return HBK_OK;
}

namespace hbk {
namespace {

struct Layout {
  std::vector<int32_t> dims, ids_send_peer, ids_recv_peer, rows_send_peer, rows_recv_peer;
  std::vector<int64_t> req_id_off, req_row_off, own_id_off, own_row_off, col_shard_off;
  int64_t req_ids = 0, own_ids = 0, req_floats = 0, own_floats = 0;
  int compute(int N, int W, const std::vector<hbk_sharded_column_t>& cols, const int32_t* S,
              const int32_t* R) {
    dims.resize(N);
    for (int c = 0; c < N; ++c) dims[c] = cols[c].dim;
    ids_send_peer.resize(W);
    ids_recv_peer.resize(W);
    rows_send_peer.resize(W);
    rows_recv_peer.resize(W);
    req_id_off.resize((size_t)N * W);
    req_row_off.resize((size_t)N * W);
    own_id_off.resize((size_t)N * W);
    own_row_off.resize((size_t)N * W);
    col_shard_off.resize((size_t)N * W);
    int rc = hbk_sharded_layout(N, W, dims.data(), S, R, ids_send_peer.data(),
                                ids_recv_peer.data(), rows_send_peer.data(),
                                rows_recv_peer.data(), req_id_off.data(), req_row_off.data(),
                                own_id_off.data(), own_row_off.data(), col_shard_off.data());
    if (rc != HBK_OK) return rc;
    req_ids = own_ids = req_floats = own_floats = 0;
    for (int q = 0; q < W; ++q) {
      req_ids += ids_send_peer[q];
      own_ids += ids_recv_peer[q];
      req_floats += rows_recv_peer[q];
      own_floats += rows_send_peer[q];
    }
    return HBK_OK;
  }
};

// One column group of the pipelined step: its own peer-major regions in the exchange buffers.
struct Group {
  int c0, c1;
  Layout lay;
  std::vector<int32_t> R;                         // [W][n_g]
  int64_t id_send, id_recv, row_send, row_recv;   // offsets of the group's regions
};

}  // namespace
}  // namespace hbk

struct hbk_sharded {
  hbk_comm_t comm;
  int W, rank, N;
  int32_t wire_dtype;
  bool id32;   // ids travel (and stay on the owner) as int32: every column is bucketized below 2^31
  bool trace;  // option sharded_trace: host-side phase times on stderr
  int n_groups;  // option sharded_groups
  bool inline_x; // option sharded_inline: exchanges on the compute stream, one column group
  std::vector<hbk_sharded_column_t> cols;
  // per-step state (kept for the backward)
  std::vector<int64_t> n_ids, n_seg;
  std::vector<int64_t> n_sent;       // ids of column c this rank put on the wire (= n_ids, or its
                                     // distinct ids when the column is deduplicated)
  std::vector<const int32_t*> row_splits;
  std::vector<int32_t> send_sizes;   // S [N][W] rows this rank requests from owner q, column c
  std::vector<int32_t> recv_sizes;   // R [W][N] rows requester q asked this rank for, column c
  std::vector<hbk::Group> groups;        // column groups of the last forward (reused backward)
  std::vector<int64_t> fwd_own_id_off;   // [W][N] where run (q, c) of the forward sits in recv_ids
  hipEvent_t ev[4][4];                   // [stage][group]: packed, ids in, gathered, rows in
  bool have_step;
  float host_us[3];     // host time of the last forward: enqueue 1-2, wait for the sizes, enqueue the rest
  // device buffers owned by the plan
  // ids_buf = ids coming in (the ids going out live in the step's PartSet), rows_buf = [rows
  // going out | rows coming in]: a run of this rank's OWN slice is addressed from the other side's
  // base (zero_copy_self) by a signed item offset
  hbk::Buffer part_ws, ids_buf, rows_buf, wire_ws, bwd_ws, runs_dev;
  char* send_ids_p;     // views into ids_buf / rows_buf, set by every forward
  char* recv_ids_p;
  float* send_rows_p;
  float* recv_rows_p;
  bool any_dedup;       // some column sends every distinct id once (hbk_sharded_column_t.dedup)
  hbk::Buffer dedup_tmp;  // backward of those columns: the requester-side IndexedSlices before they
                          // are placed in the outgoing buffer
  bool pack_early;      // option sharded_pack_early
  bool fused_half;      // fp16 wire, forward: the owner gather WRITES fp16 rows into the reply buffer and
                        // the stitch READS fp16 rows from the received one (hbk_lookup_column_t.half_io): the
                        // two cast passes and the wire workspace are gone (option sharded_wire_fused)
  bool zero_copy_grads; // the own slice of the BACKWARD's exchange stays in place too (fp32 wire only: the
                        // fp16 wire rounds the own slice's gradient rows like everybody else's, as the
                        // reference does)
  bool zero_copy_self;  // fp32 wire: the own slice never travels, not even as a device copy: the
                        // owner gather reads its ids where the pack left them and writes the rows
                        // where the stitch reads them (and the backward the other way round)
  // Stage 1-2 state (partitioned ids, shard index, size matrices), double buffered so that
  // hbk_sharded_prefetch can partition step i+1 while step i's exchanges are on the wire; the
  // set of the last forward stays untouched for its backward.
  struct PartSet {
    hbk::Buffer part_out, shard_index, sizes_dev;
    hbk::Buffer uniq, inv, nu, uniq_ws;   // deduplicated columns: distinct shard-ordered ids, their
                                          // inverse index, their counts (device), unique's workspace
    hbk::Buffer packed;                // the step's outgoing ids, peer-major per column group
    bool packed_early = false;         // ... written by run_partition (else by the forward)
    hbk::Buffer slots;                 // p2p form: the output slot of every outgoing id, same layout
    bool has_slots = false;            // ... written by run_partition's pack launch (round 6)
    int32_t* host_sizes = nullptr;     // pinned [3][N*W]: S, S^T, R as they sit on the device
    hipEvent_t done = nullptr;         // partition + size exchange + D2H copy finished
    hipEvent_t ready = nullptr;        // ... and the ids packed: the whole set is written
    std::vector<const int64_t*> ids;   // what was partitioned (match key of a prefetch)
    std::vector<int64_t> n_ids;
    bool pending = false;              // prefetched, not consumed yet
  } ps[2];
  int cur;                             // set of the last forward
  hipStream_t pre_stream;              // prefetch work runs here
  hipEvent_t step_begin;               // caller's stream at the entry of the last forward: all
                                       // readers of the OTHER set (previous step) are before it
  bool prefetch_used;                  // a prefetch was issued at some point
  // ---- p2p form of the forward (round 5; hbk_sharded_p2p_bind) ----------------------------------
  // Every rank registered its N output tensors once; peer_out[q * N + c] is where requester q's
  // column c lives as THIS process addresses it (the same process: its pointer; another process:
  // a hipIpcOpenMemHandle mapping).  A step then sends (id, output slot) pairs and the owner gather
  // stores every row straight into its place in the requester's output: no reply buffer, no rows
  // exchange, no stitch -- one random-row pass instead of two, and the rows cross the link as the
  // gather's own stores.
  // the forward in two halves (hbk_sharded_lookup_fwd_begin / _end): what the second half needs
  bool fwd_open = false;    // _begin has run, _end has not
  bool fin_wire = false, fin_hop = false, fin_p2p = false;
  int fin_use = 0;          // the PartSet of the open step
  std::chrono::steady_clock::time_point fin_t0;
  float fin_us[2] = {0.f, 0.f};
  bool p2p_opt = false;     // option sharded_p2p, read at creation
  bool p2p_bound = false;
  std::vector<float*> p2p_outs;         // [N] this rank's registered outputs
  std::vector<int32_t> p2p_strides;     // [N] their row strides (floats)
  std::vector<int64_t> p2p_rows;        // [N] their rows: remote owners store at slots < n_ids[c]
  std::vector<float*> peer_out;         // [W][N]
  std::vector<int32_t> peer_stride;     // [W][N]
  std::vector<void*> ipc_opened;        // mappings to close
  hbk::Buffer slot_send, slot_recv, token_buf, bind_buf;
  int64_t* host_runs;   // pinned [7][N*W], column-major: run starts / bases of the stitch, then
                        // run starts / id offsets / gradient offsets of the owner-side backward
};

extern "C" int hbk_sharded_create(hbk_sharded_t* plan, hbk_comm_t comm, int32_t n_cols,
                                  const hbk_sharded_column_t* cols, int32_t wire_dtype) {
  using namespace hbk;
  HBK_REQUIRE(plan != nullptr && comm != nullptr, "sharded_create: NULL argument");
  HBK_REQUIRE(n_cols >= 1 && cols != nullptr, "sharded_create: need at least one column");
  HBK_REQUIRE(wire_dtype == HBK_FLOAT || wire_dtype == HBK_HALF,
              "sharded_create: wire_dtype must be float or half");
  for (int32_t c = 0; c < n_cols; ++c) {
    HBK_REQUIRE(cols[c].dim >= 1 && cols[c].rows_local >= 0 && cols[c].bucket >= 0,
                "sharded_create: column %d: bad dim / rows / bucket", c);
    HBK_REQUIRE(cols[c].combiner >= HBK_COMBINER_SUM && cols[c].combiner <= HBK_COMBINER_SQRTN,
                "sharded_create: column %d: unknown combiner %d", c, cols[c].combiner);
    HBK_REQUIRE(cols[c].shard != nullptr || cols[c].rows_local == 0,
                "sharded_create: column %d: shard is NULL", c);
  }
  hbk_sharded* p = new hbk_sharded();
  p->comm = comm;
  p->W = hbk_comm_world_size(comm);
  p->rank = hbk_comm_rank(comm);
  p->N = n_cols;
  p->wire_dtype = wire_dtype;
  p->cols.assign(cols, cols + n_cols);
  // After the bucketize ids are < bucket, so when every bucket fits int32 the id exchange moves
  // half the bytes (the reference always sends the tensor's own dtype, nccl_collective.cc:257-259;
  // SURVEY 8e).  Option sharded_id64 keeps int64 on the wire.  (All options are read here, at
  // plan creation, not per step.)
  p->id32 = options().sharded_id64 == 0;
  p->trace = options().sharded_trace != 0;
  p->n_groups = options().sharded_groups;
  p->inline_x = options().sharded_inline != 0;
  p->p2p_opt = options().sharded_p2p != 0;
  for (int32_t c = 0; c < n_cols; ++c) {
    if (cols[c].bucket <= 0 || cols[c].bucket > 0x7fffffffll) p->id32 = false;
  }
  p->have_step = false;
  p->any_dedup = false;
  for (int32_t c = 0; c < n_cols; ++c) p->any_dedup = p->any_dedup || cols[c].dedup != 0;
  p->fused_half = wire_dtype == HBK_HALF && options().sharded_wire_fused != 0;
  p->pack_early = options().sharded_pack_early != 0;
  p->zero_copy_self = (wire_dtype == HBK_FLOAT || p->fused_half) && options().sharded_copy_self == 0;
  p->zero_copy_grads = p->zero_copy_self && wire_dtype == HBK_FLOAT;
  p->send_ids_p = p->recv_ids_p = nullptr;
  p->send_rows_p = p->recv_rows_p = nullptr;
  p->cur = 0;
  p->pre_stream = nullptr;
  p->step_begin = nullptr;
  p->prefetch_used = false;
  p->host_runs = nullptr;
  for (auto& st : p->ev) {
    for (auto& e : st) {
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
        delete p;
        return fail(HBK_INTERNAL, "sharded_create: hipEventCreate failed");
      }
    }
  }
  // (the prefetch stream is created by the first hbk_sharded_prefetch: HIP deals its streams onto a
  // few hardware queues -- 4 by default -- and a stream that is never used should not take a slot
  // next to the compute and communicator streams, see hbk_sharded_prefetch_on)
  bool ok = hipEventCreateWithFlags(&p->step_begin, hipEventDisableTiming) == hipSuccess;
  for (auto& set : p->ps) {
    ok = ok && hipEventCreateWithFlags(&set.done, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&set.ready, hipEventDisableTiming) == hipSuccess &&
         hipHostMalloc(reinterpret_cast<void**>(&set.host_sizes),
                       sizeof(int32_t) * 3 * (size_t)n_cols * p->W,
                       hipHostMallocDefault) == hipSuccess;
  }
  if (!ok ||
      hipHostMalloc(reinterpret_cast<void**>(&p->host_runs),
                    sizeof(int64_t) * 7 * (size_t)n_cols * p->W, hipHostMallocDefault) !=
          hipSuccess) {
    hbk_sharded_destroy(p);
    return fail(HBK_INTERNAL, "sharded_create: could not create streams / events / pinned memory");
  }
  *plan = p;
  return HBK_OK;
}

extern "C" int hbk_sharded_destroy(hbk_sharded_t p) {
  if (p == nullptr) return HBK_OK;
  if (p->pre_stream) (void)hipStreamSynchronize(p->pre_stream);
  for (hbk::Buffer* b : {&p->part_ws, &p->ids_buf, &p->rows_buf, &p->wire_ws, &p->bwd_ws,
                         &p->runs_dev, &p->dedup_tmp, &p->slot_send, &p->slot_recv, &p->token_buf,
                         &p->bind_buf}) {
    b->release();
  }
  for (void* m : p->ipc_opened) (void)hipIpcCloseMemHandle(m);
  p->ipc_opened.clear();
  for (auto& set : p->ps) {
    set.part_out.release();
    set.shard_index.release();
    set.sizes_dev.release();
    set.uniq.release();
    set.inv.release();
    set.nu.release();
    set.uniq_ws.release();
    set.packed.release();
    set.slots.release();
    if (set.host_sizes) (void)hipHostFree(set.host_sizes);
    if (set.done) (void)hipEventDestroy(set.done);
    if (set.ready) (void)hipEventDestroy(set.ready);
  }
  if (p->step_begin) (void)hipEventDestroy(p->step_begin);
  if (p->pre_stream) (void)hipStreamDestroy(p->pre_stream);
  if (p->host_runs) (void)hipHostFree(p->host_runs);
  for (auto& st : p->ev) {
    for (auto& e : st) (void)hipEventDestroy(e);
  }
  delete p;
  return HBK_OK;
}

namespace hbk {
namespace {

inline Seg make_seg(const void* src, void* dst, int64_t bytes, int narrow = 0) {
  Seg s;
  s.src = src;
  s.dst = dst;
  s.bytes = bytes;
  s.tile0 = 0;
  s.narrow = narrow;
  return s;
}

// one packed exchange: a single buffer, one message per peer.  With events the exchange runs on
// the communicator's stream behind `before` and records `after`; the compute stream goes on.
int exchange(hbk_sharded* p, int32_t dtype, int32_t wire, const void* in, const int32_t* send,
             void* out, const int32_t* recv, hbk_stream_t stream, hipEvent_t before = nullptr,
             hipEvent_t after = nullptr, void* wire_ws = nullptr, size_t wire_ws_bytes = 0,
             bool skip_self = false) {
  const int64_t cs[1] = {1};
  const void* vin[1] = {in};
  void* vout[1] = {out};
  if (wire != dtype && wire_ws == nullptr) {
    const size_t wws = hbk_alltoallv_wire_workspace_bytes(1, cs, send, recv, p->W);
    int rc = p->wire_ws.ensure(wws + 16);
    if (rc != HBK_OK) return rc;
    wire_ws = p->wire_ws.ptr;
    wire_ws_bytes = p->wire_ws.bytes;
  }
  return alltoallv_events(p->comm, 1, dtype, wire, HBK_TOPOLOGY_ALL, cs, vin, send, vout, recv,
                          wire_ws, wire_ws_bytes, stream, before, after, skip_self,
                          p->inline_x && before != nullptr);
}

// Number of column groups the step pipelines.  More groups hide more of the gather / stitch
// behind the exchanges (exposed compute ~ 1/G of it) at the price of G x more launches and
// smaller kernels: measured on one rank 298 us (G = 1), 318 us (G = 2), 455 us (G = 4) per
// forward step.  2 until an 8-GPU measurement says otherwise; option sharded_groups overrides (1..4).
int pipeline_groups(int n_cols, int world, int requested) {
  int g = requested >= 1 && requested <= 4 ? requested : 2;
  if (world == 1 && !(requested >= 1 && requested <= 4)) g = 1;   // nothing on the wire to hide
  return g < n_cols ? g : n_cols;
}

// (inline exchanges: nothing runs beside them, one group is all there is to schedule)
int step_groups(const hbk_sharded* p) {
  if (p->p2p_bound) return 1;   // (the p2p form gathers all columns in one launch)
  return p->inline_x && !(p->n_groups >= 1 && p->n_groups <= 4)
             ? 1 : pipeline_groups(p->N, p->W, p->n_groups);
}

// Stages 1-2 of a step into `set`: bucketize + stable partition of all columns, ONE [N x W] size
// exchange, sizes to the host (asynchronously: set.done fires when they have arrived).
int run_partition(hbk_sharded* p, hbk_sharded::PartSet& set, const int64_t* const* ids,
                  const int64_t* n_ids, hipStream_t stream) {
  const int N = p->N, W = p->W;
  int64_t total = 0;
  for (int c = 0; c < N; ++c) {
    HBK_REQUIRE(n_ids[c] >= 0 && n_ids[c] < (1ll << 31), "sharded lookup: bad n_ids[%d]", c);
    total += n_ids[c];
  }
  int rc;
  if ((rc = set.part_out.ensure((size_t)total * 8 + 8)) != HBK_OK) return rc;
  if ((rc = set.shard_index.ensure((size_t)total * 4 + 8)) != HBK_OK) return rc;
  if ((rc = set.sizes_dev.ensure((size_t)N * W * 4 * 3)) != HBK_OK) return rc;
  if ((rc = set.packed.ensure((size_t)total * (p->id32 ? 4 : 8) + 16)) != HBK_OK) return rc;
  std::vector<int64_t*> pout(N);
  std::vector<int32_t*> sizes(N), idx(N);
  std::vector<int64_t> buckets(N);
  int32_t* sizes_dev = reinterpret_cast<int32_t*>(set.sizes_dev.ptr);       // S [N][W]
  int32_t* sizes_t = sizes_dev + (size_t)N * W;                            // S^T [W][N]
  int32_t* recv_t = sizes_t + (size_t)N * W;                               // R [W][N]
  int64_t off = 0;
  for (int c = 0; c < N; ++c) {
    pout[c] = reinterpret_cast<int64_t*>(set.part_out.ptr) + off;
    idx[c] = reinterpret_cast<int32_t*>(set.shard_index.ptr) + off;
    sizes[c] = sizes_dev + (size_t)c * W;
    buckets[c] = p->cols[c].bucket;
    off += n_ids[c];
  }
  const size_t ws = hbk_partition_workspace_bytes(N, n_ids, W);
  if ((rc = p->part_ws.ensure(ws + 8)) != HBK_OK) return rc;
  rc = partition_by_modulo_fused(N, W, ids, n_ids, buckets.data(), pout.data(), sizes.data(),
                                 idx.data(), sizes_t, p->part_ws.ptr, p->part_ws.bytes, stream);
  if (rc != HBK_OK) return rc;
  // requester-side dedup: distinct ids of the flagged columns, composed index, sizes S'
  std::vector<const int64_t*> id_src(N);   // what goes on the wire: shard-ordered ids, or the distinct ones
  for (int c = 0; c < N; ++c) id_src[c] = pout[c];
  if (p->any_dedup) {
    if ((rc = set.uniq.ensure((size_t)total * 8 + 8)) != HBK_OK) return rc;
    if ((rc = set.inv.ensure((size_t)total * 4 + 8)) != HBK_OK) return rc;
    if ((rc = set.nu.ensure((size_t)N * 4 + 8)) != HBK_OK) return rc;
    HBK_HIP_OK(hipMemsetAsync(set.nu.ptr, 0, (size_t)N * 4, stream));
    std::vector<const int64_t*> uin;
    std::vector<int64_t> ulen;
    std::vector<int64_t*> uout;
    std::vector<int32_t*> iout, nout;
    DedupArgs a;
    int64_t tiles = 0, o = 0;
    int nd = 0;
    for (int c = 0; c < N; ++c) {
      if (p->cols[c].dedup != 0 && n_ids[c] > 0) {
        HBK_REQUIRE(nd < kMaxPackCols && W <= kMaxPackWorld,
                    "sharded lookup: dedup supports up to %d columns and %d ranks", kMaxPackCols,
                    kMaxPackWorld);
        int64_t* u = reinterpret_cast<int64_t*>(set.uniq.ptr) + o;
        uin.push_back(pout[c]);
        ulen.push_back(n_ids[c]);
        uout.push_back(u);
        iout.push_back(reinterpret_cast<int32_t*>(set.inv.ptr) + o);
        nout.push_back(reinterpret_cast<int32_t*>(set.nu.ptr) + c);
        id_src[c] = u;
        a.off[nd] = o;
        a.n[nd] = (int32_t)n_ids[c];
        a.tile0[nd] = (int32_t)tiles;
        a.col[nd] = (int16_t)c;
        tiles += (n_ids[c] + kDedupTile - 1) / kDedupTile;
        ++nd;
      }
      o += n_ids[c];
    }
    if (nd > 0) {
      const size_t uws = hbk_unique_workspace_bytes(nd, ulen.data());
      if ((rc = set.uniq_ws.ensure(uws + 8)) != HBK_OK) return rc;
      rc = hbk_unique_n(nd, uin.data(), ulen.data(), uout.data(), iout.data(), nout.data(),
                        set.uniq_ws.ptr, set.uniq_ws.bytes, reinterpret_cast<hbk_stream_t>(stream));
      if (rc != HBK_OK) return rc;
      a.uniq = reinterpret_cast<const int64_t*>(set.uniq.ptr);
      a.inv = reinterpret_cast<const int32_t*>(set.inv.ptr);
      a.idx = reinterpret_cast<int32_t*>(set.shard_index.ptr);
      a.nu = reinterpret_cast<const int32_t*>(set.nu.ptr);
      a.S = sizes_dev;
      a.St = sizes_t;
      a.wdiv = make_fastdiv((uint64_t)W);
      a.wdiv.d = (uint64_t)W;
      a.n_cols = nd;
      a.W = W;
      a.N = N;
      a.pad_ = 0;
      hipLaunchKernelGGL(dedup_compose_kernel, dim3((unsigned)tiles), dim3(kBlock), 0, stream, a);
      hipLaunchKernelGGL(dedup_sizes_kernel, dim3((unsigned)nd), dim3(kMaxPackWorld), 0, stream, a);
      HBK_HIP_OK(hipGetLastError());
    }
  }
  const void* sin[1] = {sizes_t};
  void* sout[1] = {recv_t};
  const int64_t cnt[1] = {(int64_t)N * W};
  rc = hbk_alltoall_n(p->comm, 1, HBK_INT32, HBK_TOPOLOGY_ALL, sin, cnt, sout,
                      reinterpret_cast<hbk_stream_t>(stream));
  if (rc != HBK_OK) return rc;
  // S, S^T and R sit back to back: one copy brings S and R to the host
  HBK_HIP_OK(hipMemcpyAsync(set.host_sizes, sizes_dev, sizeof(int32_t) * 3 * N * W,
                            hipMemcpyDeviceToHost, stream));
  HBK_HIP_OK(hipEventRecord(set.done, stream));
  // the ids go peer-major into the outgoing buffer while the host waits for `done` (pack_ids_kernel)
  set.packed_early = p->pack_early && N <= kMaxPackCols && W <= kMaxPackWorld && total > 0;
  set.has_slots = false;
  if (set.packed_early) {
    PackArgs a;
    a.slots = nullptr;
    if (p->p2p_bound) {   // the p2p form: slots ride with the pack (one column group)
      if ((rc = set.slots.ensure((size_t)total * 4 + 16)) != HBK_OK) return rc;
      a.slots = reinterpret_cast<int32_t*>(set.slots.ptr);
      set.has_slots = true;
    }
    a.S = sizes_dev;
    a.dst = set.packed.ptr;
    a.dst_items = total;
    a.n_cols = N;
    a.W = W;
    a.narrow = p->id32 ? 1 : 0;
    a.pad_ = 0;
    const int G = step_groups(p);
    int64_t tiles = 0, gbase = 0;
    for (int g = 0; g < G; ++g) {
      const int c0 = (int)((int64_t)N * g / G), c1 = (int)((int64_t)N * (g + 1) / G);
      int64_t in_group = 0;
      for (int c = c0; c < c1; ++c) {
        a.src[c] = id_src[c];
        a.idx[c] = idx[c];
        a.gbase[c] = gbase;
        a.n[c] = (int32_t)n_ids[c];
        a.tile0[c] = (int32_t)tiles;
        a.g0[c] = (int16_t)c0;
        a.g1[c] = (int16_t)c1;
        tiles += (n_ids[c] + kPackTile - 1) / kPackTile;
        in_group += n_ids[c];
      }
      gbase += in_group;
    }
    hipLaunchKernelGGL(pack_ids_kernel, dim3((unsigned)tiles), dim3(kBlock), 0, stream, a);
    HBK_HIP_OK(hipGetLastError());
  }
  HBK_HIP_OK(hipEventRecord(set.ready, stream));
  set.ids.assign(ids, ids + N);
  set.n_ids.assign(n_ids, n_ids + N);
  return HBK_OK;
}

}  // namespace
}  // namespace hbk

namespace hbk {
namespace {
struct P2pRec {
  // who owns the tensor: pid alone is not an identity (ranks in different containers or pid
  // namespaces collide on small pids), so a record also carries a hash of the host's boot id /
  // name and a random number drawn once per process; two ranks share an address space only when
  // all three agree
  uint64_t pid;
  uint64_t host;
  uint64_t nonce;
  int32_t device;     // the owner's HIP device (ranks of one process may sit on different ones)
  int32_t pad;
  int64_t rows;       // rows of the registered tensor
  uint64_t ptr;       // the output tensor in its owner's address space
  uint64_t offset;    // ... and inside the allocation the handle names
  int32_t stride;
  int32_t has_handle;
  hipIpcMemHandle_t handle;
};
uint64_t fnv1a(const char* s, size_t n, uint64_t h = 1469598103934665603ull) {
  for (size_t i = 0; i < n; ++i) h = (h ^ (unsigned char)s[i]) * 1099511628211ull;
  return h;
}
// (boot id of the kernel this process runs on + host name: equal for the ranks of one machine)
uint64_t host_identity() {
  char buf[256];
  uint64_t h = 1469598103934665603ull;
  FILE* f = fopen("/proc/sys/kernel/random/boot_id", "r");
  if (f != nullptr) {
    const size_t n = fread(buf, 1, sizeof(buf), f);
    fclose(f);
    h = fnv1a(buf, n, h);
  }
  if (gethostname(buf, sizeof(buf)) == 0) h = fnv1a(buf, strnlen(buf, sizeof(buf)), h);
  return h;
}
uint64_t process_nonce() {
  static const uint64_t nonce = [] {
    std::random_device rd;
    uint64_t v = ((uint64_t)rd() << 32) ^ (uint64_t)rd();
    v ^= (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() * 0x9e3779b97f4a7c15ull;
    return v | 1ull;
  }();
  return nonce;
}
}  // namespace
}  // namespace hbk

// Registers this rank's N output tensors for the p2p form of the forward (collective: every rank
// of the communicator calls it, with its own tensors).  One id per segment, no requester-side
// dedup, fp32 wire; every later forward of the plan must be handed exactly these tensors.
// HBK_UNIMPLEMENTED (on every rank) when some peer's memory cannot be mapped here -- the plan
// then keeps the exchange form.
extern "C" int hbk_sharded_p2p_bind(hbk_sharded_t p, float* const* outs, const int32_t* out_strides,
                                    const int64_t* out_rows, hbk_stream_t stream_) {
  using namespace hbk;
  HBK_REQUIRE(p != nullptr && outs != nullptr && out_rows != nullptr,
              "sharded_p2p_bind: NULL argument");
  hipStream_t stream = as_stream(stream_);
  const int N = p->N, W = p->W, me = p->rank;
  p->p2p_bound = false;
  p->ps[0].pending = p->ps[1].pending = false;   // (a prefetch packed for another group count)
  if (!p->p2p_opt) return fail(HBK_UNIMPLEMENTED, "sharded_p2p_bind: option sharded_p2p is off");
  HBK_REQUIRE(p->wire_dtype == HBK_FLOAT, "sharded_p2p_bind: the p2p form has no fp16 wire");
  HBK_REQUIRE(!p->any_dedup, "sharded_p2p_bind: not with requester-side dedup");
  HBK_REQUIRE(W <= kMaxPackWorldP2p, "sharded_p2p_bind: more than %d ranks", kMaxPackWorldP2p);
  for (void* m : p->ipc_opened) (void)hipIpcCloseMemHandle(m);
  p->ipc_opened.clear();
  std::vector<P2pRec> mine((size_t)N), all((size_t)N * W);
  const uint64_t pid = (uint64_t)getpid(), host = host_identity(), nonce = process_nonce();
  int my_device = 0;
  HBK_HIP_OK(hipGetDevice(&my_device));
  for (int c = 0; c < N; ++c) {
    HBK_REQUIRE(outs[c] != nullptr, "sharded_p2p_bind: outs[%d] is NULL", c);
    HBK_REQUIRE(out_rows[c] >= 0 && out_rows[c] < (1ll << 31), "sharded_p2p_bind: bad out_rows[%d]", c);
    P2pRec& r = mine[c];
    memset(&r, 0, sizeof(r));
    r.pid = pid;
    r.host = host;
    r.nonce = nonce;
    r.device = my_device;
    r.rows = out_rows[c];
    r.ptr = (uint64_t)(uintptr_t)outs[c];
    r.stride = out_strides != nullptr && out_strides[c] > 0 ? out_strides[c] : p->cols[c].dim;
    if (W > 1) {
      hipDeviceptr_t base = nullptr;
      size_t size = 0;
      if (hipMemGetAddressRange(&base, &size, outs[c]) == hipSuccess && base != nullptr) {
        r.offset = (uint64_t)((uintptr_t)outs[c] - (uintptr_t)base);
        r.has_handle = hipIpcGetMemHandle(&r.handle, base) == hipSuccess ? 1 : 0;
      }
      (void)hipGetLastError();
    }
  }
  int rc;
  const size_t rec_bytes = sizeof(P2pRec) * (size_t)N;
  if ((rc = p->bind_buf.ensure(rec_bytes * (size_t)(W + 1) + 64)) != HBK_OK) return rc;
  char* d_mine = reinterpret_cast<char*>(p->bind_buf.ptr);
  char* d_all = d_mine + rec_bytes;
  HBK_HIP_OK(hipMemcpyAsync(d_mine, mine.data(), rec_bytes, hipMemcpyHostToDevice, stream));
  std::vector<int64_t> counts((size_t)W, (int64_t)rec_bytes);
  if ((rc = hbk_allgatherv(p->comm, HBK_UINT8, d_mine, counts.data(), d_all, stream_)) != HBK_OK) {
    return rc;
  }
  HBK_HIP_OK(hipMemcpyAsync(all.data(), d_all, rec_bytes * (size_t)W, hipMemcpyDeviceToHost, stream));
  HBK_HIP_OK(hipStreamSynchronize(stream));
  p->peer_out.assign((size_t)W * N, nullptr);
  p->peer_stride.assign((size_t)W * N, 0);
  int32_t ok = 1;
  std::vector<std::pair<hipIpcMemHandle_t, void*>> opened;
  for (int q = 0; q < W && ok; ++q) {
    for (int c = 0; c < N && ok; ++c) {
      const P2pRec& r = all[(size_t)q * N + c];
      p->peer_stride[(size_t)q * N + c] = r.stride;
      const bool same_process = r.pid == pid && r.host == host && r.nonce == nonce;
      if (q == me || same_process) {   // the same address space (in-process ranks, this rank itself)
        if (r.device != my_device) {
          // an in-process rank on another GPU: its pointer is valid here only with peer access
          int can = 0;
          hipError_t e = hipDeviceCanAccessPeer(&can, my_device, r.device);
          if (e == hipSuccess && can) {
            e = hipDeviceEnablePeerAccess(r.device, 0);
            if (e == hipErrorPeerAccessAlreadyEnabled) e = hipSuccess;
          }
          (void)hipGetLastError();
          if (e != hipSuccess || !can) {
            ok = 0;
            break;
          }
        }
        p->peer_out[(size_t)q * N + c] = reinterpret_cast<float*>((uintptr_t)r.ptr);
        continue;
      }
      if (r.host != host) {   // another machine: nothing to map (IPC handles are per host)
        ok = 0;
        break;
      }
      if (!r.has_handle) {
        ok = 0;
        break;
      }
      void* base = nullptr;
      for (auto& o : opened) {
        if (memcmp(&o.first, &r.handle, sizeof(hipIpcMemHandle_t)) == 0) base = o.second;
      }
      if (base == nullptr) {
        if (hipIpcOpenMemHandle(&base, r.handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
          (void)hipGetLastError();
          ok = 0;
          break;
        }
        opened.emplace_back(r.handle, base);
        p->ipc_opened.push_back(base);
      }
      p->peer_out[(size_t)q * N + c] =
          reinterpret_cast<float*>(reinterpret_cast<char*>(base) + r.offset);
    }
  }
  if (options().sharded_p2p_test_refuse == me) ok = 0;   // test hook: "hipIpcOpenMemHandle refused"
  // every rank must come to the same answer: the minimum over the ranks of "all mapped"
  if (W > 1) {
    int32_t* d_ok = reinterpret_cast<int32_t*>(d_mine);
    HBK_HIP_OK(hipMemcpyAsync(d_ok, &ok, sizeof(ok), hipMemcpyHostToDevice, stream));
    const void* in[1] = {d_ok};
    void* out[1] = {d_ok + 4};
    const int64_t one[1] = {1};
    const size_t ws = hbk_allreduce_workspace_bytes(1, one, HBK_INT32);
    Buffer red;
    if ((rc = red.ensure(ws + 16)) != HBK_OK) return rc;
    rc = hbk_allreduce_n(p->comm, 1, HBK_INT32, 3 /* min */, in, one, out, 1.0f, red.ptr, red.bytes,
                         stream_);
    if (rc == HBK_OK) {
      rc = hipMemcpyAsync(&ok, d_ok + 4, sizeof(ok), hipMemcpyDeviceToHost, stream) == hipSuccess &&
                   hipStreamSynchronize(stream) == hipSuccess
               ? HBK_OK : HBK_INTERNAL;
    }
    red.release();
    if (rc != HBK_OK) return fail(rc, "sharded_p2p_bind: could not agree on the mapping");
  }
  if (!ok) {
    for (void* m : p->ipc_opened) (void)hipIpcCloseMemHandle(m);
    p->ipc_opened.clear();
    return fail(HBK_UNIMPLEMENTED, "sharded_p2p_bind: a peer's output memory cannot be mapped "
                                   "(hipIpcGetMemHandle / hipIpcOpenMemHandle); the plan keeps the "
                                   "exchange form");
  }
  p->p2p_outs.assign(outs, outs + N);
  p->p2p_strides.resize((size_t)N);
  p->p2p_rows.assign(out_rows, out_rows + N);
  for (int c = 0; c < N; ++c) p->p2p_strides[c] = mine[c].stride;
  p->p2p_bound = true;
  return HBK_OK;
}

extern "C" int hbk_sharded_p2p_unbind(hbk_sharded_t p) {
  using namespace hbk;
  HBK_REQUIRE(p != nullptr, "sharded_p2p_unbind: plan is NULL");
  p->p2p_bound = false;
  p->ps[0].pending = p->ps[1].pending = false;
  for (void* m : p->ipc_opened) (void)hipIpcCloseMemHandle(m);
  p->ipc_opened.clear();
  return HBK_OK;
}

// The forward in two halves.  _begin: everything up to and including the owner-side gather (partition
// or its prefetched result, the step's one host wait, id exchange, gather into the reply buffer --
// in the p2p form into the requesters' outputs, with the token exchange).  _end: rows exchange +
// stitch + combiner into `outs`.  hbk_sharded_lookup_fwd = _begin + _end.  Two plans that share
// a communicator and alternate  begin(B, step i + 1); end(A, step i)  put ids(i + 1) on the wire AHEAD
// of rows(i): plan B gathers while plan A's rows travel, plan A stitches while plan B's travel --
// the overlap of exchanges with the local gather across STEPS (forward-only use: nothing may change
// the tables between a step's _begin and its _end).
extern "C" int hbk_sharded_lookup_fwd_begin(hbk_sharded_t p, const int64_t* const* ids,
                                            const int64_t* n_ids,
                                            const int32_t* const* row_splits,
                                            const int64_t* n_segments, hbk_stream_t stream_) {
  using namespace hbk;
  HBK_REQUIRE(p != nullptr, "sharded_lookup_fwd: plan is NULL");
  HBK_REQUIRE(ids && n_ids, "sharded_lookup_fwd: NULL argument array");
  p->fwd_open = false;
  hipStream_t stream = as_stream(stream_);
  const int N = p->N, W = p->W;
  p->n_ids.assign(n_ids, n_ids + N);
  p->n_seg.resize(N);
  p->row_splits.resize(N);
  for (int c = 0; c < N; ++c) {
    HBK_REQUIRE(n_ids[c] >= 0 && n_ids[c] < (1ll << 31), "sharded_lookup_fwd: bad n_ids[%d]", c);
    p->row_splits[c] = row_splits ? row_splits[c] : nullptr;
    HBK_REQUIRE(p->row_splits[c] == nullptr || n_segments != nullptr,
                "sharded_lookup_fwd: n_segments is NULL");
    p->n_seg[c] = p->row_splits[c] ? n_segments[c] : n_ids[c];
  }
  int rc;
  // option sharded_trace: host-side time of the step's phases on stderr (us)
  const bool trace = p->trace;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto us_since = [](std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  };
  const auto t_begin = now();
  // ---- 1-2 partition + size exchange: taken from a matching prefetch, else done now -----------
  int use = -1;
  for (int k = 0; k < 2; ++k) {
    hbk_sharded::PartSet& cand = p->ps[k];
    if (!cand.pending) continue;
    bool same = true;
    for (int c = 0; c < N && same; ++c) same = cand.ids[c] == ids[c] && cand.n_ids[c] == n_ids[c];
    if (same) {
      use = k;
    }
    cand.pending = false;   // a prefetch that is not consumed by the next forward is dropped
  }
  const bool prefetched = use >= 0;
  if (!prefetched) {
    use = p->cur ^ 1;       // never the set the last forward (and its backward) lives in
    if (p->prefetch_used) {
      // a dropped prefetch may still be writing that set / the partition workspace on pre_stream
      HBK_HIP_OK(hipStreamWaitEvent(stream, p->ps[0].ready, 0));
      HBK_HIP_OK(hipStreamWaitEvent(stream, p->ps[1].ready, 0));
    }
    HBK_HIP_OK(hipEventRecord(p->step_begin, stream));
    if ((rc = run_partition(p, p->ps[use], ids, n_ids, stream)) != HBK_OK) return rc;
  } else {
    HBK_HIP_OK(hipEventRecord(p->step_begin, stream));
  }
  hbk_sharded::PartSet& set = p->ps[use];
  p->cur = use;
  const double t_enq1 = us_since(t_begin);
  HBK_HIP_OK(hipEventSynchronize(set.done));   // the step's one host wait: sizes are on the host
  // the partition's one-launch form may have given up (bounded waits, sync.hip): then the sizes
  // just read are not valid -- THIS step fails, before anything is sized from them
  // (the partition ran on the caller's stream or, prefetched, on the plan's own)
  if ((rc = sync_check("sharded_lookup_fwd", stream)) != HBK_OK) return rc;
  if (p->pre_stream != nullptr &&
      (rc = sync_check("sharded_lookup_fwd", p->pre_stream)) != HBK_OK) {
    return rc;
  }
  if (prefetched) HBK_HIP_OK(hipStreamWaitEvent(stream, set.ready, 0));
  const double t_sync = us_since(t_begin);
  p->send_sizes.assign(set.host_sizes, set.host_sizes + (size_t)N * W);
  p->recv_sizes.assign(set.host_sizes + 2 * (size_t)N * W, set.host_sizes + 3 * (size_t)N * W);
  std::vector<int64_t*> pout(N);
  std::vector<int32_t*> idx(N);
  p->n_sent.assign(N, 0);
  {
    int64_t off = 0;
    for (int c = 0; c < N; ++c) {
      pout[c] = reinterpret_cast<int64_t*>(set.part_out.ptr) + off;
      if (p->cols[c].dedup != 0 && n_ids[c] > 0) {   // the distinct ids travel (run_partition)
        pout[c] = reinterpret_cast<int64_t*>(set.uniq.ptr) + off;
      }
      idx[c] = reinterpret_cast<int32_t*>(set.shard_index.ptr) + off;
      off += n_ids[c];
      for (int q = 0; q < W; ++q) p->n_sent[c] += p->send_sizes[(size_t)c * W + q];
      HBK_REQUIRE(p->n_sent[c] <= n_ids[c], "sharded_lookup_fwd: column %d: sizes exceed its ids", c);
    }
  }
  const int32_t* S = p->send_sizes.data();
  const int32_t* R = p->recv_sizes.data();
  // ---- 3..6 pipelined over column groups ------------------------------------------------------
  // The columns are split into G groups, each with its own peer-major buffers.  The exchanges
  // run back to back on the communicator's stream; the compute stream gathers group g while the
  // ids of group g+1 are on the wire, and stitches group g while the rows of group g+1 travel:
  //   comm    : ids(0) ids(1) ...            rows(0)      rows(1) ...
  //   compute : pack(0..G-1)      gather(0)  gather(1) ..      stitch(0)   stitch(1)
  // (inline exchanges: nothing runs beside them, one group is all there is to schedule)
  const int G = step_groups(p);
  // p2p form (hbk_sharded_p2p_bind): the owner gather stores every row into the requester's
  // registered output -- (id, slot) pairs out, no rows exchange, no stitch
  const bool p2p = p->p2p_bound;
  if (p2p) {
    for (int c = 0; c < N; ++c) {
      HBK_REQUIRE(p->row_splits[c] == nullptr,
                  "sharded_lookup_fwd: column %d is ragged: a plan with registered outputs "
                  "(hbk_sharded_p2p_bind) takes one id per segment; unbind it first", c);
      // the owners store row j of this step at slot j of the REGISTERED tensor, from other ranks:
      // a batch larger than what was registered would be written out of bounds in this rank's memory
      HBK_REQUIRE(n_ids[c] <= p->p2p_rows[c],
                  "sharded_lookup_fwd: column %d: %lld ids for a registered output of %lld rows "
                  "(hbk_sharded_p2p_bind)", c, (long long)n_ids[c], (long long)p->p2p_rows[c]);
    }
  }
  std::vector<Group>& groups = p->groups;
  groups.assign(G, Group());
  int64_t tot_req_ids = 0, tot_own_ids = 0, tot_own_floats = 0, tot_req_floats = 0;
  p->fwd_own_id_off.assign((size_t)N * W, 0);
  for (int g = 0; g < G; ++g) {
    Group& gr = groups[g];
    gr.c0 = (int)((int64_t)N * g / G);
    gr.c1 = (int)((int64_t)N * (g + 1) / G);
    const int ng = gr.c1 - gr.c0;
    gr.R.resize((size_t)W * ng);
    for (int q = 0; q < W; ++q) {
      for (int c = 0; c < ng; ++c) gr.R[(size_t)q * ng + c] = R[(size_t)q * N + gr.c0 + c];
    }
    std::vector<hbk_sharded_column_t> sub(p->cols.begin() + gr.c0, p->cols.begin() + gr.c1);
    if ((rc = gr.lay.compute(ng, W, sub, S + (size_t)gr.c0 * W, gr.R.data())) != HBK_OK) return rc;
    // (capacity, not the ids actually sent: the early pack placed the groups before the host knew
    // how many ids a deduplicated column keeps; without dedup the two are the same)
    gr.id_send = tot_req_ids;
    gr.id_recv = tot_own_ids;
    gr.row_send = tot_own_floats;
    gr.row_recv = tot_req_floats;
    for (int c = gr.c0; c < gr.c1; ++c) tot_req_ids += n_ids[c];
    tot_own_ids += gr.lay.own_ids;
    tot_own_floats += gr.lay.own_floats;
    tot_req_floats += gr.lay.req_floats;
    for (int q = 0; q < W; ++q) {
      for (int c = 0; c < ng; ++c) {
        p->fwd_own_id_off[(size_t)q * N + gr.c0 + c] = gr.id_recv + gr.lay.own_id_off[(size_t)q * ng + c];
      }
    }
  }
  const size_t id_bytes = p->id32 ? 4 : 8;
  const int32_t id_dtype = p->id32 ? HBK_INT32 : HBK_INT64;
  auto up256 = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t rows_send_bytes = up256((size_t)tot_own_floats * 4 + 16);
  if ((rc = p->ids_buf.ensure((size_t)tot_own_ids * id_bytes + 16)) != HBK_OK) return rc;
  if ((rc = p->rows_buf.ensure(rows_send_bytes + (size_t)tot_req_floats * 4 + 16)) != HBK_OK) {
    return rc;
  }
  // outgoing ids: the set's own buffer (run_partition packed them, or the forward does below);
  // incoming ids: ids_buf.  Both are hipMalloc'ed (256-byte aligned): a run of this rank's OWN
  // slice is addressed from the other side's base by a whole (signed) number of items
  p->send_ids_p = reinterpret_cast<char*>(set.packed.ptr);
  p->recv_ids_p = reinterpret_cast<char*>(p->ids_buf.ptr);
  p->send_rows_p = reinterpret_cast<float*>(p->rows_buf.ptr);
  p->recv_rows_p = p->send_rows_p + rows_send_bytes / 4;
  if (p->wire_dtype == HBK_HALF && !p->fused_half) {  // staging for the largest group (exchanges are serial)
    size_t wws = 0;
    const int64_t cs1[1] = {1};
    for (const Group& gr : groups) {
      const size_t w = hbk_alltoallv_wire_workspace_bytes(1, cs1, gr.lay.rows_send_peer.data(),
                                                          gr.lay.rows_recv_peer.data(), W);
      wws = w > wws ? w : wws;
    }
    if ((rc = p->wire_ws.ensure(wws + 16)) != HBK_OK) return rc;
  }
  HBK_REQUIRE((int64_t)(rows_send_bytes / 4) + tot_req_floats < (1ll << 32),
              "sharded_lookup_fwd: more than 2^32 floats (16 GB) of rows per step on one rank");
  if ((rc = p->runs_dev.ensure(sizeof(int64_t) * 7 * (size_t)N * W)) != HBK_OK) return rc;
  if (p2p) {
    if ((rc = p->slot_send.ensure((size_t)tot_req_ids * 4 + 16)) != HBK_OK) return rc;
    if ((rc = p->slot_recv.ensure((size_t)tot_own_ids * 4 + 16)) != HBK_OK) return rc;
    if ((rc = p->token_buf.ensure((size_t)W * 8 + 16)) != HBK_OK) return rc;
  }
  char* ids_send_base = p->send_ids_p;
  char* ids_recv_base = p->recv_ids_p;
  float* rows_send_base = p->send_rows_p;
  float* rows_recv_base = p->recv_rows_p;
  const bool zc = p->zero_copy_self;
  const bool half = p->fused_half;   // fp16 rows in the exchange buffers of the forward
  const int me = p->rank;
  // element offsets of the "other side" buffers seen from the owner-side bases of the backward
  const int64_t ids_send_from_recv = (p->send_ids_p - p->recv_ids_p) / (int64_t)id_bytes;
  const int64_t rows_recv_from_send = (int64_t)(rows_send_bytes / 4);
  // run tables of the in-place stitch (column c = W runs over the group's received rows)
  {
    int64_t* h_start = p->host_runs;
    int64_t* h_base = p->host_runs + (size_t)N * W;
    for (const Group& gr : groups) {
      const int ng = gr.c1 - gr.c0;
      for (int c = 0; c < ng; ++c) {
        for (int q = 0; q < W; ++q) {
          h_start[(size_t)(gr.c0 + c) * W + q] = gr.lay.col_shard_off[(size_t)c * W + q];
          h_base[(size_t)(gr.c0 + c) * W + q] = gr.lay.req_row_off[(size_t)q * ng + c];
        }
      }
    }
    // owner side of the backward: column c = W runs over the received ids (recv_ids) and the
    // gradient rows that come back into send_rows
    int64_t* h_ostart = p->host_runs + 2 * (size_t)N * W;
    int64_t* h_oids = p->host_runs + 3 * (size_t)N * W;
    int64_t* h_ograds = p->host_runs + 4 * (size_t)N * W;
    for (const Group& gr : groups) {
      const int ng = gr.c1 - gr.c0;
      for (int c = 0; c < ng; ++c) {
        int64_t o = 0;
        for (int q = 0; q < W; ++q) {
          const size_t at = (size_t)(gr.c0 + c) * W + q;
          h_ostart[at] = o;
          h_oids[at] = gr.id_recv + gr.lay.own_id_off[(size_t)q * ng + c];
          h_ograds[at] = gr.row_send + gr.lay.own_row_off[(size_t)q * ng + c];
          if (zc && q == me) {
            // the own slice stays where the requester side put it: ids in the outgoing id buffer,
            // gradient rows (fp32 wire) in the buffer the rows came back in
            h_oids[at] = ids_send_from_recv + gr.id_send + gr.lay.req_id_off[(size_t)q * ng + c];
            if (p->zero_copy_grads) {
              h_ograds[at] = rows_recv_from_send + gr.row_recv + gr.lay.req_row_off[(size_t)q * ng + c];
            }
          }
          o += gr.R[(size_t)q * ng + c];
        }
      }
    }
    // p2p: where owner q's run of column c starts among the column's shard-ordered ids, and in the
    // peer-major slot buffer (one group: the layout of the outgoing ids)
    int64_t* h_soff = p->host_runs + 5 * (size_t)N * W;
    int64_t* h_doff = p->host_runs + 6 * (size_t)N * W;
    if (p2p) {
      const Group& gr = groups[0];
      for (int c = 0; c < N; ++c) {
        for (int q = 0; q < W; ++q) {
          h_soff[(size_t)c * W + q] = gr.lay.col_shard_off[(size_t)c * W + q];
          h_doff[(size_t)c * W + q] = gr.id_send + gr.lay.req_id_off[(size_t)q * N + c];
        }
      }
    }
    HBK_HIP_OK(hipMemcpyAsync(p->runs_dev.ptr, p->host_runs,
                              sizeof(int64_t) * (p2p ? 7 : 5) * (size_t)N * W,
                              hipMemcpyHostToDevice, stream));
  }
  // stage A: the ids of every group peer-major -- run_partition has done it (pack_ids_kernel, while
  // the host waited for the sizes); calls beyond that kernel's limits pack here, one launch for
  // all groups
  if (!set.packed_early) {
    std::vector<Seg> segs;
    segs.reserve((size_t)N * W);
    for (int g = 0; g < G; ++g) {
      const Group& gr = groups[g];
      const int ng = gr.c1 - gr.c0;
      for (int q = 0; q < W; ++q) {
        for (int c = 0; c < ng; ++c) {
          segs.push_back(make_seg(
              pout[gr.c0 + c] + gr.lay.col_shard_off[(size_t)c * W + q],
              ids_send_base + (gr.id_send + gr.lay.req_id_off[(size_t)q * ng + c]) * id_bytes,
              (int64_t)S[(size_t)(gr.c0 + c) * W + q] * 8, p->id32 ? 1 : 0));
        }
      }
    }
    if ((rc = seg_copy(segs, stream)) != HBK_OK) return rc;
  }
  // (the pack launch of run_partition has written the slots next to the ids: nothing to do here)
  const bool slots_packed = p2p && set.packed_early && set.has_slots;
  int32_t* const slot_send = reinterpret_cast<int32_t*>(slots_packed ? set.slots.ptr : p->slot_send.ptr);
  int32_t* const slot_recv = reinterpret_cast<int32_t*>(p->slot_recv.ptr);
  if (p2p && !slots_packed) {   // the output slot of every id, peer-major like the ids
    const int64_t* d_soff = reinterpret_cast<const int64_t*>(p->runs_dev.ptr) + 5 * (size_t)N * W;
    const int64_t* d_doff = d_soff + (size_t)N * W;
    int c0 = 0;
    while (c0 < N) {
      SlotArgs a;
      a.W = W;
      a.col0 = c0;
      a.pad_ = 0;
      a.shard_off = d_soff;
      a.dst_off = d_doff;
      a.slots = slot_send;
      int k = 0;
      int64_t tiles = 0;
      a.tile0[0] = 0;
      while (c0 + k < N && k < kSlotCols) {
        a.idx[k] = idx[c0 + k];
        a.n[k] = (int32_t)n_ids[c0 + k];
        tiles += (n_ids[c0 + k] + kSlotTile - 1) / kSlotTile;
        ++k;
        a.tile0[k] = (int32_t)tiles;
      }
      a.n_cols = k;
      if (tiles > 0) {
        hipLaunchKernelGGL(p2p_slots_kernel, dim3((unsigned)tiles), dim3(kBlock), 0, stream, a);
        HBK_HIP_OK(hipGetLastError());
      }
      c0 += k;
    }
  }
  // one rank with its own slice left in place: nothing goes on the wire, the step stays on the
  // compute stream (no hops to the communicator's stream and back)
  const bool wire = !(W == 1 && zc);
  const bool hop = wire && !p->inline_x;   // exchanges on the communicator's stream, ordered by events
  if (hop) HBK_HIP_OK(hipEventRecord(p->ev[0][0], stream));
  // stage B: ids exchanges, back to back on the communicator's stream
  for (int g = 0; g < G && wire; ++g) {
    const Group& gr = groups[g];
    rc = exchange(p, id_dtype, id_dtype, ids_send_base + gr.id_send * id_bytes,
                  gr.lay.ids_send_peer.data(), ids_recv_base + gr.id_recv * id_bytes,
                  gr.lay.ids_recv_peer.data(), stream_, p->ev[0][0], p->ev[1][g], nullptr, 0, zc);
    if (rc != HBK_OK) return rc;
    if (p2p) {   // the slots travel like the ids (same sizes; G = 1)
      rc = exchange(p, HBK_INT32, HBK_INT32, slot_send + gr.id_send, gr.lay.ids_send_peer.data(),
                    slot_recv + gr.id_recv, gr.lay.ids_recv_peer.data(), stream_, p->ev[0][0],
                    p->ev[3][0], nullptr, 0, zc);
      if (rc != HBK_OK) return rc;
    }
  }
  // stage C: owner gather of group g as soon as its ids are in (N_g * W virtual columns, straight
  // into the peer-major reply), then its rows go on the wire
  for (int g = 0; g < G; ++g) {
    const Group& gr = groups[g];
    const int ng = gr.c1 - gr.c0;
    if (hop) HBK_HIP_OK(hipStreamWaitEvent(stream, p->ev[1][g], 0));
    if (hop && p2p) HBK_HIP_OK(hipStreamWaitEvent(stream, p->ev[3][0], 0));
    std::vector<hbk_lookup_column_t> v;
    v.reserve((size_t)ng * W);
    for (int q = 0; q < W; ++q) {
      for (int c = 0; c < ng; ++c) {
        const int64_t n = gr.R[(size_t)q * ng + c];
        if (n == 0) continue;
        const hbk_sharded_column_t& col = p->cols[gr.c0 + c];
        hbk_lookup_column_t h;
        memset(&h, 0, sizeof(h));
        h.table = col.shard;
        h.rows = col.rows_local;
        h.dim = col.dim;
        h.ids_dtype = id_dtype;
        h.ids = ids_recv_base + (gr.id_recv + gr.lay.own_id_off[(size_t)q * ng + c]) * id_bytes;
        h.n_ids = n;
        h.n_segments = n;
        h.divisor = W;
        h.combiner = HBK_COMBINER_SUM;
        h.hot_rows = col.hot_rows;
        int64_t out_at = gr.row_send + gr.lay.own_row_off[(size_t)q * ng + c];   // elements
        float* out_base = rows_send_base;
        if (zc && q == me) {
          h.ids = ids_send_base + (gr.id_send + gr.lay.req_id_off[(size_t)q * ng + c]) * id_bytes;
          out_at = gr.row_recv + gr.lay.req_row_off[(size_t)q * ng + c];
          out_base = rows_recv_base;
        }
        if (p2p) {
          // straight into requester q's output, row = the slot that travelled with the id
          h.out = p->peer_out[(size_t)q * N + gr.c0 + c];
          h.out_stride = p->peer_stride[(size_t)q * N + gr.c0 + c];
          h.out_slots = slot_recv + gr.id_recv + gr.lay.own_id_off[(size_t)q * ng + c];
          if (zc && q == me) {
            h.out_slots = slot_send + gr.id_send + gr.lay.req_id_off[(size_t)q * ng + c];
          }
          h.hot_rows = 0;
        } else if (half) {   // the same element offsets inside the same buffers, two bytes each
          h.out = reinterpret_cast<float*>(reinterpret_cast<uint16_t*>(out_base) + out_at);
          h.half_io = HBK_LOOKUP_OUT_HALF;
        } else {
          h.out = out_base + out_at;
        }
        v.push_back(h);
      }
    }
    rc = hbk_group_lookup_fwd((int32_t)v.size(), v.data(), stream_);
    if (rc != HBK_OK) return rc;
    if (!wire) continue;
    if (hop) HBK_HIP_OK(hipEventRecord(p->ev[2][g], stream));
    if (p2p) {
      // nothing to send back: a one-int token per peer tells every requester that this owner's
      // stores into its output are done (the exchange is ordered behind the gather kernel, whose
      // end releases its stores at system scope; the requester's next kernel acquires)
      int32_t* tok = reinterpret_cast<int32_t*>(p->token_buf.ptr);
      std::vector<int32_t> ones((size_t)W, 1);
      rc = exchange(p, HBK_INT32, HBK_INT32, tok, ones.data(), tok + W, ones.data(), stream_,
                    p->ev[2][g], p->ev[2][1], nullptr, 0, zc);
      if (rc != HBK_OK) return rc;
      if (hop) HBK_HIP_OK(hipStreamWaitEvent(stream, p->ev[2][1], 0));
    }
  }
  // the second half (rows exchange, stitch + combiner) is hbk_sharded_lookup_fwd_end
  p->fin_wire = wire;
  p->fin_hop = hop;
  p->fin_p2p = p2p;
  p->fin_use = use;
  p->fin_t0 = t_begin;
  p->fin_us[0] = (float)t_enq1;
  p->fin_us[1] = (float)(t_sync - t_enq1);
  p->fwd_open = true;
  if (trace) {
    fprintf(stderr, "hbk_sharded_lookup_fwd_begin host us: enqueue partition+sizes %.1f, sync wait "
                    "%.1f, enqueue through the gather %.1f (G = %d, W = %d%s%s)\n",
            t_enq1, t_sync - t_enq1, us_since(t_begin) - t_sync, G, W,
            prefetched ? ", partition prefetched" : "", p2p ? ", p2p" : "");
  }
  return HBK_OK;
}

extern "C" int hbk_sharded_lookup_fwd_end(hbk_sharded_t p, float* const* outs,
                                          const int32_t* out_strides, hbk_stream_t stream_) {
  using namespace hbk;
  HBK_REQUIRE(p != nullptr && outs != nullptr, "sharded_lookup_fwd_end: NULL argument");
  HBK_REQUIRE(p->fwd_open, "sharded_lookup_fwd_end: no step is open (hbk_sharded_lookup_fwd_begin)");
  p->fwd_open = false;
  hipStream_t stream = as_stream(stream_);
  const int N = p->N, W = p->W;
  const bool wire = p->fin_wire, hop = p->fin_hop, p2p = p->fin_p2p;
  const bool half = p->fused_half, zc = p->zero_copy_self;
  const std::vector<Group>& groups = p->groups;
  const int G = (int)groups.size();
  hbk_sharded::PartSet& set = p->ps[p->fin_use];
  const std::vector<int64_t>& n_ids = p->n_ids;
  float* const rows_send_base = p->send_rows_p;
  float* const rows_recv_base = p->recv_rows_p;
  int rc;
  if (p2p) {
    for (int c = 0; c < N; ++c) {
      HBK_REQUIRE(outs[c] == p->p2p_outs[c] &&
                      (out_strides == nullptr || out_strides[c] == 0 ||
                       out_strides[c] == p->p2p_strides[c] ||
                       (out_strides[c] == p->cols[c].dim && p->p2p_strides[c] == p->cols[c].dim)),
                  "sharded_lookup_fwd: column %d: not the output registered with "
                  "hbk_sharded_p2p_bind", c);
    }
  }
  // the rows of group g go on the wire (its gather has recorded ev[2][g])
  for (int g = 0; g < G && wire && !p2p; ++g) {
    const Group& gr = groups[g];
    if (half) {
      rc = exchange(p, HBK_HALF, HBK_HALF, reinterpret_cast<uint16_t*>(rows_send_base) + gr.row_send,
                    gr.lay.rows_send_peer.data(),
                    reinterpret_cast<uint16_t*>(rows_recv_base) + gr.row_recv,
                    gr.lay.rows_recv_peer.data(), stream_, p->ev[2][g], p->ev[3][g], nullptr, 0, zc);
    } else {
      rc = exchange(p, HBK_FLOAT, p->wire_dtype, rows_send_base + gr.row_send,
                    gr.lay.rows_send_peer.data(), rows_recv_base + gr.row_recv,
                    gr.lay.rows_recv_peer.data(), stream_, p->ev[2][g], p->ev[3][g],
                    p->wire_ws.ptr, p->wire_ws.bytes, zc);
    }
    if (rc != HBK_OK) return rc;
  }
  // stage D: stitch + combiner of group g when its rows are in; the received rows stay
  // peer-major, column c is a W-run segmented table over them
  std::vector<int32_t*> idx((size_t)N);
  {
    int64_t off = 0;
    for (int c = 0; c < N; ++c) {
      idx[c] = reinterpret_cast<int32_t*>(set.shard_index.ptr) + off;
      off += n_ids[c];
    }
  }
  const int64_t* d_start = reinterpret_cast<const int64_t*>(p->runs_dev.ptr);
  const int64_t* d_base = d_start + (size_t)N * W;
  for (int g = 0; g < G && !p2p; ++g) {
    const Group& gr = groups[g];
    const int ng = gr.c1 - gr.c0;
    if (hop) HBK_HIP_OK(hipStreamWaitEvent(stream, p->ev[3][g], 0));
    std::vector<hbk_lookup_column_t> v(ng);
    for (int c = 0; c < ng; ++c) {
      const int cc = gr.c0 + c;
      hbk_lookup_column_t& h = v[c];
      memset(&h, 0, sizeof(h));
      h.table = half ? reinterpret_cast<const float*>(reinterpret_cast<const uint16_t*>(rows_recv_base) +
                                                      gr.row_recv)
                     : rows_recv_base + gr.row_recv;
      h.half_io = half ? HBK_LOOKUP_TABLE_HALF : 0;
      h.rows = p->n_sent[cc];   // rows that came back (the distinct ids of a deduplicated column)
      h.dim = p->cols[cc].dim;
      h.ids_dtype = HBK_INT32;
      h.ids = idx[cc];          // shard_index, composed with the inverse index where deduplicated
      h.n_ids = n_ids[cc];
      h.row_splits = p->row_splits[cc];
      h.n_segments = p->n_seg[cc];
      h.divisor = 1;
      h.combiner = p->cols[cc].combiner;
      h.out = outs[cc];
      h.out_stride = out_strides ? out_strides[cc] : 0;
      h.run_start = d_start + (size_t)cc * W;
      h.run_base = d_base + (size_t)cc * W;
      h.n_runs = W;
    }
    rc = hbk_group_lookup_fwd(ng, v.data(), stream_);
    if (rc != HBK_OK) return rc;
  }
  p->have_step = true;
  const double total = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() -
                                                                p->fin_t0).count();
  p->host_us[0] = p->fin_us[0];
  p->host_us[1] = p->fin_us[1];
  p->host_us[2] = (float)(total - p->fin_us[0] - p->fin_us[1]);
  return HBK_OK;
}

extern "C" int hbk_sharded_lookup_fwd(hbk_sharded_t p, const int64_t* const* ids,
                                      const int64_t* n_ids,
                                      const int32_t* const* row_splits,
                                      const int64_t* n_segments, float* const* outs,
                                      const int32_t* out_strides, hbk_stream_t stream_) {
  using namespace hbk;
  HBK_REQUIRE(outs != nullptr, "sharded_lookup_fwd: NULL argument array");
  const int rc = hbk_sharded_lookup_fwd_begin(p, ids, n_ids, row_splits, n_segments, stream_);
  if (rc != HBK_OK) return rc;
  return hbk_sharded_lookup_fwd_end(p, outs, out_strides, stream_);
}

// Partition (stages 1-2) of a FUTURE step on the plan's own stream, so that it overlaps whatever
// the caller's stream still has in flight (the exchanges of the step just launched).  The next
// hbk_sharded_lookup_fwd with the same id pointers and counts picks the result up; any other
// forward drops it.  Every rank must prefetch the same steps (the size exchange is a collective).
// The same on a stream of the CALLER's: no stream of the plan is involved, stream order is all the
// protection the overwritten partition state needs -- the caller enqueues this behind the _end (or
// the whole forward) of the step BEFORE the last one begun on this plan, which read that state.
// hb.embedding.PipelinedLookup calls it right behind a step's _begin on the plan's compute stream:
// the partition of the plan's next step then runs before the stream reaches the wait for the
// current step's rows, and needs no stream of its own.  (HIP maps streams onto a few hardware
// queues -- 4 by default; two streams on one queue run in order, waits included: a prefetch stream
// that shares its queue with a compute stream sits behind that stream's wait for the wire.)
extern "C" int hbk_sharded_prefetch_on(hbk_sharded_t p, const int64_t* const* ids,
                                       const int64_t* n_ids, hbk_stream_t stream_) {
  using namespace hbk;
  HBK_REQUIRE(p != nullptr && ids && n_ids, "sharded_prefetch_on: NULL argument");
  hbk_sharded::PartSet& set = p->ps[p->cur ^ 1];
  p->ps[0].pending = p->ps[1].pending = false;
  p->prefetch_used = true;
  int rc = run_partition(p, set, ids, n_ids, as_stream(stream_));
  if (rc != HBK_OK) return rc;
  set.pending = true;
  return HBK_OK;
}

extern "C" int hbk_sharded_prefetch(hbk_sharded_t p, const int64_t* const* ids,
                                    const int64_t* n_ids, void* ids_ready_event) {
  using namespace hbk;
  HBK_REQUIRE(p != nullptr && ids && n_ids, "sharded_prefetch: NULL argument");
  if (p->pre_stream == nullptr) {
    HBK_HIP_OK(hipStreamCreateWithFlags(&p->pre_stream, hipStreamNonBlocking));
  }
  hbk_sharded::PartSet& set = p->ps[p->cur ^ 1];
  p->ps[0].pending = p->ps[1].pending = false;
  p->prefetch_used = true;
  // pre_stream may start once (a) the ids exist, (b) everything of the step BEFORE the last
  // forward has drained (it read the set that is overwritten now) and (c) the last forward's own
  // partition is through with the shared partition workspace -- not after the last forward's
  // exchanges, gather and stitch: those are what this overlaps.
  if (ids_ready_event != nullptr) {
    HBK_HIP_OK(hipStreamWaitEvent(p->pre_stream, reinterpret_cast<hipEvent_t>(ids_ready_event), 0));
  }
  if (p->have_step) {
    HBK_HIP_OK(hipStreamWaitEvent(p->pre_stream, p->step_begin, 0));
    HBK_HIP_OK(hipStreamWaitEvent(p->pre_stream, p->ps[p->cur].done, 0));
  }
  int rc = run_partition(p, set, ids, n_ids, p->pre_stream);
  if (rc != HBK_OK) return rc;
  set.pending = true;
  return HBK_OK;
}

extern "C" int hbk_sharded_lookup_bwd(hbk_sharded_t p, const float* const* grads,
                                      const int32_t* grad_strides, float apply_lr,
                                      int64_t* const* unique_rows, float* const* grad_rows,
                                      int32_t* const* n_unique, hbk_stream_t stream_) {
  return hbk_sharded_lookup_bwd_apply(p, grads, grad_strides, HBK_APPLY_SGD, apply_lr,
                                      unique_rows, grad_rows, n_unique, stream_);
}

extern "C" int hbk_sharded_lookup_bwd_apply(hbk_sharded_t p, const float* const* grads,
                                            const int32_t* grad_strides, int32_t apply,
                                            float apply_lr, int64_t* const* unique_rows,
                                      float* const* grad_rows, int32_t* const* n_unique,
                                      hbk_stream_t stream_) {
  using namespace hbk;
  HBK_REQUIRE(p != nullptr, "sharded_lookup_bwd: plan is NULL");
  HBK_REQUIRE(p->have_step, "sharded_lookup_bwd: no forward step to differentiate");
  HBK_REQUIRE(grads && n_unique, "sharded_lookup_bwd: NULL argument array");
  HBK_REQUIRE((unique_rows != nullptr) == (grad_rows != nullptr),
              "sharded_lookup_bwd: unique_rows and grad_rows go together");
  HBK_REQUIRE(unique_rows != nullptr || apply_lr != 0.0f,
              "sharded_lookup_bwd: no output buffers and no optimizer step: nothing to do");
  hipStream_t stream = as_stream(stream_);
  const int N = p->N, W = p->W;
  const int32_t* R = p->recv_sizes.data();
  const int G = (int)p->groups.size();
  const bool wire = !(W == 1 && p->zero_copy_grads);   // (see the forward)
  const bool hop = wire && !p->inline_x;
  float* rows_send_base = p->send_rows_p;
  float* rows_recv_base = p->recv_rows_p;
  const int64_t* d_start = reinterpret_cast<const int64_t*>(p->runs_dev.ptr);
  const int64_t* d_base = d_start + (size_t)N * W;
  const int64_t* d_ostart = d_start + 2 * (size_t)N * W;
  const int64_t* d_oids = d_start + 3 * (size_t)N * W;
  const int64_t* d_ograds = d_start + 4 * (size_t)N * W;
  int rc;
  // The forward's buffers, layout and run tables are reused in reverse, group by group:
  //   compute : dstitch(0) dstitch(1) ..             reduce(0)   reduce(1) ..
  //   comm    :            grads(0)   grads(1) ..
  // ---- B1 d(stitch + combiner) written straight into the peer-major buffer the rows came in ----
  std::vector<int64_t> ioff(N + 1, 0);
  for (int c = 0; c < N; ++c) ioff[c + 1] = ioff[c] + p->n_ids[c];
  // deduplicated columns: scratch for their requester-side IndexedSlices (rows = positions in the
  // column's distinct-id list): int64 rows, the same as int32, gradient rows, counts
  std::vector<int64_t> t_rows(N, -1), t_vals(N, -1);
  size_t tmp_bytes = 0, dd_ws = 0;
  if (p->any_dedup) {
    for (int c = 0; c < N; ++c) {
      if (p->cols[c].dedup == 0 || p->n_sent[c] <= 0) continue;
      t_rows[c] = (int64_t)tmp_bytes;
      tmp_bytes += (size_t)p->n_sent[c] * 12 + 16;
      tmp_bytes = (tmp_bytes + 15) & ~(size_t)15;
      t_vals[c] = (int64_t)tmp_bytes;
      tmp_bytes += ((size_t)p->n_sent[c] * p->cols[c].dim * 4 + 15) & ~(size_t)15;
    }
    tmp_bytes += (size_t)N * 4 + 16;
    if ((rc = p->dedup_tmp.ensure(tmp_bytes)) != HBK_OK) return rc;
  }
  char* const tmp = reinterpret_cast<char*>(p->dedup_tmp.ptr);
  int32_t* const tmp_nu = p->any_dedup ? reinterpret_cast<int32_t*>(tmp + tmp_bytes - (size_t)N * 4 - 8)
                                       : nullptr;
  for (int g = 0; g < G; ++g) {
    const Group& gr = p->groups[g];
    const int ng = gr.c1 - gr.c0;
    // (a) deduplicated columns: duplicate positions are summed HERE, on the requester, so that one
    // gradient row per distinct id goes back -- hbk_group_lookup_bwd over the composed index (the
    // received-rows buffer was the forward's table), then every row to its place in the
    // peer-major buffer (the order of the distinct-id list)
    if (p->any_dedup) {
      std::vector<hbk_lookup_grad_column_t> dv;
      std::vector<hbk_stitch_grad_column_t> sv;
      std::vector<hbk::Seg> narrow;
      for (int c = 0; c < ng; ++c) {
        const int cc = gr.c0 + c;
        if (t_rows[cc] < 0) continue;
        const int64_t u = p->n_sent[cc];
        hbk_lookup_grad_column_t h;
        memset(&h, 0, sizeof(h));
        h.rows = u;
        h.dim = p->cols[cc].dim;
        h.ids_dtype = HBK_INT32;
        h.ids = reinterpret_cast<const int32_t*>(p->ps[p->cur].shard_index.ptr) + ioff[cc];
        h.n_ids = p->n_ids[cc];
        h.row_splits = p->row_splits[cc];
        h.n_segments = p->n_seg[cc];
        h.divisor = 1;
        h.combiner = p->cols[cc].combiner;
        h.grad_out = grads[cc];
        h.grad_stride = grad_strides ? grad_strides[cc] : 0;
        h.unique_rows = reinterpret_cast<int64_t*>(tmp + t_rows[cc]);
        h.grad_rows = reinterpret_cast<float*>(tmp + t_vals[cc]);
        h.n_unique = tmp_nu + cc;
        dv.push_back(h);
        int32_t* idx32 = reinterpret_cast<int32_t*>(tmp + t_rows[cc] + (size_t)u * 8);
        narrow.push_back(make_seg(h.unique_rows, idx32, u * 8, 1));
        hbk_stitch_grad_column_t t;
        memset(&t, 0, sizeof(t));
        t.dim = h.dim;
        t.combiner = HBK_COMBINER_SUM;
        t.n_ids = u;
        t.index = idx32;
        t.n_segments = u;
        t.grad_out = h.grad_rows;
        t.grad_rows = rows_recv_base + gr.row_recv;
        t.run_start = d_start + (size_t)cc * W;
        t.run_base = d_base + (size_t)cc * W;
        t.n_runs = W;
        sv.push_back(t);
      }
      if (!dv.empty()) {
        dd_ws = hbk_group_lookup_bwd_workspace_bytes((int32_t)dv.size(), dv.data());
        if ((rc = p->bwd_ws.ensure(dd_ws + 8)) != HBK_OK) return rc;
        rc = hbk_group_lookup_bwd((int32_t)dv.size(), dv.data(), 0.0f, p->bwd_ws.ptr,
                                  p->bwd_ws.bytes, stream_);
        if (rc != HBK_OK) return rc;
        if ((rc = seg_copy(narrow, stream)) != HBK_OK) return rc;
        rc = hbk_group_stitch_bwd((int32_t)sv.size(), sv.data(), stream_);
        if (rc != HBK_OK) return rc;
      }
    }
    std::vector<hbk_stitch_grad_column_t> v(ng);
    for (int c = 0; c < ng; ++c) {
      const int cc = gr.c0 + c;
      hbk_stitch_grad_column_t& h = v[c];
      memset(&h, 0, sizeof(h));
      h.dim = p->cols[cc].dim;
      h.combiner = p->cols[cc].combiner;
      h.n_ids = t_rows[cc] >= 0 ? 0 : p->n_ids[cc];   // (0: the column was handled above)
      h.index = reinterpret_cast<const int32_t*>(p->ps[p->cur].shard_index.ptr) + ioff[cc];
      h.row_splits = p->row_splits[cc];
      h.n_segments = t_rows[cc] >= 0 ? 0 : p->n_seg[cc];
      if (t_rows[cc] >= 0) h.row_splits = nullptr;
      h.grad_out = grads[cc];
      h.grad_stride = grad_strides ? grad_strides[cc] : 0;
      h.grad_rows = rows_recv_base + gr.row_recv;
      h.run_start = d_start + (size_t)cc * W;
      h.run_base = d_base + (size_t)cc * W;
      h.n_runs = W;
    }
    rc = hbk_group_stitch_bwd(ng, v.data(), stream_);
    if (rc != HBK_OK) return rc;
    if (hop) HBK_HIP_OK(hipEventRecord(p->ev[0][g], stream));
  }
  // ---- B2 reverse exchanges (the forward's sizes, swapped: collective.py:334-347) -------------
  for (int g = 0; g < G && wire; ++g) {
    const Group& gr = p->groups[g];
    rc = exchange(p, HBK_FLOAT, p->wire_dtype, rows_recv_base + gr.row_recv,
                  gr.lay.rows_recv_peer.data(), rows_send_base + gr.row_send,
                  gr.lay.rows_send_peer.data(), stream_, p->ev[0][g], p->ev[1][g], nullptr, 0,
                  p->zero_copy_grads);
    if (rc != HBK_OK) return rc;
  }
  // ---- B3 owner side: duplicate-row reduction (+ SGD) reading ids and gradient rows in place ---
  std::vector<hbk_lookup_grad_column_t> v(N);
  for (int c = 0; c < N; ++c) {
    int64_t n_own = 0;
    for (int q = 0; q < W; ++q) n_own += R[(size_t)q * N + c];
    hbk_lookup_grad_column_t& h = v[c];
    memset(&h, 0, sizeof(h));
    h.table = const_cast<float*>(p->cols[c].shard);
    h.accum = p->cols[c].accum;
    h.rows = p->cols[c].rows_local;
    h.dim = p->cols[c].dim;
    h.ids_dtype = p->id32 ? HBK_INT32 : HBK_INT64;
    h.ids = p->recv_ids_p;
    h.n_ids = n_own;
    h.n_segments = n_own;
    h.divisor = W;
    h.combiner = HBK_COMBINER_SUM;
    h.grad_out = rows_send_base;
    h.unique_rows = unique_rows != nullptr ? unique_rows[c] : nullptr;   // NULL: step only
    h.grad_rows = grad_rows != nullptr ? grad_rows[c] : nullptr;
    h.n_unique = n_unique[c];
    h.run_start = d_ostart + (size_t)c * W;
    h.run_ids = d_oids + (size_t)c * W;
    h.run_grads = d_ograds + (size_t)c * W;
    h.n_runs = W;
  }
  size_t ws = 0;
  for (int g = 0; g < G; ++g) {
    const Group& gr = p->groups[g];
    const size_t w = hbk_group_lookup_bwd_workspace_bytes(gr.c1 - gr.c0, v.data() + gr.c0);
    ws = w > ws ? w : ws;
  }
  if ((rc = p->bwd_ws.ensure(ws + 8)) != HBK_OK) return rc;
  for (int g = 0; g < G; ++g) {
    const Group& gr = p->groups[g];
    if (hop) HBK_HIP_OK(hipStreamWaitEvent(stream, p->ev[1][g], 0));
    rc = hbk_group_lookup_bwd_apply(gr.c1 - gr.c0, v.data() + gr.c0, apply, apply_lr,
                                    p->bwd_ws.ptr, p->bwd_ws.bytes, stream_);
    if (rc != HBK_OK) return rc;
  }
  return HBK_OK;
}

// the per-column hot_rows hints of a live plan (hbk_sharded_column_t.hot_rows): the host side turns
// them on / off from what the last backward saw (distinct rows vs ids); read by the next forward
extern "C" int hbk_sharded_set_hot_rows(hbk_sharded_t p, const int32_t* hot_rows) {
  using namespace hbk;
  HBK_REQUIRE(p != nullptr && hot_rows != nullptr, "sharded_set_hot_rows: NULL argument");
  for (int c = 0; c < p->N; ++c) p->cols[c].hot_rows = hot_rows[c] != 0 ? 1 : 0;
  return HBK_OK;
}

// host-side phases of the last forward (us): enqueueing partition + size exchange, the wait for
// the sizes (the device finishing the partition, not host work), enqueueing everything else
extern "C" int hbk_sharded_last_host_us(hbk_sharded_t p, float* out3) {
  using namespace hbk;
  HBK_REQUIRE(p != nullptr && out3 != nullptr, "sharded_last_host_us: NULL argument");
  HBK_REQUIRE(p->have_step, "sharded_last_host_us: no forward step yet");
  for (int i = 0; i < 3; ++i) out3[i] = p->host_us[i];
  return HBK_OK;
}

// capacity the caller needs for the backward outputs of column c (rows this rank owns that
// were requested in the last forward step)
extern "C" int64_t hbk_sharded_owned_ids(hbk_sharded_t p, int32_t column) {
  if (p == nullptr || !p->have_step || column < 0 || column >= p->N) return -1;
  int64_t n = 0;
  for (int q = 0; q < p->W; ++q) n += p->recv_sizes[(size_t)q * p->N + column];
  return n;
}
