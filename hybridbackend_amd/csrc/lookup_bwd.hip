// HbGroupLookupGrad: backward of the fused group lookup for gfx950 (R10) -- replaces the TF
// autodiff chain SparseSegment*Grad -> GatherV2 grad -> UnsortedSegmentSum (duplicate-id
// reduction) -> IndexedSlices(values, indices = row) of SURVEY 3.4 / hbtf/embedding/
// sharding.py:186-200 in reverse, optionally fused with the sparse SGD apply on the shard
// (sharded variables are not aggregated across ranks, hbtf/training/gradient.py:193-217).
//
// Device-scope atomics resolve at the memory side on MI355X (per-XCD L2s are not coherent):
// ~15 G/s for distinct addresses and far less for one hot address, so a scatter-add with
// global atomics costs 10-100x the forward (round-1 measurement: 0.6 ms uniform, 9 ms
// Zipf).  The design below has NO global atomics on the data path:
//
//   0 seg_of     (ragged columns) segment of every id
//   1 hist       row(j) = bucketize/`// W`; bucket = top bits of a 64-bit mix of the row;
//                per-1024-id-tile LDS histogram, all columns in one launch
//   2 scan       per column: exclusive scan of hist[bucket][tile] -> bucket starts
//   3 scatter    (row, segment) pairs grouped by bucket (LDS ticket per bucket and tile)
//   4 reduce     ONE workgroup owns a bucket, hence every table row that hashes to it: an LDS
//                hash table of the bucket's distinct rows with the fp32 accumulators next to
//                the keys (LDS-staged hot rows: a hot row is one LDS line hammered by
//                ds_add_f32, not one DRAM line); then the occupied slots are ranked with wave
//                ballots + popcounts, an output range is claimed with one atomic per wave,
//                and unique_rows / grad_rows are written with plain 16-byte stores -- plus,
//                for apply_lr != 0, the SGD update of the shard in the same pass (exclusive
//                ownership makes the read-modify-write race free).
//                A table that fills up (more distinct rows than slots: adversarial skew) is
//                flushed and the rejected pairs are re-run; then a row can appear in more
//                than one IndexedSlices entry (sum semantics preserved).
// Summation order inside a row is not fixed (LDS atomics): 1e-5 relative tolerance.
#include <alloca.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "lookup_common.h"

namespace hbk {
namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kMaxCols = 64;
constexpr int kTile = 8192;          // ids per 256-thread block in hist / scatter: few tiles keep
                                     // the [bucket][tile] histogram (and its scan) small
constexpr int kPerThread = kTile / kBlock;   // 32 ids per thread, 8 loads in flight
constexpr int kBatch = 8;
constexpr int kLdsBudget = 40 * 1024;  // bytes of LDS per reduce workgroup
constexpr int kMaxBuckets = 8192;
constexpr int kU = 8;                 // pairs per lane group and step in the reduce kernel
constexpr unsigned long long kEmptyKey = ~0ull;

struct GCol {
  const void* ids;
  const float* grad_out;     // [n_seg, dim]
  const int32_t* splits;
  int64_t* unique_rows;
  float* grad_rows;
  int32_t* n_unique;
  float* table;
  int32_t* hist;             // [P * tiles] -> exclusive offsets after the scan
  int32_t* bstart;           // [P + 1]
  int64_t* pair_row[2];      // [n_ids] ping-pong (second copy: pairs rejected by a full table)
  int32_t* pair_seg[2];
  int32_t* seg_of;           // [n_ids], ragged columns only
  IdMap map;
  int64_t n_ids;
  int64_t n_seg;
  int32_t dim;
  int32_t chunks;
  uint8_t lpr_log2, ids64, combiner, vec4;
  int32_t log2p;             // buckets = 1 << log2p
  int32_t slots_log2;        // LDS table slots of the reduce workgroup
  int32_t tile0;             // first tile (hist / scatter grids)
  int32_t bucket0;           // first block (reduce grid)
  int32_t segtile0;          // first block (seg_of grid)
};

struct GArgs {
  int32_t n_cols;
  float lr;
  GCol col[kMaxCols];
};
static_assert(sizeof(GArgs) <= 16384, "kernarg budget");

#define HBK_FIND_COL(ARGS, FIELD)                                                  \
  int ci = 0, hi__ = (ARGS).n_cols;                                                \
  while (hi__ - ci > 1) {                                                          \
    const int mid__ = (ci + hi__) >> 1;                                            \
    if ((ARGS).col[mid__].FIELD <= (int)blockIdx.x) {                              \
      ci = mid__;                                                                  \
    } else {                                                                       \
      hi__ = mid__;                                                                \
    }                                                                              \
  }                                                                                \
  const auto& c = (ARGS).col[ci];

__device__ inline uint64_t mix64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return k;
}

__device__ inline int bucket_of(uint64_t row, int log2p) {
  return log2p == 0 ? 0 : (int)(mix64(row) >> (64 - log2p));
}

// ---- 0: segment of every id (ragged columns only; the host passes just those) --------------
__global__ __launch_bounds__(kBlock) void bwd_segof_kernel(const GArgs a) {
  HBK_FIND_COL(a, segtile0)
  const int64_t s = ((int64_t)blockIdx.x - c.segtile0) * kBlock + threadIdx.x;
  if (s >= c.n_seg) return;
  const int32_t beg = c.splits[s], end = c.splits[s + 1];
  for (int32_t j = beg; j < end; ++j) c.seg_of[j] = (int32_t)s;
}

// ---- 1: per-tile bucket histogram ---------------------------------------------------------
__global__ __launch_bounds__(kBlock) void bwd_hist_kernel(const GArgs a) {
  extern __shared__ int32_t counters[];
  HBK_FIND_COL(a, tile0)
  const int P = 1 << c.log2p;
  const int tid = (int)threadIdx.x;
  const int ctile = (int)blockIdx.x - c.tile0;
  const int n_tiles = (int)((c.n_ids + kTile - 1) / kTile);
  for (int p = tid; p < P; p += kBlock) counters[p] = 0;
  __syncthreads();
  const int64_t base = (int64_t)ctile * kTile;
  for (int k0 = 0; k0 < kPerThread; k0 += kBatch) {
    int64_t id[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      const int64_t j = base + (int64_t)(k0 + k) * kBlock + tid;
      id[k] = j < c.n_ids ? load_id(c.ids, c.ids64, j) : 0;
    }
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      const int64_t j = base + (int64_t)(k0 + k) * kBlock + tid;
      if (j < c.n_ids) {
        const uint64_t r = id_to_row(c.map, id[k]);
        if (r != kNoRow) atomicAdd(&counters[bucket_of(r, c.log2p)], 1);
      }
    }
  }
  __syncthreads();
  for (int p = tid; p < P; p += kBlock) c.hist[(int64_t)p * n_tiles + ctile] = counters[p];
}

// ---- 2: per-column exclusive scan over (bucket, tile) ---------------------------------------
// Wave w owns a contiguous quarter of the entries and walks it 64 entries at a time (coalesced
// loads, shuffle scan): pass 1 totals per wave, pass 2 exclusive offsets with the carried base.
__global__ __launch_bounds__(kBlock) void bwd_scan_kernel(const GArgs a) {
  __shared__ int32_t wave_tot[kWavesPerBlock];
  const GCol& c = a.col[blockIdx.x];
  const int P = 1 << c.log2p;
  const int n_tiles = (int)((c.n_ids + kTile - 1) / kTile);
  const int32_t total = P * n_tiles;
  const int tid = (int)threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  const int32_t per_wave = ((total + kWavesPerBlock * kWave - 1) / (kWavesPerBlock * kWave)) * kWave;
  const int32_t beg = wave * per_wave;
  const int32_t end = beg + per_wave < total ? beg + per_wave : total;
  int32_t sum = 0;
  for (int32_t e0 = beg; e0 < end; e0 += kWave) {
    const int32_t e = e0 + lane;
    sum += e < end ? c.hist[e] : 0;
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off, kWave);
  if (lane == 0) wave_tot[wave] = sum;
  __syncthreads();
  int32_t carry = 0;
  for (int w = 0; w < wave; ++w) carry += wave_tot[w];
  for (int32_t e0 = beg; e0 < end; e0 += kWave) {
    const int32_t e = e0 + lane;
    const int32_t x = e < end ? c.hist[e] : 0;
    int32_t s = x;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int32_t y = __shfl_up(s, off, kWave);
      if (lane >= off) s += y;
    }
    const int32_t excl = carry + s - x;
    if (e < end) {
      c.hist[e] = excl;
      if (e % n_tiles == 0) c.bstart[e / n_tiles] = excl;
    }
    carry += __shfl(s, kWave - 1, kWave);
  }
  if (tid == 0) {
    int32_t tot = 0;
    for (int w = 0; w < kWavesPerBlock; ++w) tot += wave_tot[w];
    c.bstart[P] = tot;
    *c.n_unique = 0;
  }
}

// ---- 3: (row, segment) pairs grouped by bucket ---------------------------------------------
__global__ __launch_bounds__(kBlock) void bwd_scatter_pairs_kernel(const GArgs a) {
  extern __shared__ int32_t run[];
  HBK_FIND_COL(a, tile0)
  const int P = 1 << c.log2p;
  const int tid = (int)threadIdx.x;
  const int ctile = (int)blockIdx.x - c.tile0;
  const int n_tiles = (int)((c.n_ids + kTile - 1) / kTile);
  for (int p = tid; p < P; p += kBlock) run[p] = c.hist[(int64_t)p * n_tiles + ctile];
  __syncthreads();
  const int64_t base = (int64_t)ctile * kTile;
  for (int k0 = 0; k0 < kPerThread; k0 += kBatch) {
    int64_t id[kBatch];
    int32_t seg[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      const int64_t j = base + (int64_t)(k0 + k) * kBlock + tid;
      id[k] = 0;
      seg[k] = (int32_t)j;
      if (j < c.n_ids) {
        id[k] = load_id(c.ids, c.ids64, j);
        if (c.seg_of != nullptr) seg[k] = c.seg_of[j];
      }
    }
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      const int64_t j = base + (int64_t)(k0 + k) * kBlock + tid;
      if (j < c.n_ids) {
        const uint64_t r = id_to_row(c.map, id[k]);
        if (r != kNoRow) {
          const int32_t pos = atomicAdd(&run[bucket_of(r, c.log2p)], 1);
          c.pair_row[0][pos] = (int64_t)r;
          c.pair_seg[0][pos] = seg[k];
        }
      }
    }
  }
}

// ---- 4: one workgroup per bucket: LDS hash table with accumulators ------------------------
// An accumulator row is stored TRANSPOSED inside LDS: element (sub * VE + k) of the row lives
// at dword k * chunks + sub, so the lanes of a group hit consecutive banks for every k
// (a plain row layout puts lanes 16 bytes apart: a 4-way bank conflict on every ds_add_f32).
template <typename V>
__device__ inline void lds_add_v(float* row, int chunks, int sub, V v);
template <>
__device__ inline void lds_add_v<f32x4>(float* row, int chunks, int sub, f32x4 v) {
  atomicAdd(row + sub, v.x);
  atomicAdd(row + chunks + sub, v.y);
  atomicAdd(row + 2 * chunks + sub, v.z);
  atomicAdd(row + 3 * chunks + sub, v.w);
}
template <>
__device__ inline void lds_add_v<float>(float* row, int chunks, int sub, float v) {
  atomicAdd(row + sub, v);
}
template <typename V>
__device__ inline V lds_read_v(const float* row, int chunks, int sub);
template <>
__device__ inline f32x4 lds_read_v<f32x4>(const float* row, int chunks, int sub) {
  return f32x4{row[sub], row[chunks + sub], row[2 * chunks + sub], row[3 * chunks + sub]};
}
template <>
__device__ inline float lds_read_v<float>(const float* row, int chunks, int sub) {
  return row[sub];
}

template <typename V>
__device__ inline void bucket_reduce(const GCol& c, int bucket, float lr, char* lds) {
  constexpr int VE = sizeof(V) / 4;
  const int slots = 1 << c.slots_log2;
  const int dim = c.dim;
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(lds);       // [slots]
  float* acc = reinterpret_cast<float*>(lds + (size_t)slots * 8);               // [slots][dim]
  int32_t* slot_out = reinterpret_cast<int32_t*>(lds + (size_t)slots * (8 + 4 * dim));
  int32_t* ctrl = slot_out + slots;                                             // [0] rejected
  const int tid = (int)threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int lpr_log2 = c.lpr_log2;
  const int sub = lane & ((1 << lpr_log2) - 1);
  const int grp_lane0 = (lane >> lpr_log2) << lpr_log2;
  const bool live = sub < c.chunks;
  const int groups = kBlock >> lpr_log2;  // pairs in flight per workgroup step
  const int my_group = tid >> lpr_log2;

  const int32_t start = c.bstart[bucket];
  int32_t n_pairs = c.bstart[bucket + 1] - start;
  int src = 0;
  while (n_pairs > 0) {  // uniform: more than one pass only after a table overflow
    for (int i = tid; i < slots; i += kBlock) keys[i] = kEmptyKey;
    for (int i = tid; i < slots * dim; i += kBlock) acc[i] = 0.f;
    if (tid == 0) ctrl[0] = 0;
    __syncthreads();
    const int64_t* prow = c.pair_row[src] + start;
    const int32_t* pseg = c.pair_seg[src] + start;
    int64_t* rrow = c.pair_row[src ^ 1] + start;
    int32_t* rseg = c.pair_seg[src ^ 1] + start;
    // U pairs per lane group and step, next step's (row, segment) already in flight: the
    // bucket of a hot row holds thousands of pairs and one workgroup must stream them
    unsigned long long row_n[kU];
    int32_t seg_n[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int32_t e = u * groups + my_group;
      row_n[u] = 0;
      seg_n[u] = 0;
      if (e < n_pairs) {
        row_n[u] = (unsigned long long)prow[e];
        seg_n[u] = pseg[e];
      }
    }
    for (int32_t e0 = 0; e0 < n_pairs; e0 += groups * kU) {
      unsigned long long row[kU];
      int32_t seg[kU];
      V g[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        row[u] = row_n[u];
        seg[u] = seg_n[u];
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int32_t e = e0 + groups * kU + u * groups + my_group;
        row_n[u] = 0;
        seg_n[u] = 0;
        if (e < n_pairs) {
          row_n[u] = (unsigned long long)prow[e];
          seg_n[u] = pseg[e];
        }
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const bool has = e0 + u * groups + my_group < n_pairs;
        g[u] = zero_v<V>();
        if (has && live) {
          g[u] = __builtin_nontemporal_load(reinterpret_cast<const V*>(
              c.grad_out + (int64_t)seg[u] * dim + (int64_t)sub * VE));
        }
      }
      if (c.combiner != HBK_COMBINER_SUM && c.splits != nullptr) {
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          if (e0 + u * groups + my_group < n_pairs) {
            const int32_t n = c.splits[seg[u] + 1] - c.splits[seg[u]];
            if (c.combiner == HBK_COMBINER_MEAN) {
              g[u] = g[u] / (float)n;
            } else {
              g[u] = g[u] / sqrtf((float)n);
            }
          }
        }
      }
      // runs of equal rows inside a group's U pairs (a hot row fills its bucket with them) are
      // summed in registers and cost one probe + one LDS add
      int slot = -1;
      unsigned long long run_row = kEmptyKey;
      V run_sum = zero_v<V>();
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const bool has = e0 + u * groups + my_group < n_pairs;
        if (has && row[u] == run_row) {
          run_sum = run_sum + g[u];
          continue;
        }
        // close the previous run (wave-uniform control flow is not needed: LDS atomics only)
        if (run_row != kEmptyKey && slot >= 0 && live) {
          lds_add_v<V>(acc + (size_t)slot * dim, c.chunks, sub, run_sum);
        }
        run_row = kEmptyKey;
        if (!has) continue;
        // the group's first lane probes; the slot is broadcast to the group.  All lanes of a
        // group take the same path here (has / row are group-uniform), groups may diverge.
        int found = -1;
        if (sub == 0) {
          int h = (int)(mix64(row[u]) & (unsigned)(slots - 1));
          for (int probe = 0; probe < slots; ++probe) {
            const unsigned long long prev = atomicCAS(&keys[h], kEmptyKey, row[u]);
            if (prev == kEmptyKey || prev == row[u]) {
              found = h;
              break;
            }
            h = (h + 1) & (slots - 1);
          }
        }
        slot = __builtin_amdgcn_ds_bpermute(grp_lane0 << 2, found);
        if (slot >= 0) {
          run_row = row[u];
          run_sum = g[u];
        } else if (sub == 0) {  // table full: run this pair again after the flush
          const int32_t f = atomicAdd(&ctrl[0], 1);
          rrow[f] = (int64_t)row[u];
          rseg[f] = seg[u];
        }
      }
      if (run_row != kEmptyKey && slot >= 0 && live) {
        lds_add_v<V>(acc + (size_t)slot * dim, c.chunks, sub, run_sum);
      }
    }
    __syncthreads();
    // flush: rank the occupied slots (ballot + popcount per wave, wave totals through LDS),
    // claim the output range with ONE global atomic per workgroup
    {
      int32_t* wave_cnt = ctrl + 2;  // [kWavesPerBlock * rounds] scratch, rounds <= 8
      const int rounds = (slots + kBlock - 1) / kBlock;
      for (int r = 0; r < rounds; ++r) {
        const int s = r * kBlock + tid;
        const bool occ = s < slots && keys[s] != kEmptyKey;
        const unsigned long long m = __ballot(occ);
        if (lane == 0) wave_cnt[r * kWavesPerBlock + (tid >> 6)] = (int32_t)__builtin_popcountll(m);
      }
      __syncthreads();
      if (tid == 0) {
        int32_t tot = 0;
        for (int i = 0; i < rounds * kWavesPerBlock; ++i) {
          const int32_t x = wave_cnt[i];
          wave_cnt[i] = tot;
          tot += x;
        }
        ctrl[1] = tot > 0 ? atomicAdd(c.n_unique, tot) : 0;
      }
      __syncthreads();
      for (int r = 0; r < rounds; ++r) {
        const int s = r * kBlock + tid;
        const bool occ = s < slots && keys[s] != kEmptyKey;
        const unsigned long long m = __ballot(occ);
        if (occ) {
          const int32_t u = ctrl[1] + wave_cnt[r * kWavesPerBlock + (tid >> 6)] + rank_below(m);
          slot_out[s] = u;
          c.unique_rows[u] = (int64_t)keys[s];
        }
      }
    }
    __syncthreads();
    for (int s0 = 0; s0 < slots; s0 += groups) {
      const int s = s0 + my_group;
      if (s < slots && live && keys[s] != kEmptyKey) {
        const V v = lds_read_v<V>(acc + (size_t)s * dim, c.chunks, sub);
        *reinterpret_cast<V*>(c.grad_rows + (int64_t)slot_out[s] * dim + (int64_t)sub * VE) = v;
        if (lr != 0.0f) {
          V* t = reinterpret_cast<V*>(c.table + (int64_t)keys[s] * dim + (int64_t)sub * VE);
          // this workgroup owns the row; bypass L1 so a second flush sees the first one
          const V old = __builtin_nontemporal_load(t);
          *t = old - lr * v;
        }
      }
    }
    __syncthreads();
    n_pairs = ctrl[0];
    src ^= 1;
    __syncthreads();
  }
}

__global__ __launch_bounds__(kBlock) void bwd_reduce_kernel(const GArgs a) {
  extern __shared__ char lds_raw[];
  HBK_FIND_COL(a, bucket0)
  const int bucket = (int)blockIdx.x - c.bucket0;
  if (c.vec4) {
    bucket_reduce<f32x4>(c, bucket, a.lr, lds_raw);
  } else {
    bucket_reduce<float>(c, bucket, a.lr, lds_raw);
  }
}

// ---- d(stitch + combiner): permutation scatter (hbk_group_stitch_bwd) ----------------------
constexpr int kIters = 4;
constexpr int kMaxStitchCols = 128;

struct SCol {
  const float* grad_out;
  const int32_t* splits;
  const int32_t* index;
  float* grad_rows;
  int64_t n_seg;
  int32_t dim;
  int32_t chunks;
  uint8_t lpr_log2, combiner, vec4, pad_;
  int32_t tile0;
};

struct SArgs {
  int32_t n_cols;
  int32_t pad_;
  SCol col[kMaxStitchCols];
};
static_assert(sizeof(SArgs) <= 16384, "kernarg budget");

template <typename V>
__device__ inline void stitch_segments(const SCol& c, int64_t seg0) {
  constexpr int VE = sizeof(V) / 4;
  const int lane = lane_id();
  const int rpi = kWave >> c.lpr_log2;
  const int sub = lane & ((1 << c.lpr_log2) - 1);
  const int grp = lane >> c.lpr_log2;
  const bool live = sub < c.chunks;
  for (int it = 0; it < kIters; ++it) {
    const int64_t s = seg0 + (int64_t)it * rpi + grp;
    if (s >= c.n_seg || !live) continue;
    int32_t beg, end;
    if (c.splits != nullptr) {
      beg = c.splits[s];
      end = c.splits[s + 1];
    } else {
      beg = (int32_t)s;
      end = beg + 1;
    }
    if (end <= beg) continue;
    V g = __builtin_nontemporal_load(
        reinterpret_cast<const V*>(c.grad_out + s * (int64_t)c.dim + (int64_t)sub * VE));
    const int32_t n = end - beg;
    if (c.combiner == HBK_COMBINER_MEAN) {
      g = g / (float)n;
    } else if (c.combiner == HBK_COMBINER_SQRTN) {
      g = g / sqrtf((float)n);
    }
    for (int32_t j = beg; j < end; ++j) {
      *reinterpret_cast<V*>(c.grad_rows + (int64_t)c.index[j] * c.dim + (int64_t)sub * VE) = g;
    }
  }
}

__global__ __launch_bounds__(kBlock) void stitch_bwd_kernel(const SArgs a) {
  HBK_FIND_COL(a, tile0)
  const int64_t tile = (int)blockIdx.x - c.tile0;
  const int rpi = kWave >> c.lpr_log2;
  const int64_t seg0 = (tile * kWavesPerBlock + (threadIdx.x >> 6)) * (int64_t)(kIters * rpi);
  if (seg0 >= c.n_seg) return;
  if (c.vec4) {
    stitch_segments<f32x4>(c, seg0);
  } else {
    stitch_segments<float>(c, seg0);
  }
}

// ---- host-side planning ------------------------------------------------------------------------
inline size_t align8(size_t v) { return (v + 7) & ~(size_t)7; }

struct ColPlan {
  int slots_log2;
  int log2p;
  int64_t tiles;
  size_t lds_bytes;
};

// test hook: HBK_BWD_SLOTS_LOG2 forces a (tiny) table so the overflow / re-run path is exercised
int forced_slots_log2() {
  const char* e = getenv("HBK_BWD_SLOTS_LOG2");
  return e ? atoi(e) : -1;
}

ColPlan plan_of(int64_t n_ids, int32_t dim) {
  ColPlan p;
  int sl = 4;
  while (sl < 11 && ((size_t)2 << sl) * (12 + 4 * (size_t)dim) + 256 <= (size_t)kLdsBudget) ++sl;
  const int forced = forced_slots_log2();
  if (forced >= 1 && forced <= 11) sl = forced;
  p.slots_log2 = sl;
  const int64_t per_bucket = ((int64_t)1 << sl) / 2;  // aim at a half-full table
  int lp = 0;
  while (lp < 13 && ((int64_t)per_bucket << lp) < n_ids) ++lp;
  while (((int64_t)1 << lp) > kMaxBuckets) --lp;
  p.log2p = lp;
  p.tiles = (n_ids + kTile - 1) / kTile;
  p.lds_bytes = ((size_t)1 << sl) * (12 + 4 * (size_t)dim) + 8 + 4 * 8 * kWavesPerBlock;
  return p;
}

size_t col_workspace(const hbk_lookup_grad_column_t& h) {
  if (h.n_ids <= 0) return 0;
  const ColPlan p = plan_of(h.n_ids, h.dim);
  size_t b = align8(((size_t)p.tiles << p.log2p) * 4);   // hist
  b += align8((((size_t)1 << p.log2p) + 1) * 4);          // bstart
  b += 2 * (size_t)h.n_ids * 8;                            // pair_row x2
  b += 2 * align8((size_t)h.n_ids * 4);                    // pair_seg x2
  if (h.row_splits != nullptr) b += align8((size_t)h.n_ids * 4);
  return b;
}

}  // namespace
}  // namespace hbk

extern "C" size_t hbk_group_lookup_bwd_workspace_bytes(int32_t n_cols,
                                                       const hbk_lookup_grad_column_t* cols) {
  if (n_cols <= 0 || cols == nullptr) return 0;
  size_t total = 0;
  for (int32_t c = 0; c < n_cols; ++c) total += hbk::col_workspace(cols[c]);
  return total;
}

extern "C" int hbk_group_lookup_bwd(int32_t n_cols, const hbk_lookup_grad_column_t* cols,
                                    float apply_lr, void* workspace, size_t workspace_bytes,
                                    hbk_stream_t stream_) {
  using namespace hbk;
  hipStream_t stream = as_stream(stream_);
  HBK_REQUIRE(n_cols >= 0, "group_lookup_bwd: n_cols must be >= 0, got %d", n_cols);
  if (n_cols == 0) return HBK_OK;
  HBK_REQUIRE(cols != nullptr, "group_lookup_bwd: cols is NULL");
  for (int32_t c = 0; c < n_cols; ++c) {
    const hbk_lookup_grad_column_t& h = cols[c];
    HBK_REQUIRE(h.dim >= 1, "group_lookup_bwd: column %d: dim must be >= 1", c);
    HBK_REQUIRE(h.rows >= 0 && h.n_ids >= 0 && h.n_segments >= 0,
                "group_lookup_bwd: column %d: negative size", c);
    HBK_REQUIRE(h.n_ids < (1ll << 30), "group_lookup_bwd: column %d: more than 2^30-1 ids", c);
    HBK_REQUIRE(h.ids_dtype == HBK_INT32 || h.ids_dtype == HBK_INT64,
                "group_lookup_bwd: column %d: ids must be int32 or int64", c);
    HBK_REQUIRE(h.bucket >= 0 && h.divisor >= 1,
                "group_lookup_bwd: column %d: bad bucket/divisor", c);
    HBK_REQUIRE(h.combiner >= HBK_COMBINER_SUM && h.combiner <= HBK_COMBINER_SQRTN,
                "group_lookup_bwd: column %d: unknown combiner %d", c, h.combiner);
    HBK_REQUIRE(h.row_splits != nullptr || h.n_segments == h.n_ids,
                "group_lookup_bwd: column %d: n_segments must equal n_ids when row_splits "
                "is NULL", c);
    HBK_REQUIRE(h.n_unique != nullptr, "group_lookup_bwd: column %d: n_unique is NULL", c);
    HBK_REQUIRE(h.n_ids == 0 || (h.ids && h.grad_out && h.unique_rows && h.grad_rows),
                "group_lookup_bwd: column %d: NULL buffer", c);
    HBK_REQUIRE(apply_lr == 0.0f || h.table != nullptr || h.n_ids == 0,
                "group_lookup_bwd: column %d: table is NULL but apply_lr != 0", c);
  }
  const size_t need = hbk_group_lookup_bwd_workspace_bytes(n_cols, cols);
  HBK_REQUIRE(need == 0 || (workspace != nullptr && workspace_bytes >= need),
              "group_lookup_bwd: workspace too small: need %zu bytes, got %zu", need,
              workspace_bytes);
  HBK_REQUIRE(((uintptr_t)workspace & 7) == 0,
              "group_lookup_bwd: workspace must be 8-byte aligned");
  char* wp = reinterpret_cast<char*>(workspace);

  int32_t c0 = 0;
  while (c0 < n_cols) {
    GArgs args, seg_args;
    int32_t k = 0, ks = 0;
    int64_t tiles = 0, buckets = 0, segtiles = 0;
    size_t lds_hist = 0, lds_reduce = 0;
    while (c0 < n_cols && k < kMaxCols) {
      const hbk_lookup_grad_column_t& h = cols[c0++];
      if (h.n_ids == 0) {
        HBK_HIP_OK(hipMemsetAsync(h.n_unique, 0, sizeof(int32_t), stream));
        continue;
      }
      const ColPlan p = plan_of(h.n_ids, h.dim);
      GCol& d = args.col[k];
      memset(&d, 0, sizeof(d));
      d.ids = h.ids;
      d.grad_out = h.grad_out;
      d.splits = h.row_splits;
      d.unique_rows = h.unique_rows;
      d.grad_rows = h.grad_rows;
      d.n_unique = h.n_unique;
      d.table = h.table;
      d.hist = reinterpret_cast<int32_t*>(wp);
      wp += align8(((size_t)p.tiles << p.log2p) * 4);
      d.bstart = reinterpret_cast<int32_t*>(wp);
      wp += align8((((size_t)1 << p.log2p) + 1) * 4);
      for (int q = 0; q < 2; ++q) {
        d.pair_row[q] = reinterpret_cast<int64_t*>(wp);
        wp += (size_t)h.n_ids * 8;
      }
      for (int q = 0; q < 2; ++q) {
        d.pair_seg[q] = reinterpret_cast<int32_t*>(wp);
        wp += align8((size_t)h.n_ids * 4);
      }
      d.seg_of = nullptr;
      if (h.row_splits != nullptr) {
        d.seg_of = reinterpret_cast<int32_t*>(wp);
        wp += align8((size_t)h.n_ids * 4);
      }
      d.map = make_idmap(h.bucket, h.divisor, h.rows);
      d.n_ids = h.n_ids;
      d.n_seg = h.n_segments;
      d.dim = h.dim;
      RowShape shape;
      HBK_REQUIRE(make_rowshape(h.dim,
                                (uintptr_t)h.grad_out | (uintptr_t)h.grad_rows |
                                    (apply_lr != 0.0f ? (uintptr_t)h.table : 0),
                                &shape),
                  "group_lookup_bwd: dim %d needs more than 64 lanes per row", h.dim);
      d.chunks = shape.chunks;
      d.lpr_log2 = shape.lpr_log2;
      d.vec4 = shape.vec4;
      d.ids64 = h.ids_dtype == HBK_INT64;
      d.combiner = (uint8_t)h.combiner;
      d.log2p = p.log2p;
      d.slots_log2 = p.slots_log2;
      d.tile0 = (int32_t)tiles;
      d.bucket0 = (int32_t)buckets;
      d.segtile0 = 0;
      tiles += p.tiles;
      buckets += (int64_t)1 << p.log2p;
      HBK_REQUIRE(tiles < (1ll << 31) && buckets < (1ll << 31),
                  "group_lookup_bwd: grid too large");
      if (((size_t)4 << p.log2p) > lds_hist) lds_hist = (size_t)4 << p.log2p;
      if (p.lds_bytes > lds_reduce) lds_reduce = p.lds_bytes;
      if (h.row_splits != nullptr && h.n_segments > 0) {
        GCol& sdesc = seg_args.col[ks];
        sdesc = d;
        sdesc.segtile0 = (int32_t)segtiles;
        segtiles += (h.n_segments + kBlock - 1) / kBlock;
        ++ks;
      }
      ++k;
    }
    if (k == 0) continue;
    args.n_cols = k;
    args.lr = apply_lr;
    if (ks > 0) {
      seg_args.n_cols = ks;
      seg_args.lr = 0.f;
      hipLaunchKernelGGL(bwd_segof_kernel, dim3((unsigned)segtiles), dim3(kBlock), 0, stream,
                         seg_args);
    }
    hipLaunchKernelGGL(bwd_hist_kernel, dim3((unsigned)tiles), dim3(kBlock), lds_hist, stream,
                       args);
    hipLaunchKernelGGL(bwd_scan_kernel, dim3((unsigned)k), dim3(kBlock), 0, stream, args);
    hipLaunchKernelGGL(bwd_scatter_pairs_kernel, dim3((unsigned)tiles), dim3(kBlock), lds_hist,
                       stream, args);
    hipLaunchKernelGGL(bwd_reduce_kernel, dim3((unsigned)buckets), dim3(kBlock), lds_reduce,
                       stream, args);
    HBK_HIP_OK(hipGetLastError());
  }
  return HBK_OK;
}

// d(stitch + combiner) of the sharded pipeline (sharding.py:200 in reverse):
//   grad_rows[index[j], :] = scale(seg(j)) * grad_out[seg(j), :]
// `index` is the shard_index permutation of the forward partition, so every destination row is
// written exactly once: plain 16-byte stores, no atomics, no zeroing.
extern "C" int hbk_group_stitch_bwd(int32_t n_cols, const hbk_stitch_grad_column_t* cols,
                                    hbk_stream_t stream_) {
  using namespace hbk;
  hipStream_t stream = as_stream(stream_);
  HBK_REQUIRE(n_cols >= 0, "group_stitch_bwd: n_cols must be >= 0, got %d", n_cols);
  if (n_cols == 0) return HBK_OK;
  HBK_REQUIRE(cols != nullptr, "group_stitch_bwd: cols is NULL");
  int32_t c0 = 0;
  while (c0 < n_cols) {
    SArgs args;
    int32_t k = 0;
    int64_t t_seg = 0;
    while (c0 < n_cols && k < kMaxStitchCols) {
      const int32_t ci = c0++;
      const hbk_stitch_grad_column_t& h = cols[ci];
      HBK_REQUIRE(h.dim >= 1, "group_stitch_bwd: column %d: dim must be >= 1", ci);
      HBK_REQUIRE(h.n_ids >= 0 && h.n_segments >= 0 && h.n_ids < (1ll << 31),
                  "group_stitch_bwd: column %d: bad size", ci);
      HBK_REQUIRE(h.combiner >= HBK_COMBINER_SUM && h.combiner <= HBK_COMBINER_SQRTN,
                  "group_stitch_bwd: column %d: unknown combiner %d", ci, h.combiner);
      HBK_REQUIRE(h.row_splits != nullptr || h.n_segments == h.n_ids,
                  "group_stitch_bwd: column %d: n_segments must equal n_ids when row_splits "
                  "is NULL", ci);
      if (h.n_ids == 0 || h.n_segments == 0) continue;
      HBK_REQUIRE(h.index && h.grad_out && h.grad_rows,
                  "group_stitch_bwd: column %d: NULL buffer", ci);
      SCol& d = args.col[k];
      d.grad_out = h.grad_out;
      d.splits = h.row_splits;
      d.index = h.index;
      d.grad_rows = h.grad_rows;
      d.n_seg = h.n_segments;
      d.dim = h.dim;
      RowShape shape;
      HBK_REQUIRE(make_rowshape(h.dim, (uintptr_t)h.grad_out | (uintptr_t)h.grad_rows, &shape),
                  "group_stitch_bwd: dim %d needs more than 64 lanes per row", h.dim);
      d.chunks = shape.chunks;
      d.lpr_log2 = shape.lpr_log2;
      d.vec4 = shape.vec4;
      d.combiner = (uint8_t)h.combiner;
      d.pad_ = 0;
      const int64_t rpi = kWave >> d.lpr_log2;
      const int64_t per_block = kWavesPerBlock * kIters * rpi;
      d.tile0 = (int32_t)t_seg;
      t_seg += (h.n_segments + per_block - 1) / per_block;
      HBK_REQUIRE(t_seg < (1ll << 31), "group_stitch_bwd: grid too large");
      ++k;
    }
    if (k == 0) continue;
    args.n_cols = k;
    args.pad_ = 0;
    hipLaunchKernelGGL(stitch_bwd_kernel, dim3((unsigned)t_seg), dim3(kBlock), 0, stream, args);
    HBK_HIP_OK(hipGetLastError());
  }
  return HBK_OK;
}
