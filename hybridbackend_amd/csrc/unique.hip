// N-ary first-occurrence unique for gfx950 (R7): the owner-side `array_ops.unique` of
// hbtf/embedding/sharding.py:186 (TF Unique: values in first-occurrence order + the index of
// every input in that list).
//
// Device-scope atomics resolve at the memory side on MI355X (~15 G/s), so the first version
// (global open-addressing table, CAS + atomicMin per id) needed ~330 us for 26 x 65536 ids.
// This version keeps every atomic in LDS:
//
//   1 hist / 2 scan / 3 scatter   (key, position) pairs grouped by bucket = top bits of a
//                64-bit mix of the key (same structure as the backward, lookup_bwd.hip)
//   4 first      ONE workgroup owns a bucket, hence every key that hashes to it: LDS hash
//                table key -> minimum position (64-bit LDS CAS + ds_min_u32); every pair then
//                learns the position of its key's first occurrence.  A key that finds the
//                table full (needs > 1024 distinct keys in one bucket: adversarial hashing)
//                is resolved exactly by scanning its bucket.
//   5 count / 6 scan / 7 emit / 8 index   an id is a "first occurrence" iff first[i] == i;
//                per-1024-id-tile ballot + popcount ranks and a scan give its place in the
//                unique list -- order of first occurrence, exactly as TF's CPU kernel emits.
// All N columns share each launch (descriptors by value in the kernel-argument segment).
#include <alloca.h>
#include <stdlib.h>
#include <string.h>

#include "unique.h"

namespace hbk {
namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kPerThread = 4;
constexpr int kTile = kBlock * kPerThread;   // 1024 ids: count / emit / index tiles
constexpr int kBigTile = 4096;               // hist / scatter tiles
constexpr int kBigPerThread = kBigTile / kBlock;
constexpr int kBatch = 8;
constexpr int kMaxCols = 96;
constexpr int kSlots = 1024;
constexpr int kMaxBuckets = 8192;
constexpr unsigned long long kEmpty = ~0ull;  // key -1 never enters the table (own counter)

struct UCol {
  const int64_t* in;
  int64_t* uniq;
  int32_t* index;
  int32_t* n_unique;
  int32_t* hist;       // [P * big_tiles]
  int32_t* bstart;     // [P + 1]
  int64_t* pair_key;   // [len]
  int32_t* pair_idx;   // [len]
  int32_t* first;      // [len] position of the first occurrence of in[i]
  int32_t* upos;       // [len] place in the unique list (first occurrences only)
  int32_t* tile_off;   // [tiles + 1]
  int32_t len;
  int32_t log2p;
  int32_t tile_start;  // first 1024-id tile
  int32_t big_start;   // first 4096-id tile
  int32_t bucket0;     // first block of the per-bucket kernel
  int32_t scan0;       // first block of the scan-over-tiles kernel
};

struct UArgs {
  int32_t n_cols;
  int32_t pad_;
  UCol col[kMaxCols];
};
static_assert(sizeof(UArgs) <= 16384, "kernarg budget");

#define HBK_FIND_UCOL(FIELD)                                                       \
  int ci = 0, hi__ = a.n_cols;                                                     \
  while (hi__ - ci > 1) {                                                          \
    const int mid__ = (ci + hi__) >> 1;                                            \
    if (a.col[mid__].FIELD <= (int)blockIdx.x) {                                   \
      ci = mid__;                                                                  \
    } else {                                                                       \
      hi__ = mid__;                                                                \
    }                                                                              \
  }                                                                                \
  const UCol& c = a.col[ci];

__device__ inline uint64_t mix64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return k;
}

__device__ inline int bucket_of(uint64_t key, int log2p) {
  return log2p == 0 ? 0 : (int)(mix64(key) >> (64 - log2p));
}

// ---- 1: per-tile bucket histogram ---------------------------------------------------------
__global__ __launch_bounds__(kBlock) void unique_hist_kernel(const UArgs a) {
  extern __shared__ int32_t counters[];
  HBK_FIND_UCOL(big_start)
  const int P = 1 << c.log2p;
  const int tid = (int)threadIdx.x;
  const int ctile = (int)blockIdx.x - c.big_start;
  for (int p = tid; p < P; p += kBlock) counters[p] = 0;
  __syncthreads();
  const int64_t base = (int64_t)ctile * kBigTile;
  for (int k0 = 0; k0 < kBigPerThread; k0 += kBatch) {
    int64_t key[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      const int64_t j = base + (int64_t)(k0 + k) * kBlock + tid;
      key[k] = j < c.len ? c.in[j] : 0;
    }
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      const int64_t j = base + (int64_t)(k0 + k) * kBlock + tid;
      if (j < c.len) atomicAdd(&counters[bucket_of((uint64_t)key[k], c.log2p)], 1);
    }
  }
  __syncthreads();
  for (int p = tid; p < P; p += kBlock) c.hist[(int64_t)ctile * P + p] = counters[p];
}

// ---- 2: offsets of every (tile, bucket) run (same scheme as lookup_bwd.hip) ------------------
// hist is [tile][bucket].  2a: one thread per bucket turns its column of the matrix into the
// exclusive prefix over tiles and leaves the bucket total in bstart; 2b: one block per input
// scans the bucket totals into bucket starts.
__global__ __launch_bounds__(kBlock) void unique_scan_tiles_kernel(const UArgs a) {
  HBK_FIND_UCOL(scan0)
  const int P = 1 << c.log2p;
  const int p = ((int)blockIdx.x - c.scan0) * kBlock + (int)threadIdx.x;
  if (p >= P) return;
  const int n_tiles = (c.len + kBigTile - 1) / kBigTile;
  int32_t* h = c.hist + p;
  int32_t run = 0;
  int t = 0;
  for (; t + 4 <= n_tiles; t += 4) {
    int32_t x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = h[(int64_t)(t + k) * P];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      h[(int64_t)(t + k) * P] = run;
      run += x[k];
    }
  }
  for (; t < n_tiles; ++t) {
    const int32_t x = h[(int64_t)t * P];
    h[(int64_t)t * P] = run;
    run += x;
  }
  c.bstart[p] = run;
}

__global__ __launch_bounds__(kBlock) void unique_bucket_scan_kernel(const UArgs a) {
  __shared__ int32_t wave_tot[kWavesPerBlock];
  const UCol& c = a.col[blockIdx.x];
  const int P = 1 << c.log2p;
  const int tid = (int)threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  const int per = (P + kBlock - 1) / kBlock;
  const int beg = tid * per;
  const int end = beg + per < P ? beg + per : P;
  int32_t sum = 0;
  for (int p = beg; p < end; ++p) sum += c.bstart[p];
  int32_t incl = sum;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const int32_t y = __shfl_up(incl, off, kWave);
    if (lane >= off) incl += y;
  }
  if (lane == kWave - 1) wave_tot[wave] = incl;
  __syncthreads();
  int32_t run = incl - sum;
  for (int w = 0; w < wave; ++w) run += wave_tot[w];
  for (int p = beg; p < end; ++p) {
    const int32_t n_b = c.bstart[p];
    c.bstart[p] = run;
    run += n_b;
  }
  if (tid == kBlock - 1) c.bstart[P] = c.len;
}

// ---- 3: (key, position) pairs grouped by bucket --------------------------------------------
__global__ __launch_bounds__(kBlock) void unique_scatter_kernel(const UArgs a) {
  extern __shared__ int32_t run[];
  HBK_FIND_UCOL(big_start)
  const int P = 1 << c.log2p;
  const int tid = (int)threadIdx.x;
  const int ctile = (int)blockIdx.x - c.big_start;
  for (int p = tid; p < P; p += kBlock) run[p] = c.bstart[p] + c.hist[(int64_t)ctile * P + p];
  __syncthreads();
  const int64_t base = (int64_t)ctile * kBigTile;
  for (int k0 = 0; k0 < kBigPerThread; k0 += kBatch) {
    int64_t key[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      const int64_t j = base + (int64_t)(k0 + k) * kBlock + tid;
      key[k] = j < c.len ? c.in[j] : 0;
    }
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      const int64_t j = base + (int64_t)(k0 + k) * kBlock + tid;
      if (j < c.len) {
        const int32_t pos = atomicAdd(&run[bucket_of((uint64_t)key[k], c.log2p)], 1);
        c.pair_key[pos] = key[k];
        c.pair_idx[pos] = (int32_t)j;
      }
    }
  }
}

// ---- 4: one workgroup per bucket: first occurrence of every key ------------------------------
__global__ __launch_bounds__(kBlock) void unique_first_kernel(const UArgs a) {
  __shared__ unsigned long long keys[kSlots];
  __shared__ uint32_t first[kSlots];
  __shared__ uint32_t first_m1;  // key -1 (the table's empty marker) has its own cell
  HBK_FIND_UCOL(bucket0)
  const int bucket = (int)blockIdx.x - c.bucket0;
  const int tid = (int)threadIdx.x;
  const int32_t start = c.bstart[bucket];
  const int32_t n = c.bstart[bucket + 1] - start;
  if (n == 0) return;
  for (int i = tid; i < kSlots; i += kBlock) {
    keys[i] = kEmpty;
    first[i] = 0xffffffffu;
  }
  if (tid == 0) first_m1 = 0xffffffffu;
  __syncthreads();
  const int64_t* pkey = c.pair_key + start;
  const int32_t* pidx = c.pair_idx + start;
  for (int32_t e = tid; e < n; e += kBlock) {
    const unsigned long long key = (unsigned long long)pkey[e];
    const uint32_t idx = (uint32_t)pidx[e];
    if (key == kEmpty) {
      atomicMin(&first_m1, idx);
      continue;
    }
    int h = (int)(mix64(key) & (kSlots - 1));
    for (int probe = 0; probe < kSlots; ++probe) {
      const unsigned long long prev = atomicCAS(&keys[h], kEmpty, key);
      if (prev == kEmpty || prev == key) {
        atomicMin(&first[h], idx);
        break;
      }
      h = (h + 1) & (kSlots - 1);
    }
  }
  __syncthreads();
  for (int32_t e = tid; e < n; e += kBlock) {
    const unsigned long long key = (unsigned long long)pkey[e];
    const int32_t idx = pidx[e];
    uint32_t f = 0xffffffffu;
    if (key == kEmpty) {
      f = first_m1;
    } else {
      int h = (int)(mix64(key) & (kSlots - 1));
      for (int probe = 0; probe < kSlots; ++probe) {
        const unsigned long long k = keys[h];
        if (k == key) {
          f = first[h];
          break;
        }
        if (k == kEmpty) break;
        h = (h + 1) & (kSlots - 1);
      }
      if (f == 0xffffffffu) {  // table was full for this key: exact answer by scanning the bucket
        for (int32_t q = 0; q < n; ++q) {
          if ((unsigned long long)pkey[q] == key && (uint32_t)pidx[q] < f) f = (uint32_t)pidx[q];
        }
      }
    }
    c.first[idx] = (int32_t)f;
  }
}

// flags of the 4 consecutive ids owned by this thread: bit e set iff id i0+e is the first
// occurrence of its value
__device__ inline int first_flags(const UCol& c, int64_t i0) {
  int flags = 0;
#pragma unroll
  for (int e = 0; e < kPerThread; ++e) {
    const int64_t i = i0 + e;
    if (i < c.len && c.first[i] == (int32_t)i) flags |= 1 << e;
  }
  return flags;
}

// ---- 5: first occurrences per 1024-id tile -----------------------------------------------------
__global__ __launch_bounds__(kBlock) void unique_count_kernel(const UArgs a) {
  __shared__ int32_t wave_cnt[kWavesPerBlock];
  HBK_FIND_UCOL(tile_start)
  const int ctile = (int)blockIdx.x - c.tile_start;
  const int64_t i0 = (int64_t)ctile * kTile + (int64_t)threadIdx.x * kPerThread;
  const int n = __builtin_popcount(first_flags(c, i0));
  int total = 0;
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    total += (int)__builtin_popcountll(__ballot((n >> b) & 1)) << b;
  }
  if (lane_id() == 0) wave_cnt[threadIdx.x >> 6] = total;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) t += wave_cnt[w];
    c.tile_off[ctile] = t;
  }
}

// ---- 6: scan of the tile counts ------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void unique_scan_kernel(const UArgs a) {
  __shared__ int32_t wave_tot[kWavesPerBlock];
  __shared__ int32_t carry_s;
  const UCol& c = a.col[blockIdx.x];
  const int n_tiles = (c.len + kTile - 1) / kTile;
  const int tid = (int)threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int e0 = 0; e0 < n_tiles; e0 += kBlock) {
    const int e = e0 + tid;
    const int32_t x = e < n_tiles ? c.tile_off[e] : 0;
    int32_t s = x;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int32_t y = __shfl_up(s, off, kWave);
      if (lane >= off) s += y;
    }
    if (lane == kWave - 1) wave_tot[wave] = s;
    __syncthreads();
    int32_t wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += wave_tot[w];
    const int32_t excl = carry_s + wbase + s - x;
    if (e < n_tiles) c.tile_off[e] = excl;
    __syncthreads();
    if (tid == kBlock - 1) carry_s = excl + x;
    __syncthreads();
  }
  if (tid == 0) *c.n_unique = carry_s;
}

// ---- 7: emit the unique list in order of first occurrence --------------------------------------
__global__ __launch_bounds__(kBlock) void unique_emit_kernel(const UArgs a) {
  __shared__ int32_t wave_cnt[kWavesPerBlock];
  HBK_FIND_UCOL(tile_start)
  const int ctile = (int)blockIdx.x - c.tile_start;
  const int64_t i0 = (int64_t)ctile * kTile + (int64_t)threadIdx.x * kPerThread;
  const int flags = first_flags(c, i0);
  const int n = __builtin_popcount(flags);
  int below = 0, total = 0;
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const unsigned long long m = __ballot((n >> b) & 1);
    below += rank_below(m) << b;
    total += (int)__builtin_popcountll(m) << b;
  }
  const int wave = (int)(threadIdx.x >> 6);
  if (lane_id() == 0) wave_cnt[wave] = total;
  __syncthreads();
  int pos = c.tile_off[ctile] + below;
  for (int w = 0; w < wave; ++w) pos += wave_cnt[w];
#pragma unroll
  for (int e = 0; e < kPerThread; ++e) {
    if (flags & (1 << e)) {
      const int64_t i = i0 + e;
      c.uniq[pos] = c.in[i];
      c.upos[i] = pos;
      ++pos;
    }
  }
}

// ---- 8: index[i] = place of the first occurrence of in[i] --------------------------------------
__global__ __launch_bounds__(kBlock) void unique_index_kernel(const UArgs a) {
  HBK_FIND_UCOL(tile_start)
  const int64_t base = (int64_t)((int)blockIdx.x - c.tile_start) * kTile;
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const int64_t i = base + (int64_t)k * kBlock + threadIdx.x;
    if (i < c.len) c.index[i] = c.upos[c.first[i]];
  }
}

inline size_t align8(size_t v) { return (v + 7) & ~(size_t)7; }

inline int log2p_of(int64_t len) {
  int lp = 0;  // aim at 256 keys per bucket: a quarter-full 1024-slot table
  while (lp < 13 && ((int64_t)256 << lp) < len) ++lp;
  while (((int64_t)1 << lp) > kMaxBuckets) --lp;
  const int forced = options().unique_buckets_log2;   // option: force the bucket count
  if (forced >= 0 && forced <= 13) lp = forced;
  return lp;
}

size_t col_bytes(int64_t len) {
  if (len <= 0) return 0;
  const int lp = log2p_of(len);
  const int64_t big = (len + kBigTile - 1) / kBigTile;
  const int64_t tiles = (len + kTile - 1) / kTile;
  return align8(((size_t)big << lp) * 4) + align8((((size_t)1 << lp) + 1) * 4) +
         (size_t)len * 8 + 3 * align8((size_t)len * 4) + align8((size_t)(tiles + 1) * 4);
}

}  // namespace

size_t unique_workspace_bytes(int32_t n_cols, const int64_t* lens) {
  if (n_cols <= 0 || lens == nullptr) return 0;
  size_t total = 0;
  for (int32_t c = 0; c < n_cols; ++c) total += col_bytes(lens[c]);
  return total;
}

int unique_n_impl(int32_t n_cols, const UniqueColumn* cols, void* workspace,
                  size_t workspace_bytes, hipStream_t stream) {
  HBK_REQUIRE(n_cols >= 0, "unique_n: n_cols must be >= 0");
  if (n_cols == 0) return HBK_OK;
  HBK_REQUIRE(cols != nullptr, "unique_n: NULL argument array");
  size_t need = 0;
  for (int32_t c = 0; c < n_cols; ++c) {
    HBK_REQUIRE(cols[c].len >= 0 && cols[c].len < (1ll << 30),
                "unique_n: input %d must have fewer than 2^30 elements, got %lld", c,
                (long long)cols[c].len);
    HBK_REQUIRE(cols[c].n_unique != nullptr, "unique_n: n_unique[%d] is NULL", c);
    HBK_REQUIRE(cols[c].len == 0 || (cols[c].in && cols[c].unique_out && cols[c].index_out),
                "unique_n: NULL buffer for input %d", c);
    need += col_bytes(cols[c].len);
  }
  HBK_REQUIRE(need == 0 || (workspace != nullptr && workspace_bytes >= need),
              "unique_n: workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
  HBK_REQUIRE(((uintptr_t)workspace & 7) == 0, "unique_n: workspace must be 8-byte aligned");
  char* wp = reinterpret_cast<char*>(workspace);

  int32_t c0 = 0;
  while (c0 < n_cols) {
    UArgs args;
    int32_t k = 0;
    int64_t tiles = 0, big = 0, buckets = 0, scans = 0;
    size_t lds = 0;
    while (c0 < n_cols && k < kMaxCols) {
      const UniqueColumn& h = cols[c0++];
      if (h.len == 0) {
        HBK_HIP_OK(hipMemsetAsync(h.n_unique, 0, sizeof(int32_t), stream));
        continue;
      }
      UCol& d = args.col[k];
      memset(&d, 0, sizeof(d));
      const int lp = log2p_of(h.len);
      const int64_t cbig = (h.len + kBigTile - 1) / kBigTile;
      const int64_t ctiles = (h.len + kTile - 1) / kTile;
      d.in = h.in;
      d.uniq = h.unique_out;
      d.index = h.index_out;
      d.n_unique = h.n_unique;
      d.hist = reinterpret_cast<int32_t*>(wp);
      wp += align8(((size_t)cbig << lp) * 4);
      d.bstart = reinterpret_cast<int32_t*>(wp);
      wp += align8((((size_t)1 << lp) + 1) * 4);
      d.pair_key = reinterpret_cast<int64_t*>(wp);
      wp += (size_t)h.len * 8;
      d.pair_idx = reinterpret_cast<int32_t*>(wp);
      wp += align8((size_t)h.len * 4);
      d.first = reinterpret_cast<int32_t*>(wp);
      wp += align8((size_t)h.len * 4);
      d.upos = reinterpret_cast<int32_t*>(wp);
      wp += align8((size_t)h.len * 4);
      d.tile_off = reinterpret_cast<int32_t*>(wp);
      wp += align8((size_t)(ctiles + 1) * 4);
      d.len = (int32_t)h.len;
      d.log2p = lp;
      d.tile_start = (int32_t)tiles;
      d.big_start = (int32_t)big;
      d.bucket0 = (int32_t)buckets;
      d.scan0 = (int32_t)scans;
      scans += (((int64_t)1 << lp) + kBlock - 1) / kBlock;
      tiles += ctiles;
      big += cbig;
      buckets += (int64_t)1 << lp;
      if (((size_t)4 << lp) > lds) lds = (size_t)4 << lp;
      HBK_REQUIRE(tiles < (1ll << 31) && buckets < (1ll << 31), "unique_n: grid too large");
      ++k;
    }
    if (k == 0) continue;
    args.n_cols = k;
    args.pad_ = 0;
    const dim3 block(kBlock);
    hipLaunchKernelGGL(unique_hist_kernel, dim3((unsigned)big), block, lds, stream, args);
    hipLaunchKernelGGL(unique_scan_tiles_kernel, dim3((unsigned)scans), block, 0, stream, args);
    hipLaunchKernelGGL(unique_bucket_scan_kernel, dim3((unsigned)k), block, 0, stream, args);
    hipLaunchKernelGGL(unique_scatter_kernel, dim3((unsigned)big), block, lds, stream, args);
    hipLaunchKernelGGL(unique_first_kernel, dim3((unsigned)buckets), block, 0, stream, args);
    hipLaunchKernelGGL(unique_count_kernel, dim3((unsigned)tiles), block, 0, stream, args);
    hipLaunchKernelGGL(unique_scan_kernel, dim3((unsigned)k), block, 0, stream, args);
    hipLaunchKernelGGL(unique_emit_kernel, dim3((unsigned)tiles), block, 0, stream, args);
    hipLaunchKernelGGL(unique_index_kernel, dim3((unsigned)tiles), block, 0, stream, args);
    HBK_HIP_OK(hipGetLastError());
  }
  return HBK_OK;
}

}  // namespace hbk

extern "C" size_t hbk_unique_workspace_bytes(int32_t n_cols, const int64_t* lens) {
  return hbk::unique_workspace_bytes(n_cols, lens);
}

extern "C" int hbk_unique_n(int32_t n_cols, const int64_t* const* inputs, const int64_t* lens,
                            int64_t* const* unique_out, int32_t* const* index_out,
                            int32_t* const* n_unique, void* workspace, size_t workspace_bytes,
                            hbk_stream_t stream) {
  using namespace hbk;
  HBK_REQUIRE(n_cols >= 0, "unique_n: n_cols must be >= 0");
  if (n_cols == 0) return HBK_OK;
  HBK_REQUIRE(inputs && lens && unique_out && index_out && n_unique,
              "unique_n: NULL argument array");
  UniqueColumn* cols = (UniqueColumn*)alloca(sizeof(UniqueColumn) * (size_t)n_cols);
  for (int32_t c = 0; c < n_cols; ++c) {
    cols[c].in = inputs[c];
    cols[c].len = lens[c];
    cols[c].unique_out = unique_out[c];
    cols[c].index_out = index_out[c];
    cols[c].n_unique = n_unique[c];
    cols[c].multiplicity = nullptr;
  }
  return unique_n_impl(n_cols, cols, workspace, workspace_bytes, as_stream(stream));
}
