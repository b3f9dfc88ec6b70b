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
//                table full (needs > 2048 distinct keys in one bucket: adversarial hashing)
//                is resolved exactly by scanning its bucket.
//   5 count / 6 scan / 7 emit / 8 index   an id is a "first occurrence" iff first[i] == i;
//                per-1024-id-tile ballot + popcount ranks and a scan give its place in the
//                unique list -- order of first occurrence, exactly as TF's CPU kernel emits.
// Columns of <= 262144 ids (the owner side of a step) take FOUR launches instead of the nine:
//   group   1-3 in one: a 2048-id tile keeps its keys and their ranks inside the tile (the value
//           the LDS atomic returns) in registers, publishes its bucket counts and waits for the
//           other tiles of its column (all resident: <= 128 workgroups; sync.hip), derives the
//           bucket starts and its own offsets from all of them and stores its pairs staged
//           through LDS sorted by bucket (coalesced runs)
//   first   4, unchanged
//   order   5-7 in one: a 1024-id tile publishes its count of first occurrences and sums the
//           counts of the tiles BEFORE it (decoupled look-back: aggregate / inclusive-prefix words)
//   index   8, unchanged
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
constexpr int kBigTile = 2048;               // hist / scatter / group tiles
constexpr int kBigPerThread = kBigTile / kBlock;
constexpr int kBatch = 8;
constexpr int kMaxCols = 96;
constexpr int kSlots = 2048;
constexpr int kFirstBlock = kBlock;   // threads of the per-bucket kernel
constexpr int kMaxBuckets = 8192;
constexpr unsigned long long kEmpty = ~0ull;  // key -1 never enters the table (own counter)

// Probe builds only (-DHBK_PART_STAMPS, tools/Makefile): constant-clock stamps of the first
// workgroups of one kernel (picked by g_uni_which: 0 group, 1 first, 2 order).
#ifdef HBK_PART_STAMPS
constexpr int kUTraceBlocks = 8192, kUTraceSlots = 8;
__device__ unsigned long long g_uni_trace[kUTraceBlocks * kUTraceSlots];
__device__ int g_uni_which;
#define HBK_USTAMP(w, i)                                                                       \
  do {                                                                                         \
    if (threadIdx.x == 0 && blockIdx.x < kUTraceBlocks && g_uni_which == (w)) {                \
      g_uni_trace[blockIdx.x * kUTraceSlots + (i)] = __builtin_amdgcn_s_memrealtime();         \
    }                                                                                          \
  } while (0)
#else
#define HBK_USTAMP(w, i)
#endif

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
  int32_t sync0;       // group kernel: first of this column's [tiles][P] words
  int32_t pad_;
};

struct UArgs {
  int32_t n_cols;
  int32_t pad_;
  int32_t tile_start_v[kMaxCols];   // the columns' first blocks, packed (HBK_FIND_UCOL)
  int32_t big_start_v[kMaxCols];
  int32_t bucket0_v[kMaxCols];
  int32_t scan0_v[kMaxCols];
  UCol col[kMaxCols];
};
static_assert(sizeof(UArgs) <= 16384, "kernarg budget");

// column of this workgroup: every lane reads one column's first block (two loads cover the 96
// columns) and a ballot counts those <= blockIdx.x -- one memory round trip where a binary search
// over the descriptors takes five dependent scalar loads, cold at the start of these short kernels
#define HBK_FIND_UCOL(FIELD)                                                                   \
  int ci;                                                                                      \
  {                                                                                            \
    const int ln__ = (int)(threadIdx.x & (kWave - 1)), blk__ = (int)blockIdx.x;                \
    const int t0__ = ln__ < a.n_cols ? a.FIELD##_v[ln__] : 0x7fffffff;                         \
    const int t1__ = ln__ + kWave < a.n_cols ? a.FIELD##_v[ln__ + kWave] : 0x7fffffff;         \
    ci = (int)__builtin_popcountll(__ballot(t0__ <= blk__)) +                                  \
         (int)__builtin_popcountll(__ballot(t1__ <= blk__)) - 1;                               \
    ci = __builtin_amdgcn_readfirstlane(ci);                                                   \
  }                                                                                            \
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
        c.first[j] = (int32_t)j;   // (see unique_first_kernel: only the repeats are rewritten)
      }
    }
  }
}

// ---- 4: one workgroup per bucket: first occurrence of every key ------------------------------
__global__ __launch_bounds__(kFirstBlock) void unique_first_kernel(const UArgs a,
                                                                    const int32_t* poison) {
  __shared__ unsigned long long keys[kSlots];
  __shared__ uint32_t first[kSlots];
  __shared__ uint32_t first_m1;  // key -1 (the table's empty marker) has its own cell
  if (poisoned(poison)) return;   // the grouping launch gave up: bucket starts were never written
  HBK_FIND_UCOL(bucket0)
  const int bucket = (int)blockIdx.x - c.bucket0;
  const int tid = (int)threadIdx.x;
  HBK_USTAMP(1, 0);
  const int32_t start = c.bstart[bucket];
  const int32_t n = c.bstart[bucket + 1] - start;
  if (n == 0) return;
  HBK_USTAMP(1, 1);
  for (int i = tid; i < kSlots; i += kFirstBlock) {
    keys[i] = kEmpty;
    first[i] = 0xffffffffu;
  }
  if (tid == 0) first_m1 = 0xffffffffu;
  __syncthreads();
  HBK_USTAMP(1, 2);
  const int64_t* pkey = c.pair_key + start;
  const int32_t* pidx = c.pair_idx + start;
  // A round = kFirstKeys pairs per thread, all loads in flight at once; buckets aim at 512 pairs,
  // so one round is the rule and its registers serve both phases.  (Measured for 26 x 65536 ids:
  // 256 threads per 512-pair bucket 21 us, 128 threads 32 us: the kernel is bound by what one
  // thread does in sequence, not by the rate waves start at.)
  constexpr int kFirstKeys = 6;
  constexpr int kRound = kFirstBlock * kFirstKeys;
  unsigned long long key[kFirstKeys];
  uint32_t idx[kFirstKeys];
  for (int32_t r0 = 0; r0 < n; r0 += kRound) {
#pragma unroll
    for (int k = 0; k < kFirstKeys; ++k) {
      const int32_t e = r0 + k * kFirstBlock + tid;
      key[k] = e < n ? (unsigned long long)pkey[e] : 0ull;
      idx[k] = e < n ? (uint32_t)pidx[e] : 0u;
    }
#pragma unroll
    for (int k = 0; k < kFirstKeys; ++k) {
      if (r0 + k * kFirstBlock + tid >= n) continue;
      if (key[k] == kEmpty) {
        atomicMin(&first_m1, idx[k]);
        continue;
      }
      int h = (int)(mix64(key[k]) & (kSlots - 1));
      for (int probe = 0; probe < kSlots; ++probe) {
        const unsigned long long prev = atomicCAS(&keys[h], kEmpty, key[k]);
        if (prev == kEmpty || prev == key[k]) {
          atomicMin(&first[h], idx[k]);
          break;
        }
        h = (h + 1) & (kSlots - 1);
      }
    }
  }
  __syncthreads();
  HBK_USTAMP(1, 3);
  const bool one_round = n <= kRound;
  for (int32_t r0 = 0; r0 < n; r0 += kRound) {
    if (!one_round) {
#pragma unroll
      for (int k = 0; k < kFirstKeys; ++k) {
        const int32_t e = r0 + k * kFirstBlock + tid;
        key[k] = e < n ? (unsigned long long)pkey[e] : 0ull;
        idx[k] = e < n ? (uint32_t)pidx[e] : 0u;
      }
    }
#pragma unroll
    for (int k = 0; k < kFirstKeys; ++k) {
      if (r0 + k * kFirstBlock + tid >= n) continue;
      uint32_t f = 0xffffffffu;
      if (key[k] == kEmpty) {
        f = first_m1;
      } else {
        int h = (int)(mix64(key[k]) & (kSlots - 1));
        for (int probe = 0; probe < kSlots; ++probe) {
          const unsigned long long q = keys[h];
          if (q == key[k]) {
            f = first[h];
            break;
          }
          if (q == kEmpty) break;
          h = (h + 1) & (kSlots - 1);
        }
        if (f == 0xffffffffu) {  // table was full for this key: exact answer by scanning the bucket
          for (int32_t q = 0; q < n; ++q) {
            if ((unsigned long long)pkey[q] == key[k] && (uint32_t)pidx[q] < f) f = (uint32_t)pidx[q];
          }
        }
      }
      // first[] was filled with the identity, in order, where the ids were read (coalesced);
      // only the repeats are corrected here -- one isolated 4-byte store per id of a column
      // costs a partial line each
      if (f != idx[k]) c.first[idx[k]] = (int32_t)f;
    }
  }
  HBK_USTAMP(1, 4);
#ifdef HBK_PART_STAMPS
  __builtin_amdgcn_s_waitcnt(0x0f70);
  HBK_USTAMP(1, 5);
  HBK_USTAMP(1, 6);
  HBK_USTAMP(1, 7);
#endif
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

// ---- 1-3 in one launch ---------------------------------------------------------------------------
constexpr int kGroupMaxTiles = 128;    // 2048-id tiles per column
constexpr int kGroupMaxLog2P = 9;      // buckets per column: LDS counters

struct USync {
  int32_t* hist;        // group: per column [tiles][P] words, 0 = not published, else count + 1
  uint32_t* order;      // order: per 1024-id tile, 0 = nothing, v << 2 | 1 aggregate, | 2 inclusive
  int32_t* zero;        // words the call before left set (cleared by the group kernel)
  int64_t zero_words;
  SyncWait wait;        // bound of the waits, status / poison words, test hook
};


__global__ __launch_bounds__(kBlock, 4) void unique_group_kernel(const UArgs a, const USync y) {
  __shared__ int32_t counters[1 << kGroupMaxLog2P];   // counts, then bases
  __shared__ int32_t tot_s[1 << kGroupMaxLog2P], pre_s[1 << kGroupMaxLog2P];
  __shared__ int64_t st_key[kBigTile];     // the tile's pairs, sorted by bucket (staged scatter)
  __shared__ int32_t st_idx[kBigTile];
  __shared__ uint16_t st_b[kBigTile];
  __shared__ int32_t wave_cnt[kWavesPerBlock], n_staged;
  __shared__ int32_t wave_tot[kWavesPerBlock];
  __shared__ int32_t gave_up;
  const int tid = (int)threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  HBK_USTAMP(0, 0);
  for (int64_t j = (int64_t)blockIdx.x * kBlock + tid; j < y.zero_words;
       j += (int64_t)gridDim.x * kBlock) {
    y.zero[j] = 0;
  }
  HBK_FIND_UCOL(big_start)
  const int P = 1 << c.log2p;
  const int ctile = (int)blockIdx.x - c.big_start;
  const int n_tiles = (c.len + kBigTile - 1) / kBigTile;
  int32_t* hist = y.hist + c.sync0;
  for (int p = tid; p < P; p += kBlock) counters[p] = 0;
  if (tid == 0) gave_up = 0;
  const int64_t base = (int64_t)ctile * kBigTile;
  int64_t key[kBigPerThread];
#pragma unroll
  for (int k = 0; k < kBigPerThread; ++k) {
    const int64_t j = base + (int64_t)k * kBlock + tid;
    key[k] = j < c.len ? c.in[j] : 0;
  }
  __syncthreads();
  HBK_USTAMP(0, 1);
  // bucket and rank inside the tile's share of the bucket (what the LDS atomic returns)
  int32_t br[kBigPerThread];
#pragma unroll
  for (int k = 0; k < kBigPerThread; ++k) {
    const int64_t j = base + (int64_t)k * kBlock + tid;
    br[k] = -1;
    if (j < c.len) {
      const int b = bucket_of((uint64_t)key[k], c.log2p);
      br[k] = b | (atomicAdd(&counters[b], 1) << kGroupMaxLog2P);
    }
  }
  __syncthreads();
  HBK_USTAMP(0, 2);
  if ((int)blockIdx.x != y.wait.withhold) {
    for (int p = tid; p < P; p += kBlock) {
      __hip_atomic_store(hist + (int64_t)ctile * P + p, counters[p] + 1, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  HBK_USTAMP(0, 3);
  // first[] starts as the identity, written in order while the tile waits for the others
  // (unique_first_kernel corrects the repeats only)
#pragma unroll
  for (int k = 0; k < kBigPerThread; ++k) {
    const int64_t j = base + (int64_t)k * kBlock + tid;
    if (j < c.len) c.first[j] = (int32_t)j;
  }
  // totals of every bucket over the column's tiles and the part of the tiles before this one
  // (a thread per bucket, 16 tiles per poll), left in LDS for the scan
  const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
  bool lost = false;
#pragma unroll 1
  for (int p = tid; p < P && !lost; p += kBlock) {
    int32_t t_all = 0, t_pre = 0;
#pragma unroll 1
    for (int t0 = 0; t0 < n_tiles && !lost; t0 += 16) {
      int32_t x[16];
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          x[e] = 1;
          if (t0 + e < n_tiles) {
            x[e] = __hip_atomic_load(hist + (int64_t)(t0 + e) * P + p, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
          }
          ok = ok && x[e] != 0;
        }
        if (ok) break;
        if (__builtin_amdgcn_s_memrealtime() - t_begin > y.wait.ticks) {
          lost = true;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      if (lost) break;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        t_all += x[e] - 1;
        t_pre += t0 + e < ctile ? x[e] - 1 : 0;
      }
    }
    tot_s[p] = t_all;
    pre_s[p] = t_pre;
  }
  if (lost) gave_up = 1;
  __syncthreads();
  HBK_USTAMP(0, 4);
  if (gave_up != 0) {
    if (tid == 0) give_up(y.wait);
    return;
  }
  // bucket starts (scan of the column's totals) and the tile's own bucket offsets (scan of its
  // counts, for the staged scatter): thread t scans buckets [t * per, t * per + per)
  const int per = P >= kBlock ? P / kBlock : 1;
  const int p0 = tid * per;
  int32_t mine = 0, mine_c = 0;
  for (int q = 0; q < per; ++q) {
    mine += p0 + q < P ? tot_s[p0 + q] : 0;
    mine_c += p0 + q < P ? counters[p0 + q] : 0;
  }
  int32_t incl = mine, incl_c = mine_c;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const int32_t v = __shfl_up(incl, off, kWave);
    const int32_t w = __shfl_up(incl_c, off, kWave);
    if (lane >= off) {
      incl += v;
      incl_c += w;
    }
  }
  if (lane == kWave - 1) {
    wave_tot[wave] = incl;
    wave_cnt[wave] = incl_c;
  }
  __syncthreads();
  int32_t run = incl - mine, run_c = incl_c - mine_c;
  for (int w = 0; w < wave; ++w) {
    run += wave_tot[w];
    run_c += wave_cnt[w];
  }
  for (int q = 0; q < per; ++q) {
    const int pp = p0 + q;
    if (pp >= P) break;
    const int32_t n_b = tot_s[pp], n_c = counters[pp];
    if (ctile == 0) c.bstart[pp] = run;
    tot_s[pp] = run + pre_s[pp] - run_c;   // global position of the pair at staged slot L: + L
    pre_s[pp] = run_c;                     // first staged slot of the bucket
    run += n_b;
    run_c += n_c;
  }
  if (tid == kBlock - 1) n_staged = run_c;
  if (ctile == 0 && tid == 0) c.bstart[P] = c.len;
  __syncthreads();
  HBK_USTAMP(0, 5);
  // The tile's pairs go through LDS sorted by bucket and leave in that order: consecutive lanes
  // store consecutive positions of a bucket's run instead of 64 different buckets (= lines) per
  // store instruction -- the direct scatter spent 11 of the tile's 19 us issuing its stores.
#pragma unroll
  for (int k = 0; k < kBigPerThread; ++k) {
    if (br[k] >= 0) {
      const int b = br[k] & ((1 << kGroupMaxLog2P) - 1);
      const int L = pre_s[b] + (br[k] >> kGroupMaxLog2P);
      st_key[L] = key[k];
      st_idx[L] = (int32_t)(base + (int64_t)k * kBlock + tid);
      st_b[L] = (uint16_t)b;
    }
  }
  __syncthreads();
  const int n_st = n_staged;
#pragma unroll
  for (int k = 0; k < kBigPerThread; ++k) {
    const int L = k * kBlock + tid;
    if (L < n_st) {
      const int32_t pos = tot_s[st_b[L]] + L;
      c.pair_key[pos] = st_key[L];
      c.pair_idx[pos] = st_idx[L];
    }
  }
  HBK_USTAMP(0, 6);
#ifdef HBK_PART_STAMPS
  __builtin_amdgcn_s_waitcnt(0x0f70);
  HBK_USTAMP(0, 7);
#endif
}

// ---- 5-7 in one launch ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void unique_order_kernel(const UArgs a, const USync y) {
  __shared__ int32_t wave_cnt[kWavesPerBlock];
  __shared__ int32_t prefix_s;   // first occurrences in the tiles before this one; -1: gave up
  __shared__ int64_t st_val[kTile];   // the tile's first occurrences in list order
  static_assert(kPerThread == 4, "the dense upos stores below write four places per thread");
  // (1024-id tiles, 4 consecutive ids per thread; 4096-id tiles with 16 per thread measured 35 us
  // instead of 20 for 26 x 65536 ids: what one thread does in sequence is what counts)
  HBK_USTAMP(2, 0);
  if (poisoned(y.wait.poison)) return;   // the group kernel gave up: no pairs, no first[]
  HBK_FIND_UCOL(tile_start)
  const int ctile = (int)blockIdx.x - c.tile_start;
  const int n_tiles = (c.len + kTile - 1) / kTile;
  const int64_t i0 = (int64_t)ctile * kTile + (int64_t)threadIdx.x * kPerThread;
  int64_t val[kPerThread];   // the ids travel beside the flags: no load after the look-back
#pragma unroll
  for (int e = 0; e < kPerThread; ++e) val[e] = i0 + e < c.len ? c.in[i0 + e] : 0;
  const int flags = first_flags(c, i0);
  const int n = __builtin_popcount(flags);
  int below = 0, total = 0;
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const unsigned long long m = __ballot((n >> b) & 1);
    below += rank_below(m) << b;
    total += (int)__builtin_popcountll(m) << b;
  }
  const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
  if (lane == 0) wave_cnt[wave] = total;
  HBK_USTAMP(2, 1);   // (thread 0: loads arrived, flags counted)
  __syncthreads();
  HBK_USTAMP(2, 2);
  if (wave == 0) {
    uint32_t block_total = 0;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) block_total += (uint32_t)wave_cnt[w];
    uint32_t* words = y.order + c.tile_start;
    if (lane == 0) {
      __hip_atomic_store(words + ctile, (block_total << 2) | (ctile == 0 ? 2u : 1u),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // look back over the tiles before this one, nearest first, 64 at a time, up to the nearest
    // one that knows its inclusive prefix
    uint32_t excl = 0;
    bool lost = false;
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
    for (int look = ctile - 1; look >= 0; look -= kWave) {
      const int t = look - lane;
      uint32_t x;
      unsigned long long incl_mask;
      for (;;) {
        x = 2u;   // before the column's first tile: an inclusive prefix of nothing
        if (t >= 0) x = __hip_atomic_load(words + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        incl_mask = __ballot((x & 3u) == 2u);
        unsigned long long need = ~0ull;   // lanes whose word is needed: up to the nearest inclusive
        if (incl_mask != 0ull) {
          const int f = __builtin_ctzll(incl_mask);
          need = f == 63 ? ~0ull : ((1ull << (f + 1)) - 1ull);
        }
        if ((__ballot(x == 0u) & need) == 0ull) break;
        if (__builtin_amdgcn_s_memrealtime() - t_begin > y.wait.ticks) {
          lost = true;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      if (lost) break;
      const int f = incl_mask != 0ull ? __builtin_ctzll(incl_mask) : kWave;
      uint32_t part = lane <= f ? x >> 2 : 0u;
#pragma unroll
      for (int off = 1; off < kWave; off <<= 1) part += (uint32_t)__shfl_xor((int)part, off, kWave);
      excl += part;
      if (incl_mask != 0ull) break;
    }
    if (lane == 0) {
      prefix_s = lost ? -1 : (int32_t)excl;
      if (lost) {
        give_up(y.wait);
      } else {
        if (ctile > 0) {
          __hip_atomic_store(words + ctile, ((excl + block_total) << 2) | 2u, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        }
        if (ctile == n_tiles - 1) *c.n_unique = (int32_t)(excl + block_total);
      }
    }
  }
  HBK_USTAMP(2, 3);   // (wave 0: look-back done)
  __syncthreads();
  HBK_USTAMP(2, 4);
  if (prefix_s < 0) return;
  // The first occurrences leave through LDS in the order of the unique list, and the places go
  // out densely (-1 for the other ids): predicated 8- and 4-byte stores of every fourth element
  // touched four times the lines (4.4 of the tile's 15.6 us went into issuing them).
  int local = below;
  for (int w = 0; w < wave; ++w) local += wave_cnt[w];
  int tile_total = 0;
#pragma unroll
  for (int w = 0; w < kWavesPerBlock; ++w) tile_total += wave_cnt[w];
  const int32_t base_pos = prefix_s;
  int32_t up[kPerThread];
  {
    int r = 0;
#pragma unroll
    for (int e = 0; e < kPerThread; ++e) {
      up[e] = -1;
      if (flags & (1 << e)) {
        st_val[local + r] = val[e];
        up[e] = base_pos + local + r;
        ++r;
      }
    }
  }
  if (i0 + kPerThread <= c.len) {
    // (the workspace slices are 8-byte aligned and i0 is a multiple of 4)
    *reinterpret_cast<int2*>(c.upos + i0) = make_int2(up[0], up[1]);
    *reinterpret_cast<int2*>(c.upos + i0 + 2) = make_int2(up[2], up[3]);
  } else {
#pragma unroll
    for (int e = 0; e < kPerThread; ++e) {
      if (i0 + e < c.len) c.upos[i0 + e] = up[e];
    }
  }
  __syncthreads();
  for (int r = (int)threadIdx.x; r < tile_total; r += kBlock) c.uniq[base_pos + r] = st_val[r];
  HBK_USTAMP(2, 5);
#ifdef HBK_PART_STAMPS
  __builtin_amdgcn_s_waitcnt(0x0f70);
  HBK_USTAMP(2, 6);
  HBK_USTAMP(2, 7);
#endif
}

// ---- 8: index[i] = place of the first occurrence of in[i] --------------------------------------
__global__ __launch_bounds__(kBlock) void unique_index_kernel(const UArgs a,
                                                               const int32_t* poison) {
  if (poisoned(poison)) return;
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
  // aim at 512 keys per bucket, a quarter-full 2048-slot table (measured for 26 x 65536 ids: 256
  // keys 76 us, 512 keys 71 us, 1024 keys 80 us for the whole unique; a half-full 1024-slot table
  // costs the per-bucket kernel 29 us instead of 21: probe sequences are what it waits for)
  int lp = 0;
  while (lp < 13 && ((int64_t)512 << lp) < len) ++lp;
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
  {
    const int rc = sync_check("unique_n", stream);
    if (rc != HBK_OK) return rc;
  }
  // four launches when every column fits the group kernel (see there)
  bool onepass = options().unique_onepass != 0 && n_cols <= kMaxCols;
  size_t group_words = 0, order_words = 0;
  for (int32_t c = 0; c < n_cols && onepass; ++c) {
    if (cols[c].len == 0) continue;
    const int64_t cbig = (cols[c].len + kBigTile - 1) / kBigTile;
    const int lp = log2p_of(cols[c].len);
    onepass = cbig <= kGroupMaxTiles && lp <= kGroupMaxLog2P;
    group_words += (size_t)cbig << lp;
    order_words += (size_t)((cols[c].len + kTile - 1) / kTile);
  }
  USync sync;
  memset(&sync, 0, sizeof(sync));
  if (onepass && group_words > 0) {
    SyncTake take;
    onepass = group_words + order_words < (1u << 30) &&
              sync_take(stream, group_words + order_words, &take,
                        reinterpret_cast<const void*>(&unique_group_kernel), kBlock, kGroupMaxTiles);
    if (onepass) {
      sync.hist = take.words;
      sync.order = reinterpret_cast<uint32_t*>(take.words + group_words);
      sync.zero = take.zero;
      sync.zero_words = take.zero_words;
      sync.wait = sync_wait_of(take);
    }
  }

  int32_t c0 = 0;
  while (c0 < n_cols) {
    UArgs args;
    int32_t k = 0;
    int64_t tiles = 0, big = 0, buckets = 0, scans = 0, sync0 = 0;
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
      args.tile_start_v[k] = d.tile_start;
      args.big_start_v[k] = d.big_start;
      args.bucket0_v[k] = d.bucket0;
      args.scan0_v[k] = d.scan0;
      d.sync0 = (int32_t)sync0;
      sync0 += cbig << lp;
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
    if (onepass) {
      {
        SyncChain chain(stream);   // never beside another kernel whose tiles wait for later tiles
        hipLaunchKernelGGL(unique_group_kernel, dim3((unsigned)big), block, 0, stream, args, sync);
      }
      hipLaunchKernelGGL(unique_first_kernel, dim3((unsigned)buckets), dim3(kFirstBlock), 0, stream, args,
                         (const int32_t*)sync.wait.poison);
      hipLaunchKernelGGL(unique_order_kernel, dim3((unsigned)tiles), block, 0, stream, args, sync);
      hipLaunchKernelGGL(unique_index_kernel, dim3((unsigned)tiles), block, 0, stream, args,
                         (const int32_t*)sync.wait.poison);
      HBK_HIP_OK(hipGetLastError());
      continue;
    }
    hipLaunchKernelGGL(unique_hist_kernel, dim3((unsigned)big), block, lds, stream, args);
    hipLaunchKernelGGL(unique_scan_tiles_kernel, dim3((unsigned)scans), block, 0, stream, args);
    hipLaunchKernelGGL(unique_bucket_scan_kernel, dim3((unsigned)k), block, 0, stream, args);
    hipLaunchKernelGGL(unique_scatter_kernel, dim3((unsigned)big), block, lds, stream, args);
    hipLaunchKernelGGL(unique_first_kernel, dim3((unsigned)buckets), dim3(kFirstBlock), 0, stream, args,
                       (const int32_t*)nullptr);
    hipLaunchKernelGGL(unique_count_kernel, dim3((unsigned)tiles), block, 0, stream, args);
    hipLaunchKernelGGL(unique_scan_kernel, dim3((unsigned)k), block, 0, stream, args);
    hipLaunchKernelGGL(unique_emit_kernel, dim3((unsigned)tiles), block, 0, stream, args);
    hipLaunchKernelGGL(unique_index_kernel, dim3((unsigned)tiles), block, 0, stream, args,
                       (const int32_t*)nullptr);
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

#ifdef HBK_PART_STAMPS
extern "C" int hbk_debug_uni_trace(unsigned long long* out, int which) {
  using namespace hbk;
  HBK_HIP_OK(hipDeviceSynchronize());
  if (out != nullptr) {
    HBK_HIP_OK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_uni_trace), sizeof(g_uni_trace)));
  } else {
    static unsigned long long z[kUTraceBlocks * kUTraceSlots];
    HBK_HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_uni_trace), z, sizeof(z)));
    HBK_HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_uni_which), &which, sizeof(int)));
  }
  return HBK_OK;
}
#endif
