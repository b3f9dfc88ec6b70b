// N-ary first-occurrence unique for gfx950 (R7): the owner-side `array_ops.unique` of
// hbtf/embedding/sharding.py:186 (TF Unique: values in first-occurrence order + the index of
// every input in that list), and the duplicate-id detection the backward needs (R10).
//
//   1 insert   open-addressing table (capacity >= 2n, keys claimed with a 64-bit CAS);
//              every id atomically min-s its position into first[slot] and counts itself
//   2 count    an id is a "first occurrence" iff first[slot] == its own position; per
//              1024-id tile the flags are counted with wave ballots + popcounts
//   3 scan     per column exclusive scan of the tile counts (-> n_unique)
//   4 emit     tile-local ballot/prefix-sum ranks + tile offset = position in the unique
//              list: order of first occurrence, exactly as TF's CPU kernel emits
//   5 index    index[i] = position of the first occurrence of ids[i]
// All N columns share each launch (tile prefix in the kernel-argument segment).
#include <alloca.h>

#include "unique.h"

namespace hbk {
namespace {

constexpr int kBlock = 256;
constexpr int kPerThread = 4;
constexpr int kTile = kBlock * kPerThread;  // 1024 ids
constexpr int kMaxCols = 128;
constexpr unsigned long long kEmpty = ~0ull;  // key -1 is kept in the dedicated slot H

struct UCol {
  const int64_t* in;
  int64_t* uniq;
  int32_t* index;
  int32_t* n_unique;
  int32_t* mult;
  unsigned long long* keys;  // [H + 1]
  uint32_t* first;           // [H + 1]
  int32_t* cnt;              // [H + 1]
  uint32_t* slot_of;         // [len]
  int32_t* upos;             // [len]
  int32_t* tile_off;         // [tiles + 1]
  int32_t len;
  uint32_t hmask;            // H - 1
  int32_t tile_start;
  int32_t pad_;
};

struct UArgs {
  int32_t n_cols;
  int32_t pad_;
  UCol col[kMaxCols];
};
static_assert(sizeof(UArgs) <= 16384, "kernarg budget");

__device__ inline int find_col(const UArgs& a, int tile) {
  int ci = 0;
  while (ci + 1 < a.n_cols && a.col[ci + 1].tile_start <= tile) ++ci;
  return ci;
}

__device__ inline uint64_t mix64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return k;
}

__global__ __launch_bounds__(kBlock) void unique_insert_kernel(const UArgs a) {
  const int tile = (int)blockIdx.x;
  const UCol& c = a.col[find_col(a, tile)];
  const int64_t base = (int64_t)(tile - c.tile_start) * kTile;
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const int64_t i = base + (int64_t)k * kBlock + threadIdx.x;
    if (i >= c.len) continue;
    const unsigned long long key = (unsigned long long)c.in[i];
    uint32_t slot;
    if (key == kEmpty) {
      slot = c.hmask + 1u;
    } else {
      uint32_t h = (uint32_t)mix64(key) & c.hmask;
      for (;;) {
        const unsigned long long prev = atomicCAS(&c.keys[h], kEmpty, key);
        if (prev == kEmpty || prev == key) break;
        h = (h + 1u) & c.hmask;
      }
      slot = h;
    }
    atomicMin(&c.first[slot], (uint32_t)i);
    atomicAdd(&c.cnt[slot], 1);
    c.slot_of[i] = slot;
  }
}

// flags of the 4 consecutive ids owned by this thread: bit e set iff id base+tid*4+e is the
// first occurrence of its value
__device__ inline int first_flags(const UCol& c, int64_t i0) {
  int flags = 0;
#pragma unroll
  for (int e = 0; e < kPerThread; ++e) {
    const int64_t i = i0 + e;
    if (i < c.len && c.first[c.slot_of[i]] == (uint32_t)i) flags |= 1 << e;
  }
  return flags;
}

__global__ __launch_bounds__(kBlock) void unique_count_kernel(const UArgs a) {
  __shared__ int32_t wave_cnt[kBlock / kWave];
  const int tile = (int)blockIdx.x;
  const UCol& c = a.col[find_col(a, tile)];
  const int ctile = tile - c.tile_start;
  const int64_t i0 = (int64_t)ctile * kTile + (int64_t)threadIdx.x * kPerThread;
  const int n = __builtin_popcount(first_flags(c, i0));
  // wave total via ballots of the count's bits (n <= 4: 3 bits)
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
    for (int w = 0; w < kBlock / kWave; ++w) t += wave_cnt[w];
    c.tile_off[ctile] = t;
  }
}

__global__ __launch_bounds__(kBlock) void unique_scan_kernel(const UArgs a) {
  __shared__ int32_t wave_tot[kBlock / kWave];
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

__global__ __launch_bounds__(kBlock) void unique_emit_kernel(const UArgs a) {
  __shared__ int32_t wave_cnt[kBlock / kWave];
  const int tile = (int)blockIdx.x;
  const UCol& c = a.col[find_col(a, tile)];
  const int ctile = tile - c.tile_start;
  const int64_t i0 = (int64_t)ctile * kTile + (int64_t)threadIdx.x * kPerThread;
  const int flags = first_flags(c, i0);
  const int n = __builtin_popcount(flags);
  // exclusive prefix of n over the lanes of this wave: ballot per bit + popcount below
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
      if (c.mult != nullptr) c.mult[pos] = c.cnt[c.slot_of[i]];
      ++pos;
    }
  }
}

__global__ __launch_bounds__(kBlock) void unique_index_kernel(const UArgs a) {
  const int tile = (int)blockIdx.x;
  const UCol& c = a.col[find_col(a, tile)];
  const int64_t base = (int64_t)(tile - c.tile_start) * kTile;
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const int64_t i = base + (int64_t)k * kBlock + threadIdx.x;
    if (i < c.len) c.index[i] = c.upos[c.first[c.slot_of[i]]];
  }
}

inline size_t align8(size_t v) { return (v + 7) & ~(size_t)7; }

inline uint64_t table_slots(int64_t len) {
  uint64_t h = 64;
  while (h < (uint64_t)len * 2) h <<= 1;
  return h;
}

struct Layout {
  size_t ff_bytes;    // section initialised to 0xFF: keys + first of every column
  size_t zero_bytes;  // section initialised to 0: cnt of every column
  size_t raw_bytes;   // slot_of, upos, tile_off
  size_t total() const { return ff_bytes + zero_bytes + raw_bytes; }
};

Layout layout_of(int32_t n_cols, const int64_t* lens) {
  Layout l = {0, 0, 0};
  for (int32_t c = 0; c < n_cols; ++c) {
    if (lens[c] <= 0) continue;
    const uint64_t h = table_slots(lens[c]);
    const int64_t tiles = (lens[c] + kTile - 1) / kTile;
    l.ff_bytes += (h + 1) * 8 + align8((h + 1) * 4);
    l.zero_bytes += align8((h + 1) * 4);
    l.raw_bytes += 2 * align8((size_t)lens[c] * 4) + align8((size_t)(tiles + 1) * 4);
  }
  return l;
}

}  // namespace

size_t unique_workspace_bytes(int32_t n_cols, const int64_t* lens) {
  if (n_cols <= 0 || lens == nullptr) return 0;
  return layout_of(n_cols, lens).total();
}

int unique_n_impl(int32_t n_cols, const UniqueColumn* cols, void* workspace,
                  size_t workspace_bytes, hipStream_t stream) {
  HBK_REQUIRE(n_cols >= 0, "unique_n: n_cols must be >= 0");
  if (n_cols == 0) return HBK_OK;
  HBK_REQUIRE(cols != nullptr, "unique_n: NULL argument array");
  int64_t* lens = (int64_t*)alloca(sizeof(int64_t) * (size_t)n_cols);
  for (int32_t c = 0; c < n_cols; ++c) {
    lens[c] = cols[c].len;
    HBK_REQUIRE(cols[c].len >= 0 && cols[c].len < (1ll << 30),
                "unique_n: input %d must have fewer than 2^30 elements, got %lld", c,
                (long long)cols[c].len);
    HBK_REQUIRE(cols[c].n_unique != nullptr, "unique_n: n_unique[%d] is NULL", c);
    HBK_REQUIRE(cols[c].len == 0 || (cols[c].in && cols[c].unique_out && cols[c].index_out),
                "unique_n: NULL buffer for input %d", c);
  }
  const Layout l = layout_of(n_cols, lens);
  HBK_REQUIRE(l.total() == 0 || (workspace != nullptr && workspace_bytes >= l.total()),
              "unique_n: workspace too small: need %zu bytes, got %zu", l.total(),
              workspace_bytes);
  HBK_REQUIRE(((uintptr_t)workspace & 7) == 0, "unique_n: workspace must be 8-byte aligned");
  char* ff = reinterpret_cast<char*>(workspace);
  char* zero = ff + l.ff_bytes;
  char* raw = zero + l.zero_bytes;
  if (l.ff_bytes) HBK_HIP_OK(hipMemsetAsync(ff, 0xff, l.ff_bytes, stream));
  if (l.zero_bytes) HBK_HIP_OK(hipMemsetAsync(zero, 0, l.zero_bytes, stream));

  int32_t c0 = 0;
  while (c0 < n_cols) {
    UArgs args;
    int32_t k = 0;
    int64_t tiles = 0;
    while (c0 < n_cols && k < kMaxCols) {
      const UniqueColumn& h = cols[c0++];
      if (h.len == 0) {
        HBK_HIP_OK(hipMemsetAsync(h.n_unique, 0, sizeof(int32_t), stream));
        continue;
      }
      UCol& d = args.col[k];
      const uint64_t slots = table_slots(h.len);
      const int64_t ctiles = (h.len + kTile - 1) / kTile;
      d.in = h.in;
      d.uniq = h.unique_out;
      d.index = h.index_out;
      d.n_unique = h.n_unique;
      d.mult = h.multiplicity;
      d.keys = reinterpret_cast<unsigned long long*>(ff);
      ff += (slots + 1) * 8;
      d.first = reinterpret_cast<uint32_t*>(ff);
      ff += align8((slots + 1) * 4);
      d.cnt = reinterpret_cast<int32_t*>(zero);
      zero += align8((slots + 1) * 4);
      d.slot_of = reinterpret_cast<uint32_t*>(raw);
      raw += align8((size_t)h.len * 4);
      d.upos = reinterpret_cast<int32_t*>(raw);
      raw += align8((size_t)h.len * 4);
      d.tile_off = reinterpret_cast<int32_t*>(raw);
      raw += align8((size_t)(ctiles + 1) * 4);
      d.len = (int32_t)h.len;
      d.hmask = (uint32_t)(slots - 1);
      d.tile_start = (int32_t)tiles;
      d.pad_ = 0;
      tiles += ctiles;
      HBK_REQUIRE(tiles < (1ll << 31), "unique_n: grid too large");
      ++k;
    }
    if (k == 0) continue;
    args.n_cols = k;
    args.pad_ = 0;
    const dim3 grid((unsigned)tiles), block(kBlock);
    hipLaunchKernelGGL(unique_insert_kernel, grid, block, 0, stream, args);
    hipLaunchKernelGGL(unique_count_kernel, grid, block, 0, stream, args);
    hipLaunchKernelGGL(unique_scan_kernel, dim3((unsigned)k), block, 0, stream, args);
    hipLaunchKernelGGL(unique_emit_kernel, grid, block, 0, stream, args);
    hipLaunchKernelGGL(unique_index_kernel, grid, block, 0, stream, args);
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
