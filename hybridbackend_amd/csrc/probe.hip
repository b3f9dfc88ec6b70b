// HbLookup cache probe for gfx950 (R11): wave64 redesign of the reference's 32-lane
// warp-cooperative slab probe (hbtf/embedding/lookup_functors.cu.cc:54-149, op
// hbtf/embedding/lookup_ops.cc:38-145).  The per-key outcome is what the reference computes:
//   slab = murmur3_hash32(key) % slab_count               (hybridbackend/common/murmur3.cu.h:32-77)
//   hit  = first slot of the slab holding the key  -> slab*slab_size + slot
//   miss = the slab holds an EMPTY (INT64_MIN, service.py:87) slot, or all slabs were probed
//   otherwise continue with slab+1 (wrapping).
// Mapping for CDNA4: a group of G = pow2(slab_size) adjacent lanes owns one key and reads one
// slab per step (slot = lane in group, coalesced 8*slab_size bytes); match / empty are found
// with one 64-bit ballot masked to the group, so a wave64 probes 64/G keys concurrently
// instead of one (the reference serialises its 32 keys through one slab read at a time).
#include "common.h"

namespace hbk {

__host__ __device__ inline uint32_t rotl32(uint32_t x, int r) {
  return (x << r) | (x >> (32 - r));
}

// murmur3_hash32<int64, seed 0>: two 4-byte blocks, no tail, len = 8.
__host__ __device__ inline uint32_t murmur3_hash32_i64(int64_t key) {
  const uint32_t c1 = 0xcc9e2d51u, c2 = 0x1b873593u;
  uint32_t h1 = 0;
  uint32_t blocks[2] = {(uint32_t)((uint64_t)key & 0xffffffffu),
                        (uint32_t)((uint64_t)key >> 32)};
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    uint32_t k1 = blocks[i];
    k1 *= c1;
    k1 = rotl32(k1, 15);
    k1 *= c2;
    h1 ^= k1;
    h1 = rotl32(h1, 13);
    h1 = h1 * 5 + 0xe6546b64u;
  }
  h1 ^= 8u;
  h1 ^= h1 >> 16;
  h1 *= 0x85ebca6bu;
  h1 ^= h1 >> 13;
  h1 *= 0xc2b2ae35u;
  h1 ^= h1 >> 16;
  return h1;
}

namespace {

constexpr int kBlock = 256;
constexpr long long kEmptyKey = (long long)0x8000000000000000ull;

__global__ __launch_bounds__(kBlock) void murmur3_kernel(const int64_t* keys, int64_t n,
                                                         uint32_t* out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) out[i] = murmur3_hash32_i64(keys[i]);
}

// A group of pow2(slab_size) lanes owns a key and reads one slab per probe; a wave takes kProbeKeys
// keys per group with ALL first probes in flight at once (one key per group and wave = one memory
// round trip per wave: 1.7 M keys took 160 us, bound by the rate waves start and one latency each).
// The rare key whose first slab neither holds it nor has an empty slot goes on slab by slab.
constexpr int kProbeKeys = 8;

__global__ __launch_bounds__(kBlock) void cache_probe_kernel(
    const int64_t* __restrict__ keys_cache, FastDiv slab_div, int32_t slab_size,
    int32_t group_log2, const int64_t* __restrict__ keys, int64_t n_keys,
    int64_t* __restrict__ hit_slot, int32_t* __restrict__ n_miss) {
  const int lane = lane_id();
  const int gsize = 1 << group_log2;
  const int sub = lane & (gsize - 1);
  const int grp = lane >> group_log2;
  const int groups_per_wave = kWave >> group_log2;
  const int64_t wave_global = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int64_t i0 = wave_global * groups_per_wave * kProbeKeys + grp;   // + u * groups_per_wave
  const unsigned long long group_mask =
      (gsize == 64 ? ~0ull : ((1ull << gsize) - 1ull)) << (grp << group_log2);
  const int64_t slab_count = (int64_t)slab_div.d;
  const bool in_slab = sub < slab_size;

  int64_t key[kProbeKeys], slab[kProbeKeys];
  long long read_key[kProbeKeys];
#pragma unroll
  for (int u = 0; u < kProbeKeys; ++u) {
    const int64_t i = i0 + (int64_t)u * groups_per_wave;
    key[u] = i < n_keys ? keys[i] : 0;
  }
#pragma unroll
  for (int u = 0; u < kProbeKeys; ++u) {
    const int64_t i = i0 + (int64_t)u * groups_per_wave;
    slab[u] = (int64_t)fastmod((uint64_t)murmur3_hash32_i64(key[u]), slab_div);
    read_key[u] = 0;
    if (i < n_keys && in_slab) read_key[u] = keys_cache[slab[u] * slab_size + sub];
  }
  int32_t missed = 0;
#pragma unroll
  for (int u = 0; u < kProbeKeys; ++u) {
    const int64_t i = i0 + (int64_t)u * groups_per_wave;
    bool active = i < n_keys;
    int64_t result = -1;
    int64_t probed = 0;
    long long rk = read_key[u];
    for (;;) {
      const bool live = active && in_slab;
      const unsigned long long match = __ballot(live && rk == key[u]) & group_mask;
      const unsigned long long empty = __ballot(live && rk == kEmptyKey) & group_mask;
      if (active) {
        if (match != 0ull) {
          result = slab[u] * slab_size + (__builtin_ctzll(match) - (grp << group_log2));
          active = false;
        } else if (empty != 0ull) {
          active = false;
        } else {
          ++probed;
          slab[u] = slab[u] + 1 == slab_count ? 0 : slab[u] + 1;
          if (probed >= slab_count) active = false;
        }
      }
      if (!__any(active)) break;
      rk = 0;
      if (active && in_slab) rk = keys_cache[slab[u] * slab_size + sub];
    }
    if (i < n_keys && sub == 0) {
      hit_slot[i] = result;
      missed += result < 0 ? 1 : 0;
    }
  }
  if (n_miss != nullptr) {
    // one atomic per wave: the lanes' miss counts summed across the wave
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) missed += __shfl_xor(missed, off, kWave);
    if (lane == 0 && missed != 0) atomicAdd(n_miss, missed);
  }
}

// ---- compaction of the per-key result into the op's four lists (stable: key order kept) -----
constexpr int kTileKeys = 1024;  // keys per 256-thread block

__global__ __launch_bounds__(kBlock) void probe_count_kernel(const int64_t* hit_slot, int64_t n,
                                                             int32_t* tile_hits) {
  __shared__ int32_t wave_hits[kBlock / kWave];
  const int lane = lane_id(), wave = (int)threadIdx.x >> 6;
  const int64_t base = (int64_t)blockIdx.x * kTileKeys;
  int32_t hits = 0;
#pragma unroll
  for (int k = 0; k < kTileKeys / kBlock; ++k) {
    const int64_t i = base + (int64_t)k * kBlock + threadIdx.x;
    hits += (int32_t)__builtin_popcountll(__ballot(i < n && hit_slot[i] >= 0));
  }
  if (lane == 0) wave_hits[wave] = hits;
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t t = 0;
    for (int w = 0; w < kBlock / kWave; ++w) t += wave_hits[w];
    tile_hits[blockIdx.x] = t;
  }
}

// one block: exclusive scan of the tile hit counts (in place) + the two totals
__global__ __launch_bounds__(kBlock) void probe_scan_kernel(int32_t* tile_hits, int64_t n_tiles,
                                                            int64_t n_keys, int32_t* counts) {
  __shared__ int32_t wave_tot[kBlock / kWave];
  __shared__ int32_t carry_s;
  const int tid = (int)threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int64_t e0 = 0; e0 < n_tiles; e0 += kBlock) {
    const int64_t e = e0 + tid;
    const int32_t x = e < n_tiles ? tile_hits[e] : 0;
    int32_t s = x;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int32_t y = __shfl_up(s, off, kWave);
      if (lane >= off) s += y;
    }
    if (lane == kWave - 1) wave_tot[wave] = s;
    __syncthreads();
    int32_t run = carry_s + s - x;
    for (int w = 0; w < wave; ++w) run += wave_tot[w];
    if (e < n_tiles) tile_hits[e] = run;
    __syncthreads();
    if (tid == kBlock - 1) carry_s = run + x;
    __syncthreads();
  }
  if (tid == 0) {
    counts[0] = carry_s;
    counts[1] = (int32_t)(n_keys - carry_s);
  }
}

__global__ __launch_bounds__(kBlock) void probe_emit_kernel(
    const int64_t* hit_slot, const int64_t* keys, int64_t n, const int32_t* tile_hits,
    int32_t* hit_keys_indices, int64_t* hit_cache_indices, int32_t* miss_keys_indices,
    int64_t* miss_keys) {
  __shared__ int32_t wave_hits[kTileKeys / kWave];
  const int tid = (int)threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  const int64_t base = (int64_t)blockIdx.x * kTileKeys;
  // chunk q of 64 keys = pass k, wave w: q = k * 4 + w (key order inside the tile)
  int64_t slot[kTileKeys / kBlock];
  unsigned long long mask[kTileKeys / kBlock];
#pragma unroll
  for (int k = 0; k < kTileKeys / kBlock; ++k) {
    const int64_t i = base + (int64_t)k * kBlock + tid;
    slot[k] = i < n ? hit_slot[i] : -1;
    mask[k] = __ballot(i < n && slot[k] >= 0);
    if (lane == 0) wave_hits[k * (kBlock / kWave) + wave] = (int32_t)__builtin_popcountll(mask[k]);
  }
  __syncthreads();
  const int32_t hit_base = tile_hits[blockIdx.x];
  const int32_t miss_base = (int32_t)(base - hit_base);
#pragma unroll
  for (int k = 0; k < kTileKeys / kBlock; ++k) {
    const int q = k * (kBlock / kWave) + wave;
    int32_t before = 0;
    for (int j = 0; j < q; ++j) before += wave_hits[j];
    const int64_t i = base + (int64_t)k * kBlock + tid;
    if (i >= n) continue;
    const int32_t in_chunk = rank_below(mask[k]);
    if (slot[k] >= 0) {
      const int32_t pos = hit_base + before + in_chunk;
      hit_keys_indices[pos] = (int32_t)i;
      hit_cache_indices[pos] = slot[k];
    } else {
      const int32_t pos = miss_base + (q * kWave - before) + (lane - in_chunk);
      miss_keys_indices[pos] = (int32_t)i;
      miss_keys[pos] = keys[i];
    }
  }
}

}  // namespace
}  // namespace hbk

extern "C" size_t hbk_cache_lookup_workspace_bytes(int64_t n_keys) {
  if (n_keys <= 0) return 0;
  const int64_t tiles = (n_keys + hbk::kTileKeys - 1) / hbk::kTileKeys;
  return (size_t)n_keys * 8 + (size_t)tiles * 4 + 16;
}

extern "C" int hbk_cache_lookup(const int64_t* keys_cache, int64_t slab_count,
                                int32_t slab_size, const int64_t* keys, int64_t n_keys,
                                int32_t* hit_keys_indices, int64_t* hit_cache_indices,
                                int32_t* miss_keys_indices, int64_t* miss_keys, int32_t* counts,
                                void* workspace, size_t workspace_bytes, hbk_stream_t stream) {
  using namespace hbk;
  HBK_REQUIRE(counts != nullptr, "cache_lookup: counts is NULL");
  HBK_REQUIRE(n_keys >= 0 && n_keys < (1ll << 31), "cache_lookup: n_keys out of range");
  if (n_keys == 0) {
    HBK_HIP_OK(hipMemsetAsync(counts, 0, 2 * sizeof(int32_t), as_stream(stream)));
    return HBK_OK;
  }
  HBK_REQUIRE(hit_keys_indices && hit_cache_indices && miss_keys_indices && miss_keys,
              "cache_lookup: NULL output");
  const size_t need = hbk_cache_lookup_workspace_bytes(n_keys);
  HBK_REQUIRE(workspace != nullptr && workspace_bytes >= need && ((uintptr_t)workspace & 7) == 0,
              "cache_lookup: workspace too small or misaligned: need %zu bytes", need);
  int64_t* hit_slot = reinterpret_cast<int64_t*>(workspace);
  int32_t* tile_hits = reinterpret_cast<int32_t*>(hit_slot + n_keys);
  int rc = hbk_cache_probe(keys_cache, slab_count, slab_size, keys, n_keys, hit_slot, nullptr,
                           stream);
  if (rc != HBK_OK) return rc;
  const int64_t tiles = (n_keys + kTileKeys - 1) / kTileKeys;
  hipLaunchKernelGGL(probe_count_kernel, dim3((unsigned)tiles), dim3(kBlock), 0,
                     as_stream(stream), hit_slot, n_keys, tile_hits);
  hipLaunchKernelGGL(probe_scan_kernel, dim3(1), dim3(kBlock), 0, as_stream(stream), tile_hits,
                     tiles, n_keys, counts);
  hipLaunchKernelGGL(probe_emit_kernel, dim3((unsigned)tiles), dim3(kBlock), 0,
                     as_stream(stream), hit_slot, keys, n_keys, tile_hits, hit_keys_indices,
                     hit_cache_indices, miss_keys_indices, miss_keys);
  HBK_HIP_OK(hipGetLastError());
  return HBK_OK;
}

extern "C" int hbk_murmur3_hash32(const int64_t* keys, int64_t n_keys, uint32_t* out,
                                  hbk_stream_t stream) {
  using namespace hbk;
  HBK_REQUIRE(n_keys >= 0, "murmur3_hash32: n_keys must be >= 0");
  if (n_keys == 0) return HBK_OK;
  HBK_REQUIRE(keys && out, "murmur3_hash32: NULL buffer");
  const int64_t blocks = (n_keys + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(murmur3_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, as_stream(stream),
                     keys, n_keys, out);
  HBK_HIP_OK(hipGetLastError());
  return HBK_OK;
}

extern "C" int hbk_cache_probe(const int64_t* keys_cache, int64_t slab_count,
                               int32_t slab_size, const int64_t* keys, int64_t n_keys,
                               int64_t* hit_slot, int32_t* n_miss, hbk_stream_t stream) {
  using namespace hbk;
  HBK_REQUIRE(slab_size >= 1 && slab_size <= kWave,
              "cache_probe: cache_slab_size must be in [1, 64], got %d", slab_size);
  HBK_REQUIRE(slab_count >= 1, "cache_probe: keys_cache must hold at least one slab");
  HBK_REQUIRE(n_keys >= 0, "cache_probe: n_keys must be >= 0");
  if (n_miss != nullptr) {
    HBK_HIP_OK(hipMemsetAsync(n_miss, 0, sizeof(int32_t), as_stream(stream)));
  }
  if (n_keys == 0) return HBK_OK;
  HBK_REQUIRE(keys_cache && keys && hit_slot, "cache_probe: NULL buffer");
  int group_log2 = 0;
  while ((1 << group_log2) < slab_size) ++group_log2;
  const int64_t keys_per_block = (int64_t)(kBlock >> group_log2) * kProbeKeys;
  const int64_t blocks = (n_keys + keys_per_block - 1) / keys_per_block;
  FastDiv sd = make_fastdiv((uint64_t)slab_count);
  sd.d = (uint64_t)slab_count;
  hipLaunchKernelGGL(cache_probe_kernel, dim3((unsigned)blocks), dim3(kBlock), 0,
                     as_stream(stream), keys_cache, sd, slab_size, group_log2, keys, n_keys,
                     hit_slot, n_miss);
  HBK_HIP_OK(hipGetLastError());
  return HBK_OK;
}
