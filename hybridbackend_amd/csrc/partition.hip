// HbPartitionByModulo[N] / HbPartitionByDualModuloStage{One,Two}[N] for gfx950 (R2, R3).
//
// Semantics = the reference CPU functor (a STABLE counting sort):
//   hbtf/distribute/partition/partition_by_modulo_functors.cc:39-70
//   hbtf/distribute/partition/partition_by_dual_modulo_functors.cc:37-91
// The reference CUDA kernels take their slot with atomicAdd and are therefore not
// order-stable (partition_by_modulo_functors.cu.cc:49-53); bit-exactness against the CPU
// path needs the stable form, built here from wavefront ballots and prefix sums:
//
//   A  tile histogram   one wave64 per 1024-id tile, all N columns in one launch
//                       (no max_len x N thread waste, cu.cc:139-149); LDS counters
//   B  scan             per column: exclusive scan of hist[p][tile] in (p, tile) order ->
//                       global start of every (shard, tile) run; sizes[p] falls out
//   C  stable scatter   same tiling; per 64-id chunk a match-any loop (ballot of lanes with
//                       the leader's shard) gives each id its rank inside the wave, the LDS
//                       running counter gives the base; chunks are taken in order, so
//                       input order is preserved inside every shard
//
// P <= 8 and columns of <= 256 tiles (the single-node step: 26 x 65536 ids over 8 ranks) take
// ONE launch instead of the three: a wave keeps its 1024 ids and their ranks inside the tile in
// registers, publishes the tile's counts, waits for the other tiles of its column to publish
// theirs (every wave of a column is resident: see partition_onepass_kernel), derives its bases
// from all of them and scatters from registers -- the ids are read once.
//
// Column descriptors travel in the kernel-argument segment: no pinned pointer tables and
// no H2D copies per call (cu.cc:283-306).  `%` never reaches the 64-bit software divide:
// shards come from a multiply-high with a host-computed magic (common.h).
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.h"

namespace hbk {
namespace {

constexpr int kChunks = 16;                   // 64-id chunks per tile
constexpr int kTile = kChunks * kWave;        // 1024 ids per wave
constexpr int kMaxColsPerLaunch = 128;
constexpr int kMaxPartitions = 16384;         // LDS counters: 64 KB

struct PartCol {
  const void* in;
  void* out;
  int32_t* sizes;
  int32_t* indices;
  int32_t len;
  int32_t tile_start;  // first tile of this column inside the launch group
  FastDiv bucket;      // d > 0: ids are bucketized (R1, floor-mod) on the fly; outputs hold the
                       // bucketized ids (the sharded driver fuses `% embedding_size` here)
  int32_t global_col;  // column index inside the whole call (for sizes_t)
  int32_t pad_;
};

struct ShardFn {
  FastDiv pre;    // dual: P*M; plain: unused (d == 0)
  FastDiv part;   // P
  FastDiv mod;    // dual stage 2: M
  int32_t stage;  // 0 plain modulo, 1 / 2 dual
  int32_t num_partitions;
};

struct PartArgs {
  int32_t n_cols;
  int32_t total_tiles;
  int32_t sub_tiles;     // 1024-id passes per wave: a tile is sub_tiles * 1024 ids
  int32_t fixed_max;     // P <= fixed_max: one ballot per shard (default 8; HBK_PART_FIXED tunes)
  int32_t* hist;  // [sum over columns of P * tiles_c]; column c starts at P * tile_start[c]
  int32_t* sizes_t;      // optional [P][n_total_cols] transposed copy of the sizes
  int32_t n_total_cols;
  int32_t uniform_tiles;  // > 0: every column of the launch has this many tiles (col = tile / it)
  ShardFn fn;
  int32_t tile0[kMaxColsPerLaunch];   // first tile of every column, packed (find_col reads 4 lines)
  PartCol col[kMaxColsPerLaunch];
};
static_assert(sizeof(PartArgs) <= 16384, "kernarg budget");

template <typename T>
__device__ inline T bucketize(T v, const FastDiv& b) {
  if (b.d == 0) return v;
  if constexpr (std::is_signed<T>::value) {
    return (T)floormod_i64((int64_t)v, b);
  } else {
    return (T)fastmod((uint64_t)v, b);
  }
}

template <typename T>
__device__ inline uint32_t shard_of(T v, const ShardFn& f) {
  uint64_t r;
  if (f.stage == 0) {
    if constexpr (std::is_signed<T>::value) {
      r = floormod_i64((int64_t)v, f.part);
    } else {
      r = fastmod((uint64_t)v, f.part);
    }
    return (uint32_t)r;
  }
  if constexpr (std::is_signed<T>::value) {
    r = floormod_i64((int64_t)v, f.pre);
  } else {
    r = fastmod((uint64_t)v, f.pre);
  }
  if (f.stage == 1) return (uint32_t)fastmod(r, f.part);
  return (uint32_t)fastdiv(r, f.mod);
}

__device__ inline int find_col(const PartArgs& a, int tile) {
  // every lane reads one descriptor's first tile (two independent loads cover 128 columns) and a
  // ballot counts those <= tile: one memory round trip instead of a binary search's 7 dependent
  // scalar loads, each a cold miss at the start of these short kernels
  if (a.uniform_tiles > 0) return tile / a.uniform_tiles;   // no memory round trip at all
  const int lane = lane_id();
  const int n = a.n_cols;
  const int t0 = lane < n ? a.tile0[lane] : 0x7fffffff;
  const int t1 = lane + kWave < n ? a.tile0[lane + kWave] : 0x7fffffff;
  const int ci = (int)__builtin_popcountll(__ballot(t0 <= tile)) +
                 (int)__builtin_popcountll(__ballot(t1 <= tile)) - 1;
  return __builtin_amdgcn_readfirstlane(ci);
}


// One ballot per shard.  With the shard count known at compile time the P compares, popcounts and
// selects of a chunk are independent straight-line code the scheduler can interleave; as a
// runtime loop every iteration waits for its own v_cmp -> s_bcnt1 round trip (measured: the loop
// form cost 26 of the 45 us of a 10 M-id histogram).
template <int NP>
__device__ inline int32_t count_shards_fixed(uint32_t shard, int lane) {
  int32_t add = 0;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int n = (int)__builtin_popcountll(__ballot(shard == (uint32_t)p));
    add = lane == p ? n : add;
  }
  return add;
}

__device__ inline int32_t count_shards(uint32_t shard, int P, int lane) {
  switch (P) {
    case 1: return count_shards_fixed<1>(shard, lane);
    case 2: return count_shards_fixed<2>(shard, lane);
    case 3: return count_shards_fixed<3>(shard, lane);
    case 4: return count_shards_fixed<4>(shard, lane);
    case 5: return count_shards_fixed<5>(shard, lane);
    case 6: return count_shards_fixed<6>(shard, lane);
    case 7: return count_shards_fixed<7>(shard, lane);
    case 8: return count_shards_fixed<8>(shard, lane);
    default: break;
  }
  int32_t add = 0;
  for (int p = 0; p < P; ++p) {
    const int n = (int)__builtin_popcountll(__ballot(shard == (uint32_t)p));
    add = lane == p ? n : add;
  }
  return add;
}

// position of this lane's id inside the output (stable): running base of its shard + rank among
// the lanes of the chunk with the same shard; lane p's `my_run` advances by the shard's count
template <int NP>
__device__ inline int32_t place_fixed(uint32_t shard, int lane, int32_t& my_run) {
  int32_t pos = 0, add = 0;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const unsigned long long same = __ballot(shard == (uint32_t)p);
    const int32_t base_p = __builtin_amdgcn_readlane(my_run, p);
    pos = shard == (uint32_t)p ? base_p + rank_below(same) : pos;
    add = lane == p ? (int32_t)__builtin_popcountll(same) : add;
  }
  my_run += add;
  return pos;
}

__device__ inline int32_t place(uint32_t shard, int P, int lane, int32_t& my_run) {
  switch (P) {
    case 1: return place_fixed<1>(shard, lane, my_run);
    case 2: return place_fixed<2>(shard, lane, my_run);
    case 3: return place_fixed<3>(shard, lane, my_run);
    case 4: return place_fixed<4>(shard, lane, my_run);
    case 5: return place_fixed<5>(shard, lane, my_run);
    case 6: return place_fixed<6>(shard, lane, my_run);
    case 7: return place_fixed<7>(shard, lane, my_run);
    case 8: return place_fixed<8>(shard, lane, my_run);
    default: break;
  }
  int32_t pos = 0, add = 0;
  for (int p = 0; p < P; ++p) {
    const unsigned long long same = __ballot(shard == (uint32_t)p);
    const int32_t base_p = __builtin_amdgcn_readlane(my_run, p);
    pos = shard == (uint32_t)p ? base_p + rank_below(same) : pos;
    add = lane == p ? (int32_t)__builtin_popcountll(same) : add;
  }
  my_run += add;
  return pos;
}

// Rank of a lane's id among the ids of its shard, P <= 8, from one ballot per BIT of the shard
// (<= 3) instead of one per shard: the lane builds the mask of its shard's lanes from the bit
// ballots taken plain or inverted; the counts of the chunks before live in four uniform words of
// two 16-bit fields each (scalar registers, advanced with scalar popcounts of the eight
// shard masks), so the vector unit sees ~25 instructions per chunk and no VALU -> SALU -> VALU
// round trip sits on its critical path.  (The one-ballot-per-shard form above measured 8.3 us of
// a wave's 16 us in the one-pass kernel.)
struct Run8 {
  uint32_t w[4];   // w[s >> 1] >> 16 (s & 1): ids of shard s in the chunks so far
};

__device__ inline int32_t place_bits(uint32_t shard, bool valid, Run8& run) {
  const unsigned long long all = __ballot(valid);
  const unsigned long long b0 = __ballot(valid && (shard & 1u) != 0);
  const unsigned long long b1 = __ballot(valid && (shard & 2u) != 0);
  const unsigned long long b2 = __ballot(valid && (shard & 4u) != 0);
  // lanes of my shard = lanes whose three bits all equal mine: per 32-bit half,
  // all & ~((b0 ^ m0) | (b1 ^ m1) | (b2 ^ m2)) with m_i = bit i of my shard spread over the word
  const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int)shard, 0, 1);
  const uint32_t m1 = (uint32_t)__builtin_amdgcn_sbfe((int)shard, 1, 1);
  const uint32_t m2 = (uint32_t)__builtin_amdgcn_sbfe((int)shard, 2, 1);
  const uint32_t lo = (uint32_t)all & ~(((uint32_t)b0 ^ m0) | ((uint32_t)b1 ^ m1) | ((uint32_t)b2 ^ m2));
  const uint32_t hi = (uint32_t)(all >> 32) &
                      ~(((uint32_t)(b0 >> 32) ^ m0) | ((uint32_t)(b1 >> 32) ^ m1) | ((uint32_t)(b2 >> 32) ^ m2));
  const int32_t below = (int32_t)__builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
  uint32_t w = (shard & 2u) ? run.w[1] : run.w[0];
  w = (shard & 4u) ? ((shard & 2u) ? run.w[3] : run.w[2]) : w;
  const int32_t before = (int32_t)((w >> ((shard & 1u) * 16u)) & 0xffffu);
  // advance the uniform counts: popcounts of the shard masks (all scalar)
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    const unsigned long long m = all & ((h & 1) ? b1 : ~b1) & ((h & 2) ? b2 : ~b2);
    run.w[h] += (uint32_t)__builtin_popcountll(m & ~b0) |
                ((uint32_t)__builtin_popcountll(m & b0) << 16);
  }
  return before + below;
}

// the counts alone (histogram kernel): four ballots and scalar popcounts per chunk
__device__ inline void count_bits(uint32_t shard, bool valid, Run8& run) {
  const unsigned long long all = __ballot(valid);
  const unsigned long long b0 = __ballot(valid && (shard & 1u) != 0);
  const unsigned long long b1 = __ballot(valid && (shard & 2u) != 0);
  const unsigned long long b2 = __ballot(valid && (shard & 4u) != 0);
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    const unsigned long long m = all & ((h & 1) ? b1 : ~b1) & ((h & 2) ? b2 : ~b2);
    run.w[h] += (uint32_t)__builtin_popcountll(m & ~b0) |
                ((uint32_t)__builtin_popcountll(m & b0) << 16);
  }
}

// lane p's view of the uniform counts: the count of shard p (lanes >= 8: 0)
__device__ inline int32_t run_of_lane(const Run8& run, int lane) {
  uint32_t w = (lane & 2) ? run.w[1] : run.w[0];
  w = (lane & 4) ? ((lane & 2) ? run.w[3] : run.w[2]) : w;
  return lane < 8 ? (int32_t)((w >> ((lane & 1) * 16)) & 0xffffu) : 0;
}

// 8 < P <= 64: the lanes holding the same shard are found from one ballot per BIT of the shard
// (<= 6) instead of one loop iteration per distinct shard in the chunk (up to 64, each waiting
// for the previous one): every lane ANDs the ballots, taken plain or inverted by its own bits,
// into the mask of its shard's lanes -- and a second time with the bits of its LANE number into
// the mask of shard `lane`, whose running counter it keeps.
template <int NBITS>
__device__ inline void shard_masks(uint32_t shard, int lane, unsigned long long& mine,
                                   unsigned long long& for_lane) {
  mine = for_lane = __ballot(shard != 0xffffffffu);
#pragma unroll
  for (int b = 0; b < NBITS; ++b) {
    const unsigned long long bal = __ballot(((shard >> b) & 1u) != 0);
    mine &= ((shard >> b) & 1u) ? bal : ~bal;
    for_lane &= ((lane >> b) & 1) ? bal : ~bal;
  }
}

__device__ inline void shard_masks_n(uint32_t shard, int nbits, int lane, unsigned long long& mine,
                                     unsigned long long& for_lane) {
  switch (nbits) {
    case 1: shard_masks<1>(shard, lane, mine, for_lane); break;
    case 2: shard_masks<2>(shard, lane, mine, for_lane); break;
    case 3: shard_masks<3>(shard, lane, mine, for_lane); break;
    case 4: shard_masks<4>(shard, lane, mine, for_lane); break;
    case 5: shard_masks<5>(shard, lane, mine, for_lane); break;
    default: shard_masks<6>(shard, lane, mine, for_lane); break;
  }
}

// ---- A: per-tile histogram ------------------------------------------------------
// (hist and scatter: kTileWaves independent waves per workgroup while the counters fit registers,
// P <= 64 -- the dispatcher starts ~300 workgroups per us whatever their size, and 26 x 1 M ids
// are 26 624 tiles; one wave per workgroup when the counters need LDS)
constexpr int kTileWaves = 4;

template <typename T>
__global__ __launch_bounds__(kTileWaves* kWave) void partition_hist_kernel(const PartArgs a) {
  extern __shared__ int32_t counters[];
  const int tile = (int)blockIdx.x * (int)(blockDim.x >> 6) +
                   __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (tile >= a.total_tiles) return;
  const int ci = find_col(a, tile);
  const PartCol& c = a.col[ci];
  const int P = a.fn.num_partitions;
  const int lane = lane_id();
  const int ctile = tile - c.tile_start;
  const int tile_ids = a.sub_tiles * kTile;
  const int n_tiles = (c.len + tile_ids - 1) / tile_ids;
  const T* in = reinterpret_cast<const T*>(c.in);
  const int64_t len = c.len;
  // uniform descriptor fields live in registers: re-reading them from the kernel-argument
  // segment inside the unrolled chunk loops costs a scalar-memory round trip each time
  const ShardFn fn = a.fn;
  const FastDiv bk = c.bucket;
  const bool small_p = P <= kWave;
  int nbits = 1;
  while ((1 << nbits) < P) ++nbits;
  const int fixed_max = a.fixed_max;
  int32_t cnt = 0;  // P <= 64: lane p counts the ids of shard p in this tile
  unsigned long long lane_cnt = 0;   // P <= 8: this lane's ids per shard, eight 8-bit fields
  const bool bits8 = P <= 8;
  const bool pow2_plain = fn.stage == 0 && fn.part.kind == 1 && bk.d == 0;
  const uint32_t pow2_mask = (uint32_t)fn.part.d - 1u;
  if (!small_p) {
    for (int p = lane; p < P; p += kWave) counters[p] = 0;
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): zeroing done before the atomics
  }
  for (int sb = 0; sb < a.sub_tiles; ++sb) {
    const int64_t base = (int64_t)ctile * tile_ids + (int64_t)sb * kTile;
    if (base >= len) break;
    T v[kChunks];
#pragma unroll
    for (int k = 0; k < kChunks; ++k) {
      const int64_t i = base + k * kWave + lane;
      v[k] = i < len ? in[i] : T(0);
    }
#pragma unroll
    for (int k = 0; k < kChunks; ++k) {
      const int64_t i = base + k * kWave + lane;
      const uint32_t shard =
          !(i < len) ? 0xffffffffu
                     : pow2_plain ? (uint32_t)v[k] & pow2_mask   // floor-mod by a power of two
                                  : shard_of<T>(bucketize<T>(v[k], bk), fn);
      if (bits8) {
        // every lane counts ITS ids in eight 8-bit fields of one 64-bit word: no cross-lane work
        // per id (the ballots + scalar popcounts per chunk were most of this kernel: 25 us for the
        // 10 M ids of the reference's benchmark shape, whose 40 MB stream in 8)
        if (i < len) lane_cnt += 1ull << (shard * 8u);
      } else if (P <= fixed_max) {
        cnt += count_shards(shard, P, lane);
      } else if (small_p) {
        unsigned long long mine, for_lane;
        shard_masks_n(shard, nbits, lane, mine, for_lane);
        cnt += (int32_t)__builtin_popcountll(for_lane);
      } else if (i < len) {
        atomicAdd(&counters[shard], 1);
      }
    }
  }
  int32_t* hist = a.hist + (int64_t)P * c.tile_start;
  if (bits8) {
    // wave sum of the lanes' counters: bytes spread to 16-bit fields (<= 64 lanes x 128 ids), one
    // butterfly over four dwords, lane p keeps the field of shard p
    auto spread = [](uint32_t x) -> unsigned long long {
      return (unsigned long long)(x & 0xffu) | ((unsigned long long)(x & 0xff00u) << 8) |
             ((unsigned long long)(x & 0xff0000u) << 16) | ((unsigned long long)(x & 0xff000000u) << 24);
    };
    unsigned long long lo = spread((uint32_t)lane_cnt), hi = spread((uint32_t)(lane_cnt >> 32));
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      lo += ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(lo >> 32), off, kWave) << 32) |
            (uint32_t)__shfl_xor((int)(uint32_t)lo, off, kWave);
      hi += ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(hi >> 32), off, kWave) << 32) |
            (uint32_t)__shfl_xor((int)(uint32_t)hi, off, kWave);
    }
    const unsigned long long w = (lane & 4) ? hi : lo;
    cnt = lane < 8 ? (int32_t)((w >> ((lane & 3) * 16)) & 0xffffu) : 0;
  }
  if (small_p) {
    if (lane < P) hist[(int64_t)lane * n_tiles + ctile] = cnt;
    return;
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  for (int p = lane; p < P; p += kWave) hist[(int64_t)p * n_tiles + ctile] = counters[p];
}

// ---- B: per-column exclusive scan of hist in (shard, tile) order ------------------
constexpr int kScanBlock = 256;

__global__ __launch_bounds__(kScanBlock) void partition_scan_kernel(const PartArgs a) {
  __shared__ int32_t wave_tot[kScanBlock / kWave];
  __shared__ int32_t carry_s;
  const PartCol& c = a.col[blockIdx.x];
  const int P = a.fn.num_partitions;
  const int tile_ids = a.sub_tiles * kTile;
  const int n_tiles = (c.len + tile_ids - 1) / tile_ids;
  const int64_t total = (int64_t)P * n_tiles;
  int32_t* hist = a.hist + (int64_t)P * c.tile_start;
  const int tid = (int)threadIdx.x;
  const int lane = tid & (kWave - 1), wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int64_t e0 = 0; e0 < total; e0 += kScanBlock) {
    const int64_t e = e0 + tid;
    const int32_t x = e < total ? hist[e] : 0;
    // inclusive scan inside the wave
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
    const int32_t carry = carry_s;
    const int32_t excl = carry + wbase + s - x;
    if (e < total) {
      hist[e] = excl;
      // first tile of shard p: its exclusive offset is the shard's start
      if (n_tiles > 0 && e % n_tiles == 0) {
        const int p = (int)(e / n_tiles);
        // sizes[p-1] = start[p] - start[p-1] is finished below once all starts are known;
        // stash the start in sizes[p] for now
        c.sizes[p] = excl;
      }
    }
    __syncthreads();
    if (tid == kScanBlock - 1) carry_s = excl + x;
    __syncthreads();
  }
  // sizes[p] currently holds start[p]; turn into counts: start[p+1] - start[p], start[P] = len
  __syncthreads();
  for (int p0 = 0; p0 < P; p0 += kScanBlock) {
    const int p = p0 + tid;
    int32_t cnt = 0;
    if (p < P && n_tiles > 0) {
      const int32_t st = c.sizes[p];
      const int32_t nx = (p + 1 < P) ? c.sizes[p + 1] : c.len;
      cnt = nx - st;
    }
    __syncthreads();
    if (p < P) {
      c.sizes[p] = cnt;
      if (a.sizes_t != nullptr) a.sizes_t[(int64_t)p * a.n_total_cols + c.global_col] = cnt;
    }
    __syncthreads();
  }
}

// ---- C: stable scatter --------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kTileWaves* kWave) void partition_scatter_kernel(const PartArgs a) {
  extern __shared__ int32_t run_lds[];
  volatile int32_t* run = run_lds;  // cross-lane hand-off inside one wave: keep every access
  const int tile = (int)blockIdx.x * (int)(blockDim.x >> 6) +
                   __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (tile >= a.total_tiles) return;
  const int ci = find_col(a, tile);
  const PartCol& c = a.col[ci];
  const int P = a.fn.num_partitions;
  const int lane = lane_id();
  const int ctile = tile - c.tile_start;
  const int tile_ids = a.sub_tiles * kTile;
  const int n_tiles = (c.len + tile_ids - 1) / tile_ids;
  const int32_t* hist = a.hist + (int64_t)P * c.tile_start;
  const T* in = reinterpret_cast<const T*>(c.in);
  T* out = reinterpret_cast<T*>(c.out);
  int32_t* indices = c.indices;
  const int64_t len = c.len;
  const ShardFn fn = a.fn;     // uniform descriptor fields in registers (see the histogram)
  const FastDiv bk = c.bucket;
  // running position of every shard: lane p's register for P <= 64, LDS beyond; it carries over
  // the 1024-id passes of the tile, which are taken in order (stability)
  int32_t my_run = 0;
  Run8 run8;   // P <= 8: ids of every shard in the chunks of this tile so far (uniform)
  run8.w[0] = run8.w[1] = run8.w[2] = run8.w[3] = 0u;
  int nbits = 1;
  while ((1 << nbits) < P) ++nbits;
  const int fixed_max = a.fixed_max;
  if (P <= kWave) {
    my_run = lane < P ? hist[(int64_t)lane * n_tiles + ctile] : 0;
  } else {
    for (int p = lane; p < P; p += kWave) run[p] = hist[(int64_t)p * n_tiles + ctile];
    __builtin_amdgcn_s_waitcnt(0xc07f);  // run[] initialised
  }
  for (int sb = 0; sb < a.sub_tiles; ++sb) {
    const int64_t base = (int64_t)ctile * tile_ids + (int64_t)sb * kTile;
    if (base >= len) break;
    T v[kChunks];
#pragma unroll
    for (int k = 0; k < kChunks; ++k) {
      const int64_t i = base + k * kWave + lane;
      v[k] = i < len ? bucketize<T>(in[i], bk) : T(0);
    }
    if (P <= 8) {
      // the single-node case: one ballot per bit of the shard, counts in scalar registers
#pragma unroll
      for (int k = 0; k < kChunks; ++k) {
        const int64_t i = base + k * kWave + lane;
        const bool valid = i < len;
        const uint32_t shard = valid ? shard_of<T>(v[k], fn) : 0u;
        const int32_t pos = __shfl(my_run, (int)shard, kWave) + place_bits(shard, valid, run8);
        if (valid) {
          out[pos] = v[k];
          indices[i] = pos;
        }
      }
    } else if (P <= fixed_max) {
      // one ballot per shard, all compares of a chunk independent
#pragma unroll
      for (int k = 0; k < kChunks; ++k) {
        const int64_t i = base + k * kWave + lane;
        const bool valid = i < len;
        const uint32_t shard = valid ? shard_of<T>(v[k], fn) : 0xffffffffu;
        const int32_t pos = place(shard, P, lane, my_run);
        if (valid) {
          out[pos] = v[k];
          indices[i] = pos;
        }
      }
    } else if (P <= kWave) {
      // one ballot per bit of the shard; the base comes from lane `shard` by ds_bpermute
#pragma unroll
      for (int k = 0; k < kChunks; ++k) {
        const int64_t i = base + k * kWave + lane;
        const bool valid = i < len;
        const uint32_t shard = valid ? shard_of<T>(v[k], fn) : 0xffffffffu;
        unsigned long long mine, for_lane;
        shard_masks_n(shard, nbits, lane, mine, for_lane);
        const int32_t pos = __shfl(my_run, (int)(shard & 63u), kWave) + rank_below(mine);
        my_run += (int32_t)__builtin_popcountll(for_lane);
        if (valid) {
          out[pos] = v[k];
          indices[i] = pos;
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < kChunks; ++k) {
        const int64_t i = base + k * kWave + lane;
        const bool valid = i < len;
        const uint32_t shard = valid ? shard_of<T>(v[k], fn) : 0xffffffffu;
        unsigned long long todo = __ballot(valid);
        int32_t pos = 0;
        while (todo != 0ull) {
          const int leader = __builtin_ctzll(todo);
          const uint32_t s = (uint32_t)__shfl((int)shard, leader, kWave);
          const unsigned long long same = __ballot(shard == s);
          const int32_t base_s = run[s];  // every lane reads the same address: LDS broadcast
          if (shard == s) pos = base_s + rank_below(same);
          __builtin_amdgcn_s_waitcnt(0xc07f);
          if (lane == leader) run[s] = base_s + (int32_t)__builtin_popcountll(same);
          todo &= ~same;
        }
        if (valid) {
          out[pos] = v[k];
          indices[i] = pos;
        }
      }
    }
  }
}


// ---- A+B+C in one launch (P <= 8, <= kOneMaxTiles tiles per column) ----------------------------
// Every wave publishes hist[p][tile] = count + 1 (0 = not there yet; the words are zero when the
// kernel starts) and polls the words of ITS COLUMN until all are there: lane t reads tile t's
// word of every shard, the totals and the prefix over the earlier tiles are two sums per shard.
// A wave waits for tiles launched AFTER it, so those must find a slot while it spins: workgroups
// are dispatched in index order, hence when the dispatcher is stalled all resident waves of this
// kernel belong to the one column that is not fully dispatched -- at most kOneMaxTiles of the
// ~5000 one-wave slots of the chip (8 such kernels side by side still fit).  The wait is bounded
// all the same: after kSyncWaitTicks of the 100 MHz clock a wave raises *status (host memory),
// skips its stores, and the next entry call reports HBK_INTERNAL instead of the box hanging.
// Probe builds only (-DHBK_PART_STAMPS, tools/Makefile): constant-clock stamps of the one-pass
// kernel's waves, read back by hbk_debug_part_trace().
#ifdef HBK_PART_STAMPS
constexpr int kPTraceBlocks = 8192, kPTraceSlots = 8;
__device__ unsigned long long g_part_trace[kPTraceBlocks * kPTraceSlots];
#define HBK_PSTAMP(i)                                                                          \
  do {                                                                                         \
    const unsigned tr_ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                  \
    if ((threadIdx.x & 63) == 0 && tr_ < kPTraceBlocks) {                                      \
      g_part_trace[tr_ * kPTraceSlots + (i)] = __builtin_amdgcn_s_memrealtime();               \
    }                                                                                          \
  } while (0)
#else
#define HBK_PSTAMP(i)
#endif
constexpr int kOneMaxTiles = 256;
constexpr int kOneMaxGrid = 3072;   // tiles of one call
constexpr int kOneMaxP = 8;

struct OnePass {
  int32_t* zero;        // words the call before this one left set (the other half), or NULL
  int64_t zero_words;
  SyncWait wait;        // bound of the waits, status / poison words, test hook
};

constexpr int kOneWaves = 4;   // waves per workgroup, each on a tile of its own (nothing shared:
                               // the dispatcher starts ~350 workgroups per us whatever their size)

// one wave: its LDS instructions run in order; keep the compiler from moving them across
__device__ inline void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <typename T>
__global__ __launch_bounds__(kOneWaves* kWave) void partition_onepass_kernel(const PartArgs a,
                                                                             const OnePass o) {
  __shared__ int32_t xs_all[kOneWaves][kOneMaxP * kWave];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  int32_t* xs = xs_all[wave];
  const int tile = (int)blockIdx.x * kOneWaves + wave;
  const int lane = lane_id();
  HBK_PSTAMP(0);
  if (tile >= a.total_tiles) return;
  for (int64_t j = (int64_t)tile * kWave + lane; j < o.zero_words; j += (int64_t)a.total_tiles * kWave) {
    o.zero[j] = 0;
  }
  const int ci = find_col(a, tile);
  const PartCol& c = a.col[ci];
  const int P = a.fn.num_partitions;
  const int ctile = tile - c.tile_start;
  const int n_tiles = (c.len + kTile - 1) / kTile;
  const T* in = reinterpret_cast<const T*>(c.in);
  const int64_t len = c.len;
  const ShardFn fn = a.fn;
  const FastDiv bk = c.bucket;
  int32_t* hist = a.hist + (int64_t)P * c.tile_start;
  const int64_t base = (int64_t)ctile * kTile;

  T v[kChunks];
#pragma unroll
  for (int k = 0; k < kChunks; ++k) {
    const int64_t i = base + k * kWave + lane;
    v[k] = i < len ? in[i] : T(0);
  }
  if (tile == 0) {
    // columns without ids have no tile: their sizes are written here
    for (int e = 0; e < a.n_cols; ++e) {
      const PartCol& z = a.col[e];
      if (z.len == 0 && lane < P) {
        z.sizes[lane] = 0;
        if (a.sizes_t != nullptr) a.sizes_t[(int64_t)lane * a.n_total_cols + z.global_col] = 0;
      }
    }
  }
  HBK_PSTAMP(1);   // descriptor known, loads issued
#ifdef HBK_PART_STAMPS
  __builtin_amdgcn_s_waitcnt(0x0f70);              // vmcnt(0)
#endif
  HBK_PSTAMP(2);   // ids here
  // rank of every id among the ids of its shard inside the tile; lane p counts shard p
  int32_t pk[kChunks];
  // (1) shards: the single-node case -- plain modulo by a power of two, nothing to bucketize --
  // is a mask; the general case takes the multiply-high forms
  if (fn.stage == 0 && fn.part.kind == 1 && bk.d == 0) {
    const uint32_t mask = (uint32_t)fn.part.d - 1u;
#pragma unroll
    for (int k = 0; k < kChunks; ++k) pk[k] = (int32_t)((uint32_t)v[k] & mask);
  } else {
#pragma unroll
    for (int k = 0; k < kChunks; ++k) {
      v[k] = bucketize<T>(v[k], bk);
      pk[k] = (int32_t)shard_of<T>(v[k], fn);
    }
  }
  // (2) ranks inside the tile
  Run8 run;
  run.w[0] = run.w[1] = run.w[2] = run.w[3] = 0u;
#pragma unroll
  for (int k = 0; k < kChunks; ++k) {
    const bool valid = base + k * kWave + lane < len;
    const uint32_t shard = valid ? (uint32_t)pk[k] : 0u;
    const int32_t r = place_bits(shard, valid, run);
    pk[k] = valid ? (r | (int32_t)(shard << 16)) : -1;
  }
  // lane p holds the tile's count of shard p
  const int32_t my_run = run_of_lane(run, lane);
  HBK_PSTAMP(3);   // ranks done
  if (lane < P && tile != o.wait.withhold) {
    __hip_atomic_store(hist + (int64_t)lane * n_tiles + ctile, my_run + 1, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  }
  // wait for the column, 64 tiles at a time: lane p ends with the total of shard p and its count
  // in the tiles before this one
  int32_t tot = 0, pre = 0;
  const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
  for (int g0 = 0; g0 < n_tiles; g0 += kWave) {
    const int t = g0 + lane;
    int32_t x[kOneMaxP];
    for (;;) {
      bool ok = true;
#pragma unroll
      for (int p = 0; p < kOneMaxP; ++p) {
        x[p] = 1;
        if (p < P && t < n_tiles) {
          x[p] = __hip_atomic_load(hist + (int64_t)p * n_tiles + t, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        ok = ok && x[p] != 0;
      }
      if (__ballot(!ok) == 0ull) break;
      if (__builtin_amdgcn_s_memrealtime() - t_begin > o.wait.ticks) {
        if (lane == 0) give_up(o.wait);
        return;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    wave_sync();
#pragma unroll
    for (int p = 0; p < kOneMaxP; ++p) xs[p * kWave + lane] = x[p] - 1;
    wave_sync();
    // lane (p, j) = 8 p + j adds tiles 8 j .. 8 j + 7 of shard p, three butterfly steps finish
    const int sp = lane >> 3, sj = lane & 7;
    int32_t s = 0, q = 0;
#pragma unroll
    for (int i2 = 0; i2 < 8; ++i2) {
      const int tt = sj * 8 + i2;
      const int32_t y = xs[sp * kWave + tt];
      s += y;
      q += g0 + tt < ctile ? y : 0;
    }
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
      s += __shfl_xor(s, off, kWave);
      q += __shfl_xor(q, off, kWave);
    }
    const int32_t s_p = __shfl(s, (lane & 7) * 8, kWave);
    const int32_t q_p = __shfl(q, (lane & 7) * 8, kWave);
    if (lane < kOneMaxP) {
      tot += s_p;
      pre += q_p;
    }
  }
  HBK_PSTAMP(4);   // the column's counts are in
  // start of shard p = total of the shards below it
  int32_t inc = tot;
#pragma unroll
  for (int off = 1; off < kOneMaxP; off <<= 1) {
    const int32_t y = __shfl_up(inc, off, kWave);
    if (lane >= off) inc += y;
  }
  const int32_t my_base = inc - tot + pre;
  if (ctile == 0 && lane < P) {
    c.sizes[lane] = tot;
    if (a.sizes_t != nullptr) a.sizes_t[(int64_t)lane * a.n_total_cols + c.global_col] = tot;
  }
  T* out = reinterpret_cast<T*>(c.out);
  int32_t* indices = c.indices;
#pragma unroll
  for (int k = 0; k < kChunks; ++k) {
    const int64_t i = base + k * kWave + lane;
    const bool valid = pk[k] >= 0;
    const int32_t pos = __shfl(my_base, valid ? pk[k] >> 16 : 0, kWave) + (pk[k] & 0xffff);
    if (valid) {
      out[pos] = v[k];
      indices[i] = pos;
    }
  }
  HBK_PSTAMP(5);   // stores issued
#ifdef HBK_PART_STAMPS
  __builtin_amdgcn_s_waitcnt(0x0f70);
  HBK_PSTAMP(6);   // stores done
  HBK_PSTAMP(7);
#endif
}

template <typename T>
int launch_onepass(const PartArgs& args, const OnePass& o, hipStream_t stream) {
  SyncChain chain(stream);   // never beside another kernel whose tiles wait for later tiles
  hipLaunchKernelGGL(partition_onepass_kernel<T>,
                     dim3((unsigned)((args.total_tiles + kOneWaves - 1) / kOneWaves)),
                     dim3(kOneWaves * kWave), 0, stream, args, o);
  HBK_HIP_OK(hipGetLastError());
  return HBK_OK;
}

template <typename T>
int launch_group(const PartArgs& args, int P, hipStream_t stream) {
  const size_t lds = P > kWave ? (size_t)P * sizeof(int32_t) : 0;
  const int wpb = P > kWave ? 1 : kTileWaves;   // LDS counters belong to one wave
  const unsigned blocks = (unsigned)((args.total_tiles + wpb - 1) / wpb);
  if (args.total_tiles > 0) {
    hipLaunchKernelGGL(partition_hist_kernel<T>, dim3(blocks), dim3(wpb * kWave), lds, stream, args);
    HBK_HIP_OK(hipGetLastError());
  }
  hipLaunchKernelGGL(partition_scan_kernel, dim3((unsigned)args.n_cols), dim3(kScanBlock), 0,
                     stream, args);
  HBK_HIP_OK(hipGetLastError());
  if (args.total_tiles > 0) {
    hipLaunchKernelGGL(partition_scatter_kernel<T>, dim3(blocks), dim3(wpb * kWave), lds, stream,
                       args);
    HBK_HIP_OK(hipGetLastError());
  }
  return HBK_OK;
}

// A wave may take several 1024-id passes (sub_tiles); measured slower than one pass per wave at
// every size (10 M ids: 69 us with 1 pass, 89 us with 4; 1.7 M ids: 33 vs 73 us), so the default
// is 1 and the option partition_sub_tiles is only a tuning knob.
int sub_tiles_of(int32_t n_cols, const int64_t* lens) {
  (void)n_cols;
  (void)lens;
  int64_t sub = 1;
  if (options().partition_sub_tiles >= 1) sub = options().partition_sub_tiles;
  return sub > 8 ? 8 : (int)sub;
}

int64_t tiles_of(int64_t len, int sub) {
  const int64_t tile_ids = (int64_t)kTile * sub;
  return (len + tile_ids - 1) / tile_ids;
}

int partition_impl(const char* what, int32_t n_cols, int32_t dtype, int32_t P,
                   int32_t modulus, int32_t stage, const void* const* inputs,
                   const int64_t* lens, void* const* outputs, int32_t* const* sizes,
                   int32_t* const* indices, void* workspace, size_t workspace_bytes,
                   hipStream_t stream, const int64_t* buckets = nullptr,
                   int32_t* sizes_t = nullptr) {
  HBK_REQUIRE(n_cols >= 0, "%s: n_cols must be >= 0", what);
  HBK_REQUIRE(P >= 1, "%s: num_partitions must be >= 1, got %d", what, P);
  HBK_REQUIRE(P <= kMaxPartitions, "%s: num_partitions %d > %d unsupported", what, P,
              kMaxPartitions);
  HBK_REQUIRE(dtype == HBK_INT32 || dtype == HBK_INT64 || dtype == HBK_UINT32 ||
                  dtype == HBK_UINT64,
              "%s: T must be one of int32, int64, uint32, uint64", what);
  if (stage != 0) {
    HBK_REQUIRE(stage == 1 || stage == 2, "%s: stage must be 1 or 2", what);
    HBK_REQUIRE(modulus >= 1, "%s: modulus must be >= 1, got %d", what, modulus);
    HBK_REQUIRE((int64_t)P * modulus < (1ll << 31), "%s: num_partitions * modulus overflows",
                what);
  }
  if (n_cols == 0) return HBK_OK;
  HBK_REQUIRE(inputs && lens && outputs && sizes && indices, "%s: NULL argument array", what);
  size_t need = hbk_partition_workspace_bytes(n_cols, lens, P);
  HBK_REQUIRE(need == 0 || (workspace != nullptr && workspace_bytes >= need),
              "%s: workspace too small: need %zu bytes, got %zu", what, need, workspace_bytes);
  for (int32_t c = 0; c < n_cols; ++c) {
    HBK_REQUIRE(lens[c] >= 0 && lens[c] < (1ll << 31),
                "%s: input %d must have fewer than 2^31 elements (rank-1 int32 indices), got %lld",
                what, c, (long long)lens[c]);
    HBK_REQUIRE(sizes[c] != nullptr, "%s: sizes[%d] is NULL", what, c);
    HBK_REQUIRE(lens[c] == 0 || (inputs[c] && outputs[c] && indices[c]),
                "%s: NULL buffer for input %d", what, c);
  }

  {
    const int rc = sync_check(what, stream);
    if (rc != HBK_OK) return rc;
  }
  ShardFn fn;
  fn.stage = stage;
  fn.num_partitions = P;
  fn.part = make_fastdiv((uint64_t)P);
  fn.pre = make_fastdiv(stage ? (uint64_t)P * (uint64_t)modulus : 1);
  fn.mod = make_fastdiv(stage ? (uint64_t)modulus : 1);

  int32_t* hist = reinterpret_cast<int32_t*>(workspace);
  const int sub = sub_tiles_of(n_cols, lens);
  // one launch when every column fits the one-pass kernel (see there); its words must read zero
  int64_t all_tiles = 0;
  bool onepass = options().partition_onepass != 0 && sub == 1 && P <= kOneMaxP &&
                 P <= options().partition_fixed_max;
  for (int32_t c = 0; c < n_cols; ++c) {
    const int64_t t = tiles_of(lens[c], sub);
    onepass = onepass && t <= kOneMaxTiles;
    all_tiles += t;
  }
  // (beyond what is resident at once the waves wait for slots as much as for counts: the
  // reference's benchmark shape, 9766 tiles, takes 68 us in one launch and 56 us in three)
  onepass = onepass && all_tiles > 0 && all_tiles <= kOneMaxGrid;
  OnePass one;
  one.zero = nullptr;
  one.zero_words = 0;
  memset(&one.wait, 0, sizeof(one.wait));
  if (onepass) {
    const size_t words = (size_t)all_tiles * (size_t)P;
    SyncTake take;
    const void* fn_ptr = nullptr;
    switch (dtype) {
      case HBK_INT32: fn_ptr = reinterpret_cast<const void*>(&partition_onepass_kernel<int32_t>); break;
      case HBK_UINT32: fn_ptr = reinterpret_cast<const void*>(&partition_onepass_kernel<uint32_t>); break;
      case HBK_INT64: fn_ptr = reinterpret_cast<const void*>(&partition_onepass_kernel<int64_t>); break;
      default: fn_ptr = reinterpret_cast<const void*>(&partition_onepass_kernel<uint64_t>); break;
    }
    if (!sync_take(stream, words, &take, fn_ptr, kOneWaves * kWave,
                   (kOneMaxTiles + kOneWaves - 1) / kOneWaves)) {
      // the stream is being captured into a graph (a graph replays ONE recorded launch: no state
      // may alternate between calls), or there is no memory for the words: three launches
      onepass = false;
    } else {
      hist = take.words;
      one.zero = take.zero;
      one.zero_words = take.zero_words;
      one.wait = sync_wait_of(take);
    }
  }
  int32_t c0 = 0;
  while (c0 < n_cols) {
    PartArgs args;
    args.fn = fn;
    args.hist = hist;
    args.sizes_t = sizes_t;
    args.n_total_cols = n_cols;
    args.uniform_tiles = 0;
    args.fixed_max = options().partition_fixed_max;
    args.sub_tiles = sub;
    int32_t k = 0;
    int64_t tiles = 0;
    while (c0 < n_cols && k < kMaxColsPerLaunch) {
      PartCol& d = args.col[k];
      d.in = inputs[c0];
      d.out = outputs[c0];
      d.sizes = sizes[c0];
      d.indices = indices[c0];
      d.len = (int32_t)lens[c0];
      d.bucket = make_fastdiv(buckets ? (uint64_t)buckets[c0] : 0);
      d.bucket.d = buckets ? (uint64_t)buckets[c0] : 0;
      d.global_col = c0;
      d.pad_ = 0;
      d.tile_start = (int32_t)tiles;
      args.tile0[k] = (int32_t)tiles;
      {
        const int64_t t = tiles_of(lens[c0], sub);
        if (k == 0) args.uniform_tiles = (int32_t)t;
        if (t != args.uniform_tiles) args.uniform_tiles = 0;
      }
      tiles += tiles_of(lens[c0], sub);
      HBK_REQUIRE(tiles < (1ll << 31), "%s: too many tiles", what);
      ++k;
      ++c0;
    }
    args.n_cols = k;
    args.total_tiles = (int32_t)tiles;
    int rc;
    if (onepass && tiles > 0) {
      switch (dtype) {
        case HBK_INT32: rc = launch_onepass<int32_t>(args, one, stream); break;
        case HBK_INT64: rc = launch_onepass<int64_t>(args, one, stream); break;
        case HBK_UINT32: rc = launch_onepass<uint32_t>(args, one, stream); break;
        default: rc = launch_onepass<uint64_t>(args, one, stream); break;
      }
      one.zero_words = 0;   // the first launch of the call clears the other half
    } else {
      switch (dtype) {
        case HBK_INT32: rc = launch_group<int32_t>(args, P, stream); break;
        case HBK_INT64: rc = launch_group<int64_t>(args, P, stream); break;
        case HBK_UINT32: rc = launch_group<uint32_t>(args, P, stream); break;
        default: rc = launch_group<uint64_t>(args, P, stream); break;
      }
    }
    if (rc != HBK_OK) return rc;
    hist += tiles * P;
  }
  return HBK_OK;
}

}  // namespace

// internal entry (sharded.hip): partition by modulo with the bucketize fused in and a second,
// transposed copy of the sizes ([P][n_cols]: the send layout of the size exchange)
int partition_by_modulo_fused(int32_t n_cols, int32_t num_partitions, const int64_t* const* inputs,
                              const int64_t* lens, const int64_t* buckets,
                              int64_t* const* outputs, int32_t* const* sizes,
                              int32_t* const* indices, int32_t* sizes_t, void* workspace,
                              size_t workspace_bytes, hipStream_t stream) {
  if (buckets != nullptr) {
    for (int32_t c = 0; c < n_cols; ++c) {
      if (buckets[c] < 0) return fail(HBK_INVALID_ARGUMENT, "partition: bucket must be >= 0");
    }
  }
  return partition_impl("partition_by_modulo_fused", n_cols, HBK_INT64, num_partitions, 1, 0,
                        reinterpret_cast<const void* const*>(inputs), lens,
                        reinterpret_cast<void* const*>(outputs), sizes, indices, workspace,
                        workspace_bytes, stream, buckets, sizes_t);
}

}  // namespace hbk

extern "C" size_t hbk_partition_workspace_bytes(int32_t n_cols, const int64_t* lens,
                                                int32_t num_partitions) {
  if (n_cols <= 0 || lens == nullptr || num_partitions < 1) return 0;
  const int sub = hbk::sub_tiles_of(n_cols, lens);
  int64_t tiles = 0;
  for (int32_t c = 0; c < n_cols; ++c) {
    if (lens[c] > 0) tiles += hbk::tiles_of(lens[c], sub);
  }
  return (size_t)tiles * (size_t)num_partitions * sizeof(int32_t);
}

extern "C" int hbk_partition_by_modulo_n(int32_t n_cols, int32_t dtype,
                                         int32_t num_partitions, const void* const* inputs,
                                         const int64_t* lens, void* const* outputs,
                                         int32_t* const* sizes, int32_t* const* indices,
                                         void* workspace, size_t workspace_bytes,
                                         hbk_stream_t stream) {
  return hbk::partition_impl("partition_by_modulo_n", n_cols, dtype, num_partitions, 1, 0,
                             inputs, lens, outputs, sizes, indices, workspace,
                             workspace_bytes, hbk::as_stream(stream));
}

extern "C" int hbk_partition_by_dual_modulo_n(int32_t n_cols, int32_t dtype,
                                              int32_t num_partitions, int32_t modulus,
                                              int32_t stage, const void* const* inputs,
                                              const int64_t* lens, void* const* outputs,
                                              int32_t* const* sizes, int32_t* const* indices,
                                              void* workspace, size_t workspace_bytes,
                                              hbk_stream_t stream) {
  hbk::fail(HBK_OK, "");
  if (stage != 1 && stage != 2) {
    return hbk::fail(HBK_INVALID_ARGUMENT, "partition_by_dual_modulo_n: stage must be 1 or 2");
  }
  return hbk::partition_impl("partition_by_dual_modulo_n", n_cols, dtype, num_partitions,
                             modulus, stage, inputs, lens, outputs, sizes, indices, workspace,
                             workspace_bytes, hbk::as_stream(stream));
}

#ifdef HBK_PART_STAMPS
extern "C" int hbk_debug_part_trace(unsigned long long* out, int reset) {
  using namespace hbk;
  HBK_HIP_OK(hipDeviceSynchronize());
  if (out != nullptr) {
    HBK_HIP_OK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_part_trace), sizeof(g_part_trace)));
  }
  if (reset) {
    static unsigned long long z[kPTraceBlocks * kPTraceSlots];
    HBK_HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_part_trace), z, sizeof(z)));
  }
  return HBK_OK;
}
#endif
