// HbGroupLookup forward: fused multi-table bucketize -> HBM row gather -> segment
// combiner for gfx950 (R1 + R7/R8 + R9 of DESIGN.md; replaces the per-column
// FloorMod / Unique / GatherV2 x3 / SparseSegment* chain the reference builds in
// hbtf/embedding/sharding.py:171-205 and docs/tutorial/ranking/data.py:179-193).
//
// Bandwidth-bound gather, no contraction -> no MFMA.  Design for CDNA4:
//   * one launch for all N columns: a block finds its column with a wave-uniform scan of
//     the tile prefix held in the kernel-argument segment (SGPR loads, no H2D pointer
//     tables as in partition_by_modulo_functors.cu.cc:283-306);
//   * a row of `dim` floats is owned by LPR = pow2(dim/4) adjacent lanes, each moving one
//     16-byte chunk (global_load_dwordx4): a wave64 covers 64/LPR rows per instruction and
//     keeps U independent row loads in flight per lane before the first use;
//   * ids are read once per wave, fully coalesced (one id per lane), turned into row
//     numbers with a multiply-high floor-mod (no 64-bit software divide) and handed to
//     the owning lanes with ds_bpermute;
//   * ids and outputs are streamed with non-temporal accesses so the L2/MALL capacity is
//     left to the table rows.
#include <stdlib.h>

#include <vector>

#include "lookup_common.h"

namespace hbk {
namespace {

constexpr int kBlock = 256;           // 4 waves
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kMaxColsPerLaunch = 128; // LookupArgs travels by value: 14.5 KB of kernarg (gfx950 takes >256 KB)
constexpr int kU = 2;                 // independent row loads per lane (one-id-per-segment path);
                                      // 2 beats 4 by 4% at batch 65536 (shorter tail), equal at 262144
                                      // (tools/tune_lookup.hip, profiles/r01_tune_lookup.txt)
constexpr int kSegIters = 4;          // segments per lane group and block (CSR path)

struct ColArg {
  const float* table;
  const void* ids;
  const int32_t* splits;
  float* out;
  int64_t n_seg;
  IdMap map;
  int32_t dim;
  int32_t chunks;   // elements of width `vec` per row
  uint8_t lpr_log2; // lanes per row = 1 << lpr_log2
  uint8_t ids64;
  uint8_t combiner;
  uint8_t vec4;     // 1: 16-byte chunks, 0: 4-byte chunks (dim % 4 != 0 or unaligned)
  int32_t n_runs;   // > 0: segmented table (see hbk_lookup_column_t)
  int32_t out_stride;  // floats between output rows (>= dim)
  const int64_t* run_start;
  const int64_t* run_base;
  const int32_t* out_slots;   // != NULL: segment s is written to row out_slots[s] of `out` (a permutation
                              // scatter: the owner gather of the sharded step's p2p form)
};

// float offset of logical row r inside the table
template <bool RUNS>
__device__ inline uint64_t row_offset(const ColArg& c, uint64_t r) {
  if (!RUNS) return r * (uint64_t)c.dim;
  int k = 0;
  while (k + 1 < c.n_runs && (uint64_t)c.run_start[k + 1] <= r) ++k;
  return (uint64_t)c.run_base[k] + (r - (uint64_t)c.run_start[k]) * (uint64_t)c.dim;
}

// fp16 rows on one side of a gather (the wire format of the sharded step's embedding exchange,
// hbtf/distribute/nccl/nccl_alltoallv.cc:56-88 + hbtf/common/cast.cu.cc:84-285, fused into the
// kernels on either side of it): HALF = 1 the OUTPUT rows are half (owner gather -> reply
// buffer, fp32 -> fp16 round to nearest even as the reference's cast), HALF = 2 the TABLE rows
// are half (stitch over the received buffer; sums stay fp32).  Offsets and strides count elements.
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <typename V, int HALF>
__device__ inline V load_row_chunk(const float* table, uint64_t off) {
  if (HALF != 2) return *reinterpret_cast<const V*>(table + off);
  const _Float16* t = reinterpret_cast<const _Float16*>(table) + off;
  if (sizeof(V) == 16) {
    const f16x4 h = *reinterpret_cast<const f16x4*>(t);
    f32x4 v = {(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    return *reinterpret_cast<V*>(&v);
  }
  float f = (float)*t;
  return *reinterpret_cast<V*>(&f);
}

template <typename V, int HALF>
__device__ inline void store_out_chunk(float* out, int64_t off, V v) {
  if (HALF != 1) {
    __builtin_nontemporal_store(v, reinterpret_cast<V*>(out + off));
    return;
  }
  _Float16* o = reinterpret_cast<_Float16*>(out) + off;
  if (sizeof(V) == 16) {
    const f32x4 f = *reinterpret_cast<const f32x4*>(&v);
    const f16x4 h = {(_Float16)f[0], (_Float16)f[1], (_Float16)f[2], (_Float16)f[3]};   // RNE
    __builtin_nontemporal_store(h, reinterpret_cast<f16x4*>(o));
  } else {
    *o = (_Float16)*reinterpret_cast<const float*>(&v);
  }
}

struct LookupArgs {
  int32_t n_cols;
  int32_t hot_mode;   // hot-row kernel only: 1 = stage repeated rows in LDS, 2 = large tiles only
  int32_t xcd;        // != 0: every XCD takes a contiguous range of the tiles (xcd_contiguous)
  int32_t interleave; // != 0: every column has the same number of tiles and tile b belongs to
                      // column b % n_cols (row tiles outermost: the columns of one dense output
                      // block are written side by side, see the host side)
  int32_t tile_start[kMaxColsPerLaunch + 1];
  ColArg col[kMaxColsPerLaunch];
};
static_assert(sizeof(LookupArgs) <= 24576, "kernarg budget");

// ---------------------------------------------------------------------------------
// one id per segment (Criteo scalar columns): out[s,:] = table[row(ids[s]),:]
template <typename V, int U, bool RUNS, int HALF, bool SLOT = false, bool D16 = false>
__device__ inline void gather_rows(const ColArg& c, int64_t wave_row0) {
  constexpr int VE = sizeof(V) / 4;
  const int lane = lane_id();
  const int lpr_log2 = D16 ? 2 : c.lpr_log2;   // (D16: the host has looked -- rows of 16 floats, int64 ids)
  const int rpi = kWave >> lpr_log2;  // rows per wave instruction
  const int sub = lane & ((1 << lpr_log2) - 1);
  const int grp = lane >> lpr_log2;
  const int64_t n_seg = c.n_seg;

  // ids: slot q (0 <= q < U*rpi) lives in register q>>6 of lane q&63
  uint64_t rowreg[U];
  int32_t slotreg[U];   // SLOT: where the segment's row goes (read with the id, handed over like it)
  const int n_slots = U * rpi;
#pragma unroll
  for (int k = 0; k < U; ++k) {
    rowreg[k] = kNoRow;
    slotreg[k] = 0;
    const int q = k * kWave + lane;
    const int64_t s = wave_row0 + q;
    if (q < n_slots && s < n_seg) {
      rowreg[k] = id_to_row(c.map, load_id(c.ids, D16 ? 1 : c.ids64, s));
      if (SLOT) slotreg[k] = __builtin_nontemporal_load(c.out_slots + s);
    }
  }

  V v[U];
  const bool live = D16 ? true : sub < c.chunks;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int q0 = u * rpi;  // multiple of rpi (a power of two <= 64): q0>>6 is uniform
    const int k = q0 >> 6;
    uint64_t src = rowreg[0];
#pragma unroll
    for (int kk = 1; kk < U; ++kk) src = (k == kk) ? rowreg[kk] : src;
    const uint64_t r = shfl_u64(src, (q0 & (kWave - 1)) + grp);
    v[u] = zero_v<V>();
    if (live && r != kNoRow) {
      v[u] = load_row_chunk<V, HALF>(c.table, row_offset<RUNS>(c, r) + (uint64_t)sub * VE);
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t s = wave_row0 + u * rpi + grp;
    int64_t dst = s;
    if (SLOT) {
      const int q0 = u * rpi;
      const int k = q0 >> 6;
      int32_t src = slotreg[0];
#pragma unroll
      for (int kk = 1; kk < U; ++kk) src = (k == kk) ? slotreg[kk] : src;
      dst = (int64_t)__shfl(src, (q0 & (kWave - 1)) + grp, kWave);
    }
    if (live && s < n_seg) {
      store_out_chunk<V, HALF>(c.out, dst * (int64_t)c.out_stride + (int64_t)sub * VE, v[u]);
    }
  }
}

// ---------------------------------------------------------------------------------
// ragged segments (row_splits): out[s,:] = combine_j table[row(ids[j]),:], in order of j
template <typename V, bool RUNS, int HALF>
__device__ inline void combine_segments(const ColArg& c, int64_t wave_seg0) {
  constexpr int VE = sizeof(V) / 4;
  const int lane = lane_id();
  const int lpr_log2 = c.lpr_log2;
  const int lpr = 1 << lpr_log2;
  const int rpi = kWave >> lpr_log2;
  const int sub = lane & (lpr - 1);
  const int grp = lane >> lpr_log2;
  const int grp_lane0 = grp << lpr_log2;
  const int64_t n_seg = c.n_seg;
  const bool live = sub < c.chunks;

  for (int it = 0; it < kSegIters; ++it) {
    const int64_t s = wave_seg0 + (int64_t)it * rpi + grp;
    int32_t beg = 0, end = 0;
    if (s < n_seg) {
      beg = c.splits[s];
      end = c.splits[s + 1];
    }
    V acc = zero_v<V>();
    // group-cooperative id fetch: lane `sub` of the group owns id j0 + sub
    int32_t j0 = beg;
    uint64_t myrow = kNoRow;
    if (j0 + sub < end) myrow = id_to_row(c.map, load_id(c.ids, c.ids64, j0 + sub));
    while (__any(j0 < end)) {
      // prefetch the next chunk of ids while this chunk's rows are in flight
      uint64_t nextrow = kNoRow;
      const int32_t j1 = j0 + lpr;
      if (j1 + sub < end) nextrow = id_to_row(c.map, load_id(c.ids, c.ids64, j1 + sub));
      const int32_t cnt = end - j0;  // ids of this group still to add (may be <= 0)
      for (int t0 = 0; t0 < lpr; t0 += 4) {
        V v[4];
        bool p[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int tt = t0 + t;
          const uint64_t r = shfl_u64(myrow, grp_lane0 + (tt & (lpr - 1)));
          p[t] = tt < lpr && tt < cnt;
          v[t] = zero_v<V>();
          if (p[t] && live && r != kNoRow) {
            v[t] = load_row_chunk<V, HALF>(c.table, row_offset<RUNS>(c, r) + (uint64_t)sub * VE);
          }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (p[t]) acc = acc + v[t];
        }
      }
      myrow = nextrow;
      j0 = j1;
    }
    const int32_t n = end - beg;
    if (n > 0 && c.combiner == HBK_COMBINER_MEAN) {
      acc = acc / (float)n;
    } else if (n > 0 && c.combiner == HBK_COMBINER_SQRTN) {
      acc = acc / sqrtf((float)n);
    }
    if (live && s < n_seg) {
      store_out_chunk<V, HALF>(c.out, s * (int64_t)c.out_stride + (int64_t)sub * VE, acc);
    }
  }
}

// One instantiation per (ragged?, 16-byte chunks?, segmented table?) so that the common case --
// one id per sample, dim % 4 == 0, plain table -- carries none of the other paths' code; the host
// launches each kind present in the call with the columns of that kind.
template <bool CSR, typename V, bool RUNS, int HALF = 0, bool SLOT = false, bool D16 = false>
__global__ __launch_bounds__(kBlock) void group_lookup_fwd_kernel(const LookupArgs a) {
  const int b = xcd_contiguous((int)blockIdx.x, (int)gridDim.x, a.xcd);
  // last column whose first tile is <= b.  A binary search over the kernel-argument table is
  // up to 7 DEPENDENT scalar loads (each a cold miss at kernel start); here every lane reads
  // one entry (two independent loads cover 128 columns) and a ballot counts the entries <= b:
  // one memory round trip.
  int ci;
  int64_t tile;
  if (a.interleave) {
    ci = b % a.n_cols;
    tile = b / a.n_cols;
  } else {
    const int lane = (int)threadIdx.x & (kWave - 1);
    const int n = a.n_cols;
    const int t0 = lane < n ? a.tile_start[lane] : 0x7fffffff;
    const int t1 = lane + kWave < n ? a.tile_start[lane + kWave] : 0x7fffffff;
    ci = (int)__builtin_popcountll(__ballot(t0 <= b)) +
         (int)__builtin_popcountll(__ballot(t1 <= b)) - 1;
    ci = __builtin_amdgcn_readfirstlane(ci);
    tile = b - a.tile_start[ci];
  }
  const ColArg& c = a.col[ci];
  const int wave = (int)(threadIdx.x >> 6);
  const int rpi = D16 ? 16 : kWave >> c.lpr_log2;
  if (!CSR) {
    const int64_t row0 = (tile * kWavesPerBlock + wave) * (int64_t)(kU * rpi);
    if (row0 >= c.n_seg) return;
    gather_rows<V, kU, RUNS, HALF, SLOT, D16>(c, row0);
  } else {
    const int64_t seg0 = (tile * kWavesPerBlock + wave) * (int64_t)(kSegIters * rpi);
    if (seg0 >= c.n_seg) return;
    combine_segments<V, RUNS, HALF>(c, seg0);
  }
}

// ---------------------------------------------------------------------------------
// LDS-staged hot rows (config 4: dim 128, Zipf ids; north_star's "LDS-staged hot rows", the
// reference's analogue is the slab cache in front of the table, hbtf/embedding/
// lookup_functors.cu.cc:54-149).  One id per segment, wide rows (dim >= 64, 16-byte chunks), tables
// of < 2^32 rows.  A workgroup owns a tile of 256 segments: (1) the tile's rows enter an LDS hash
// table (32-bit CAS) with a count per row; (2) rows that occur more than once get one of S =
// min(64, 3072 / dim) staging slots; (3) every staged row is fetched ONCE into LDS; (4) a segment
// whose row is staged is served from LDS, the others gather from the table as before.  With
// Zipf(1.2) ids about half of a tile's segments name one of its ~13 repeated rows: half of the
// L1 row reads go away; what stays are the stores.  24.6 KB of LDS = 6 workgroups per CU on
// purpose: with 17 KB (8 per CU) the same kernel measured 248 us instead of 215 on config 4 --
// the store stream likes fewer, longer writers (the plain store probe says the same: 151 us from
// 8192 workgroups, 183 us from 2048 x 4).
constexpr int kHotTile = 256;          // segments per workgroup
constexpr int kHotSlots = 1024;        // LDS hash slots (a tile holds <= 256 distinct rows)
constexpr int kHotStageFloats = 4096;  // 16 KB of staged rows
constexpr int kHotMaxStage = 64;
constexpr int kHotU = 4;               // row loads in flight per lane
constexpr uint32_t kHotEmpty = 0xffffffffu;   // (rows are < 2^32 - 1: host check)

__device__ inline uint32_t hot_mix(uint32_t k) {
  k ^= k >> 16;
  k *= 0x85ebca6bu;
  k ^= k >> 13;
  return k;
}

__global__ __launch_bounds__(kBlock) void group_lookup_fwd_hot_kernel(const LookupArgs a) {
  typedef f32x4 V;
  __shared__ uint32_t keys[kHotSlots];
  __shared__ int32_t cnt[kHotSlots];       // pairs of the slot's row; then its staging slot or -1
  __shared__ uint16_t slot_of[kHotTile];
  __shared__ uint16_t stage_slot[kHotMaxStage];
  __shared__ float stage[kHotStageFloats];
  __shared__ int32_t n_staged;
  const int b = xcd_contiguous((int)blockIdx.x, (int)gridDim.x, a.xcd);
  const int tid = (int)threadIdx.x;
  int ci;
  int64_t seg0;
  if (a.interleave) {
    ci = b % a.n_cols;
    seg0 = (int64_t)(b / a.n_cols) * kHotTile;
  } else {
    const int lane = tid & (kWave - 1);
    const int n = a.n_cols;
    const int t0 = lane < n ? a.tile_start[lane] : 0x7fffffff;
    const int t1 = lane + kWave < n ? a.tile_start[lane + kWave] : 0x7fffffff;
    ci = (int)__builtin_popcountll(__ballot(t0 <= b)) +
         (int)__builtin_popcountll(__ballot(t1 <= b)) - 1;
    ci = __builtin_amdgcn_readfirstlane(ci);
    seg0 = (int64_t)(b - a.tile_start[ci]) * kHotTile;
  }
  const ColArg& c = a.col[ci];
  const int64_t n_seg = c.n_seg;
  const bool staging = a.hot_mode == 1;

  // (0) one id per thread
  uint64_t row = kNoRow;
  if (seg0 + tid < n_seg) row = id_to_row(c.map, load_id(c.ids, c.ids64, seg0 + tid));
  for (int i = tid; i < kHotSlots; i += kBlock) {
    keys[i] = kHotEmpty;
    cnt[i] = 0;
  }
  if (tid == 0) n_staged = 0;
  __syncthreads();
  // (1) rows -> slots, one count per row
  int slot = 0xffff;
  if (row != kNoRow) {
    const uint32_t r32 = (uint32_t)row;
    int h = (int)(hot_mix(r32) & (kHotSlots - 1));
    for (;;) {
      const uint32_t k = keys[h];
      if (k == r32) break;
      if (k == kHotEmpty) {
        const uint32_t prev = atomicCAS(&keys[h], kHotEmpty, r32);
        if (prev == kHotEmpty || prev == r32) break;
      }
      h = (h + 1) & (kHotSlots - 1);
    }
    slot = h;
    if (staging) atomicAdd(&cnt[h], 1);
  }
  slot_of[tid] = (uint16_t)slot;
  __syncthreads();
  // (2) repeated rows take the staging slots, first come first served; cnt[] becomes the map
  // hash slot -> staging slot
  const int S = kHotStageFloats / c.dim < kHotMaxStage ? kHotStageFloats / c.dim : kHotMaxStage;
  for (int i = tid; i < kHotSlots; i += kBlock) {
    int st = -1;
    if (cnt[i] >= 2) {
      const int idx = atomicAdd(&n_staged, 1);
      if (idx < S) {
        st = idx;
        stage_slot[idx] = (uint16_t)i;
      }
    }
    cnt[i] = st;
  }
  __syncthreads();
  const int ns = n_staged < S ? n_staged : S;
  const int lpr_log2 = c.lpr_log2;
  const int sub = tid & ((1 << lpr_log2) - 1);
  const int grp = tid >> lpr_log2;
  const int groups = kBlock >> lpr_log2;
  const bool live = sub < c.chunks;
  // (3) every staged row is fetched once
  for (int i0 = 0; i0 < ns; i0 += kHotU * groups) {
    V v[kHotU];
#pragma unroll
    for (int u = 0; u < kHotU; ++u) {
      const int i = i0 + u * groups + grp;
      v[u] = zero_v<V>();
      if (i < ns && live) {
        const uint64_t r = keys[stage_slot[i]];
        v[u] = *reinterpret_cast<const V*>(c.table + r * (uint64_t)c.dim + (uint64_t)sub * 4);
      }
    }
#pragma unroll
    for (int u = 0; u < kHotU; ++u) {
      const int i = i0 + u * groups + grp;
      if (i < ns && live) *reinterpret_cast<V*>(&stage[(size_t)i * c.dim + (size_t)sub * 4]) = v[u];
    }
  }
  if (ns > 0) __syncthreads();   // uniform
  // (4) the tile's segments: staged rows from LDS, the others from the table.  (A software
  // pipeline -- batch i + 1 requested before batch i is stored -- measured 5 % slower.)
  for (int s0 = 0; s0 < kHotTile; s0 += kHotU * groups) {
    if (seg0 + s0 >= n_seg) break;   // uniform
    V v[kHotU];
#pragma unroll
    for (int u = 0; u < kHotU; ++u) {
      const int sl = s0 + u * groups + grp;
      v[u] = zero_v<V>();
      if (seg0 + sl < n_seg && live) {
        const int q = (int)slot_of[sl];
        if (q != 0xffff) {
          const int st = cnt[q];
          if (st >= 0) {
            v[u] = *reinterpret_cast<const V*>(&stage[(size_t)st * c.dim + (size_t)sub * 4]);
          } else {
            v[u] = *reinterpret_cast<const V*>(c.table + (uint64_t)keys[q] * (uint64_t)c.dim +
                                               (uint64_t)sub * 4);
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kHotU; ++u) {
      const int64_t s = seg0 + s0 + u * groups + grp;
      if (s < n_seg && live) {
        __builtin_nontemporal_store(
            v[u], reinterpret_cast<V*>(c.out + s * (int64_t)c.out_stride + (int64_t)sub * 4));
      }
    }
  }
}

template <bool CSR, typename V, bool RUNS, int HALF = 0, bool SLOT = false>
void launch_kind(const LookupArgs& args, unsigned tiles, hipStream_t stream) {
  hipLaunchKernelGGL((group_lookup_fwd_kernel<CSR, V, RUNS, HALF, SLOT>), dim3(tiles), dim3(kBlock), 0,
                     stream, args);
}

void launch_by_kind(int kind, const LookupArgs& args, unsigned tiles, hipStream_t stream) {
  if (kind == 39) {  // kind 0 with every column 16 floats wide and int64 ids (the headline's shape): constants
    hipLaunchKernelGGL((group_lookup_fwd_kernel<false, f32x4, false, 0, false, true>), dim3(tiles), dim3(kBlock), 0,
                       stream, args);
    return;
  }
  if (kind == 8) {  // wide rows through the hot-row kernel (option fwd_hot_rows)
    hipLaunchKernelGGL(group_lookup_fwd_hot_kernel, dim3(tiles), dim3(kBlock), 0, stream, args);
    return;
  }
  switch (kind) {  // bit 0 ragged, bit 1 scalar chunks, bit 2 segmented table, bit 4 fp16 rows, bit 5 output slots
    case 32: launch_kind<false, f32x4, false, 0, true>(args, tiles, stream); return;   // out[out_slots[s]]
    case 34: launch_kind<false, float, false, 0, true>(args, tiles, stream); return;
    case 16: launch_kind<false, f32x4, false, 1>(args, tiles, stream); return;   // half output
    case 18: launch_kind<false, float, false, 1>(args, tiles, stream); return;
    case 20: launch_kind<false, f32x4, true, 2>(args, tiles, stream); return;    // half table
    case 21: launch_kind<true, f32x4, true, 2>(args, tiles, stream); return;
    case 22: launch_kind<false, float, true, 2>(args, tiles, stream); return;
    case 23: launch_kind<true, float, true, 2>(args, tiles, stream); return;
    default: break;
  }
  switch (kind) {
    case 0: launch_kind<false, f32x4, false>(args, tiles, stream); break;
    case 1: launch_kind<true, f32x4, false>(args, tiles, stream); break;
    case 2: launch_kind<false, float, false>(args, tiles, stream); break;
    case 3: launch_kind<true, float, false>(args, tiles, stream); break;
    case 4: launch_kind<false, f32x4, true>(args, tiles, stream); break;
    case 5: launch_kind<true, f32x4, true>(args, tiles, stream); break;
    case 6: launch_kind<false, float, true>(args, tiles, stream); break;
    default: launch_kind<true, float, true>(args, tiles, stream); break;
  }
}

}  // namespace
}  // namespace hbk

extern "C" int hbk_group_lookup_fwd(int32_t n_cols, const hbk_lookup_column_t* cols,
                                    hbk_stream_t stream) {
  using namespace hbk;
  HBK_REQUIRE(n_cols >= 0, "group_lookup_fwd: n_cols must be >= 0, got %d", n_cols);
  HBK_REQUIRE(n_cols == 0 || cols != nullptr, "group_lookup_fwd: cols is NULL");
  for (int32_t c = 0; c < n_cols; ++c) {
    const hbk_lookup_column_t& h = cols[c];
    HBK_REQUIRE(h.dim >= 1, "group_lookup_fwd: column %d: dim must be >= 1, got %d", c, h.dim);
    HBK_REQUIRE(h.dim <= 1024, "group_lookup_fwd: column %d: dim %d > 1024 unsupported", c,
                h.dim);
    HBK_REQUIRE(h.rows >= 0 && h.n_ids >= 0 && h.n_segments >= 0,
                "group_lookup_fwd: column %d: negative size", c);
    HBK_REQUIRE(h.ids_dtype == HBK_INT32 || h.ids_dtype == HBK_INT64,
                "group_lookup_fwd: column %d: ids must be int32 or int64", c);
    HBK_REQUIRE(h.bucket >= 0, "group_lookup_fwd: column %d: bucket must be >= 0", c);
    HBK_REQUIRE(h.divisor >= 1, "group_lookup_fwd: column %d: divisor must be >= 1", c);
    HBK_REQUIRE(h.combiner >= HBK_COMBINER_SUM && h.combiner <= HBK_COMBINER_SQRTN,
                "group_lookup_fwd: column %d: unknown combiner %d", c, h.combiner);
    HBK_REQUIRE(h.row_splits != nullptr || h.n_segments == h.n_ids,
                "group_lookup_fwd: column %d: n_segments (%lld) must equal n_ids (%lld) "
                "when row_splits is NULL",
                c, (long long)h.n_segments, (long long)h.n_ids);
    HBK_REQUIRE(h.n_segments == 0 ||
                    ((h.table || h.rows == 0) && h.out && (h.ids || h.n_ids == 0)),
                "group_lookup_fwd: column %d: NULL buffer", c);
    HBK_REQUIRE(h.n_runs >= 0 && (h.n_runs == 0 || (h.run_start && h.run_base)),
                "group_lookup_fwd: column %d: bad segmented-table description", c);
    HBK_REQUIRE(h.n_ids < (1ll << 31) && h.n_segments < (1ll << 31),
                "group_lookup_fwd: column %d: more than 2^31-1 ids/segments", c);
    HBK_REQUIRE(h.half_io == 0 ||
                    (h.half_io == HBK_LOOKUP_OUT_HALF && h.row_splits == nullptr && h.n_runs == 0) ||
                    (h.half_io == HBK_LOOKUP_TABLE_HALF && h.n_runs > 0),
                "group_lookup_fwd: column %d: half_io %d: fp16 output rows need one id per segment "
                "and a plain table, fp16 table rows a segmented table", c, h.half_io);
    HBK_REQUIRE(h.out_slots == nullptr ||
                    (h.row_splits == nullptr && h.n_runs == 0 && h.half_io == 0),
                "group_lookup_fwd: column %d: out_slots needs one id per segment, a plain table and "
                "fp32 rows", c);
  }

  const int hot_mode = options().fwd_hot_rows;
  // every column is classified ONCE (row shape, kernel kind); the launches then walk only the
  // kinds that occur (the sharded owner gather passes N x W virtual columns: 17 passes over 208
  // descriptors, each redoing the checks, were microseconds of the step's enqueue path)
  struct Classified {
    RowShape shape;
    int kind;   // -1: nothing to do (no segments)
  };
  std::vector<Classified> cls((size_t)(n_cols > 0 ? n_cols : 1));
  uint64_t kinds_present = 0;
  for (int32_t c = 0; c < n_cols; ++c) {
    const hbk_lookup_column_t& h = cols[c];
    cls[c].kind = -1;
    if (h.n_segments == 0) continue;
    HBK_REQUIRE(h.out_stride == 0 || h.out_stride >= h.dim,
                "group_lookup_fwd: out_stride %d is smaller than dim %d", h.out_stride, h.dim);
    // a strided output keeps 16-byte chunks only if every row start stays 16-byte aligned
    // (a half buffer is held to 8 bytes where an fp32 one is held to 16: its address counts
    // twice in the alignment test; a row stride must be a multiple of 4 elements either way)
    const uintptr_t table_bits = (uintptr_t)h.table * (h.half_io == HBK_LOOKUP_TABLE_HALF ? 2 : 1);
    const uintptr_t out_bits = (uintptr_t)h.out * (h.half_io == HBK_LOOKUP_OUT_HALF ? 2 : 1) |
                               ((uintptr_t)(uint32_t)h.out_stride * 4);
    HBK_REQUIRE(make_rowshape(h.dim, table_bits | out_bits, &cls[c].shape),
                "group_lookup_fwd: dim %d needs more than 64 lanes per row "
                "(unaligned or dim %% 4 != 0 with dim > 64 is unsupported)", h.dim);
    const RowShape& shape = cls[c].shape;
    int col_kind = (h.row_splits != nullptr ? 1 : 0) | (shape.vec4 ? 0 : 2) |
                   (h.n_runs > 0 ? 4 : 0);
    // one id per segment, plain table, wide 16-byte-chunk rows: the hot-row kernel when asked
    if (h.half_io != 0) col_kind |= 16;
    if (h.out_slots != nullptr) col_kind |= 32;
    if (col_kind == 0 && (hot_mode > 0 || h.hot_rows != 0) && h.dim >= 64 && shape.lpr_log2 <= 6 &&
        h.dim <= kHotStageFloats && h.rows < 0xffffffffll) {
      col_kind = 8;
    }
    if (col_kind == 0 && options().fwd_d16 != 0 && h.dim == 16 && h.ids_dtype == HBK_INT64 && shape.lpr_log2 == 2 &&
        shape.chunks == 4) {
      col_kind = 39;
    }
    cls[c].kind = col_kind;
    kinds_present |= 1ull << col_kind;
  }
  for (int kind = 0; kind < 40; ++kind) {
    if (((kinds_present >> kind) & 1ull) == 0ull) continue;
    int32_t c0 = 0;
    while (c0 < n_cols) {
      LookupArgs args;
      args.hot_mode = hot_mode > 0 ? hot_mode : 1;
      args.xcd = 0;
      args.interleave = 0;
      int32_t k = 0;
      int64_t tiles = 0;
      bool same_tiles = true, one_block = true;   // (see args.interleave below)
      uintptr_t block_lo = 0, block_hi = 0;   // lowest / highest output address of the launch
      int64_t small_lookups = 0, all_lookups = 0;   // (tables of <= 2 MB: see args.xcd below)
      args.tile_start[0] = 0;
      while (c0 < n_cols && k < kMaxColsPerLaunch) {
        const int32_t ci = c0++;
        if (cls[ci].kind != kind) continue;
        const hbk_lookup_column_t& h = cols[ci];
        const RowShape& shape = cls[ci].shape;
        const int col_kind = kind;
        ColArg& d = args.col[k];
        d.table = h.table;
        d.ids = h.ids;
        d.splits = h.row_splits;
        d.out = h.out;
        d.n_seg = h.n_segments;
        d.map = make_idmap(h.bucket, h.divisor, h.rows);
        d.dim = h.dim;
        d.vec4 = shape.vec4;
        d.chunks = shape.chunks;
        d.lpr_log2 = shape.lpr_log2;
        d.ids64 = h.ids_dtype == HBK_INT64;
        d.combiner = (uint8_t)h.combiner;
        d.n_runs = h.n_runs;
        d.out_stride = h.out_stride > 0 ? h.out_stride : h.dim;
        d.run_start = h.run_start;
        d.run_base = h.run_base;
        d.out_slots = h.out_slots;
        const int64_t rpi = kWave >> d.lpr_log2;
        const int64_t per_block =
            col_kind == 8 ? kHotTile : kWavesPerBlock * rpi * (h.row_splits ? kSegIters : kU);
        const int64_t col_tiles = (h.n_segments + per_block - 1) / per_block;
        same_tiles = same_tiles && (k == 0 || col_tiles == args.tile_start[1]);
        // "one dense block": every column's rows are strided and start inside the first row of
        // the lowest column's block
        one_block = one_block && d.out_stride > h.dim && h.half_io == 0;
        const uintptr_t out_at = reinterpret_cast<uintptr_t>(h.out);
        block_lo = k == 0 || out_at < block_lo ? out_at : block_lo;
        block_hi = k == 0 || out_at > block_hi ? out_at : block_hi;
        one_block = one_block && (block_hi - block_lo) / sizeof(float) + (uintptr_t)h.dim <=
                                     (uintptr_t)d.out_stride;
        tiles += col_tiles;
        HBK_REQUIRE(tiles < (1ll << 31), "group_lookup_fwd: grid too large");
        all_lookups += h.n_ids;
        if (h.rows * (int64_t)h.dim * 4 <= (2ll << 20)) small_lookups += h.n_ids;
        ++k;
        args.tile_start[k] = (int32_t)tiles;
      }
      if (k == 0) continue;
      args.n_cols = k;
      // Tiles to XCDs (xcd_contiguous): whole columns per XCD
      //  * for the hot-row kernel -- a hot row staged by many tiles of its column is then fetched
      //    through ONE L2 instead of eight.  Toggled inside one process on the same tensors
      //    (tools/sweep.py, SWEEP_J_XCD; between processes the same setting differs by up to 15 %
      //    on config 4): hot-row tiles Zipf 221 -> 213-220 us, without the staging (mode 2) 244 ->
      //    223-230, uniform ids 351-355 -> 331-347;
      //  * for a launch with SMALL tables (>= 20 % of its lookups go to tables of <= 2 MB): dealt
      //    round robin every L2 holds its own copy of every small table, per XCD a table lives in
      //    one -- config-5 shape (rows 1e3..1e7) 1299-1352 -> 1264-1268 us.
      // Other launches of the per-wave kernels gain nothing (config 2 56.7 us and config 4 Zipf
      // 235-241 either way) or lose (ragged dim 16 over 6.4 MB tables 316-324 -> 329, one row per
      // column 202 -> 209) and keep the round-robin placement.  Option fwd_xcd: 0 never, 1 these
      // rules, 2 always.
      const int xcd_opt = options().fwd_xcd;
      args.xcd = xcd_opt == 2 || (xcd_opt == 1 && (kind == 8 || 5 * small_lookups >= all_lookups))
                     ? 1 : 0;
      // Row tiles outermost when the columns fill ONE dense [B, sum dim] block (DenseFeatures):
      // the workgroups in flight then write whole rows of the block side by side instead of
      // dim-wide pieces a row stride apart.  Option fwd_interleave: 0 never, 1 per-wave kernels,
      // 2 the hot-row tiles too, 3 any launch of equal tile counts.
      const int il = options().fwd_interleave;
      if (k > 1 && same_tiles && il > 0 &&
          (il == 3 || (one_block && (kind != 8 || il == 2)))) {
        args.interleave = 1;
      }
      launch_by_kind(kind, args, (unsigned)tiles, as_stream(stream));
      HBK_HIP_OK(hipGetLastError());
    }
  }
  return HBK_OK;
}
