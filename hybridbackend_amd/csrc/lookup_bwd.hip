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
//   1 hist       row(j) = bucketize/`// W`; bucket = high bits of a 32-bit mix of the row;
//                per-1024-id-tile LDS histogram, all columns in one launch
//   2 scan       hist[tile][bucket]: prefix over tiles per bucket, then bucket starts
//   3 scatter    (row, segment) pairs grouped by bucket (LDS ticket per bucket and tile)
//                (1-3 are ONE launch for columns of <= 512 buckets and <= 64 tiles: bwd_group_kernel)
//   4 reduce     ONE workgroup owns a bucket, hence every table row that hashes to it.  Per
//                chunk of <= 512 pairs: (a) the rows are entered into an LDS hash table (64-bit
//                CAS) with one ticket per pair and slot; (b) one packed block scan over the PAIRS
//                (new rows | rows with several pairs | their pair counts) hands every new row its
//                output position (one global atomic per workgroup claims the range, issued right
//                after (a)); (c) one round of gradient loads, 6 rows in flight per lane (3 with the
//                optimizer step): a row with ONE pair in the chunk -- the common case -- is written
//                straight from the registers the gradient arrived in, plain 16-byte stores, with
//                the optimizer step when apply_lr != 0 (exclusive ownership makes the
//                read-modify-write race free); (d) rows with several pairs are counting-sorted by
//                slot and the sorted pairs are walked FLAT: a lane group takes consecutive
//                positions, all rows in flight, sums runs of one slot in registers, and the group
//                where a run starts adds the head partials of its successors (LDS) -- the
//                "LDS-staged hot rows": a row of 400 pairs is 50 lane groups' partials.  A row
//                that spans chunks is accumulated into its output row by the owning workgroup.  The table
//                takes new rows only while it has room; pairs whose row found no room are left for
//                another PASS over the bucket with an emptied table (handled pairs are struck out
//                of the pair buffer), so every row is emitted exactly once whatever the bucket
//                holds -- adversarial hashing or 100 M ids in one column cost passes, never
//                correctness (the fused Adagrad step depends on that).
//   5 split      a bucket far above the average size holds a hot row (Zipf heads: one row can
//                own 20% of a column).  One workgroup has ~32 KB of loads in flight, so it would
//                sum such a row at ~15 GB/s while the rest of the chip idles.  The scan kernel
//                therefore lists every bucket with more than split_t pairs; the reduce grid has
//                spare workgroups that take the 2nd, 3rd.. range of split_t pairs of a listed
//                bucket.  The workgroups of a split bucket run steps (a)-(d) on their range but
//                emit (row, partial sum) entries into the bucket's own slice of a partials
//                buffer; a merge launch then runs the same steps over those entries (one
//                workgroup per split bucket) and emits the final rows.  A table row still has
//                exactly one emitting workgroup, so rows stay unique and the fused SGD apply
//                stays race free.
//   4b dense     (round 3) columns whose batch covers the table densely take row-RANGE buckets and
//                direct-indexed LDS bitmaps instead of the hash table (dense_reduce, below).
//   4c rowsort   (round 4) columns whose batch is dense in the table (rows <= 8-16 x ids: ragged columns,
//                small / medium tables) take row-RANGE buckets of ~1800 pairs and the row-sorted job of
//                lookup_bwd_rowsort.h: bitmap ranks, LDS counting sort by row, equal shares of the
//                sorted order walked with 12 gradient rows in flight per lane.
//   0b scale     (round 4) ragged columns with mean / sqrtn: the seg-of launch also writes every segment's
//                gradient row times the combiner's factor; the stages behind it see a SUM column.
//   3b staged    (round 4) large columns of <= 1024 buckets: the scatter's pairs leave sorted by bucket.
// Where lines live (round 3): gradient rows are loaded with PLAIN loads (a 128-byte line holds two
// rows of dim 16, a ragged column re-reads its rows), the reduce jobs and the large columns'
// scatter tiles go to the XCDs in contiguous ranges (whole columns per XCD: xcd_contiguous,
// GArgs.xcd / xcd_w) so that both rows of a gradient line -- and the pieces of a pair-array line
// -- meet in ONE L2.
// Summation order inside a row is not fixed (pair order comes from LDS tickets): 1e-5 relative.
// Option bwd_deterministic fixes it -- id order, the sequential fp32 sum, rows ascending: 1 = the row-sorted
// jobs in their in-order form (lookup_bwd_rowsort.h, DET) with a count launch in front of the reduce,
// 2 (and the columns 1 does not fit) = a stable sort of the batch's pairs (lookup_bwd_det.h).
#include <alloca.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <type_traits>
#include <mutex>
#include <vector>

#include "lookup_common.h"

#ifndef HBK_BWD_WAVES
#define HBK_BWD_WAVES 4
#endif

namespace hbk {
// det_prims.hip: the stable radix sort and the scan of the deterministic backward (rocPRIM)
size_t det_sort_temp_bytes(size_t n, int end_bit);
int det_sort_pairs(void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out,
                   const uint32_t* vals_in, uint32_t* vals_out, size_t n, int end_bit,
                   hipStream_t stream);
size_t det_scan_temp_bytes(size_t n);
int det_scan(void* temp, size_t temp_bytes, const int32_t* in, int32_t* out, size_t n,
             hipStream_t stream);
namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kMaxCols = 64;
#ifndef HBK_BWD_TILE
#define HBK_BWD_TILE 2048
#endif
constexpr int kTile = HBK_BWD_TILE;  // ids per 256-thread block in hist / scatter (config 2
                                     // backward: 4096 149 us, 2048 139, 1024 145, 512 158)
constexpr int kPerThread = kTile / kBlock;
constexpr int kBatch = kPerThread < 8 ? kPerThread : 8;   // loads in flight per thread
constexpr int kMaxBuckets = 16384;   // 64 KB of LDS counters in hist / scatter
// The reduce stage works in TEAMS: kTeam threads own one bucket.  kTeam = 256 (shipped): a team
// is the workgroup.  kTeam = 64 (-DHBK_BWD_TEAM=64): a team is ONE WAVE, a workgroup holds four
// independent teams with their own LDS tables and there is no workgroup barrier anywhere in the
// reduce kernel (lanes of a wave execute LDS instructions in lockstep: a fence that keeps the
// compiler from reordering them is all the synchronisation a team needs), so a CU runs 16
// independent chains of dependent memory round trips instead of 4.  Measured on the config-2
// backward (profiles/r02_bwd_reduce_trace.txt): a wave-team job lives 17 us instead of 21 and the
// reduce kernel takes the same ~95 us either way (jobs x life / resident teams + one life of
// tail), while four times as many buckets cost the grouping kernels 19 us (scatter 26 -> 39, scan
// 6 -> 12): 164 vs 142 us in total.
#ifndef HBK_BWD_TEAM
#define HBK_BWD_TEAM 256
#endif
constexpr int kTeam = HBK_BWD_TEAM;
constexpr int kTeams = kBlock / kTeam;  // teams per workgroup
constexpr int kCP = 2 * kTeam;        // pairs per chunk in the reduce kernel
#ifndef HBK_BWD_RS_CAP
#define HBK_BWD_RS_CAP 2048
#endif
constexpr int kRsCap = HBK_BWD_RS_CAP;   // pairs per chunk of the row-sorted reduce (4c, lookup_bwd_rowsort.h)
#ifndef HBK_BWD_SLOTX
#define HBK_BWD_SLOTX 2
#endif
constexpr int kSlots = HBK_BWD_SLOTX * kCP;   // LDS hash-table slots
constexpr int kRoom = kSlots - kTeam - 8;   // rows enter the table while it holds fewer (a
                                           // team's concurrent inserts overshoot by < kTeam)
static_assert(kTeam == 64 || kTeam == kBlock, "a team is one wave or the whole workgroup");

// all threads of a team have finished their LDS accesses before anyone goes on
__device__ inline void team_sync() {
  if (kTeam == kBlock) {
    __syncthreads();
  } else {
    // one wave: the hardware runs its LDS instructions in order; keep the compiler from moving
    // LDS accesses across this point and wait for the ones issued
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
}
#ifndef HBK_BWD_PRE
#define HBK_BWD_PRE 3
#endif
// gradient rows a lane keeps in flight: 2 x kPre without the optimizer step, kPre with it.  Four
// (8 / 4 rows) made the kernels spill 64-188 bytes per lane of loop-invariant state: 63 MB of
// scratch writes per config-2 launch (TCC_EA0_WRREQ_64B 2.85 M where the rows account for 1.86 M),
// 117 us instead of 110.
constexpr int kPre = HBK_BWD_PRE;
#ifndef HBK_BWD_WIDE_W
#define HBK_BWD_WIDE_W 5   // width of the WIDE walk (rows of >= 16 lanes, with the optimizer step).  2 x kPre = 6 spills
                          // 28 VGPRs in the SGD instantiation, 5 spills 12: config 4 + SGD 423-428 -> 412-416 us, the
                          // config-5 mix + SGD 3.39-3.60 -> 3.22-3.46 ms in-box (Adagrad the same)
#endif
constexpr int kLdsRowPairs = kCP / 8; // multi-pair slots are summed in LDS rows when they hold at
                                      // most this many pairs together
#ifndef HBK_BWD_HOT
#define HBK_BWD_HOT 4
#endif
constexpr int kHotTries = HBK_BWD_HOT; // ballot rounds that look for a hot row inside a wave
constexpr int kHotMin = 8;             // lanes sharing a row that make the wave reduce it first
constexpr unsigned long long kEmptyKey = ~0ull;
constexpr int64_t kDonePair = -1;      // a pair struck out of the pair buffer (rows are >= 0)
constexpr uint16_t kNoSlot = 0xffff;

// Phase timing of the reduce kernel (probe builds only: tools/Makefile builds a second library with
// -DHBK_BWD_STAMPS; never defined in the product build): thread 0 of every workgroup adds the
// shader-clock cycles of each phase into a global table read back by hbk_debug_bwd_stamps().
#ifdef HBK_BWD_STAMPS
constexpr int kTraceBlocks = 8192, kTraceSlots = 8;
__device__ unsigned long long g_bwd_trace[kTraceBlocks * kTraceSlots];
__device__ unsigned long long g_bwd_sub[kTraceBlocks * 4];   // sub-stamps inside phase (a)
#define HBK_SUBSTAMP(i)                                                                    \
  do {                                                                                     \
    if (threadIdx.x == 0 && blockIdx.x < kTraceBlocks) {                                   \
      g_bwd_sub[blockIdx.x * 4 + (i)] = __builtin_amdgcn_s_memrealtime();                  \
    }                                                                                      \
  } while (0)
#define HBK_STAMP_BEGIN()                                                                  \
  if (threadIdx.x == 0 && blockIdx.x < kTraceBlocks) {                                     \
    g_bwd_trace[blockIdx.x * kTraceSlots] = __builtin_amdgcn_s_memrealtime();                  \
  }
#define HBK_STAMP(i)                                                                       \
  do {                                                                                     \
    if (threadIdx.x == 0 && blockIdx.x < kTraceBlocks) {                                   \
      g_bwd_trace[blockIdx.x * kTraceSlots + (i)] = __builtin_amdgcn_s_memrealtime();          \
    }                                                                                      \
  } while (0)
#define HBK_STAMP_ARG
#define HBK_STAMP_PASS
#else
#define HBK_SUBSTAMP(i)
#define HBK_STAMP_BEGIN()
#define HBK_STAMP(i)
#define HBK_STAMP_ARG
#define HBK_STAMP_PASS
#endif

struct GCol {
  const void* ids;
  const float* grad_out;     // [n_seg, dim]
  const int32_t* splits;
  int64_t* unique_rows;
  float* grad_rows;
  int32_t* n_unique;
  int32_t no_emit;           // unique_rows / grad_rows are scratch: the caller passed none (step only)
  int32_t sync0;             // group kernel: first of the column's [tiles][n_buckets] sync words
  int32_t* counter;          // the column's claim counter, ALONE on its 256-byte line of the workspace:
                             // every reduce job claims its output range with one atomic on it, and
                             // atomics on one line are served by one memory channel at ~80 per us --
                             // 26 caller-side counters side by side in one line made the whole reduce
                             // kernel wait for that channel (3822 claims = its ~95 us)
  float* table;
  float* accum;              // Adagrad accumulator [rows, dim] (apply kind 2)
  int32_t* hist;             // [P * tiles] -> exclusive offsets after the scan
  int32_t* bstart;           // [P + 1]
  int64_t* pair_row[1];      // [n_ids] rows of the pairs, grouped by bucket
  int32_t* pair_seg[1];      // [n_ids] their segments
  int32_t* seg_of;           // [n_ids], ragged columns only
  float* scaled;             // seg-of / histogram launch: [n_seg, dim] gradient rows times the combiner's 1 / n,
                             // 1 / sqrt(n) -- what the reduce stage then reads as a SUM column's gradient
  const float* raw_grad;     // scaled != NULL in the histogram launch: the caller's gradient rows, their
  int32_t raw_stride;        //   stride and the combiner (grad_out / grad_stride / combiner of such a
  int32_t raw_combiner;      //   column already describe the scaled rows)
  int4* desc;                // [n_buckets + e_max] job of every reduce workgroup slot of the column:
                             // {first pair, pairs of the bucket, bucket or -1, range index}
  int32_t* work;             // [2 * e_max] (bucket, range index >= 1) of the spare workgroups
  int32_t* n_extra;          // -> entries of `work` (written by the scan kernel)
  int32_t* pcount;           // [P] partial entries of a split bucket
  int64_t* part_rows;        // [n_ids] partial entries; bucket b owns [bstart[b], bstart[b+1])
  float* part_vals;          // [n_ids, dim]
  const int64_t* run_start;  // segmented inputs (n_runs > 0): see hbk_lookup_grad_column_t
  const int64_t* run_ids;
  const int64_t* run_grads;
  IdMap map;
  int64_t n_ids;
  int64_t n_seg;
  int32_t dim;
  int32_t chunks;
  uint8_t lpr_log2, ids64, combiner, vec4;
  int32_t n_buckets;         // any count in [1, kMaxBuckets]: bucket = mulhi(mix64(row), count)
  int32_t tile0;             // first tile (hist / scatter grids)
  int32_t bucket0;           // first block (reduce grid)
  int32_t segtile0;          // first block (seg_of grid)
  int32_t split_t;           // pairs per workgroup of a split bucket
  int32_t e_max;             // spare workgroups of the column (>= sum over buckets of extras)
  int32_t merge0;            // first block (merge grid)
  int32_t scan0;             // first block (scan-over-tiles grid)
  int32_t n_runs;
  int32_t grad_stride;       // floats between rows of grad_out (>= dim)
  uint32_t dense_mul;        // != 0: DENSE column -- a bucket is a row RANGE, bucket =
                             // mulhi(row, dense_mul), and the reduce stage indexes its LDS tables
                             // directly with row - first row of the range (4b); 0: hashed buckets
  int32_t tpitch;            // floats between rows of table / accum (hbk_lookup_grad_column_t.table_pitch; dim by default)
  uint8_t rowsort;           // dense column whose buckets take the row-sorted reduce (4c)
  uint8_t det;               // deterministic jobs (option bwd_deterministic, lookup_bwd_rowsort.h): the one-launch
                             // grouping leaves every tile's offset inside its buckets in hist[], like the scan
  uint8_t pad_;
  uint8_t packed;            // row-sorted columns (rows < 2^32): a pair is ONE word of pair_row[],
                             // row << 32 | gradient row (segment / float offset); pair_seg[] is not used
};

struct GArgs {
  int32_t n_cols;
  float lr;
  int32_t apply;             // 0 none / SGD by lr, 2 Adagrad
  int32_t xcd;               // bit k: the reduce launch of bucket kind k deals its jobs to the XCDs
                             // in contiguous ranges (xcd_contiguous); chosen per launch on the host
  // first block of every column in the grids below (copies of the GCol fields, kept together so
  // that a block finds its column with ONE wave-wide load + ballot instead of a binary search of
  // dependent scalar loads -- each a cold miss at kernel start)
  int32_t tile0[kMaxCols];
  int32_t bucket0[kMaxCols];
  int32_t merge0[kMaxCols];
  int32_t scan0[kMaxCols];
  int32_t segtile0[kMaxCols];
  // bit k of xcd_w: the reduce launch of kind k gives XCD x the job slots [xcd_start[k][x],
  // xcd_start[k][x + 1]) -- ranges of equal WORK (mixed columns), block b = x + 8 i takes the
  // i-th of them and leaves when the range is shorter
  int32_t xcd_w;
  int32_t stage_p;           // staged scatter: LDS words per bucket array (the launch group's largest bucket count)
  int32_t xcd_start[10][9];
  GCol col[kMaxCols];
};
static_assert(sizeof(GArgs) <= 24576, "kernarg budget");
static_assert(kMaxCols <= kWave, "one lane per column in HBK_FIND_COL");
static_assert(kSlots <= 65536 && kCP <= 65536, "16-bit LDS indices");

// Reduce jobs go to the XCDs in contiguous ranges (xcd_contiguous, lookup_common.h): consecutive
// jobs are the buckets of ONE column -- dealt round robin, a column's gradient rows were read
// through all eight L2s, every 128-byte line of 64-byte rows once per XCD that meets one of its
// halves; now both halves of a line (dim 16; four rows of dim 8) and the 8 reads of a ragged
// column's segment gradients meet in one 4 MB L2 (if the line lives that long: config 2 -10 % of
// the reduce kernel's read requests, -3.5 us).
#define HBK_FIND_COL(ARGS, FIELD)                                                  \
  int ci;                                                                          \
  {                                                                                \
    const int l__ = (int)threadIdx.x & (kWave - 1);                                \
    const int v__ = l__ < (ARGS).n_cols ? (ARGS).FIELD[l__] : 0x7fffffff;          \
    ci = (int)__builtin_popcountll(__ballot(v__ <= (int)blockIdx.x)) - 1;          \
    ci = __builtin_amdgcn_readfirstlane(ci);                                       \
  }                                                                                \
  const auto& c = (ARGS).col[ci];

// 32-bit avalanche of a row number (murmur3's finalizer on the folded halves): every bit of the
// result depends on every bit of the row.  The grouping kernels hash every id twice (histogram and
// scatter) and the reduce kernel once more for its LDS table; a 64-bit mixer costs two 64x64
// multiplies = ~10 quarter-rate 32-bit multiplies per id, this one two.
__device__ inline uint32_t mix32(uint64_t row) {
  uint32_t k = (uint32_t)row ^ ((uint32_t)(row >> 32) * 0x9e3779b1u);
  k ^= k >> 16;
  k *= 0x85ebca6bu;
  k ^= k >> 13;
  k *= 0xc2b2ae35u;
  k ^= k >> 16;
  return k;
}

// bucket = high bits of the mix scaled to any bucket count; the LDS table uses the LOW bits.
// Dense columns (rows / ids small enough that a bucket's row range fits the LDS bitmaps of 4b):
// bucket = floor(row * M / 2^32), M = floor(2^32 * P / rows) -- monotone in the row, so a bucket
// is the row range [ceil(b 2^32 / M), ceil((b + 1) 2^32 / M)).
__device__ inline int bucket_of(const GCol& c, uint64_t row) {
  if (c.dense_mul != 0) return (int)__umulhi((uint32_t)row, c.dense_mul);
  return (int)__umulhi(mix32(row), (uint32_t)c.n_buckets);
}

// Segmented inputs: position j of the column -> where its id and its gradient row live.  A
// thread visits increasing j, so the cursor only moves forward (runs are thousands of ids long).
struct RunCursor {
  int k = -1;
  int64_t next = 0;         // first position of the following run
  int64_t id_delta = 0;     // id of position j: ids[j + id_delta]
  int64_t grad_delta = 0;   // gradient row of position j: grad_out + grad_delta + j * dim
};

__device__ inline void run_seek(const GCol& c, int64_t j, RunCursor& rc) {
  while (j >= rc.next) {
    ++rc.k;
    const int64_t start = c.run_start[rc.k];
    rc.next = rc.k + 1 < c.n_runs ? c.run_start[rc.k + 1] : (int64_t)1 << 62;
    rc.id_delta = c.run_ids[rc.k] - start;
    rc.grad_delta = c.run_grads[rc.k] - start * c.dim;
  }
}

template <typename V>
__device__ inline void scale_segments(const GCol& c, const float* grad, int32_t stride, int combiner,
                                      int64_t s0, int n, const int32_t* sp) {
  constexpr int VE = sizeof(V) / 4;
  constexpr int kU = 4;   // rows in flight per lane (one at a time: 32 round trips per block at dim 128)
  const int tid = (int)threadIdx.x;
  const int sub = tid & ((1 << c.lpr_log2) - 1);
  const int groups = kBlock >> c.lpr_log2;
  if (sub >= c.chunks) return;
  for (int s = tid >> c.lpr_log2; s < n; s += kU * groups) {
    V g[kU];
    int32_t len[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int ss = s + u * groups < n ? s + u * groups : n - 1;   // (clamped: straight-line loads)
      len[u] = s + u * groups < n ? sp[ss + 1] - sp[ss] : 0;
      g[u] = __builtin_nontemporal_load(reinterpret_cast<const V*>(
          grad + (s0 + ss) * (int64_t)stride + (int64_t)sub * VE));
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (len[u] <= 0) continue;   // (no id names the segment: nobody reads its row)
      const V v = combiner == HBK_COMBINER_MEAN ? g[u] / (float)len[u]
                                                : g[u] / sqrtf((float)len[u]);
      *reinterpret_cast<V*>(c.scaled + (s0 + s + u * groups) * (int64_t)c.dim + (int64_t)sub * VE) = v;
    }
  }
}

// ---- 0a: the segment of an id, found inside the grouping kernels (round 5) -----------------------
// A tile is kTile CONSECUTIVE ids of a ragged column, so the segments they belong to are one
// contiguous range [s_lo, s_hi] of the row splits.  Two waves find its ends with a 64-ary search
// over the column's splits (a wave probes 64 splits per round trip: 3 rounds for 65536 segments).
// Every segment of (s_lo, s_hi] STARTS inside the tile: the workgroup counts the starts per id
// position in LDS (one LDS atomic per segment; empty segments share a position and just add to
// its count) and ONE inclusive scan over the tile's kTile positions turns the counts into
//     segment(j) = s_lo + starts at positions <= j
// -- no search per id (a first version searched the splits per id: 9 dependent LDS reads for each
// of a thread's 8 ids made the staged scatter 135 us instead of 115).  Replaces the seg-of launch
// and its [n_ids] array (54 MB written and read per 13.6 M ids).  The LDS words are the staging
// area the tile only fills later.
// last s in [0, n_seg) with splits[s] <= j (empty segments share their start with the next one and
// are skipped by taking the last); wave-uniform result.
__device__ inline int32_t seg_search_wave(const int32_t* splits, int64_t n_seg, int32_t j) {
  const int lane = (int)threadIdx.x & (kWave - 1);
  int64_t lo = 0, n = n_seg;   // answer in [lo, lo + n)
  while (n > 1) {
    const int64_t step = (n + kWave - 1) / kWave;
    const int64_t s = lo + (int64_t)lane * step;
    const bool le = s < lo + n && splits[s] <= j;
    int k = (int)__builtin_popcountll(__ballot(le));
    if (k < 1) k = 1;   // (j before splits[0]: not a valid position; stay in range)
    const int64_t nlo = lo + (int64_t)(k - 1) * step;
    const int64_t rest = lo + n - nlo;
    lo = nlo;
    n = rest < step ? rest : step;
  }
  return (int32_t)lo;
}

// Called by every thread of the workgroup (barriers inside).  seg_lds: kTile int32 of LDS that
// end up holding segment(base + p) - s_lo for every position p of the tile; ends: 2 + kWavesPerBlock
// int32 of LDS.  Returns s_lo.  j_first / j_last: first and last id position of the tile.
__device__ inline int32_t tile_segments(const GCol& c, int64_t j_first, int64_t j_last,
                                        int32_t* seg_lds, int32_t* ends) {
  static_assert(kTile == kBlock * kPerThread && kPerThread % 4 == 0, "whole int4 per thread");
  const int tid = (int)threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  if (wave == 0) {
    const int32_t s = seg_search_wave(c.splits, c.n_seg, (int32_t)j_first);
    if (tid == 0) ends[0] = s;
  } else if (wave == 1) {
    const int32_t s = seg_search_wave(c.splits, c.n_seg, (int32_t)j_last);
    if (tid == kWave) ends[1] = s;
  }
  int4* const v4 = reinterpret_cast<int4*>(seg_lds) + tid * (kPerThread / 4);
#pragma unroll
  for (int q = 0; q < kPerThread / 4; ++q) v4[q] = make_int4(0, 0, 0, 0);
  __syncthreads();
  const int32_t s_lo = ends[0], s_hi = ends[1];
  for (int32_t s = s_lo + 1 + tid; s <= s_hi; s += kBlock) {
    const int64_t p = (int64_t)c.splits[s] - j_first;     // in (0, kTile) by the searches' ends
    if (p >= 0 && p < kTile) atomicAdd(&seg_lds[p], 1);
  }
  __syncthreads();
  // inclusive scan over the tile's positions: thread t owns [t * kPerThread, (t + 1) * kPerThread)
  int32_t x[kPerThread];
#pragma unroll
  for (int q = 0; q < kPerThread / 4; ++q) {
    const int4 v = v4[q];
    x[4 * q] = v.x;
    x[4 * q + 1] = v.y;
    x[4 * q + 2] = v.z;
    x[4 * q + 3] = v.w;
  }
#pragma unroll
  for (int q = 1; q < kPerThread; ++q) x[q] += x[q - 1];
  int32_t incl = x[kPerThread - 1];
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const int32_t y = __shfl_up(incl, o, kWave);
    if (lane >= o) incl += y;
  }
  if (lane == kWave - 1) ends[2 + wave] = incl;
  __syncthreads();
  int32_t before = incl - x[kPerThread - 1];
  for (int w = 0; w < wave; ++w) before += ends[2 + w];
#pragma unroll
  for (int q = 0; q < kPerThread / 4; ++q) {
    v4[q] = make_int4(x[4 * q] + before, x[4 * q + 1] + before, x[4 * q + 2] + before,
                      x[4 * q + 3] + before);
  }
  __syncthreads();
  return s_lo;
}

// ---- 0: segment of every id (ragged columns only; the host passes just those) --------------
// A workgroup takes kBlock consecutive segments, i.e. ONE contiguous range of ids: their row
// splits go to LDS and every thread finds the segment of ids tid, tid + kBlock, .. of the range by
// a binary search there -- consecutive lanes store consecutive ids.  (One thread per segment
// walking its own ids stored 4 bytes per lane ~32 bytes apart: 67 us for 13.6 M ids where the
// 54 MB take ~15.)
__global__ __launch_bounds__(kBlock) void bwd_segof_kernel(const GArgs a) {
  __shared__ int32_t sp[kBlock + 1];
  HBK_FIND_COL(a, segtile0)
  const int64_t s0 = ((int64_t)blockIdx.x - c.segtile0) * kBlock;
  const int tid = (int)threadIdx.x;
  const int n = c.n_seg - s0 < kBlock ? (int)(c.n_seg - s0) : kBlock;   // segments of this block
  if (tid < n) sp[tid] = c.splits[s0 + tid];
  if (tid == 0) sp[n] = c.splits[s0 + n];
  __syncthreads();
  const int32_t lo = sp[0], hi = c.seg_of != nullptr ? sp[n] : sp[0];   // (no array: scaling only)
  for (int32_t j = lo + tid; j < hi; j += kBlock) {
    // the segment s with sp[s] <= j < sp[s + 1]: the last s with sp[s] <= j (empty segments share
    // their start with the next one and are skipped by taking the last)
    int s = 0, e = n;   // answer in [s, e)
    while (e - s > 1) {
      const int mid = (s + e) >> 1;
      if (sp[mid] <= j) {
        s = mid;
      } else {
        e = mid;
      }
    }
    c.seg_of[j] = (int32_t)(s0 + s);
  }
  // mean / sqrtn: the gradient row of every segment of this block, divided ONCE (round 4).  The
  // reduce stage used to divide per PAIR -- 8 pairs per segment, four correctly rounded divisions
  // each, a third of its vector instructions on ragged columns -- and fetched the segment's length
  // with two more loads per pair; now it sums rows that are already scaled.  Same operation on the
  // same operands, done once: bit-identical results.
  if (c.scaled != nullptr) {
    if (c.vec4) {
      scale_segments<f32x4>(c, c.grad_out, c.grad_stride, c.combiner, s0, n, sp);
    } else {
      scale_segments<float>(c, c.grad_out, c.grad_stride, c.combiner, s0, n, sp);
    }
  }
}

// ---- 1: per-tile bucket histogram ---------------------------------------------------------
template <bool SIMPLE>   // (see bwd_group_kernel)
__global__ __launch_bounds__(kBlock) void bwd_hist_kernel(const GArgs a) {
  extern __shared__ int32_t counters[];
  HBK_FIND_COL(a, tile0)
  const int P = c.n_buckets;
  const int tid = (int)threadIdx.x;
  const int ctile = (int)blockIdx.x - c.tile0;
  const int64_t base = (int64_t)ctile * kTile;
  RunCursor rc;
  int64_t id[kBatch];
  auto load_batch = [&](int k0) {
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      const int64_t j = base + (int64_t)(k0 + k) * kBlock + tid;
      id[k] = 0;
      if (j < c.n_ids) {
        if (!SIMPLE && c.n_runs > 0) run_seek(c, j, rc);
        id[k] = load_id(c.ids, SIMPLE ? 1 : c.ids64, SIMPLE ? j : j + rc.id_delta);
      }
    }
  };
  load_batch(0);   // the ids travel while the counters are cleared
  for (int p = tid; p < P; p += kBlock) counters[p] = 0;
  __syncthreads();
  for (int k0 = 0; k0 < kPerThread; k0 += kBatch) {
    if (k0 > 0) load_batch(k0);
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      const int64_t j = base + (int64_t)(k0 + k) * kBlock + tid;
      if (j < c.n_ids) {
        const uint64_t r = id_to_row(c.map, id[k]);
        if (r != kNoRow) atomicAdd(&counters[SIMPLE ? (int)__umulhi((uint32_t)r, c.dense_mul) : bucket_of(c, r)], 1);
      }
    }
  }
  __syncthreads();
  for (int p = tid; p < P; p += kBlock) c.hist[(int64_t)ctile * P + p] = counters[p];
  // ragged mean / sqrtn column: this tile's share of the SEGMENTS -- their gradient rows times the
  // combiner's factor, written once for the reduce stage (0b) -- rides with the histogram launch
  // (round 5): the histogram is instruction-bound, the scaling is a 2 x n_seg x dim x 4-byte stream,
  // and the launch of its own (38 us for 26 x 65536 segments of dim 16) goes away.
  if (c.scaled != nullptr) {
    __shared__ int32_t sp[kBlock + 1];
    const int64_t n_tiles = (c.n_ids + kTile - 1) / kTile;
    const int64_t per = (c.n_seg + n_tiles - 1) / n_tiles;
    const int64_t s_begin = (int64_t)ctile * per;
    const int64_t s_end = s_begin + per < c.n_seg ? s_begin + per : c.n_seg;
    for (int64_t s0 = s_begin; s0 < s_end; s0 += kBlock) {
      const int n = s_end - s0 < kBlock ? (int)(s_end - s0) : kBlock;
      __syncthreads();   // (sp is reused)
      if (tid < n) sp[tid] = c.splits[s0 + tid];
      if (tid == 0) sp[n] = c.splits[s0 + n];
      __syncthreads();
      if (c.vec4) {
        scale_segments<f32x4>(c, c.raw_grad, c.raw_stride, c.raw_combiner, s0, n, sp);
      } else {
        scale_segments<float>(c, c.raw_grad, c.raw_stride, c.raw_combiner, s0, n, sp);
      }
    }
  }
}

// ---- 2: offsets of every (tile, bucket) run ---------------------------------------------------
// hist is [tile][bucket].  2a: one thread per bucket walks its column of the matrix (coalesced
// across the threads of a block, addresses independent of the data: loads stream) and leaves the
// exclusive prefix over tiles in place and the bucket total in bstart.  2b: one block per table
// column scans the <= 16384 bucket totals into bucket starts, clears the counters and lists the
// extra ranges of every bucket above split_t pairs (step 5).  Replaces a one-block serial walk
// over all P x tiles entries (0.4 ms per launch on the 200-column config).
// prefix over the tiles of bucket p (in place), bucket total -> bstart[p]
__device__ inline void scan_tiles_of_bucket(const GCol& c, int p) {
  const int P = c.n_buckets;
  const int n_tiles = (int)((c.n_ids + kTile - 1) / kTile);
  int32_t* h = c.hist + p;
  int32_t run = 0;
  int t = 0;
  // 16 tiles per memory round trip (the loads of a step cannot pass the stores of the step before:
  // same array; with 4 per step a 256-tile column was 64 dependent round trips, 23 us).  Round 5
  // tried, for the ragged case's 11.6 us: 64 tiles per step (11.1 us), one wave per workgroup = four
  // times the CUs (12.0), a WAVE per bucket with a shuffle scan over the tiles (two round trips per
  // 256 tiles, but every lane touches a line of its own: 28.7 us) -- a thread per bucket with
  // coalesced rows of the matrix stays.
  for (; t + 16 <= n_tiles; t += 16) {
    int32_t x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = h[(int64_t)(t + k) * P];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      h[(int64_t)(t + k) * P] = run;
      run += x[k];
    }
  }
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

__global__ __launch_bounds__(kBlock) void bwd_scan_tiles_kernel(const GArgs a) {
  HBK_FIND_COL(a, scan0)
  const int p = ((int)blockIdx.x - c.scan0) * kBlock + (int)threadIdx.x;
  if (p < c.n_buckets) scan_tiles_of_bucket(c, p);
}

// bucket totals (bstart[0..P)) -> bucket starts, job descriptors, list of extra ranges; one
// workgroup per column
__device__ inline void scan_buckets_of_column(const GCol& c, int32_t* wave_tot, int32_t* n_extra_lds) {
  int32_t& n_extra = *n_extra_lds;
  const int P = c.n_buckets;
  const int tid = (int)threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  const int per = (P + kBlock - 1) / kBlock;   // contiguous buckets per thread (<= 64)
  const int beg = tid * per;
  const int end = beg + per < P ? beg + per : P;
  if (tid == 0) n_extra = 0;
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
    c.desc[p] = make_int4(run, n_b, p, 0);
    c.pcount[p] = 0;
    if (n_b > c.split_t) {
      const int32_t extras = (n_b - 1) / c.split_t;
      const int32_t base = atomicAdd(&n_extra, extras);
      for (int32_t e = 0; e < extras; ++e) {
        c.work[2 * (base + e)] = p;
        c.work[2 * (base + e) + 1] = e + 1;
        c.desc[P + base + e] = make_int4(run, n_b, p, e + 1);
      }
    }
    run += n_b;
  }
  __syncthreads();
  for (int e = n_extra + tid; e < c.e_max; e += kBlock) c.desc[P + e] = make_int4(0, 0, -1, 0);
  if (tid == kBlock - 1) {
    c.bstart[P] = run;   // the last thread's running sum is the column total
    c.counter[0] = 0;
    c.counter[1] = 0;      // merge blocks of the column that are done (bwd_merge_kernel)
    *c.n_extra = n_extra;
  }
}

__global__ __launch_bounds__(kBlock) void bwd_scan_kernel(const GArgs a) {
  __shared__ int32_t wave_tot[kWavesPerBlock];
  __shared__ int32_t n_extra;
  scan_buckets_of_column(a.col[blockIdx.x], wave_tot, &n_extra);
}

// both steps in one launch when every column of the call is small (<= 1024 buckets, few tiles:
// a thread walks the tiles of up to four buckets): one workgroup per column
__global__ __launch_bounds__(kBlock) void bwd_scan_fused_kernel(const GArgs a) {
  __shared__ int32_t wave_tot[kWavesPerBlock];
  __shared__ int32_t n_extra;
  const GCol& c = a.col[blockIdx.x];
  for (int p = (int)threadIdx.x; p < c.n_buckets; p += kBlock) scan_tiles_of_bucket(c, p);
  __syncthreads();   // bstart[] written above is read below by other threads of this workgroup
  scan_buckets_of_column(c, wave_tot, &n_extra);
}

// ---- 3: (row, segment) pairs grouped by bucket ---------------------------------------------
// Tiles go to the XCDs in contiguous ranges (kXcdScatterBit of GArgs.xcd): a tile adds ~2 pairs to
// each of a large column's ~1000 buckets, so a 128-byte line of the pair arrays is filled by ~9
// CONSECUTIVE tiles -- dealt round robin they sit on eight XCDs and every L2 writes its own
// partial copy of the line back; on one XCD the pieces merge in its L2 first.
constexpr int kXcdScatterBit = 30;

__global__ __launch_bounds__(kBlock) void bwd_scatter_pairs_kernel(const GArgs a) {
  extern __shared__ int32_t run[];
  const int blk = xcd_contiguous((int)blockIdx.x, (int)gridDim.x, (a.xcd >> kXcdScatterBit) & 1);
  int ci;
  {
    const int l__ = (int)threadIdx.x & (kWave - 1);
    const int v__ = l__ < a.n_cols ? a.tile0[l__] : 0x7fffffff;
    ci = (int)__builtin_popcountll(__ballot(v__ <= blk)) - 1;
    ci = __builtin_amdgcn_readfirstlane(ci);
  }
  const GCol& c = a.col[ci];
  const int P = c.n_buckets;
  const int tid = (int)threadIdx.x;
  const int ctile = blk - c.tile0;
  const int64_t base = (int64_t)ctile * kTile;
  // ragged column without a seg-of array: the tile's row splits in LDS (0a)
  __shared__ int32_t sp_lds[kTile];
  __shared__ int32_t seg_ends[2 + kWavesPerBlock];
  const bool seg_inline = c.splits != nullptr && c.seg_of == nullptr;   // block-uniform
  int32_t s_lo = 0;
  if (seg_inline) {
    const int64_t j_last = (base + kTile < c.n_ids ? base + kTile : c.n_ids) - 1;
    s_lo = tile_segments(c, base, j_last, sp_lds, seg_ends);
  }
  RunCursor rc;
  int64_t id[kBatch];
  int32_t seg[kBatch];
  auto load_batch = [&](int k0) {
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      const int64_t j = base + (int64_t)(k0 + k) * kBlock + tid;
      id[k] = 0;
      seg[k] = (int32_t)j;
      if (j < c.n_ids) {
        if (c.n_runs > 0) {
          // the pair carries the float offset of its gradient row (host checks < 2^32)
          run_seek(c, j, rc);
          if (!c.det) seg[k] = (int32_t)(uint32_t)(rc.grad_delta + j * c.dim);   // (deterministic: the position, see rowsort_reduce)
        }
        id[k] = load_id(c.ids, c.ids64, j + rc.id_delta);
        if (c.seg_of != nullptr) {
          seg[k] = c.seg_of[j];
        } else if (seg_inline) {
          seg[k] = s_lo + sp_lds[(k0 + k) * kBlock + tid];
        }
      }
    }
  };
  load_batch(0);   // the ids travel beside the bucket offsets
  for (int p = tid; p < P; p += kBlock) run[p] = c.bstart[p] + c.hist[(int64_t)ctile * P + p];
  __syncthreads();
  for (int k0 = 0; k0 < kPerThread; k0 += kBatch) {
    if (k0 > 0) load_batch(k0);
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      const int64_t j = base + (int64_t)(k0 + k) * kBlock + tid;
      if (j < c.n_ids) {
        const uint64_t r = id_to_row(c.map, id[k]);
        if (r != kNoRow) {
          const int32_t pos = atomicAdd(&run[bucket_of(c, r)], 1);
          if (c.packed) {
            c.pair_row[0][pos] = (int64_t)((r << 32) | (uint64_t)(uint32_t)seg[k]);
          } else {
            c.pair_row[0][pos] = (int64_t)r;
            c.pair_seg[0][pos] = seg[k];
          }
        }
      }
    }
  }
}

// ---- 3b: the same, staged through LDS (round 4) ---------------------------------------------------
// Launch groups whose columns all have <= kStageMaxBuckets buckets (every row-sorted column: ~300
// buckets per 524288 ids): the tile's pairs are sorted by bucket in LDS first and leave in that
// order -- consecutive lanes store consecutive positions of a bucket's run.  The direct scatter
// above issues 64 requests of 8 / 4 bytes per store instruction (every lane another bucket, i.e.
// another line): 27 M requests for the 13.6 M pairs of the ragged case, which is what its 180 us
// were -- the L2's request rate, not bytes.
constexpr int kStageMaxBuckets = 1024;

// LDS (round 5): the bucket arrays are sized by the launch group's largest bucket count and the
// segment staging exists only for groups with unpacked columns -- 36.9 KB (4 workgroups per CU;
// the kernel needs 56 VGPRs) became 16 + 4 + 8 P / 1024 KB: ~22.5 KB at 300 buckets = 7 per CU.
// The kernel is bound by the life of a tile (loads -> LDS tickets -> barrier -> stage -> barrier ->
// stores), not by bytes: more resident tiles are throughput.
template <bool SIMPLE>   // (see bwd_group_kernel)
__global__ __launch_bounds__(kBlock, 4) void bwd_scatter_staged_kernel(const GArgs a) {
  extern __shared__ int32_t stage_dyn[];
  int32_t* const counters = stage_dyn;             // [stage_p] pairs of the tile per bucket; then: global
                                                   // position of staged slot L of the bucket - L
  int32_t* const first = stage_dyn + a.stage_p;    // [stage_p] first staged slot of the bucket
  int32_t* const st_seg = stage_dyn + 2 * a.stage_p;   // [kTile], only in groups with unpacked columns
  __shared__ int64_t st_row[kTile];                // the tile's pairs, sorted by bucket
  // (SIMPLE: a staged pair's bucket is recomputed from its row -- 4 KB of LDS less: 8 resident tiles per CU instead of 7)
  __shared__ uint16_t st_b[SIMPLE ? 1 : kTile];
  __shared__ int32_t wave_cnt[kWavesPerBlock];
  __shared__ int32_t n_staged;
  const int blk = xcd_contiguous((int)blockIdx.x, (int)gridDim.x, (a.xcd >> kXcdScatterBit) & 1);
  int ci;
  {
    const int l__ = (int)threadIdx.x & (kWave - 1);
    const int v__ = l__ < a.n_cols ? a.tile0[l__] : 0x7fffffff;
    ci = (int)__builtin_popcountll(__ballot(v__ <= blk)) - 1;
    ci = __builtin_amdgcn_readfirstlane(ci);
  }
  const GCol& c = a.col[ci];
  const int P = c.n_buckets;
  const int tid = (int)threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  const int ctile = blk - c.tile0;
  const int64_t base = (int64_t)ctile * kTile;
  // ragged column without a seg-of array: the tile's row splits go to LDS (0a) -- into the staging
  // area, which is only filled after the segments have been found
  __shared__ int32_t seg_ends[2 + kWavesPerBlock];
  int32_t* const sp_lds = reinterpret_cast<int32_t*>(st_row);
  const bool seg_inline = c.splits != nullptr && (SIMPLE || c.seg_of == nullptr);   // block-uniform
  // everything the tile needs from memory is requested up front -- the ids and where the tile's share
  // of every bucket starts in the pair arrays -- so the segment search below runs under those loads
  // instead of in front of them.  (Tried: the histogram launch leaves every tile's first / last
  // segment for this kernel, which then starts with one round trip instead of four -- no gain at 7
  // resident tiles per CU, and the histogram launch pays for the search: 494 vs 493 us.)
  RunCursor rc;
  int64_t id[kPerThread];
  int32_t seg[kPerThread];
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const int64_t j = base + (int64_t)k * kBlock + tid;
    id[k] = 0;
    seg[k] = (int32_t)j;
    if (j < c.n_ids) {
      if (!SIMPLE && c.n_runs > 0) {
        run_seek(c, j, rc);
        if (!c.det) seg[k] = (int32_t)(uint32_t)(rc.grad_delta + j * c.dim);   // (deterministic: the position, see rowsort_reduce)
      }
      id[k] = load_id(c.ids, SIMPLE ? 1 : c.ids64, SIMPLE ? j : j + rc.id_delta);
      if (!SIMPLE && c.seg_of != nullptr) seg[k] = c.seg_of[j];
    }
  }
  const int per = (P + kBlock - 1) / kBlock;   // <= 4
  const int beg = tid * per;
  const int end = beg + per < P ? beg + per : P;
  int32_t gpos[kStageMaxBuckets / kBlock];
#pragma unroll
  for (int q = 0; q < kStageMaxBuckets / kBlock; ++q) {
    const int p = beg + q;
    gpos[q] = q < per && p < end ? c.bstart[p] + c.hist[(int64_t)ctile * P + p] : 0;
  }
  if (seg_inline) {
    const int64_t j_last = (base + kTile < c.n_ids ? base + kTile : c.n_ids) - 1;
    const int32_t s_lo = tile_segments(c, base, j_last, sp_lds, seg_ends);
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) seg[k] = s_lo + sp_lds[k * kBlock + tid];
  }   // (sp_lds is the staging area: the barriers below come before its first staged pair)
  for (int p = tid; p < P; p += kBlock) counters[p] = 0;
  __syncthreads();
  int32_t br[kPerThread];   // bucket | rank << 10, -1: no row
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const int64_t j = base + (int64_t)k * kBlock + tid;
    br[k] = -1;
    if (j < c.n_ids) {
      const uint64_t r = id_to_row(c.map, id[k]);
      id[k] = (int64_t)r;
      if (r != kNoRow) {
        const int b = SIMPLE ? (int)__umulhi((uint32_t)r, c.dense_mul) : bucket_of(c, r);
        br[k] = b | (atomicAdd(&counters[b], 1) << 10);
      }
    }
  }
  __syncthreads();
  // first staged slot of every bucket: scan of the tile's counts (thread t: buckets [beg, end))
  int32_t sum_c = 0;
#pragma unroll
  for (int q = 0; q < kStageMaxBuckets / kBlock; ++q) {
    if (q < per && beg + q < end) sum_c += counters[beg + q];
  }
  int32_t incl_c = sum_c;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const int32_t w = __shfl_up(incl_c, off, kWave);
    if (lane >= off) incl_c += w;
  }
  if (lane == kWave - 1) wave_cnt[wave] = incl_c;
  __syncthreads();
  int32_t run_c = incl_c - sum_c;
  for (int w = 0; w < wave; ++w) run_c += wave_cnt[w];
#pragma unroll
  for (int q = 0; q < kStageMaxBuckets / kBlock; ++q) {
    const int p = beg + q;
    if (q < per && p < end) {
      const int32_t n_c = counters[p];   // (read and replaced by the one thread that owns p)
      first[p] = run_c;
      counters[p] = gpos[q] - run_c;
      run_c += n_c;
    }
  }
  if (tid == kBlock - 1) n_staged = run_c;   // the last thread's running count: pairs of the tile
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    if (br[k] >= 0) {
      const int b = br[k] & 1023;
      const int L = first[b] + (br[k] >> 10);
      if (SIMPLE || c.packed) {   // (block-uniform) one word per pair: row << 32 | gradient row
        st_row[L] = (int64_t)(((uint64_t)id[k] << 32) | (uint64_t)(uint32_t)seg[k]);
      } else {
        st_row[L] = id[k];
        st_seg[L] = seg[k];
      }
      if (!SIMPLE) st_b[L] = (uint16_t)b;
    }
  }
  __syncthreads();
  const int n_st = n_staged;
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const int L = k * kBlock + tid;
    if (L < n_st) {
      const int sb = SIMPLE ? (int)__umulhi((uint32_t)((uint64_t)st_row[L] >> 32), c.dense_mul) : (int)st_b[L];
      const int32_t pos = counters[sb] + L;
      c.pair_row[0][pos] = st_row[L];
      if (!SIMPLE && !c.packed) c.pair_seg[0][pos] = st_seg[L];
    }
  }
}

// ---- 1-3 in one launch ---------------------------------------------------------------------------
// Columns of <= 512 buckets and <= 64 tiles (every column of a 65536-id step): a tile keeps its
// rows, their buckets and their ranks inside the tile's share of the bucket (what the LDS atomic
// returns) in registers, publishes its bucket counts (count + 1 into words that read zero when
// the kernel starts, sync.hip) and waits for the other tiles of its column -- all resident:
// workgroups start in index order, so when the dispatcher stalls every waiting workgroup belongs
// to the one column that is not fully started, <= 64 of the chip's > 1000 slots -- then derives
// the bucket starts and its own offsets from all of them and scatters from registers.  Tile 0 of a
// column also writes what the scan kernel wrote: bucket starts, job descriptors, extra ranges of
// split buckets, the cleared counters.  The ids are read once and two launches disappear
// (hist 9 us + scan 6 us + scatter 26 us -> measured in profiles/).
#ifdef HBK_BWD_STAMPS
__device__ unsigned long long g_grp_trace[kTraceBlocks * kTraceSlots];
#define HBK_GSTAMP(i)                                                                       \
  do {                                                                                      \
    if (threadIdx.x == 0 && blockIdx.x < kTraceBlocks) {                                    \
      g_grp_trace[blockIdx.x * kTraceSlots + (i)] = __builtin_amdgcn_s_memrealtime();       \
    }                                                                                       \
  } while (0)
#else
#define HBK_GSTAMP(i)
#endif

constexpr int kGroupMaxBuckets = 2 * kBlock;   // (64 tiles of 2048 ids at ~448 per bucket: 293)

struct GSync {
  int32_t* hist;        // per column [tiles][n_buckets] words: 0 = not published, else count + 1
  int32_t* zero;        // words the call before left set
  int64_t zero_words;
  SyncWait wait;        // bound of the waits, status / poison words, test hook
};

// (5 waves per SIMD: all 832 workgroups of a 26 x 65536 call resident at once; at 145 VGPRs the
// last 64 started 21 us late and the kernel took as long as the three launches it replaces;
// with the staged scatter's 28 KB of LDS four workgroups fit a CU: 1024 slots)
// SIMPLE (the host has looked at every column of the launch group): no segmented inputs, no seg-of
// arrays, packed pairs, int64 ids, row-range buckets -- config 2, the benchmark's ragged columns:
// those questions are constants in this instantiation.
template <bool SIMPLE>
__global__ __launch_bounds__(kBlock, 4) void bwd_group_kernel(const GArgs a, const GSync y) {
  __shared__ int32_t counters[kGroupMaxBuckets];   // pairs of the tile per bucket
  __shared__ int32_t tot_s[kGroupMaxBuckets], pre_s[kGroupMaxBuckets];   // bucket totals / before
                                                   // this tile; then position deltas / first slots
  __shared__ int64_t st_row[kTile];                          // the tile's pairs, sorted by bucket
  extern __shared__ int32_t st_seg[];                        // [kTile], only in groups with unpacked columns
  __shared__ uint16_t st_b[kTile];
  __shared__ int32_t wave_tot[kWavesPerBlock], wave_cnt[kWavesPerBlock];
  __shared__ int32_t n_extra, gave_up, n_staged;
  const int tid = (int)threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  HBK_GSTAMP(0);
  for (int64_t j = (int64_t)blockIdx.x * kBlock + tid; j < y.zero_words;
       j += (int64_t)gridDim.x * kBlock) {
    y.zero[j] = 0;
  }
  HBK_FIND_COL(a, tile0)
  const int P = c.n_buckets;
  const int ctile = (int)blockIdx.x - c.tile0;
  const int n_tiles = (int)((c.n_ids + kTile - 1) / kTile);
  const int64_t base = (int64_t)ctile * kTile;
  int32_t* hist = y.hist + c.sync0;
  // ragged column without a seg-of array: the tile's row splits in LDS (0a; the staging area is
  // only filled after the wait)
  __shared__ int32_t seg_ends[2 + kWavesPerBlock];
  int32_t* const sp_lds = reinterpret_cast<int32_t*>(st_row);
  const bool seg_inline = c.splits != nullptr && (SIMPLE || c.seg_of == nullptr);   // block-uniform
  // (the ids are requested before the segment search: they travel under it)
  RunCursor rc;
  int64_t id[kPerThread];
  int32_t seg[kPerThread];
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const int64_t j = base + (int64_t)k * kBlock + tid;
    id[k] = 0;
    seg[k] = (int32_t)j;
    if (j < c.n_ids) {
      if (!SIMPLE && c.n_runs > 0) {
        run_seek(c, j, rc);
        if (!c.det) seg[k] = (int32_t)(uint32_t)(rc.grad_delta + j * c.dim);   // (deterministic: the position, see rowsort_reduce)
      }
      id[k] = load_id(c.ids, SIMPLE ? 1 : c.ids64, SIMPLE ? j : j + rc.id_delta);
      if (!SIMPLE && c.seg_of != nullptr) seg[k] = c.seg_of[j];
    }
  }
  if (seg_inline) {
    const int64_t j_last = (base + kTile < c.n_ids ? base + kTile : c.n_ids) - 1;
    const int32_t s_lo = tile_segments(c, base, j_last, sp_lds, seg_ends);
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) seg[k] = s_lo + sp_lds[k * kBlock + tid];
  }
  for (int p = tid; p < P; p += kBlock) counters[p] = 0;
  if (tid == 0) {
    n_extra = 0;
    gave_up = 0;
  }
  __syncthreads();
  HBK_GSTAMP(1);            // counters cleared, loads issued
  int32_t br[kPerThread];   // bucket | rank << 10, -1: no row
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const int64_t j = base + (int64_t)k * kBlock + tid;
    br[k] = -1;
    if (j < c.n_ids) {
      const uint64_t r = id_to_row(c.map, id[k]);
      id[k] = (int64_t)r;
      if (r != kNoRow) {
        const int b = SIMPLE ? (int)__umulhi((uint32_t)r, c.dense_mul) : bucket_of(c, r);
        br[k] = b | (atomicAdd(&counters[b], 1) << 10);
      }
    }
  }
  __syncthreads();
  HBK_GSTAMP(2);            // ids arrived, ranks taken
  if ((int)blockIdx.x != y.wait.withhold) {
    for (int p = tid; p < P; p += kBlock) {
      __hip_atomic_store(hist + (int64_t)ctile * P + p, counters[p] + 1, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  HBK_GSTAMP(3);            // published
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
  HBK_GSTAMP(4);            // the column's counts are in
  if (gave_up != 0) {
    if (tid == 0) give_up(y.wait);
    return;
  }
  // bucket starts (scan of the column's totals) and, for the staged scatter, the tile's own
  // bucket offsets (scan of its counts): thread t scans buckets [t * per, t * per + per)
  const int per = (P + kBlock - 1) / kBlock;   // <= 4
  const int beg = tid * per;
  const int end = beg + per < P ? beg + per : P;
  int32_t sum = 0, sum_c = 0;
  for (int p = beg; p < end; ++p) {
    sum += tot_s[p];
    sum_c += counters[p];
  }
  int32_t incl = sum, incl_c = sum_c;
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
  int32_t run = incl - sum, run_c = incl_c - sum_c;
  for (int w = 0; w < wave; ++w) {
    run += wave_tot[w];
    run_c += wave_cnt[w];
  }
  const bool first_tile = ctile == 0;
  for (int p = beg; p < end; ++p) {
    const int32_t n_b = tot_s[p];
    const int32_t n_c = counters[p];
    if (c.det) c.hist[(int64_t)ctile * P + p] = pre_s[p];   // where the tile's share of the bucket begins
    tot_s[p] = run + pre_s[p] - run_c;   // global position of the tile's pair at staged slot L: + L
    pre_s[p] = run_c;                    // first staged slot of the bucket
    if (first_tile) {
      c.bstart[p] = run;
      c.desc[p] = make_int4(run, n_b, p, 0);
      c.pcount[p] = 0;
      if (n_b > c.split_t) {
        const int32_t extras = (n_b - 1) / c.split_t;
        const int32_t e0 = atomicAdd(&n_extra, extras);
        for (int32_t e = 0; e < extras; ++e) {
          c.work[2 * (e0 + e)] = p;
          c.work[2 * (e0 + e) + 1] = e + 1;
          c.desc[P + e0 + e] = make_int4(run, n_b, p, e + 1);
        }
      }
    }
    run += n_b;
    run_c += n_c;
  }
  if (tid == kBlock - 1) n_staged = run_c;   // the last thread's running count: pairs of the tile
  __syncthreads();
  if (first_tile) {
    for (int e = n_extra + tid; e < c.e_max; e += kBlock) c.desc[P + e] = make_int4(0, 0, -1, 0);
    if (tid == kBlock - 1) {
      c.bstart[P] = run;   // the last thread's running sum is the column total
      c.counter[0] = 0;
      c.counter[1] = 0;    // merge blocks of the column that are done (bwd_merge_kernel)
      *c.n_extra = n_extra;
    }
  }
  HBK_GSTAMP(5);            // offsets known
  // The tile's pairs go through LDS sorted by bucket and leave in that order: consecutive lanes
  // store consecutive positions of a bucket's run (~14 pairs = 112 + 56 bytes) instead of 64
  // different buckets, i.e. 64 different lines, per store instruction (the direct scatter spent
  // 6.3 us of the tile's 22 us issuing its 16 store instructions).
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    if (br[k] >= 0) {
      const int b = br[k] & 1023;
      const int L = pre_s[b] + (br[k] >> 10);
      if (SIMPLE || c.packed) {   // (block-uniform) one word per pair: row << 32 | gradient row
        st_row[L] = (int64_t)(((uint64_t)id[k] << 32) | (uint64_t)(uint32_t)seg[k]);
      } else {
        st_row[L] = id[k];
        st_seg[L] = seg[k];
      }
      st_b[L] = (uint16_t)b;
    }
  }
  __syncthreads();
  const int n_st = n_staged;
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const int L = k * kBlock + tid;
    if (L < n_st) {
      const int32_t pos = tot_s[st_b[L]] + L;
      c.pair_row[0][pos] = st_row[L];
      if (!SIMPLE && !c.packed) c.pair_seg[0][pos] = st_seg[L];
    }
  }
  HBK_GSTAMP(6);            // stores issued
#ifdef HBK_BWD_STAMPS
  __builtin_amdgcn_s_waitcnt(0x0f70);
  HBK_GSTAMP(7);            // stores landed
#endif
}

// ---- 4: one workgroup per bucket ---------------------------------------------------------------
struct ReduceLds {
  unsigned long long keys[kSlots];
  int32_t cnt[kSlots];       // tickets = pairs of the slot in this chunk; bit 30: row is new in it
  int32_t off[kSlots];       // multi-pair slots: start of the slot's pairs in `order` (-> end by (d))
  int32_t slot_out[kSlots];  // output row of the slot, -1 = none yet
  int32_t segs[kCP];         // segment (= row of grad_out) of every pair of the chunk
  int32_t nseg[kCP];         // ragged columns with mean / sqrtn: ids of that segment (the divisor)
  uint16_t active[kCP];      // slots with several pairs in this chunk
  uint16_t order[kCP];       // their pair indices grouped by slot
  uint16_t pslot[kCP];       // slot of every pair; kNoSlot: struck out earlier / left for a later pass
  uint16_t emitted[kSlots];  // slots of the rows this pass has emitted (jobs of several chunks)
  int32_t wave_tot[kTeam / kWave];
  int32_t n_active, n_new, n_single, base_u, occupied, occupied_before, n_left, lds_rows,
      n_emitted, emit0, n_multi;
  float red[kTeam * 4];      // partial sums handed between lane groups, one 16-byte chunk per thread
  float carry[2][kWave * 4]; // sum of the slot that runs on into the next round of the sorted walk
};

constexpr int32_t kNewBit = 1 << 30;

// What a workgroup reduces and where the result goes: the pairs of a bucket into the final
// IndexedSlices, a range of a split bucket into its partial entries, or those entries into the
// final IndexedSlices (merge).
struct ReduceJob {
  int64_t* prow;             // rows of the pairs (handled pairs are struck out when a bucket needs
                             // more than one pass)
  const int32_t* pseg;       // their gradient rows; NULL: pair i reads gradient row i
  const float* grad;         // [*, dim]
  int32_t n_pairs;
  bool scale;                // apply the combiner's 1/n, 1/sqrt(n)
  bool seg_is_offset;        // pseg holds float offsets into grad (segmented inputs)
  int32_t stride;            // floats between rows of grad
  int64_t* out_rows;
  float* out_vals;
  int32_t* out_counter;      // claimed with one atomic per chunk
  int32_t out_base;          // added to the claimed index
  float lr;                  // != 0: fused optimizer step on the table row
  int32_t apply;             // HBK_APPLY_SGD | HBK_APPLY_ADAGRAD
  bool packed;               // prow holds row << 32 | gradient row (GCol.packed); pseg is not read
  bool no_emit;              // "step only" (the caller wants no IndexedSlices) and the job is one
                             // chunk: rows are stepped where their sums sit in registers, nothing
                             // is written out and no output range is claimed
};

// How a gradient chunk is loaded: PLAIN.  The non-temporal hint these loads carried through
// round 2 tells L2 not to keep the line -- but a 128-byte line holds two rows of dim 16 (four of
// dim 8), asked for by two different buckets a few microseconds apart, and a ragged column reads
// every segment gradient ~8 times: with plain loads (and whole columns per XCD, xcd_contiguous)
// config 2 100.5 -> 91.7 us, SGD step only 150 -> 131.5, ragged dim 16 / 64 + SGD 774 -> 704 /
// 1367 -> 1269, over 10 M rows 1037 -> 901 (tools/scratch/bwd_cache_variants.sh, two visits each).
// Write-through stores of the output rows (so that they do not push gradient lines out of L2)
// change nothing (HBK_BWD_OUT_WT).  -DHBK_BWD_GRAD_NT=1 brings the hint back (probe builds).
#ifndef HBK_BWD_GRAD_NT
#define HBK_BWD_GRAD_NT 0
#endif
#if HBK_BWD_GRAD_NT
#define HBK_GRAD_LOAD(P) __builtin_nontemporal_load(P)
#else
#define HBK_GRAD_LOAD(P) (*(P))
#endif

// the pairs of a bucket are read once (probe builds: -DHBK_BWD_PAIRS_NT=1 non-temporal loads)
#ifndef HBK_BWD_PAIRS_NT
#define HBK_BWD_PAIRS_NT 0
#endif
#if HBK_BWD_PAIRS_NT
#define HBK_PAIR_LOAD(P) __builtin_nontemporal_load(P)
#else
#define HBK_PAIR_LOAD(P) (*(P))
#endif

// table / accumulator rows of the optimizer step (probe builds: -DHBK_BWD_STEP_NT=0 plain loads)
#ifndef HBK_BWD_STEP_NT
#define HBK_BWD_STEP_NT 1
#endif
#if HBK_BWD_STEP_NT
#define HBK_STEP_LOAD(P) __builtin_nontemporal_load(P)
#else
#define HBK_STEP_LOAD(P) (*(P))
#endif

template <typename V>
__device__ inline V load_grad(const GCol& c, const ReduceJob& job, int32_t seg, int sub) {
  constexpr int VE = sizeof(V) / 4;
  const int64_t off = job.seg_is_offset ? (int64_t)(uint32_t)seg : (int64_t)seg * job.stride;
#ifdef HBK_BWD_ABLATE_LOADS   // probe builds: what the kernel costs without its gradient traffic
  V g = zero_v<V>() + (float)(off & 7);
#else
  V g = HBK_GRAD_LOAD(reinterpret_cast<const V*>(job.grad + off + (int64_t)sub * VE));
#endif
  if (job.scale && c.combiner != HBK_COMBINER_SUM && c.splits != nullptr) {
    const int32_t n = c.splits[seg + 1] - c.splits[seg];
    g = c.combiner == HBK_COMBINER_MEAN ? g / (float)n : g / sqrtf((float)n);
  }
  return g;
}

// The gradient chunk of a pair with the combiner's divisor taken from LDS (hashed path: the
// segment lengths of a chunk are fetched once, in (a)): no memory wait between two gradient loads.
template <typename V>
__device__ inline V load_grad_lds(const GCol& c, const ReduceJob& job, int32_t seg, int sub,
                                  bool scaled, int32_t n) {
  constexpr int VE = sizeof(V) / 4;
  const int64_t off = job.seg_is_offset ? (int64_t)(uint32_t)seg : (int64_t)seg * job.stride;
#ifdef HBK_BWD_ABLATE_LOADS
  V g = zero_v<V>() + (float)(off & 7);
#else
  V g = HBK_GRAD_LOAD(reinterpret_cast<const V*>(job.grad + off + (int64_t)sub * VE));
#endif
  if (scaled) g = c.combiner == HBK_COMBINER_MEAN ? g / (float)n : g / sqrtf((float)n);
  return g;
}

// The same in two steps: the gradient chunk and -- ragged columns with mean / sqrtn -- the length
// of its segment are REQUESTED here; the division waits (scale_grad) until the caller has issued
// all its loads.  (load_grad divides at once: a wait for memory between any two loads of a round.)
template <typename V>
__device__ inline V load_grad_raw(const GCol& c, const ReduceJob& job, int32_t seg, int sub,
                                  int32_t* n) {
  constexpr int VE = sizeof(V) / 4;
  const int64_t off = job.seg_is_offset ? (int64_t)(uint32_t)seg : (int64_t)seg * job.stride;
  const V g = HBK_GRAD_LOAD(reinterpret_cast<const V*>(job.grad + off + (int64_t)sub * VE));
  *n = 0;
  if (job.scale && c.combiner != HBK_COMBINER_SUM && c.splits != nullptr) {
    *n = c.splits[seg + 1] - c.splits[seg];
  }
  return g;
}
template <typename V>
__device__ inline V scale_grad(const GCol& c, V g, int32_t n) {
  if (n == 0) return g;
  return c.combiner == HBK_COMBINER_MEAN ? g / (float)n : g / sqrtf((float)n);
}

// The sparse optimizer step of one row chunk (this workgroup owns the row):
//   SGD      var -= lr * g                                        (GradientDescentOptimizer)
//   Adagrad  accum += g * g;  var -= lr * g * (1 / sqrt(accum))   (AdagradOptimizer's sparse apply,
//            docs/tutorial/ranking/taobao/train.py:115; g = the row's deduplicated gradient)
template <typename V>
__device__ inline V rsqrt_v(V a);
template <>
__device__ inline float rsqrt_v<float>(float a) { return 1.0f / sqrtf(a); }
template <>
__device__ inline f32x4 rsqrt_v<f32x4>(f32x4 a) {
  return f32x4{1.0f / sqrtf(a.x), 1.0f / sqrtf(a.y), 1.0f / sqrtf(a.z), 1.0f / sqrtf(a.w)};
}

// A row chunk stored write-through (sc0 sc1): the line does not stay in the XCD's L2 (probe builds,
// -DHBK_BWD_OUT_WT=1: do the output rows push the gradient lines out of L2?)
#ifndef HBK_BWD_OUT_WT
#define HBK_BWD_OUT_WT 0
#endif
__device__ inline void store_wt(f32x4* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ inline void store_wt(float* p, float v) {
  asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}

// out row u += (or =) v
template <typename V>
__device__ inline void emit_row(const GCol& c, const ReduceJob& job, int32_t u, bool is_new,
                                int sub, V v) {
  constexpr int VE = sizeof(V) / 4;
  V* o = reinterpret_cast<V*>(job.out_vals + (int64_t)u * c.dim + (int64_t)sub * VE);
#ifdef HBK_BWD_ABLATE_STORES   // probe builds: what the kernel costs without its output traffic
  if (u == 0x7fffffff) *o = v;
  (void)is_new;
#else
  // rows are owned by this workgroup; bypass L1 when re-reading what an earlier chunk wrote
#if HBK_BWD_OUT_WT
  if (is_new) {
    store_wt(o, v);
  } else {
    *o = __builtin_nontemporal_load(o) + v;
  }
#else
  *o = is_new ? v : __builtin_nontemporal_load(o) + v;
#endif
#endif
}

// The sparse optimizer step of one row chunk (this workgroup owns the row), from values already
// in registers: g = the row's deduplicated gradient, tv / av = table / accumulator row
//   SGD      var -= lr * g                                        (GradientDescentOptimizer)
//   Adagrad  accum += g * g;  var -= lr * g * (1 / sqrt(accum))   (AdagradOptimizer's sparse apply,
//            docs/tutorial/ranking/taobao/train.py:115)
template <typename V>
__device__ inline void step_row(const GCol& c, bool adagrad, float lr, int64_t toff, V g, V tv, V av) {
  if (adagrad) {
    const V acc = av + g * g;
    *reinterpret_cast<V*>(c.accum + toff) = acc;
    *reinterpret_cast<V*>(c.table + toff) = tv - (lr * g) * rsqrt_v<V>(acc);
  } else {
    *reinterpret_cast<V*>(c.table + toff) = tv - lr * g;
  }
}

// emit + (lr != 0) the optimizer step with the loads it needs: the rows of the
// LDS-row path -- few, or long to sum, so the extra round trip is not on the critical path
template <typename V, int STEP>
__device__ inline void emit_step_row(const GCol& c, const ReduceJob& job, float lr, int32_t u,
                                     bool is_new, int64_t row, int sub, V v) {
  constexpr int VE = sizeof(V) / 4;
  if (!(STEP && job.no_emit)) emit_row<V>(c, job, u, is_new, sub, v);
  if (STEP && lr != 0.0f) {
    constexpr bool adagrad = STEP == 2;
    const int64_t toff = row * c.tpitch + (int64_t)sub * VE;
    const V tv = HBK_STEP_LOAD(reinterpret_cast<const V*>(c.table + toff));
    V av = zero_v<V>();
    if (adagrad) av = HBK_STEP_LOAD(reinterpret_cast<const V*>(c.accum + toff));
    step_row<V>(c, adagrad, lr, toff, v, tv, av);
  }
}

// Find `row` in the LDS table or enter it while the table has room.  Returns the slot, or -1 when
// the row is absent and the table takes no more rows (the pair waits for the next pass).
// `entered` is set when the row is new in the table: the CALLER adds the new rows of a wave to
// L.occupied with one atomic (every lane adding its own 1 to that one LDS word serialised 512
// same-address atomics per chunk: 4.4 of a workgroup's 20 us).
__device__ inline int table_slot(ReduceLds& L, unsigned long long row, bool* entered) {
  int h = (int)(mix32(row) & (kSlots - 1));
  for (;;) {
    const unsigned long long k = L.keys[h];
    if (k == row) return h;
    if (k == kEmptyKey) {
      // racy read of the fill level (the waves add their new rows after every pair of the chunk
      // loop): concurrent inserts overshoot kRoom by < kTeam rows, the table keeps >= 8 empty
      // slots, so every probe sequence ends
      if (*(volatile int32_t*)&L.occupied >= kRoom) return -1;
      const unsigned long long prev = atomicCAS(&L.keys[h], kEmptyKey, row);
      if (prev == kEmptyKey) {
        *entered = true;
        return h;
      }
      if (prev == row) return h;
    }
    h = (h + 1) & (kSlots - 1);
  }
}

// Look `row` up without entering it (the table is still: no insert runs beside this).
__device__ inline int table_find(const ReduceLds& L, unsigned long long row) {
  int h = (int)(mix32(row) & (kSlots - 1));
  for (;;) {
    const unsigned long long k = L.keys[h];
    if (k == row) return h;
    if (k == kEmptyKey) return -1;
    h = (h + 1) & (kSlots - 1);
  }
}

// STEP: 0 = no fused optimizer step (none of its loads, registers and branches), 1 = SGD,
// 2 = Adagrad (accumulator rows as well).  WIDE (only with a step): the columns of >= 16 lanes per
// row, whose sorted walk keeps the full width -- an instantiation of its own so that neither form
// carries the other's registers.
template <typename V, int STEP, bool WIDE = false>
__device__ inline void bucket_reduce(const GCol& c, const ReduceJob& job, ReduceLds& L) {
  constexpr int VE = sizeof(V) / 4;
  const int tid = (int)threadIdx.x & (kTeam - 1);   // inside the team
  const int lane = tid & (kWave - 1), wave = tid >> 6;
  const int lpr_log2 = c.lpr_log2;
  const int sub = lane & ((1 << lpr_log2) - 1);
  const bool live = sub < c.chunks;
  const int groups = kTeam >> lpr_log2 > 0 ? kTeam >> lpr_log2 : 1;
  const int my_group = tid >> lpr_log2;

  const int32_t n_pairs = job.n_pairs;
  if (n_pairs <= 0) return;
  int64_t* prow = job.prow;
  const int32_t* pseg = job.pseg;

  team_sync();   // a workgroup may run several jobs: the previous one is done with the table
  for (int i = tid; i < kSlots; i += kTeam) {
    L.keys[i] = kEmptyKey;
    L.slot_out[i] = -1;
    L.cnt[i] = 0;
  }
  if (tid == 0) {
    L.occupied = 0;
    L.occupied_before = 0;
    L.n_left = 0;
  }
  team_sync();
  HBK_STAMP(2);

  // The optimizer step is taken once per row.  A job of one chunk (almost all: buckets aim at 7/8
  // of a chunk) steps a row right where its sum sits in registers.  In a job of several chunks a
  // row may span chunks and is summed across them in its output row; stepping chunk by chunk
  // would round differently from table -= lr * grad_row and is wrong for Adagrad, so such a job
  // only emits, lists the rows a pass emitted, and steps them after the pass.
  // the combiner's 1/n, 1/sqrt(n) applies (ragged column, mean / sqrtn, not a merge of partials)
  const bool scaled = job.scale && c.combiner != HBK_COMBINER_SUM && c.splits != nullptr;
  const bool defer = STEP && job.lr != 0.0f && n_pairs > kCP;
  const float lr_now = !STEP || defer ? 0.0f : job.lr;
  constexpr bool adagrad = STEP == 2;
  if (tid == 0) L.n_emitted = 0;
  int32_t first_cb = 0;        // first chunk that may still hold unhandled pairs
  bool striking = false;       // this pass left pairs behind: handled pairs are struck out
  for (;;) {                   // passes; one unless the bucket holds more rows than the table
    int32_t first_left = -1;
    for (int32_t cb = first_cb; cb < n_pairs; cb += kCP) {
      const int32_t n_chunk = n_pairs - cb < kCP ? n_pairs - cb : kCP;
      if (tid == 0) {
        L.n_single = 0;
      }

      // the LDS rows that take the sums of multi-pair slots in (c)
      *reinterpret_cast<f32x4*>(&L.red[(size_t)tid * 4]) = f32x4{0.f, 0.f, 0.f, 0.f};

      // (a) rows of the chunk -> slots; one ticket per pair and slot.  Same-address LDS atomics
      // serialise, so a hot row is first reduced inside the wave: up to kHotTries times the first
      // pending lane's row is matched with a ballot; a group of >= kHotMin lanes lets its leader
      // probe once and take all tickets, the other lanes get slot and ticket by broadcast.
      int hs_[kCP / kTeam], tk_[kCP / kTeam], rk_[kCP / kTeam];
      unsigned long long row_[kCP / kTeam];
      HBK_SUBSTAMP(0);
      // all pairs of the chunk are requested before the first is worked on (inside the loop below
      // the second pair's loads waited behind the LDS work of the first: two memory round trips)
      int64_t r_in[kCP / kTeam];
      int32_t seg_in[kCP / kTeam];
#pragma unroll
      for (int k = 0; k < kCP / kTeam; ++k) {
        const int e = k * kTeam + tid;
        r_in[k] = kDonePair;
        seg_in[k] = cb + e;
        if (e < n_chunk) {
          r_in[k] = HBK_PAIR_LOAD(prow + cb + e);
          if (pseg != nullptr) seg_in[k] = HBK_PAIR_LOAD(pseg + cb + e);
        }
      }
#ifdef HBK_BWD_STAMPS
      __builtin_amdgcn_s_waitcnt(0x0f70);
      HBK_SUBSTAMP(1);
#endif
      // mean / sqrtn over ragged segments: the segment lengths are requested HERE, one round trip
      // for the whole chunk beside the insert work, and kept in LDS.  (Divided inside the gradient
      // loads -- load, wait for the length, divide, next load -- the rows of a round travelled one
      // memory round trip at a time.)
      int32_t n_in[kCP / kTeam];
      if (scaled) {   // uniform
#pragma unroll
        for (int k = 0; k < kCP / kTeam; ++k) {
          n_in[k] = 1;
          if (k * kTeam + tid < n_chunk) n_in[k] = c.splits[seg_in[k] + 1] - c.splits[seg_in[k]];
        }
      }
#pragma unroll
      for (int k = 0; k < kCP / kTeam; ++k) {
        const int e = k * kTeam + tid;
        bool valid = e < n_chunk;
        unsigned long long row = 0;
        if (valid) {
          const int64_t r = r_in[k];
          L.segs[e] = seg_in[k];
          valid = r != kDonePair;
          row = (unsigned long long)r;
        }
        int h = -2, ticket = 0;    // -2: not looked up yet, -1: no room in this pass
        bool entered = false;      // this lane entered a new row into the table
        unsigned long long todo = __ballot(valid);
        for (int t = 0; t < kHotTries && todo != 0ull; ++t) {
          const int leader = __builtin_ctzll(todo);
          const unsigned long long r =
              ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(row >> 32), leader) << 32) |
              (unsigned)__builtin_amdgcn_readlane((int)row, leader);
          const unsigned long long same = __ballot(valid && row == r) & todo;
          todo &= ~same;
          const int n_same = (int)__builtin_popcountll(same);
          // wave-uniform.  Uniform ids: the first lane's row is shared with nobody and the search
          // ends after one round (it would cost 5 % of the backward otherwise); skewed ids: the
          // first lane most likely holds a repeated row and the search goes on
#ifndef HBK_BWD_HOT_NOEARLY
          if (n_same == 1 && t == 0) break;
#endif
          if (n_same < kHotMin) continue;
          int hs = -1, base = 0;
          if (lane == leader) {
            hs = table_slot(L, r, &entered);
            if (hs >= 0) base = atomicAdd(&L.cnt[hs], n_same);
          }
          hs = __builtin_amdgcn_readlane(hs, leader);
          base = __builtin_amdgcn_readlane(base, leader);
          if ((same >> lane) & 1ull) {
            h = hs;
            ticket = base + rank_below(same);
          }
        }
        if (valid && h == -2) {
          h = table_slot(L, row, &entered);
          if (h >= 0) ticket = atomicAdd(&L.cnt[h], 1);
        }
        {
          // the wave's new rows enter the fill level with one atomic
          const unsigned long long fresh = __ballot(entered);
          if (fresh != 0ull && lane == (int)__builtin_ctzll(fresh)) {
            atomicAdd(&L.occupied, (int32_t)__builtin_popcountll(fresh));
          }
        }
        if (e < n_chunk) L.pslot[e] = valid && h >= 0 ? (uint16_t)h : kNoSlot;
        if (valid && h < 0) atomicAdd(&L.n_left, 1);
        hs_[k] = valid && h >= 0 ? h : (valid ? -2 : -1);   // -2: found no room
        tk_[k] = ticket;
        row_[k] = row;
      }
      if (scaled) {
#pragma unroll
        for (int k = 0; k < kCP / kTeam; ++k) {
          if (k * kTeam + tid < n_chunk) L.nseg[k * kTeam + tid] = n_in[k];
        }
      }
      HBK_SUBSTAMP(2);
      team_sync();
      HBK_SUBSTAMP(3);
      if (L.n_left > 0) {   // uniform: nobody changes it before the next pass
        // A pair that found no room may belong to a row another lane entered in the same instant
        // (it read the slot before that lane's CAS and the fill level after it): the next pass
        // would enter the row a second time and emit it twice.  The table is still now: look again.
#pragma unroll
        for (int k = 0; k < kCP / kTeam; ++k) {
          if (hs_[k] == -2) {
            const int h = table_find(L, row_[k]);
            if (h >= 0) {
              tk_[k] = atomicAdd(&L.cnt[h], 1);
              L.pslot[k * kTeam + tid] = (uint16_t)h;
            }
            hs_[k] = h >= 0 ? h : -1;
          }
        }
        team_sync();
      }
      HBK_STAMP(3);

      // One global atomic per workgroup and chunk claims the output range of the new rows.  A
      // returning device-scope atomic under this load takes several microseconds: it is issued
      // as soon as the count is known -- the rows the table has gained in this chunk -- and its
      // round trip runs beside the scan of (b) and the gradient loads of (c).
      int32_t claimed = 0;
      if (tid == kTeam - 1) {
        const int32_t n_new = L.occupied - L.occupied_before;
        L.occupied_before = L.occupied;
        if (STEP && job.no_emit) {
          // only the count is wanted: fire and forget, nobody waits for this atomic
          if (n_new > 0) __hip_atomic_fetch_add(job.out_counter, n_new, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (n_new > 0) {
          claimed = atomicAdd(job.out_counter, n_new);
        }
      }

      // (b) one packed exclusive scan over the PAIRS that hold ticket 0 (one per slot of the
      // chunk): new rows | slots with several pairs << 10 | their pairs << 20
      {
        int32_t pk[kCP / kTeam], sum = 0;
#pragma unroll
        for (int k = 0; k < kCP / kTeam; ++k) {
          pk[k] = 0;
          if (hs_[k] >= 0 && tk_[k] == 0) {
            const int32_t n = L.cnt[hs_[k]];
            const bool is_new = L.slot_out[hs_[k]] < 0;
            // a new row with one pair is stored straight from registers in (c); rows with
            // several pairs, and rows an earlier chunk already emitted (read-modify-write), go
            // through the sorted walk of (d)
            pk[k] = (is_new ? 1 : 0) | (n > 1 || !is_new ? (1 << 10) | (n << 20) : 0);
          }
          sum += pk[k];
        }
        int32_t incl = sum;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
          const int32_t y = __shfl_up(incl, o, kWave);
          if (lane >= o) incl += y;
        }
        if (lane == kWave - 1) L.wave_tot[wave] = incl;
        team_sync();
        int32_t run = incl - sum;
        for (int w = 0; w < wave; ++w) run += L.wave_tot[w];
        if (tid == kTeam - 1) {
          const int32_t tot = run + sum;
          L.n_new = tot & 1023;
          L.n_active = (tot >> 10) & 1023;
          L.n_multi = tot >> 20;
          // few multi-pair slots holding few pairs (the usual case): their pairs are summed in LDS
          // rows by ds_add_f32; many, or a slot with many pairs (skewed ids: same-address LDS
          // atomics serialise): they are sorted by slot and walked / summed by the whole
          // workgroup, see (d)
          L.lds_rows = ((tot >> 10) & 1023) * c.dim <= kTeam * 4 && (tot >> 20) <= kLdsRowPairs;
        }
        team_sync();
        const bool lds_rows_b = L.lds_rows != 0;
#pragma unroll
        for (int k = 0; k < kCP / kTeam; ++k) {
          rk_[k] = -1;
          if (!lds_rows_b) {
            // skewed ids: most pairs belong to multi-pair slots and go through (d); (c) then walks
            // the LIST of single-pair new rows (kept in order[], which (d) only fills later), one
            // LDS ticket per wave
            const unsigned long long single = __ballot(pk[k] == 1);
            if (single != 0ull) {
              int base = 0;
              const int first = __builtin_ctzll(single);
              if (lane == first) base = atomicAdd(&L.n_single, (int)__builtin_popcountll(single));
              base = __builtin_amdgcn_readlane(base, first);
              if (pk[k] == 1) L.order[base + rank_below(single)] = (uint16_t)(k * kTeam + tid);
            }
          }
          if (pk[k] != 0) {
            const int s = hs_[k];
            if (pk[k] & 1) {
              // the output position is base_u + rank; base_u comes from the claiming atomic, whose
              // round trip runs beside the gradient loads of (c): until then the slot holds the rank
              rk_[k] = run & 1023;
              L.slot_out[s] = -2 - rk_[k];
              L.cnt[s] |= kNewBit;
            }
            if (pk[k] >> 10) {
              const int m = (run >> 10) & 1023;
              L.active[m] = (uint16_t)s;
              L.off[s] = lds_rows_b ? m : run >> 20;
            }
          }
          run += pk[k];
        }
      }
      team_sync();
      HBK_STAMP(4);

      // (c) one round of gradient loads, kPre rows in flight per lane.  A NEW row with ONE pair in
      // the chunk -- the common case -- is stored straight from the registers its gradient
      // arrived in; the pairs of multi-pair slots are added into the slot's LDS row (ds_add_f32)
      // when there are few such slots, else left to (d).
      const int n_active = L.n_active;   // uniform
      const bool lds_rows = L.lds_rows != 0;
      int32_t base_u = 0;
      // lds_rows: every pair of the chunk is fetched here; else only the listed single-pair rows.
      // kDepth rows in flight per lane; between the loads and the stores a lane keeps nothing but
      // the rows and one flag word (slot, ticket count and output position are read again from
      // LDS), so that the deep variant fits the register budget without spilling.
      constexpr int kDepth = STEP == 2 ? kPre - 1 : STEP ? kPre : 2 * kPre;   // (Adagrad: a third row spills 136 B/lane)
      const int n_fetch = lds_rows ? n_chunk : L.n_single;
      for (int e0 = 0; e0 == 0 || e0 < n_fetch; e0 += kDepth * groups) {
        V pre[kDepth], tv[STEP ? kDepth : 1], av[STEP == 2 ? kDepth : 1];
        uint32_t single = 0, multi = 0;     // bit k: row k of this round is a single new row / an
                                            // LDS-summed pair
#pragma unroll
        for (int k = 0; k < kDepth; ++k) {
          const int i = e0 + k * groups + my_group;
          pre[k] = zero_v<V>();
          if (i < n_fetch && live) {
            const int e = lds_rows ? i : (int)L.order[i];
            const int sidx = (int)L.pslot[e];
            if (sidx != (int)kNoSlot) {
              const bool is_single = L.cnt[sidx] == (kNewBit | 1);
              if (is_single || lds_rows) {
                pre[k] = load_grad_lds<V>(c, job, L.segs[e], sub, scaled, L.nseg[e]);
                single |= is_single ? 1u << k : 0u;
                multi |= is_single ? 0u : 1u << k;
              }
              if (STEP && is_single && lr_now != 0.0f) {
                // the table (and accumulator) row of the step travels with the gradient
                const int64_t toff = (int64_t)L.keys[sidx] * c.tpitch + (int64_t)sub * VE;
                tv[STEP ? k : 0] =
                    HBK_STEP_LOAD(reinterpret_cast<const V*>(c.table + toff));
                if (STEP == 2) {
                  av[STEP == 2 ? k : 0] =
                      HBK_STEP_LOAD(reinterpret_cast<const V*>(c.accum + toff));
                }
              }
            }
          }
        }
        if (e0 == 0) {
          if (tid == kTeam - 1) {
            L.base_u = job.out_base + claimed;
            L.emit0 = L.n_emitted;
            L.n_emitted += L.n_new;
          }
          team_sync();
          base_u = L.base_u;
        }
#pragma unroll
        for (int k = 0; k < kDepth; ++k) {
          if (((single | multi) >> k & 1u) == 0) continue;
          const int i = e0 + k * groups + my_group;
          const int sidx = (int)L.pslot[lds_rows ? i : (int)L.order[i]];
          if (single >> k & 1u) {
            if (!(STEP && job.no_emit)) {
              emit_row<V>(c, job, base_u + (-2 - L.slot_out[sidx]), true, sub, pre[k]);
            }
            if (STEP && lr_now != 0.0f) {
              const int64_t toff = (int64_t)L.keys[sidx] * c.tpitch + (int64_t)sub * VE;
              step_row<V>(c, adagrad, lr_now, toff, pre[k], tv[STEP ? k : 0],
                          STEP == 2 ? av[STEP == 2 ? k : 0] : zero_v<V>());
            }
          } else {
            float* r = &L.red[(size_t)L.off[sidx] * c.dim + (size_t)sub * VE];
#pragma unroll
            for (int q = 0; q < VE; ++q) atomicAdd(r + q, reinterpret_cast<const float*>(&pre[k])[q]);
          }
        }
      }
      // the new rows' slots get their absolute output position (later chunks, (d), the optimizer
      // step read it), and the row numbers go out
      team_sync();
#pragma unroll
      for (int k = 0; k < kCP / kTeam; ++k) {
        if (rk_[k] >= 0) {
          const int32_t u = base_u + rk_[k];
          L.slot_out[hs_[k]] = u;
          if (!(STEP && job.no_emit)) job.out_rows[u] = (int64_t)row_[k];
          if (defer) L.emitted[L.emit0 + rk_[k]] = (uint16_t)hs_[k];
        }
      }
      if (lds_rows && n_active > 0) {
        team_sync();
        for (int m = my_group; m < n_active; m += groups) {
          if (!live) continue;
          const int sidx = L.active[m];
          emit_step_row<V, STEP>(c, job, lr_now, L.slot_out[sidx], (L.cnt[sidx] & kNewBit) != 0,
                           (int64_t)L.keys[sidx], sub,
                           *reinterpret_cast<const V*>(&L.red[(size_t)m * c.dim + (size_t)sub * VE]));
        }
      }
      HBK_STAMP(5);
      if (n_active > 0 && !lds_rows) {
        // (d) counting sort of the pairs of multi-pair slots (off[] ends up as the end of every
        // slot's run); tickets of a hot slot are taken once per wave and split by ballot rank
#pragma unroll
        for (int k = 0; k < kCP / kTeam; ++k) {
          const int e = k * kTeam + tid;
          const int h = hs_[k];
          const bool valid = h >= 0 && L.cnt[h] != (kNewBit | 1);
          int pos = -1;
          unsigned long long todo = __ballot(valid);
          for (int t = 0; t < kHotTries && todo != 0ull; ++t) {
            const int leader = __builtin_ctzll(todo);
            const int hs = __builtin_amdgcn_readlane(h, leader);
            const unsigned long long same = __ballot(valid && h == hs) & todo;
            todo &= ~same;
            const int n_same = (int)__builtin_popcountll(same);
            if (n_same < kHotMin) continue;
            int base = 0;
            if (lane == leader) base = atomicAdd(&L.off[hs], n_same);
            base = __builtin_amdgcn_readlane(base, leader);
            if ((same >> lane) & 1ull) pos = base + rank_below(same);
          }
          if (valid) {
            if (pos < 0) pos = atomicAdd(&L.off[h], 1);
            L.order[pos] = (uint16_t)e;
          }
        }
        team_sync();

        // The sorted pairs are walked FLAT: per round a lane group takes kW consecutive positions
        // of the sorted order, all their gradient rows in flight at once, and sums runs of one slot
        // in registers.  A slot's pairs span neighbouring lane groups (and rounds): the group where
        // a run starts owns it and adds the head partials its successors leave in LDS; a slot that
        // runs past the round's end is carried in an LDS row into the next round.  One emit per
        // slot whatever its length -- 3 pairs or a hot row's 400 -- and no lane group walks a
        // slot's pairs one memory round trip at a time (with 33 pairs per row that walk was 20 of
        // a workgroup's 37 us; it also replaces the whole-workgroup sum of hot rows).
        auto walk = [&](auto width_tag) {
          constexpr int kW = decltype(width_tag)::value;
          constexpr int kB = kW > (STEP == 2 ? 2 : kPre) ? 2 : kW;   // positions whose step rows travel together
          const int n_multi = L.n_multi;
          const int per_round = groups * kW;
          int par = 0;
          for (int r0 = 0; r0 < n_multi; r0 += per_round, par ^= 1) {
            const int q0 = r0 + my_group * kW;   // my first position
            V g[kW];
            int32_t sl[kW];                       // slot of every position, -1 beyond the end
  #pragma unroll
            for (int w = 0; w < kW; ++w) {
              const int q = q0 + w;
              sl[w] = -1;
              g[w] = zero_v<V>();
              if (q < n_multi) {
                const int e = (int)L.order[q];
                sl[w] = (int)L.pslot[e];
                if (live) g[w] = load_grad_lds<V>(c, job, L.segs[e], sub, scaled, L.nseg[e]);
              }
            }
            // head: the run at my first position (it may have started in a group before me)
            {
              V head = zero_v<V>();
              bool in_head = sl[0] >= 0;
  #pragma unroll
              for (int w = 0; w < kW; ++w) {
                in_head = in_head && sl[w] == sl[0];
                if (in_head) head = head + g[w];
              }
              *reinterpret_cast<V*>(&L.red[(size_t)tid * VE]) = head;
            }
            team_sync();
            const int round_end = r0 + per_round;
            V acc = zero_v<V>();
            int cur = -1;
            bool owned = false;
            uint32_t fin = 0;   // bit w: a run I own ends at my position w and is complete: its sum is
                                // left in g[w] (the registers its last gradient row arrived in)
  #pragma unroll
            for (int w = 0; w <= kW; ++w) {
              const int sw = w < kW ? sl[w] : -1;
              if (sw != cur) {
                if (w > 0 && cur >= 0 && owned) {
                  // the run of `cur` ends at my position w - 1
                  const int32_t s_end = L.off[cur];
                  const int32_t s_beg = s_end - (L.cnt[cur] & (kNewBit - 1));
                  V total = acc;
                  if (w == kW) {
                    // it reached the end of my range: the heads of the groups it goes on in
                    const int lim = s_end < round_end ? s_end : round_end;
                    for (int q = q0 + kW; q < lim; q += kW) {
                      const int gp = (q - r0) / kW;
                      total = total + *reinterpret_cast<const V*>(
                                          &L.red[(((size_t)gp << lpr_log2) + sub) * VE]);
                    }
                  }
                  if (s_beg < r0) {   // it began in an earlier round (only the round's first run can)
                    total = total + *reinterpret_cast<const V*>(&L.carry[par ^ 1][(size_t)sub * VE]);
                  }
                  if (s_end > round_end) {
                    *reinterpret_cast<V*>(&L.carry[par][(size_t)sub * VE]) = total;   // goes on
                  } else {
                    g[w > 0 ? w - 1 : 0] = total;
                    fin |= 1u << (w > 0 ? w - 1 : 0);
                  }
                }
                cur = sw;
                acc = zero_v<V>();
                // a run that starts inside my range is mine; the one at my first position is mine
                // when the slot starts there or when I am the round's first group (it is carried in)
                owned = w > 0 || my_group == 0 ||
                        (sw >= 0 && L.off[sw] - (L.cnt[sw] & (kNewBit - 1)) >= q0);
              }
              if (w < kW && sw >= 0) acc = acc + g[w];
            }
            // the finished rows leave together: the table (and accumulator) rows of the optimizer
            // step are requested for all of them before the first is used
            if (STEP && lr_now != 0.0f) {
    #pragma unroll
              for (int w0 = 0; w0 < kW; w0 += kB) {
                if (((fin >> w0) & ((1u << kB) - 1u)) == 0u || !live) continue;
                V tv[kB], av[STEP == 2 ? kB : 1];
  #pragma unroll
                for (int b = 0; b < kB; ++b) {
                  const int w = w0 + b;
                  tv[b] = zero_v<V>();
                  if (w < kW && (fin >> w & 1u)) {
                    const int64_t toff = (int64_t)L.keys[sl[w]] * c.tpitch + (int64_t)sub * VE;
                    tv[b] = HBK_STEP_LOAD(reinterpret_cast<const V*>(c.table + toff));
                    if (STEP == 2) {
                      av[STEP == 2 ? b : 0] =
                          HBK_STEP_LOAD(reinterpret_cast<const V*>(c.accum + toff));
                    }
                  }
                }
  #pragma unroll
                for (int b = 0; b < kB; ++b) {
                  const int w = w0 + b;
                  if (w < kW && (fin >> w & 1u)) {
                    const int s = sl[w];
                    if (!job.no_emit) {
                      emit_row<V>(c, job, L.slot_out[s], (L.cnt[s] & kNewBit) != 0, sub, g[w]);
                    }
                    const int64_t toff = (int64_t)L.keys[s] * c.tpitch + (int64_t)sub * VE;
                    step_row<V>(c, adagrad, lr_now, toff, g[w], tv[b],
                                STEP == 2 ? av[STEP == 2 ? b : 0] : zero_v<V>());
                  }
                }
              }
            } else {
  #pragma unroll
              for (int w = 0; w < kW; ++w) {
                if ((fin >> w & 1u) && live) {
                  const int s = sl[w];
                  emit_row<V>(c, job, L.slot_out[s], (L.cnt[s] & kNewBit) != 0, sub, g[w]);
                }
              }
            }
            team_sync();
          }
        };
        // Width of the walk with the optimizer step: as many positions as the step's table /
        // accumulator rows leave registers for (kPre with SGD, 2 with Adagrad) -- or, for WIDE rows
        // (dim >= 64: <= 16 lane groups, i.e. 10-28 rounds of two barriers and two memory round
        // trips per chunk at the narrow width), HBK_BWD_WIDE_W positions with the step's rows requested
        // two positions at a time (a round finishes ~1 row per lane group).  Narrow rows lose with
        // that form (ragged dim 16: 937 vs 782 us: more dependent table round trips per round).
        if constexpr (STEP != 0 && WIDE) {
          walk(std::integral_constant<int, HBK_BWD_WIDE_W>());
        } else {
          walk(std::integral_constant<int, STEP == 2 ? 2 : STEP ? kPre : 2 * kPre>());
        }
      }
      team_sync();

      // chunk done: the tickets go back to zero; once a pass has left pairs behind, the pairs it
      // did handle are struck out of the pair buffer so that the next pass skips them
      const bool left = L.n_left > 0;   // uniform: written before the barriers above
      if (left && first_left < 0) first_left = cb;
      striking = striking || left;
#pragma unroll
      for (int k = 0; k < kCP / kTeam; ++k) {
        if (hs_[k] >= 0) {
          L.cnt[hs_[k]] = 0;
          if (striking) prow[cb + k * kTeam + tid] = kDonePair;
        }
      }
      team_sync();
      HBK_STAMP(6);
    }
    // the pass is over (job of several chunks): one optimizer step per row the pass emitted.  All
    // loads of a round first (gradient sums, table rows, accumulator rows of kAp rows per lane
    // group), then the stores: a load behind a store to a possibly aliasing row would serialise
    // the round into one memory round trip per row.
    if (STEP && defer) {
      constexpr int kAp = STEP == 2 ? 2 : 4;   // (Adagrad: four rows of three vectors spill)
      const int n_emitted = L.n_emitted;   // written before the last barrier
      for (int i0 = 0; i0 < n_emitted; i0 += kAp * groups) {
        int64_t toff[kAp];
        V g[kAp], tv[kAp], av[kAp];
#pragma unroll
        for (int k = 0; k < kAp; ++k) {
          const int i = i0 + k * groups + my_group;
          toff[k] = -1;
          g[k] = tv[k] = av[k] = zero_v<V>();
          if (i < n_emitted && live) {
            const int s = L.emitted[i];
            toff[k] = (int64_t)L.keys[s] * c.tpitch + (int64_t)sub * VE;
            g[k] = __builtin_nontemporal_load(reinterpret_cast<const V*>(
                job.out_vals + (int64_t)L.slot_out[s] * c.dim + (int64_t)sub * VE));
            tv[k] = HBK_STEP_LOAD(reinterpret_cast<const V*>(c.table + toff[k]));
            if (adagrad) {
              av[k] = HBK_STEP_LOAD(reinterpret_cast<const V*>(c.accum + toff[k]));
            }
          }
        }
#pragma unroll
        for (int k = 0; k < kAp; ++k) {
          if (toff[k] >= 0) step_row<V>(c, adagrad, job.lr, toff[k], g[k], tv[k], av[k]);
        }
      }
    }
    if (first_left < 0) break;     // uniform
    // another pass for the rows that found no room, with an emptied table
    team_sync();
    for (int i = tid; i < kSlots; i += kTeam) {
      L.keys[i] = kEmptyKey;
      L.slot_out[i] = -1;
    }
    if (tid == 0) {
      L.occupied = 0;
      L.occupied_before = 0;
      L.n_left = 0;
      L.n_emitted = 0;
    }
    first_cb = first_left;
    team_sync();
  }
}

// ---- 4b: dense columns -- one workgroup per ROW RANGE, LDS tables indexed by the row ---------------
// When a column's batch covers its table densely enough (rows / ids <~ 36: all of config 2, most
// of config 5) a bucket is a contiguous range of <= kDenseSpan rows and the reduce stage needs no
// hash table, no CAS, no probe chains and no tickets:
//   A  every pair sets its row's bit in a PRESENT bitmap (ds_or with return); a pair that finds
//      the bit set also sets it in a DUP bitmap (rows with several pairs);
//   B  one packed scan over the bitmap words (popcounts): rows before every word.  The rank of a
//      row among the bucket's rows = prefix + popcount of the lower bits of its word -- its output
//      position once the workgroup's ONE global atomic has claimed the range; the rows leave
//      SORTED, and their numbers are written straight from the bitmap (consecutive lanes store
//      consecutive entries);
//   C  a pair whose row has no other pair (the common case) moves its gradient row from the
//      registers it arrived in to its output row, with the optimizer step when there is one (the
//      table / accumulator row travels in the same round);
//   D  the pairs of DUP rows are summed in LDS rows (ds_add_f32; a lane group first adds up what it
//      holds for one row in registers), kRedFloats / dim rows per round, and every such row is
//      emitted -- and stepped -- once, from the finished sum.
// Every row is complete when it is emitted whatever the number of chunks: no rows spanning chunks,
// no deferred step, no second pass.  Jobs of several chunks (a bucket above kCP pairs, the ranges
// of a split bucket, the merge of their partial entries) read their pairs once per stage.
// workgroups per CU the dense kernels are compiled for: 5 (96 VGPRs) for the lean instantiation,
// 4 (128 VGPRs) for the one with the sorted walk (it spills at 96)
#ifndef HBK_BWD_DENSE_WAVES
#define HBK_BWD_DENSE_WAVES(SORT) ((SORT) ? 4 : 5)
#endif
#ifndef HBK_BWD_DENSE_WALK
#define HBK_BWD_DENSE_WALK(STEP) ((STEP) == 2 ? 2 : (STEP) ? 3 : 4)
#endif
// probe switches (tools/scratch/bwd_dense_variants.sh builds the library with them turned off)
#ifndef HBK_DENSE_FOLD
#define HBK_DENSE_FOLD 0          // few dup pairs ride with the single rows' loads
#endif
#ifndef HBK_DENSE_EARLY_CLAIM
#define HBK_DENSE_EARLY_CLAIM 0   // output range claimed before the scan (rows = pairs - repeats)
#endif
#ifndef HBK_DENSE_SORT
#define HBK_DENSE_SORT 1          // many dup pairs: sorted flat walk instead of LDS float atomics
#endif
constexpr int kDenseSpan = 16384;                 // rows of a bucket's range (bits per bitmap)
constexpr int kDenseWords = kDenseSpan / 32;
constexpr int kDenseWPT = kDenseWords / kBlock;   // bitmap words per thread in the scan
constexpr int kRedFloats = 2048;                  // LDS floats that take the sums of dup rows
constexpr int kSortMin = 48;                      // dup pairs of a chunk above which they are sorted by row (D)
constexpr int32_t kDupBit = 1 << 30;
static_assert(kDenseWords % kBlock == 0, "whole words per thread");

struct DenseLds {
  uint32_t present[kDenseWords];
  uint32_t dup[kDenseWords];
  uint32_t pre[kDenseWords];     // rows before word w: present (low 16 bits) | dup (high 16 bits)
  int32_t seg[kCP];              // gradient row of every pair of the chunk
  int32_t code[kCP];             // rank of the pair's row among the bucket's rows, or kDupBit | its
                                 // rank among the dup rows
  uint16_t off[kCP];             // row - first row of the range
  uint16_t dlist[kCP];           // pairs of the chunk whose dup row is summed in this round; sorted
                                 // by row when there are many (D)
  uint16_t doff[kCP];            // row - first row of the range, of every dup row of the round
  int32_t dcnt[kCP];             // pairs of dup row m of the round in this chunk (tickets)
  uint32_t drun[kCP];            // its run in the sorted list: first (low 16 bits) | end (high 16)
  float red[kRedFloats];         // sums of dup rows across chunks / of the few-pairs case
  float heads[kBlock * 4];       // sorted walk: what a lane group holds of the run at its first position
  float carry[2][kWave * 4];     // sorted walk: the run that goes on into the next round
  int32_t wave_tot[kWavesPerBlock];
  int32_t n_again, n_dup, base_u;   // pairs that found their row's bit set; dup rows; output base
  int32_t n_dlist[2];
};

// first row of bucket b of a dense column: the smallest r with mulhi(r, M) >= b
__device__ inline uint64_t dense_first_row(uint32_t M, int b) {
  return (((uint64_t)(uint32_t)b << 32) + M - 1) / M;
}

// D of dense_reduce for MANY dup pairs: the chunk's list of them is sorted by row (dlist / drun)
// and walked flat, sums in registers (see there).  The call sits in a branch marked unlikely: the
// mere presence of this code made the common path -- which never enters it -- 9 us slower on the
// config-2 backward (probe builds: allocation / scheduling of the surrounding code); out of line
// (noinline) was worse still: the call's register convention spilled 40 registers in the hot loop.
struct WalkArgs {
  const float* grad;
  const int32_t* splits;
  float* out_vals;
  float* table;
  float* accum;
  int32_t stride, dim, chunks, n_dl, d0, base_u;
  uint32_t base;
  float lr;
  int32_t tpitch;
  uint8_t lpr_log2, combiner, seg_is_offset, scale, one_chunk, no_emit;
};

template <typename V, int STEP>
__device__ inline void dense_walk(const WalkArgs w_, DenseLds& L) {
  constexpr int VE = sizeof(V) / 4;
  constexpr bool adagrad = STEP == 2;
  // the fields the helpers below read (the rest of the structs is never touched)
  GCol c;
  c.splits = w_.splits;
  c.table = w_.table;
  c.accum = w_.accum;
  c.dim = w_.dim;
  c.tpitch = w_.tpitch;
  c.chunks = w_.chunks;
  c.lpr_log2 = w_.lpr_log2;
  c.combiner = w_.combiner;
  ReduceJob job;
  job.grad = w_.grad;
  job.stride = w_.stride;
  job.seg_is_offset = w_.seg_is_offset != 0;
  job.scale = w_.scale != 0;
  job.out_vals = w_.out_vals;
  job.no_emit = w_.no_emit != 0;
  const int n_dl = w_.n_dl, d0 = w_.d0;
  const uint32_t base = w_.base;
  const int32_t base_u = w_.base_u;
  const float lr = w_.lr;
  const bool one_chunk = w_.one_chunk != 0;
  const int tid = (int)threadIdx.x;
  const int lpr_log2 = c.lpr_log2;
  const int sub = tid & ((1 << lpr_log2) - 1);
  const bool live = sub < c.chunks;
  const int groups = kBlock >> lpr_log2;
  const int my_group = tid >> lpr_log2;
  const bool emit = !(STEP && job.no_emit);
  {
  constexpr int kW = HBK_BWD_DENSE_WALK(STEP);   // (register budget: 96 VGPRs = 5 workgroups per CU)
  const int per_round = groups * kW;
  int cpar = 0;
  for (int r0 = 0; r0 < n_dl; r0 += per_round, cpar ^= 1) {
    const int q0 = r0 + my_group * kW;   // my first position
    V g[kW];
    int sl[kW];                           // dup row of every position, -1 beyond the end
    int32_t n_[kW];
#pragma unroll
    for (int w = 0; w < kW; ++w) {
      const int q = q0 + w;
      sl[w] = -1;
      n_[w] = 0;
      g[w] = zero_v<V>();
      if (q < n_dl) {
        const int e = (int)L.dlist[q];
        sl[w] = (L.code[e] & (kDupBit - 1)) - d0;
        if (live) g[w] = load_grad_raw<V>(c, job, L.seg[e], sub, &n_[w]);
      }
    }
    if (job.scale && c.combiner != HBK_COMBINER_SUM && c.splits != nullptr) {   // uniform
#pragma unroll
      for (int w = 0; w < kW; ++w) g[w] = scale_grad<V>(c, g[w], n_[w]);
    }
    {
      V head = zero_v<V>();
      bool in_head = sl[0] >= 0;
#pragma unroll
      for (int w = 0; w < kW; ++w) {
        in_head = in_head && sl[w] == sl[0];
        if (in_head) head = head + g[w];
      }
      *reinterpret_cast<V*>(&L.heads[(size_t)tid * VE]) = head;
    }
    __syncthreads();
    const int round_end = r0 + per_round;
    V acc = zero_v<V>();
    int cur = -1;
    bool owned = false;
    uint32_t fin = 0;   // bit w: a run I own ends at my position w, complete: its sum is in g[w]
#pragma unroll
    for (int w = 0; w <= kW; ++w) {
      const int sw = w < kW ? sl[w] : -1;
      if (sw != cur) {
        if (w > 0 && cur >= 0 && owned) {
          const uint32_t rn = L.drun[cur];
          const int s_beg = (int)(rn & 0xffffu), s_end = (int)(rn >> 16);
          V total = acc;
          if (w == kW) {
            // it reached the end of my range: the heads of the groups it goes on in
            const int lim = s_end < round_end ? s_end : round_end;
            for (int q = q0 + kW; q < lim; q += kW) {
              const int gp = (q - r0) / kW;
              total = total + *reinterpret_cast<const V*>(
                                  &L.heads[(((size_t)gp << lpr_log2) + sub) * VE]);
            }
          }
          if (s_beg < r0) {   // it began in an earlier round (only the round's first run can)
            total = total + *reinterpret_cast<const V*>(&L.carry[cpar ^ 1][(size_t)sub * VE]);
          }
          if (s_end > round_end) {
            *reinterpret_cast<V*>(&L.carry[cpar][(size_t)sub * VE]) = total;   // goes on
          } else {
            g[w > 0 ? w - 1 : 0] = total;
            fin |= 1u << (w > 0 ? w - 1 : 0);
          }
        }
        cur = sw;
        acc = zero_v<V>();
        // a run that starts inside my range is mine; the one at my first position is mine
        // when the row starts there or when I am the round's first group (it is carried in)
        owned = w > 0 || my_group == 0 || (sw >= 0 && (int)(L.drun[sw] & 0xffffu) >= q0);
      }
      if (w < kW && sw >= 0) acc = acc + g[w];
    }
    if (!one_chunk) {
      // a job of several chunks: the row's pairs in the other chunks are still to come (or
      // already there): the chunk's sum joins the row's LDS sum; one owner per row and chunk
#pragma unroll
      for (int w = 0; w < kW; ++w) {
        if ((fin >> w & 1u) && live) {
          V* r = reinterpret_cast<V*>(&L.red[(size_t)sl[w] * c.dim + (size_t)sub * VE]);
          *r = *r + g[w];
        }
      }
    } else {
      // the row is complete: it leaves from the registers its sum sits in, with the
      // optimizer step (table / accumulator rows requested for all finished rows first)
      int32_t u_[kW];
      uint32_t off_[kW];
      V tv[STEP ? kW : 1], av[STEP == 2 ? kW : 1];
#pragma unroll
      for (int w = 0; w < kW; ++w) {
        u_[w] = 0;
        off_[w] = 0;
        if ((fin >> w & 1u) && live) {
          const uint32_t off = L.doff[sl[w]];
          const int ww = (int)(off >> 5);
          off_[w] = off;
          u_[w] = base_u + (int32_t)(L.pre[ww] & 0xffffu) +
                  __builtin_popcount(L.present[ww] & ((1u << (off & 31u)) - 1u));
          if (STEP && lr != 0.0f) {
            const int64_t toff = (int64_t)(base + off) * c.tpitch + (int64_t)sub * VE;
            tv[STEP ? w : 0] =
                HBK_STEP_LOAD(reinterpret_cast<const V*>(c.table + toff));
            if (STEP == 2) {
              av[STEP == 2 ? w : 0] =
                  HBK_STEP_LOAD(reinterpret_cast<const V*>(c.accum + toff));
            }
          }
        }
      }
#pragma unroll
      for (int w = 0; w < kW; ++w) {
        if ((fin >> w & 1u) && live) {
          if (emit) emit_row<V>(c, job, u_[w], true, sub, g[w]);
          if (STEP && lr != 0.0f) {
            const int64_t toff = (int64_t)(base + off_[w]) * c.tpitch + (int64_t)sub * VE;
            step_row<V>(c, adagrad, lr, toff, g[w], tv[STEP ? w : 0],
                        STEP == 2 ? av[STEP == 2 ? w : 0] : zero_v<V>());
          }
        }
      }
    }
    __syncthreads();
  }
  }
}

// The first chunk of a job's pairs, requested ahead of the job (persistent workgroups: while the
// job before is still running).
struct FirstPairs {
  int64_t row[kCP / kBlock];
  int32_t seg[kCP / kBlock];
};

__device__ inline void load_first_pairs(const ReduceJob& job, FirstPairs& f) {
#pragma unroll
  for (int k = 0; k < kCP / kBlock; ++k) {
    const int32_t e = k * kBlock + (int)threadIdx.x;
    f.row[k] = -1;
    f.seg[k] = e;
    if (e < job.n_pairs) {
      f.row[k] = HBK_PAIR_LOAD(job.prow + e);
      if (job.pseg != nullptr) f.seg[k] = HBK_PAIR_LOAD(job.pseg + e);
    }
  }
}

struct NoHook {
  __device__ void operator()() const {}
};

// `first`: the job's first chunk of pairs (already requested).  `hook`: called once, behind
// stage B -- where a persistent workgroup requests the NEXT job's pairs, so that they travel
// beside this job's gradient rows.
// SORT: the instantiation that counting-sorts many dup pairs and walks them flat (D); without it
// every dup pair goes through the LDS float atomics -- right for any input, slow for many -- and
// the common path of a column with few repeated rows is ~10 % faster for not carrying that code.
template <typename V, int STEP, bool SORT, typename Hook = NoHook>
__device__ inline void dense_reduce(const GCol& c, const ReduceJob& job, DenseLds& L, int bucket,
                                    const FirstPairs& first, Hook hook = Hook()) {
  constexpr int VE = sizeof(V) / 4;
  constexpr int PT = kCP / kBlock;   // pairs per thread and chunk
  const int tid = (int)threadIdx.x;
  const int lane = tid & (kWave - 1), wave = tid >> 6;
  const int lpr_log2 = c.lpr_log2;
  const int sub = lane & ((1 << lpr_log2) - 1);
  const bool live = sub < c.chunks;
  const int groups = kBlock >> lpr_log2;
  const int my_group = tid >> lpr_log2;
  const int32_t n_pairs = job.n_pairs;
  if (n_pairs <= 0) return;
  const int64_t* prow = job.prow;
  const int32_t* pseg = job.pseg;
  const bool one_chunk = n_pairs <= kCP;   // the pairs stay in registers between the stages
  const float lr = STEP ? job.lr : 0.0f;
  constexpr bool adagrad = STEP == 2;
  const bool emit = !(STEP && job.no_emit);
  const bool dscaled = job.scale && c.combiner != HBK_COMBINER_SUM && c.splits != nullptr;   // uniform

  int64_t r_in[PT];
  int32_t seg_in[PT];
  auto load_pairs = [&](int32_t cb) {
#pragma unroll
    for (int k = 0; k < PT; ++k) {
      const int32_t e = cb + k * kBlock + tid;
      r_in[k] = -1;
      seg_in[k] = e;
      if (e < n_pairs) {
        r_in[k] = HBK_PAIR_LOAD(prow + e);
        if (pseg != nullptr) seg_in[k] = HBK_PAIR_LOAD(pseg + e);
      }
    }
  };
#pragma unroll
  for (int k = 0; k < PT; ++k) {   // (requested by the caller; they travel while the bitmaps are cleared)
    r_in[k] = first.row[k];
    seg_in[k] = first.seg[k];
  }
  const uint32_t M = c.dense_mul;
  const uint32_t base = (uint32_t)dense_first_row(M, bucket);
  uint64_t lim = dense_first_row(M, bucket + 1);
  if (lim > c.map.rows) lim = c.map.rows;
  const int words = (int)((lim - base + 31) >> 5);   // <= kDenseWords (host: plan_of)

  __syncthreads();   // a workgroup may run several jobs (merge): the previous one is done with L
  for (int w = tid; w < words; w += kBlock) {
    L.present[w] = 0u;
    L.dup[w] = 0u;
  }
  if (SORT) {
    for (int i = tid; i < kCP; i += kBlock) L.dcnt[i] = 0;
  }
  if (tid == 0) {
    L.n_dlist[0] = 0;
    L.n_dlist[1] = 0;
    L.n_again = 0;
  }
  __syncthreads();
  HBK_STAMP(2);

  // A: rows -> bitmaps.  A bit that is already set is not set again: the pairs of a hot row
  // would serialise on its word (64 same-address LDS atomics per instruction).
  // The pairs that find their bit set are counted: rows of the bucket = pairs - those, known
  // right behind the barrier, a scan earlier than the ranks.
  int n_again = 0;   // (wave-uniform)
  for (int32_t cb = 0; cb < n_pairs; cb += kCP) {
    if (cb > 0) load_pairs(cb);
#pragma unroll
    for (int k = 0; k < PT; ++k) {
      bool again = false;
      if (r_in[k] >= 0) {
        const uint32_t off = (uint32_t)r_in[k] - base;
        const int w = (int)(off >> 5);
        const uint32_t bit = 1u << (off & 31u);
        uint32_t old = L.present[w];
        if ((old & bit) == 0u) old = atomicOr(&L.present[w], bit);
        again = (old & bit) != 0u;
        if (again && (L.dup[w] & bit) == 0u) atomicOr(&L.dup[w], bit);
      }
      n_again += (int)__builtin_popcountll(__ballot(again));
    }
  }
  if (lane == 0 && n_again > 0) atomicAdd(&L.n_again, n_again);
  __syncthreads();
  HBK_STAMP(3);

  // One global atomic per workgroup claims the output range of the bucket's rows; a returning
  // device-scope atomic takes microseconds under load: it is issued as soon as the count is known
  // and its round trip runs beside the scan of B and the gradient loads of C.  Step only: just
  // the count is wanted, nobody waits for it.
  int32_t claimed = 0;
  if (HBK_DENSE_EARLY_CLAIM && tid == kBlock - 1) {
    const int32_t n_rows = n_pairs - L.n_again;
    if (!emit) {
      __hip_atomic_fetch_add(job.out_counter, n_rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      claimed = atomicAdd(job.out_counter, n_rows);
    }
  }

  // B: rows before every word, present and dup counts packed in one scan (both <= kDenseSpan < 2^16)
  {
    uint32_t cnt[kDenseWPT], sum = 0;
#pragma unroll
    for (int q = 0; q < kDenseWPT; ++q) {
      const int w = tid * kDenseWPT + q;
      cnt[q] = 0;
      if (w < words) {
        cnt[q] = (uint32_t)__builtin_popcount(L.present[w]) |
                 ((uint32_t)__builtin_popcount(L.dup[w]) << 16);
      }
      sum += cnt[q];
    }
    uint32_t incl = sum;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const uint32_t y = (uint32_t)__shfl_up((int)incl, o, kWave);
      if (lane >= o) incl += y;
    }
    if (lane == kWave - 1) L.wave_tot[wave] = (int32_t)incl;
    __syncthreads();
    uint32_t run = incl - sum;
    for (int w = 0; w < wave; ++w) run += (uint32_t)L.wave_tot[w];
#pragma unroll
    for (int q = 0; q < kDenseWPT; ++q) {
      const int w = tid * kDenseWPT + q;
      if (w < words) L.pre[w] = run;
      run += cnt[q];
    }
    if (tid == kBlock - 1) {
      L.n_dup = (int32_t)(run >> 16);
      if (!HBK_DENSE_EARLY_CLAIM) {
        const int32_t n_rows = (int32_t)(run & 0xffffu);
        if (!emit) {
          __hip_atomic_fetch_add(job.out_counter, n_rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          claimed = atomicAdd(job.out_counter, n_rows);
        }
      }
    }
  }
  hook();   // (the next job's pairs are requested here)
  __syncthreads();
  HBK_STAMP(4);

  const int n_dup = L.n_dup;
  int cap = kRedFloats / c.dim;   // dup rows summed per round
  if (cap > kCP) cap = kCP;
  const int n_rounds = n_dup > 0 ? (n_dup + cap - 1) / cap : 1;
  // gradient rows a lane keeps in flight (register budget: the step's table / accumulator rows
  // travel with them)
  constexpr int kDepth = STEP == 2 ? 3 : STEP ? 4 : 8;
  int32_t base_u = 0;
  bool have_base = !emit;   // step only: no output positions
  int par = 0;
  for (int round = 0; round < n_rounds; ++round) {
    const int d0 = round * cap;
    const int d1 = n_dup < d0 + cap ? n_dup : d0 + cap;
    for (int i = tid; i < (d1 - d0) * c.dim; i += kBlock) L.red[i] = 0.0f;
    bool used_red = false;                                      // LDS sums to flush after the round
    const bool last_use = one_chunk && round + 1 == n_rounds;   // nobody takes tickets after this
    for (int32_t cb = 0; cb < n_pairs; cb += kCP, par ^= 1) {
      const int32_t n_chunk = n_pairs - cb < kCP ? n_pairs - cb : kCP;
      if (!one_chunk) {
        load_pairs(cb);
        __syncthreads();   // the walks of the chunk before are over: seg / code / dlist are free
      }
      // per pair: output rank of its row; dup pairs of this round go on the chunk's list
      int m_[PT], tk_[PT];   // listed pairs: dup row inside the round, ticket inside the row
#pragma unroll
      for (int k = 0; k < PT; ++k) {
        const int e = k * kBlock + tid;
        bool listed = false;
        m_[k] = -1;
        tk_[k] = 0;
        if (r_in[k] >= 0) {
          const uint32_t off = (uint32_t)r_in[k] - base;
          const int w = (int)(off >> 5);
          const uint32_t below = (1u << (off & 31u)) - 1u;
          const uint32_t pw = L.present[w], dw = L.dup[w], pr = L.pre[w];
          int32_t code = (int32_t)(pr & 0xffffu) + __builtin_popcount(pw & below);
          if ((dw >> (off & 31u)) & 1u) {
            const int32_t dr = (int32_t)(pr >> 16) + __builtin_popcount(dw & below);
            code = kDupBit | dr;
            if (dr >= d0 && dr < d1) {
              listed = true;
              m_[k] = dr - d0;
              if (SORT) tk_[k] = atomicAdd(&L.dcnt[dr - d0], 1);
              L.doff[dr - d0] = (uint16_t)off;   // (every pair of the row writes the same value)
            }
          }
          L.seg[e] = seg_in[k];
          L.code[e] = code;
          L.off[e] = (uint16_t)off;
        }
        const unsigned long long m = __ballot(listed);
        if (m != 0ull) {
          const int first = __builtin_ctzll(m);
          int at = 0;
          if (lane == first) at = atomicAdd(&L.n_dlist[par], (int)__builtin_popcountll(m));
          at = __builtin_amdgcn_readlane(at, first);
          if (listed) L.dlist[at + rank_below(m)] = (uint16_t)e;
        }
      }
      if (tid == 0) L.n_dlist[par ^ 1] = 0;   // the next chunk's list (its last users are past the
                                              // barrier above, or there is no chunk before)
      __syncthreads();
      HBK_STAMP(5);

      // this round's dup pairs of the chunk.  Many: they are counting-sorted by row here (the
      // tickets taken above) and walked in D; the tickets leave the registers before C needs them
      const int n_dl = L.n_dlist[par];
      const int nd = d1 - d0;
      const bool sorted = SORT && HBK_DENSE_SORT && n_dl > kSortMin;   // uniform
      if (__builtin_expect(sorted, 0)) {
        {
          int32_t cn[PT], sum = 0;
#pragma unroll
          for (int k = 0; k < PT; ++k) {
            const int i = tid * PT + k;
            cn[k] = i < nd ? L.dcnt[i] : 0;
            sum += cn[k];
          }
          int32_t incl = sum;
#pragma unroll
          for (int o = 1; o < kWave; o <<= 1) {
            const int32_t y = __shfl_up(incl, o, kWave);
            if (lane >= o) incl += y;
          }
          if (lane == kWave - 1) L.wave_tot[wave] = incl;
          __syncthreads();
          int32_t run = incl - sum;
          for (int w = 0; w < wave; ++w) run += L.wave_tot[w];
#pragma unroll
          for (int k = 0; k < PT; ++k) {
            const int i = tid * PT + k;
            if (i < nd) L.drun[i] = (uint32_t)run | ((uint32_t)(run + cn[k]) << 16);
            run += cn[k];
          }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PT; ++k) {
          if (m_[k] >= 0) L.dlist[(L.drun[m_[k]] & 0xffffu) + tk_[k]] = (uint16_t)(k * kBlock + tid);
        }
        __syncthreads();
        if (!last_use) {
          for (int i = tid; i < nd; i += kBlock) L.dcnt[i] = 0;   // (next read: behind a barrier)
        }
      } else if (SORT && !last_use) {
        for (int i = tid; i < nd; i += kBlock) L.dcnt[i] = 0;   // (next read: behind a barrier)
      }

      // C: the pairs that are alone on their row (first round only).  When the chunk's dup pairs
      // are few (not sorted), their gradient rows travel in the same round and go into the rows'
      // LDS sums (ds_add_f32) when they arrive.
      if (round == 0) {
        const bool fold = HBK_DENSE_FOLD && !sorted && n_dl > 0;   // uniform
        used_red = used_red || fold;
        HBK_SUBSTAMP(0);
        for (int e0 = 0; e0 < n_chunk; e0 += kDepth * groups) {
          V g[kDepth], tv[STEP ? kDepth : 1], av[STEP == 2 ? kDepth : 1];
          int32_t n_[kDepth];
          uint32_t mask = 0, dmask = 0;
#pragma unroll
          for (int k = 0; k < kDepth; ++k) {
            const int i = e0 + k * groups + my_group;
            g[k] = zero_v<V>();
            n_[k] = 0;
            if (i < n_chunk && live) {
              const int32_t code = L.code[i];
              if ((code & kDupBit) == 0) {
                mask |= 1u << k;
                g[k] = load_grad_raw<V>(c, job, L.seg[i], sub, &n_[k]);
                if (STEP && lr != 0.0f) {
                  const int64_t toff = (int64_t)(base + L.off[i]) * c.tpitch + (int64_t)sub * VE;
                  tv[STEP ? k : 0] =
                      HBK_STEP_LOAD(reinterpret_cast<const V*>(c.table + toff));
                  if (STEP == 2) {
                    av[STEP == 2 ? k : 0] =
                        HBK_STEP_LOAD(reinterpret_cast<const V*>(c.accum + toff));
                  }
                }
              } else if (fold && (code & (kDupBit - 1)) < d1) {   // (round 0: d0 = 0)
                dmask |= 1u << k;
                g[k] = load_grad_raw<V>(c, job, L.seg[i], sub, &n_[k]);
              }
            }
          }
          HBK_SUBSTAMP(1);
          if (!have_base) {   // uniform
            if (tid == kBlock - 1) L.base_u = job.out_base + claimed;
            __syncthreads();
            base_u = L.base_u;
            have_base = true;
          }
          HBK_SUBSTAMP(2);
#ifdef HBK_BWD_STAMPS
          __builtin_amdgcn_s_waitcnt(0x0f70);   // (probe) the gradient rows are here
          HBK_SUBSTAMP(3);
#endif
#pragma unroll
          for (int k = 0; k < kDepth; ++k) {
            if (((mask | dmask) >> k & 1u) == 0) continue;
            const int i = e0 + k * groups + my_group;
            if (dscaled) g[k] = scale_grad<V>(c, g[k], n_[k]);
            if (dmask >> k & 1u) {
              float* r = &L.red[(size_t)(L.code[i] & (kDupBit - 1)) * c.dim + (size_t)sub * VE];
#pragma unroll
              for (int q = 0; q < VE; ++q) atomicAdd(r + q, reinterpret_cast<const float*>(&g[k])[q]);
              continue;
            }
            if (emit) emit_row<V>(c, job, base_u + L.code[i], true, sub, g[k]);
            if (STEP && lr != 0.0f) {
              const int64_t toff = (int64_t)(base + L.off[i]) * c.tpitch + (int64_t)sub * VE;
              step_row<V>(c, adagrad, lr, toff, g[k], tv[STEP ? k : 0],
                          STEP == 2 ? av[STEP == 2 ? k : 0] : zero_v<V>());
            }
          }
        }
      }

      // D: this round's dup pairs.
      if (__builtin_expect(sorted, 0)) {
        // Many: LDS float atomics are slow (~3 clocks per lane and add on a CU, measured: with 3
        // pairs per row the whole backward took twice as long as with hashed buckets), so the
        // pairs are counting-sorted by row (the tickets taken above) and the sorted list is walked
        // FLAT, sums in registers: per round a lane group takes kDepth consecutive positions, all
        // gradient rows in flight at once; the group where a run starts owns it and adds what its
        // successors hold of it (heads, LDS); a run that goes on past the round's end is carried.
        WalkArgs wa;
        wa.grad = job.grad;
        wa.splits = c.splits;
        wa.out_vals = job.out_vals;
        wa.table = c.table;
        wa.accum = c.accum;
        wa.stride = job.stride;
        wa.dim = c.dim;
        wa.tpitch = c.tpitch;
        wa.chunks = c.chunks;
        wa.n_dl = n_dl;
        wa.d0 = d0;
        wa.base_u = base_u;
        wa.base = base;
        wa.lr = lr;
        wa.lpr_log2 = c.lpr_log2;
        wa.combiner = c.combiner;
        wa.seg_is_offset = job.seg_is_offset;
        wa.scale = job.scale;
        wa.one_chunk = one_chunk;
        wa.no_emit = job.no_emit;
        dense_walk<V, STEP>(wa, L);
      } else if (round > 0 || !HBK_DENSE_FOLD) {
        // Few, and not the first round (wide rows: only 2048 / dim sums fit LDS at a time; the
        // first round's pairs went with C): summed into the rows' LDS sums by ds_add_f32; what a
        // lane group holds for one row in consecutive registers is added up first.
        used_red = used_red || n_dl > 0;
        for (int i0 = 0; i0 < n_dl; i0 += kDepth * groups) {
          V g[kDepth];
          int dr[kDepth];
          int32_t n_[kDepth];
#pragma unroll
          for (int k = 0; k < kDepth; ++k) {
            const int i = i0 + k * groups + my_group;
            dr[k] = -1;
            n_[k] = 0;
            g[k] = zero_v<V>();
            if (i < n_dl && live) {
              const int e = (int)L.dlist[i];
              dr[k] = (L.code[e] & (kDupBit - 1)) - d0;
              g[k] = load_grad_raw<V>(c, job, L.seg[e], sub, &n_[k]);
            }
          }
#pragma unroll
          for (int k = 0; k < kDepth; ++k) {
            if (dscaled) g[k] = scale_grad<V>(c, g[k], n_[k]);
          }
#pragma unroll
          for (int k = 1; k < kDepth; ++k) {
            if (dr[k] >= 0 && dr[k] == dr[k - 1]) {
              g[k] = g[k] + g[k - 1];
              dr[k - 1] = -1;
            }
          }
#pragma unroll
          for (int k = 0; k < kDepth; ++k) {
            if (dr[k] >= 0) {
              float* r = &L.red[(size_t)dr[k] * c.dim + (size_t)sub * VE];
#pragma unroll
              for (int q = 0; q < VE; ++q) atomicAdd(r + q, reinterpret_cast<const float*>(&g[k])[q]);
            }
          }
        }
      }
    }
    if (!one_chunk || used_red) {   // uniform
      __syncthreads();   // the round's sums are complete
      for (int m = d0 + my_group; m < d1; m += groups) {
        if (!live) continue;
        const uint32_t off = L.doff[m - d0];
        const int w = (int)(off >> 5);
        const int32_t rank = (int32_t)(L.pre[w] & 0xffffu) +
                             __builtin_popcount(L.present[w] & ((1u << (off & 31u)) - 1u));
        emit_step_row<V, STEP>(c, job, lr, base_u + rank, true, (int64_t)(base + off), sub,
                               *reinterpret_cast<const V*>(&L.red[(size_t)(m - d0) * c.dim +
                                                                  (size_t)sub * VE]));
      }
    }
    if (round + 1 < n_rounds) __syncthreads();   // the next round clears the sums, takes tickets
  }
  HBK_STAMP(6);
  // the row numbers, sorted, straight from the bitmap
  if (emit) {
    for (int w = tid; w < words; w += kBlock) {
      uint32_t m = L.present[w];
      int64_t* o = job.out_rows + base_u + (int32_t)(L.pre[w] & 0xffffu);
      const int64_t r0 = (int64_t)base + 32 * (int64_t)w;
      while (m != 0u) {
        *o++ = r0 + __builtin_ctz(m);
        m &= m - 1u;
      }
    }
  }
}

// Two instantiations (16-byte / 4-byte chunks) so the common one keeps its registers low; a job
// whose column is of the other kind is skipped.  Every workgroup's job is one 16-byte descriptor
// written by the scan kernel (slots [0, P) of a column are its buckets = range 0 of a split
// bucket, the e_max spare slots take the listed extra ranges): one scalar load instead of a chain
// of dependent loads (column, list of extras, bucket start, bucket end) at the head of every
// workgroup's critical path.  (Persistent workgroups that fetch the next job's pairs while the
// current one runs were tried: the state carried around the loop spills, 224 us vs 143.)
// MODE: 0 hashed buckets, 1 row-range buckets with the bitmaps of 4b, 2 row-sorted (4c)
template <typename V, int MODE>
__device__ inline bool decode_job(const GArgs& a, int my_b0, int vb, const int4& d, float lr,
                                  int* ci_out, ReduceJob* job) {
  if (d.z < 0 || d.y <= 0) return false;   // wave-uniform
  int ci = (int)__builtin_popcountll(__ballot(my_b0 <= vb)) - 1;
  ci = __builtin_amdgcn_readfirstlane(ci);
  const GCol& c = a.col[ci];
  if ((c.vec4 != 0) != (sizeof(V) == 16) || (c.dense_mul != 0) != (MODE != 0) ||
      (c.rowsort != 0) != (MODE == 2)) {
    return false;
  }
  *ci_out = ci;
  const int32_t start = d.x, n_b = d.y, bucket = d.z, range = d.w;
  job->no_emit = false;
  job->packed = c.packed != 0;
  job->grad = c.grad_out;
  job->scale = true;
  job->seg_is_offset = c.n_runs > 0;
  job->stride = c.grad_stride;
  if (n_b > c.split_t) {
    const int32_t lo = range * c.split_t;
    job->prow = c.pair_row[0] + start + lo;
    job->pseg = c.packed ? nullptr : c.pair_seg[0] + start + lo;
    job->n_pairs = n_b - lo < c.split_t ? n_b - lo : c.split_t;
    job->out_rows = c.part_rows;
    job->out_vals = c.part_vals;
    job->out_counter = c.pcount + bucket;
    job->out_base = start;
    job->lr = 0.0f;
    job->apply = HBK_APPLY_SGD;
  } else {
    job->prow = c.pair_row[0] + start;
    job->pseg = c.packed ? nullptr : c.pair_seg[0] + start;
    job->n_pairs = n_b;
    job->out_rows = c.unique_rows;
    job->out_vals = c.grad_rows;
    job->out_counter = c.counter;
    job->out_base = 0;
    job->lr = lr;
    job->apply = a.apply;
    // step only: a dense job emits every row complete whatever its chunk count; a hashed one may
    // see a row span chunks (deferred step from the emitted rows)
    job->no_emit = c.no_emit != 0 && lr != 0.0f &&
                   (MODE == 1 || n_b <= (MODE == 2 ? kRsCap : kCP));
  }
  return true;
}

// The kernels are instantiated per row kind (16-byte / 4-byte chunks), optimizer (none / SGD /
// Adagrad) and bucket kind (hashed / dense); the host sorts a launch group's columns by kind, so
// the job slots of one kind are one range [slot0, slot0 + grid) and no workgroup starts for
// nothing.
template <typename V, int STEP, bool WIDE>
__global__ __launch_bounds__(kBlock, HBK_BWD_WAVES) void bwd_reduce_kernel(const GArgs a,
                                                                          const int4* desc,
                                                                          int slot0, int total,
                                                                          const int32_t* poison) {
  __shared__ ReduceLds lds[kTeams];
  if (poisoned(poison)) return;   // the grouping launch gave up: no descriptors, no pairs
  HBK_STAMP_BEGIN()
  const int lane = (int)threadIdx.x & (kWave - 1);
  const int team = (int)threadIdx.x / kTeam;
  constexpr int kKind = 2 * (WIDE ? 3 : 2) + (sizeof(V) == 4 ? 1 : 0);   // (the host's ColInfo.kind)
  int jb = (int)blockIdx.x;
  if ((a.xcd_w >> kKind) & 1) {
    const int x = jb & 7;
    jb = a.xcd_start[kKind][x] + (jb >> 3);
    if (jb >= a.xcd_start[kKind][x + 1]) return;   // (the whole workgroup)
  } else {
    jb = xcd_contiguous(jb, (int)gridDim.x, (a.xcd >> kKind) & 1);
  }
  const int vb = slot0 + jb * kTeams + team;   // the team's job slot
  if (vb >= total) return;                          // team-uniform; no workgroup barrier follows
  // two independent loads (the job, the columns' first slots): one round trip
  const int4 d = desc[vb];
  const int my_b0 = lane < a.n_cols ? a.bucket0[lane] : 0x7fffffff;
  ReduceJob job;
  int ci;
  if (!decode_job<V, 0>(a, my_b0, vb, d, a.lr, &ci, &job)) return;
  HBK_STAMP(1);
  bucket_reduce<V, STEP, WIDE>(a.col[ci], job, lds[team]);
  HBK_STAMP(7);
}


// (Persistent workgroups that request the next job's descriptor and pairs while the current job
// runs -- the two dependent round trips at the head of a job are 5 of its ~14 us -- measured 5 %
// SLOWER on the config-2 backward and 10 % slower on ragged columns: the registers that carry the
// next job's pairs cost more than the round trips they hide.)
template <typename V, int STEP, bool SORT>
__global__ __launch_bounds__(kBlock, HBK_BWD_DENSE_WAVES(SORT)) void bwd_dense_kernel(
    const GArgs a, const int4* desc, int slot0, int total, const int32_t* poison) {
  __shared__ DenseLds lds;
  if (poisoned(poison)) return;   // the grouping launch gave up: no descriptors, no pairs
  HBK_STAMP_BEGIN()
  const int lane = (int)threadIdx.x & (kWave - 1);
  constexpr int kKind = 2 * (SORT ? 1 : 0) + (sizeof(V) == 4 ? 1 : 0);   // (the host's ColInfo.kind)
  int jb = (int)blockIdx.x;
  if ((a.xcd_w >> kKind) & 1) {
    const int x = jb & 7;
    jb = a.xcd_start[kKind][x] + (jb >> 3);
    if (jb >= a.xcd_start[kKind][x + 1]) return;   // (the whole workgroup)
  } else {
    jb = xcd_contiguous(jb, (int)gridDim.x, (a.xcd >> kKind) & 1);
  }
  const int vb = slot0 + jb;
  if (vb >= total) return;
  const int4 d = desc[vb];
  const int my_b0 = lane < a.n_cols ? a.bucket0[lane] : 0x7fffffff;
  ReduceJob job;
  int ci;
  if (!decode_job<V, 1>(a, my_b0, vb, d, a.lr, &ci, &job)) return;
  FirstPairs first;
  load_first_pairs(job, first);
  HBK_STAMP(1);
  dense_reduce<V, STEP, SORT>(a.col[ci], job, lds, d.z, first);
  HBK_STAMP(7);
}

#include "lookup_bwd_rowsort.h"

// The partial entries of a split bucket -> final rows.  Every split bucket has exactly one entry
// with range index 1 in the column's list of extra ranges; the column's merge blocks (at most
// kMergeBlocks) share that list round robin.
constexpr int kMergeBlocks = 8;

#define HBK_FIND_COL_AT(ARGS, FIELD, BLOCK)                                        \
  int ci;                                                                          \
  {                                                                                \
    const int l__ = (int)threadIdx.x & (kWave - 1);                                \
    const int v__ = l__ < (ARGS).n_cols ? (ARGS).FIELD[l__] : 0x7fffffff;          \
    ci = (int)__builtin_popcountll(__ballot(v__ <= (BLOCK))) - 1;                  \
    ci = __builtin_amdgcn_readfirstlane(ci);                                       \
  }                                                                                \
  const auto& c = (ARGS).col[ci];

// the column's row count goes from its workspace counter to the caller's n_unique: by the last of
// the column's merge blocks to get here (no launch of its own).  Every claim of a block has
// returned before the block counts itself done, so the last one reads the final count.
__device__ inline void merge_done(const GCol& c, int blocks) {
  __syncthreads();
  if (c.det) {   // (uniform) deterministic: the sum of the buckets' row counts (bwd_rowsort_count_kernel); one merge block
    if (threadIdx.x < kWave) {
      int32_t n = 0;
      for (int p = (int)threadIdx.x; p < c.n_buckets; p += kWave) n += c.pcount[p];
#pragma unroll
      for (int o = kWave / 2; o > 0; o >>= 1) n += __shfl_xor(n, o, kWave);
      if (threadIdx.x == 0) *c.n_unique = n;
    }
    return;
  }
  if (threadIdx.x == 0) {
    const int32_t done = __hip_atomic_fetch_add(c.counter + 1, 1, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
    if (done == blocks - 1) {
      *c.n_unique = __hip_atomic_load(c.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__device__ inline void merge_job(const GArgs& a, const GCol& c, int bucket, ReduceJob* job) {
  const int32_t start = c.bstart[bucket];
  job->prow = c.part_rows + start;
  job->pseg = nullptr;
  job->packed = false;
  job->grad = c.part_vals + (int64_t)start * c.dim;
  job->n_pairs = c.pcount[bucket];
  job->scale = false;
  job->seg_is_offset = false;
  job->stride = c.dim;
  job->out_rows = c.unique_rows;
  job->out_vals = c.grad_rows;
  job->out_counter = c.counter;
  job->out_base = 0;
  job->lr = a.lr;
  job->apply = a.apply;
  job->no_emit = false;
}

template <typename V, int STEP, bool WIDE>
__global__ __launch_bounds__(kBlock, HBK_BWD_WAVES) void bwd_merge_kernel(const GArgs a, int block0,
                                                                         const int32_t* poison) {
  __shared__ ReduceLds lds[kTeams];
  if (poisoned(poison)) return;
  const int block = block0 + (int)blockIdx.x;
  HBK_FIND_COL_AT(a, merge0, block)
  const int team = (int)threadIdx.x / kTeam;
  const int n_extra = *c.n_extra;
  const int blocks = c.e_max < kMergeBlocks ? c.e_max : kMergeBlocks;
  for (int e = (block - c.merge0) * kTeams + team; e < n_extra; e += blocks * kTeams) {
    if (c.work[2 * e + 1] != 1) continue;
    ReduceJob job;
    merge_job(a, c, c.work[2 * e], &job);
    bucket_reduce<V, STEP, WIDE>(c, job, lds[team]);
  }
  merge_done(c, blocks);
}

template <typename V, int STEP>
__global__ __launch_bounds__(kBlock, HBK_BWD_DENSE_WAVES(true)) void bwd_dense_merge_kernel(const GArgs a,
                                                                                     int block0,
                                                                                     const int32_t* poison) {
  __shared__ DenseLds lds;
  if (poisoned(poison)) return;
  const int block = block0 + (int)blockIdx.x;
  HBK_FIND_COL_AT(a, merge0, block)
  const int n_extra = *c.n_extra;
  const int blocks = c.e_max < kMergeBlocks ? c.e_max : kMergeBlocks;
  for (int e = block - c.merge0; e < n_extra; e += blocks) {
    if (c.work[2 * e + 1] != 1) continue;   // uniform
    ReduceJob job;
    merge_job(a, c, c.work[2 * e], &job);
    FirstPairs first;
    load_first_pairs(job, first);
    dense_reduce<V, STEP, true>(c, job, lds, c.work[2 * e], first);
  }
  merge_done(c, blocks);
}

template <typename V, int STEP>
__global__ __launch_bounds__(kBlock, 4) void bwd_rowsort_merge_kernel(const GArgs a, int block0,
                                                                     const int32_t* poison) {
  __shared__ RsLds lds;
  if (poisoned(poison)) return;
  const int block = block0 + (int)blockIdx.x;
  HBK_FIND_COL_AT(a, merge0, block)
  const int n_extra = *c.n_extra;
  const int blocks = c.e_max < kMergeBlocks ? c.e_max : kMergeBlocks;
  for (int e = block - c.merge0; e < n_extra; e += blocks) {
    if (c.work[2 * e + 1] != 1) continue;   // uniform
    ReduceJob job;
    merge_job(a, c, c.work[2 * e], &job);
    rowsort_reduce<V, STEP>(c, job, lds, c.work[2 * e]);
  }
  merge_done(c, blocks);
}

#include "lookup_bwd_det.h"

// ---- d(stitch + combiner): permutation scatter (hbk_group_stitch_bwd) ----------------------
constexpr int kIters = 4;
constexpr int kMaxStitchCols = 128;

struct SCol {
  const float* grad_out;
  const int32_t* splits;
  const int32_t* index;
  float* grad_rows;
  const int64_t* run_start;  // segmented destination (n_runs > 0)
  const int64_t* run_base;
  int32_t n_runs;
  int32_t grad_stride;       // floats between rows of grad_out
  int64_t n_seg;
  int32_t dim;
  int32_t chunks;
  uint8_t lpr_log2, combiner, vec4, pad_;
  int32_t tile0;
};

struct SArgs {
  int32_t n_cols;
  int32_t pad_;
  int32_t tile0[kMaxStitchCols];   // first block of every column (see GArgs)
  SCol col[kMaxStitchCols];
};
static_assert(sizeof(SArgs) <= 20480, "kernarg budget");
static_assert(kMaxStitchCols <= 2 * kWave, "two lanes-worth of columns in the stitch search");

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
        reinterpret_cast<const V*>(c.grad_out + s * (int64_t)c.grad_stride + (int64_t)sub * VE));
    const int32_t n = end - beg;
    if (c.combiner == HBK_COMBINER_MEAN) {
      g = g / (float)n;
    } else if (c.combiner == HBK_COMBINER_SQRTN) {
      g = g / sqrtf((float)n);
    }
    for (int32_t j = beg; j < end; ++j) {
      const int64_t r = c.index[j];
      int64_t off = r * c.dim;
      if (c.n_runs > 0) {
        int k = 0;
        while (k + 1 < c.n_runs && c.run_start[k + 1] <= r) ++k;
        off = c.run_base[k] + (r - c.run_start[k]) * c.dim;
      }
      *reinterpret_cast<V*>(c.grad_rows + off + (int64_t)sub * VE) = g;
    }
  }
}

__global__ __launch_bounds__(kBlock) void stitch_bwd_kernel(const SArgs a) {
  int ci;
  {
    const int l = (int)threadIdx.x & (kWave - 1);
    const int t0 = l < a.n_cols ? a.tile0[l] : 0x7fffffff;
    const int t1 = l + kWave < a.n_cols ? a.tile0[l + kWave] : 0x7fffffff;
    ci = (int)__builtin_popcountll(__ballot(t0 <= (int)blockIdx.x)) +
         (int)__builtin_popcountll(__ballot(t1 <= (int)blockIdx.x)) - 1;
    ci = __builtin_amdgcn_readfirstlane(ci);
  }
  const SCol& c = a.col[ci];
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
  int n_buckets;
  int64_t tiles;
  int32_t split_t;   // a bucket above this many pairs is reduced by several workgroups
  int32_t e_max;
  uint32_t dense_mul;   // != 0: row-range buckets (4b)
  bool dense_sort;      // the dense instantiation with the sorted walk (many repeated rows expected)
  bool rowsort;         // row-range buckets of ~7/8 kRsCap pairs reduced by the row-sorted job (4c)
};

// options (hbk_set_option): bwd_buckets_log2 forces the bucket count to 1 << value (0 = one
// bucket per column) so that multi-chunk buckets, rows spanning chunks and the multi-pass path
// can be exercised; bwd_bucket_pairs sets the aimed pairs per bucket (tuning); bwd_dense = 0
// keeps hashed buckets for every column
int forced_log2p() { return options().bwd_buckets_log2; }

ColPlan plan_of(int64_t n_ids, int32_t dim, int64_t rows, bool ragged, bool det = false) {
  ColPlan p;
  // Aim at 7/8 of a chunk per bucket: bucket sizes are Poisson around the aim, so ~0.1 % of the
  // buckets need a second (short) chunk, while the per-workgroup fixed cost (table init,
  // 1024-slot scan) is spread over as many pairs as possible.  Measured, config 2 backward:
  // aim 256 171 us, 320 165, 384 157, 448 147, 512 149.
  int64_t target = kCP * 7 / 8;
  if (options().bwd_bucket_pairs > 0) target = options().bwd_bucket_pairs;
  int64_t nb = (n_ids + target - 1) / target;
  if (nb < 1) nb = 1;
  if (nb > kMaxBuckets) nb = kMaxBuckets;
  const int forced = forced_log2p();
  if (forced >= 0 && forced <= 14) nb = (int64_t)1 << forced;
  // Dense: bucket = floor(row * M / 2^32) with M = floor(2^32 P / rows) (< 2^32: P <= rows; any
  // smaller M is still a monotone map into [0, P)); a bucket then spans at most ceil(2^32 / M)
  // rows, which must fit the reduce stage's LDS bitmaps.
  // Where dense pays (measured through the C ABI, 26 x 65536 ids unless said; profiles/r03_*):
  //  * narrow rows only (dim <= 32: >= 32 rows per load instruction of the workgroup).  Wide rows
  //    lose to the hashed path: dim 128 uniform 417 vs 389 us, Zipf 482 vs 351 us; dim 64 with 3
  //    pairs per row 210 vs 165 us;
  //  * columns with one id per sample.  Ragged ones (8 ids per sample, 100 k rows) 780-880 vs 780 us;
  //  * few repeated rows expected (ids <= rows / 8; uniform config 2: ~14 of a bucket's 450 pairs):
  //    the LEAN instantiation, 103-107 vs 111-114 us (+ SGD 170 vs 175, step only 150-156 vs 162-165).
  //    With more repeats the instantiation with the sorted walk is level with the hashed path or a
  //    little ahead on 26 equal columns (99 / 116 vs 107 / 126 us at 33 / 330 pairs per row) but
  //    behind it in the config-5 mix (200 columns: 4.58 vs 4.36 ms with SGD, 6.09 vs 5.53 ms with
  //    Adagrad -- six kernel kinds per launch group instead of two): hashed.
  // bwd_dense = 2 forces dense (with the sorted walk) wherever the row range fits (tests).
  p.dense_mul = 0;
  p.dense_sort = true;
  p.rowsort = false;
  // Row-sorted buckets (4c): columns whose batch is dense in the table, rows <= bwd_rowsort_ratio
  // x ids (default 8: where the lean bitmaps of 4b end) -- ragged columns, small and medium tables,
  // any dim.  A bucket aims at 7/8 of kRsCap pairs and must span <= kRsSpan rows, so sparser
  // columns would get smaller jobs (ratio 16: ~1000 pairs).  bwd_dense = 3 forces it wherever the
  // row range fits (tests).
  {
    const int dense_opt = options().bwd_dense;
    // narrow rows (dim <= 32) gain up to FOUR times the ratio (config 2, rows = 15 x ids: 106 -> 93 us;
    // ragged dim 16, 10 M rows = 19 x ids, + SGD: 834-889 -> 683-703 us in-box at 32 x, 719 at 16 x,
    // 704 at 64 x; the config-5 mix +- 1 %); wide rows with skewed ids lose there (config 4 Zipf,
    // dim 128: 343 -> 447 us)
    const int64_t ratio = (int64_t)options().bwd_rowsort_ratio * (dim <= 32 ? 4 : 1);
    // deterministic columns (`det`: option bwd_deterministic = 1 or the column's flag): row-sorted jobs wherever the row range fits -- the one
    // reduce kind with an in-order form (lookup_bwd_rowsort.h) -- and no bucket is split over
    // workgroups (partial sums joined by a merge are not the sequential sum)
    const bool want = det || dense_opt == 3 || (dense_opt == 1 && ratio > 0 && rows <= ratio * n_ids);
    if (kTeam == kBlock && want && rows >= 1 && rows < (1ll << 32)) {
      int64_t rs_target = (int64_t)kRsCap * 7 / 8;
      // wide rows: a lane group is 16-64 lanes, a workgroup holds 4-16 of them, and a job of 1792
      // pairs is a serial walk of 100-450 positions per lane group; option bwd_rowsort_pos (> 0)
      // sizes the job for that many positions per lane group instead
      if (options().bwd_rowsort_pos > 0) {
        int lanes = 1;
        while (lanes * 4 < dim) lanes *= 2;   // (16-byte chunks; 4-byte chunks only make the job smaller)
        int64_t by_pos = (int64_t)options().bwd_rowsort_pos * (kBlock / lanes);
        if (by_pos < 256) by_pos = 256;
        if (by_pos < rs_target) rs_target = by_pos;
      }
      if (options().bwd_bucket_pairs > 0 && options().bwd_bucket_pairs < rs_target) {
        rs_target = options().bwd_bucket_pairs;
      }
      int64_t P = (n_ids + rs_target - 1) / rs_target;
      if (forced >= 0 && forced <= 14) P = (int64_t)1 << forced;
      const int64_t by_span = (rows + kRsSpan - 2) / (kRsSpan - 1);   // buckets of < kRsSpan rows
      if (P < by_span) P = by_span;
      if (P < 1) P = 1;
      if (P > rows) P = rows;
      if (P <= kMaxBuckets) {
        uint64_t M = ((uint64_t)P << 32) / (uint64_t)rows;
        if (M > 0xffffffffull) M = 0xffffffffull;
        if (M >= 1 && (((uint64_t)1 << 32) + M - 1) / M <= (uint64_t)kRsSpan) {
          p.dense_mul = (uint32_t)M;
          p.rowsort = true;
          p.n_buckets = (int)P;
          p.tiles = (n_ids + kTile - 1) / kTile;
          int64_t t = 2 * ((n_ids + P - 1) / P) + 128;
          if (t < 2 * kRsCap) t = 2 * kRsCap;
          if (options().bwd_split_pairs > 0) t = options().bwd_split_pairs;
          if (det) t = 0x7fffffff;
          p.split_t = (int32_t)t;
          p.e_max = (int32_t)(n_ids / t + 1);
          return p;
        }
      }
    }
  }
  const bool forced_dense = options().bwd_dense == 2;
  const bool eligible = forced_dense || (dim <= 32 && !ragged && n_ids <= rows / 8);
  if (kTeam == kBlock && options().bwd_dense != 0 && eligible && rows >= 1 && rows < (1ll << 32)) {
    const int64_t P = nb < rows ? nb : rows;
    uint64_t M = ((uint64_t)P << 32) / (uint64_t)rows;
    if (M > 0xffffffffull) M = 0xffffffffull;
    if (M >= 1 && (((uint64_t)1 << 32) + M - 1) / M <= (uint64_t)kDenseSpan) {
      p.dense_mul = (uint32_t)M;
      p.dense_sort = forced_dense;
      nb = P;
    }
  }
  p.n_buckets = (int)nb;
  p.tiles = (n_ids + kTile - 1) / kTile;
  // hashing keeps ordinary buckets near n / P pairs; twice that (and >= 2 chunks) means a hot row
  // (measured, config 4 backward: ranges of 1024 pairs 391 us, 2048 417 us, 4096 466 us)
  int64_t t = 2 * ((n_ids + nb - 1) / nb) + 128;
  if (t < 2 * kCP) t = 2 * kCP;
  if (options().bwd_split_pairs > 0) t = options().bwd_split_pairs;
  p.split_t = (int32_t)t;
  p.e_max = (int32_t)(n_ids / t + 1);
  return p;
}

// row-sorted columns keep a pair as ONE word (GCol.packed)
inline bool pairs_packed(const ColPlan& p) { return p.rowsort && options().bwd_pairs_packed != 0; }

size_t col_workspace(const hbk_lookup_grad_column_t& h, bool det = false) {
  if (h.n_ids <= 0) return 0;
  const ColPlan p = plan_of(h.n_ids, h.dim, h.rows, h.row_splits != nullptr, det);
  size_t b = align8(((size_t)p.tiles * p.n_buckets) * 4);   // hist
  b += align8(((size_t)p.n_buckets + 1) * 4);          // bstart
  b += (size_t)h.n_ids * 8;                                // pair_row
  if (!pairs_packed(p)) b += align8((size_t)h.n_ids * 4);  // pair_seg
  if (h.row_splits != nullptr && options().bwd_seg_inline == 0) {
    b += align8((size_t)h.n_ids * 4);                      // seg_of
  }
  if (h.row_splits != nullptr && h.combiner != HBK_COMBINER_SUM) {
    b += align8((size_t)h.n_segments * h.dim * 4) + 16;   // the segments' scaled gradient rows
  }
  b += ((size_t)p.n_buckets + p.e_max) * sizeof(int4);     // desc (carved from the call's head)
  b += 256;                                                // the claim counter's own line
  b += align8((size_t)p.e_max * 8) + 8;                    // work, n_extra
  b += align8(((size_t)p.n_buckets) * 4);                 // pcount
  b += (size_t)h.n_ids * 8;                                // part_rows
  b += align8((size_t)h.n_ids * h.dim * 4) + 16;           // part_vals (16-byte aligned)
  if (h.grad_rows == nullptr) {                            // step only: scratch for the rows of the
    b += (size_t)h.n_ids * 8;                              // few jobs that must emit (several
    b += align8((size_t)h.n_ids * h.dim * 4) + 16;         // chunks, split buckets)
  }
  return b;
}

// deterministic mode, option value 1: the columns whose buckets fit the row-sorted jobs take their
// in-order form (lookup_bwd_rowsort.h); the others -- tables too large for row-range buckets -- and
// everything under option value 2 go through the sort of lookup_bwd_det.h
// 0: the default forms; 1: in id order, by the row-sorted jobs where they fit; 2: in id order, by sorting
inline int det_mode(const hbk_lookup_grad_column_t& h) {
  const int opt = options().bwd_deterministic;
  if (opt != 0) return opt;
  return (h.flags & HBK_GRAD_DETERMINISTIC) != 0 ? 1 : 0;
}
inline bool det_rowsort(const hbk_lookup_grad_column_t& h) {
  if (det_mode(h) != 1 || h.n_ids <= 0) return false;
  return plan_of(h.n_ids, h.dim, h.rows, h.row_splits != nullptr, true).rowsort;
}
// the columns of a call by the form they take: `fast` the in-order row-sorted jobs, `slow` the sort,
// `plain` the default forms (empty columns of a deterministic call ride with the sort, which clears
// their counts)
inline void det_split(int32_t n_cols, const hbk_lookup_grad_column_t* cols,
                      std::vector<hbk_lookup_grad_column_t>* fast,
                      std::vector<hbk_lookup_grad_column_t>* slow,
                      std::vector<hbk_lookup_grad_column_t>* plain) {
  for (int32_t c = 0; c < n_cols; ++c) {
    (det_mode(cols[c]) == 0 ? plain : det_rowsort(cols[c]) ? fast : slow)->push_back(cols[c]);
  }
}
inline bool any_deterministic(int32_t n_cols, const hbk_lookup_grad_column_t* cols) {
  if (options().bwd_deterministic != 0) return true;
  for (int32_t c = 0; c < n_cols; ++c) {
    if ((cols[c].flags & HBK_GRAD_DETERMINISTIC) != 0) return true;
  }
  return false;
}

// ---- host side of the deterministic backward (lookup_bwd_det.h) -----------------------------------
struct DetLayout {
  size_t keys, skeys, vals, svals, heads, ranks, temp, temp_bytes, total;
};
inline int bits_for(uint64_t v) {   // bits that hold every value in [0, v]
  int b = 1;
  while (b < 64 && (v >> b) != 0) ++b;
  return b;
}
// the call's launch groups run one after the other on the caller's stream and share one set of
// arrays sized for the largest of them (at most the call's ids)
DetLayout det_layout(int32_t n_cols, const hbk_lookup_grad_column_t* cols) {
  size_t n = 0;
  uint64_t max_rows = 1;
  for (int32_t c = 0; c < n_cols; ++c) {
    if (cols[c].n_ids <= 0) continue;
    n += (size_t)cols[c].n_ids;
    if ((uint64_t)cols[c].rows > max_rows) max_rows = (uint64_t)cols[c].rows;
  }
  DetLayout l = {};
  if (n == 0) return l;
  const int end_bit = bits_for(max_rows) + bits_for(kMaxCols - 1);
  const size_t t_sort = det_sort_temp_bytes(n, end_bit < 64 ? end_bit : 64);
  const size_t t_scan = det_scan_temp_bytes(n);
  size_t at = 256;
  auto take = [&](size_t bytes) {
    const size_t here = at;
    at += (bytes + 255) & ~(size_t)255;
    return here;
  };
  l.keys = take(n * 8);
  l.skeys = take(n * 8);
  l.vals = take(n * 4);
  l.svals = take(n * 4);
  l.heads = take(n * 4);
  l.ranks = take(n * 4);
  l.temp_bytes = t_sort > t_scan ? t_sort : t_scan;
  l.temp = take(l.temp_bytes);
  l.total = at;
  return l;
}

int det_backward(int32_t n_cols, const hbk_lookup_grad_column_t* cols, int32_t apply, float lr,
                 void* workspace, hipStream_t stream) {
  const DetLayout l = det_layout(n_cols, cols);
  for (int32_t c = 0; c < n_cols; ++c) {
    if (cols[c].n_ids == 0) HBK_HIP_OK(hipMemsetAsync(cols[c].n_unique, 0, sizeof(int32_t), stream));
  }
  if (l.total == 0) return HBK_OK;
  HBK_REQUIRE(l.temp_bytes > 256, "deterministic backward: the sort / scan primitives cannot be "
                                  "sized on this device");
  char* ws = reinterpret_cast<char*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  uint64_t* keys = reinterpret_cast<uint64_t*>(ws + l.keys);
  uint64_t* skeys = reinterpret_cast<uint64_t*>(ws + l.skeys);
  uint32_t* vals = reinterpret_cast<uint32_t*>(ws + l.vals);
  uint32_t* svals = reinterpret_cast<uint32_t*>(ws + l.svals);
  int32_t* heads = reinterpret_cast<int32_t*>(ws + l.heads);
  int32_t* ranks = reinterpret_cast<int32_t*>(ws + l.ranks);
  int32_t c0 = 0;
  while (c0 < n_cols) {
    DArgs a;
    memset(&a, 0, sizeof(a));
    int k = 0;
    int64_t total = 0, tiles = 0;
    uint64_t max_rows = 1;
    int max_dim = 1;
    bool vec4 = true;     // every row of the group in 16-byte chunks (dims, strides, addresses)
    while (c0 < n_cols && k < kMaxCols) {
      const hbk_lookup_grad_column_t& h = cols[c0++];
      if (h.n_ids <= 0) continue;
      HBK_REQUIRE(h.dim <= kDetLanes * kDetMaxE, "deterministic backward: dim %d > %d", h.dim,
                  kDetLanes * kDetMaxE);
      HBK_REQUIRE(h.rows < (1ll << 40), "deterministic backward: more than 2^40 rows");
      DCol& d = a.col[k];
      d.ids = h.ids;
      d.grad = h.grad_out;
      d.splits = h.row_splits;
      d.unique_rows = h.unique_rows;
      d.grad_rows = h.grad_rows;
      d.table = h.table;
      d.accum = h.accum;
      d.run_start = h.run_start;
      d.run_ids = h.run_ids;
      d.run_grads = h.run_grads;
      d.map = make_idmap(h.bucket, h.divisor, h.rows);
      d.n_ids = h.n_ids;
      d.n_seg = h.n_segments;
      d.dim = h.dim;
      d.grad_stride = h.grad_stride > 0 ? h.grad_stride : h.dim;
      d.tpitch = h.table_pitch > 0 ? h.table_pitch : h.dim;
      d.n_runs = h.n_runs;
      d.ids64 = h.ids_dtype == HBK_INT64;
      d.combiner = (uint8_t)h.combiner;
      a.n_unique[k] = h.n_unique;
      a.tile0[k] = (int32_t)tiles;
      a.base[k] = (int32_t)total;
      tiles += (h.n_ids + kDetTile - 1) / kDetTile;
      total += h.n_ids;
      if ((uint64_t)h.rows > max_rows) max_rows = (uint64_t)h.rows;
      if (h.dim > max_dim) max_dim = h.dim;
      uintptr_t bits = (uintptr_t)h.grad_out | (uintptr_t)h.grad_rows |
                       ((uintptr_t)(uint32_t)d.grad_stride * 4);
      if (lr != 0.0f) bits |= (uintptr_t)h.table | (uintptr_t)h.accum | ((uintptr_t)(uint32_t)h.table_pitch * 4);
      vec4 = vec4 && h.dim % 4 == 0 && bits % 16 == 0 && h.n_runs == 0;
      ++k;
    }
    if (k == 0) continue;
    HBK_REQUIRE(total < (1ll << 31), "deterministic backward: more than 2^31-1 ids in one launch group");
    for (int q = k; q <= kMaxCols; ++q) a.base[q] = (int32_t)total;
    a.n_cols = k;
    a.row_bits = bits_for(max_rows);        // the all-ones row field (> every row) marks "no row"
    a.apply = apply;
    a.lr = lr;
    a.total = total;
    const int end_bit = a.row_bits + (k > 1 ? bits_for((uint64_t)k - 1) : 0);
    // 1 keys in id order
    a.keys = keys;
    a.vals = vals;
    a.heads = heads;
    a.ranks = ranks;
    hipLaunchKernelGGL(det_keys_kernel, dim3((unsigned)tiles), dim3(kBlock), 0, stream, a);
    HBK_HIP_OK(hipGetLastError());
    // 2 stable sort by (column, row)
    int rc = det_sort_pairs(ws + l.temp, l.temp_bytes, keys, skeys, vals, svals, (size_t)total, end_bit,
                            stream);
    if (rc != HBK_OK) return rc;
    // 3 heads and their ranks
    a.keys = skeys;
    a.vals = svals;
    const unsigned tiles_p = (unsigned)((total + kDetTile - 1) / kDetTile);
    hipLaunchKernelGGL(det_heads_kernel, dim3(tiles_p), dim3(kBlock), 0, stream, a);
    HBK_HIP_OK(hipGetLastError());
    rc = det_scan(ws + l.temp, l.temp_bytes, heads, ranks, (size_t)total, stream);
    if (rc != HBK_OK) return rc;
    // 4 every row's run, front to back
    if (vec4) {
      det_launch_reduce<f32x4>(a, max_dim / 4, stream);
    } else {
      det_launch_reduce<float>(a, max_dim, stream);
    }
    HBK_HIP_OK(hipGetLastError());
    hipLaunchKernelGGL(det_counts_kernel, dim3(1), dim3(kWave), 0, stream, a);
    HBK_HIP_OK(hipGetLastError());
  }
  return HBK_OK;
}

}  // namespace
}  // namespace hbk

#ifdef HBK_BWD_STAMPS
extern "C" int hbk_debug_grp_trace(unsigned long long* out, int reset) {
  using namespace hbk;
  HBK_HIP_OK(hipDeviceSynchronize());
  if (out != nullptr) {
    HBK_HIP_OK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_grp_trace), sizeof(g_grp_trace)));
  }
  if (reset) {
    static unsigned long long z[kTraceBlocks * kTraceSlots];
    HBK_HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_grp_trace), z, sizeof(z)));
  }
  return HBK_OK;
}
// probe builds: constant-clock (100 MHz) stamps of the first 8192 reduce workgroups, 8 each
extern "C" int hbk_debug_bwd_sub(unsigned long long* out) {
  using namespace hbk;
  HBK_HIP_OK(hipDeviceSynchronize());
  HBK_HIP_OK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bwd_sub), sizeof(g_bwd_sub)));
  return HBK_OK;
}
extern "C" int hbk_debug_bwd_trace(unsigned long long* out, int reset) {
  using namespace hbk;
  HBK_HIP_OK(hipDeviceSynchronize());
  if (out != nullptr) {
    HBK_HIP_OK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bwd_trace), sizeof(g_bwd_trace)));
  }
  if (reset) {
    static unsigned long long z[kTraceBlocks * kTraceSlots];
    HBK_HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_trace), z, sizeof(z)));
  }
  return HBK_OK;
}
#endif

extern "C" size_t hbk_group_lookup_bwd_workspace_bytes(int32_t n_cols,
                                                       const hbk_lookup_grad_column_t* cols) {
  if (n_cols <= 0 || cols == nullptr) return 0;
  if (hbk::any_deterministic(n_cols, cols)) {
    std::vector<hbk_lookup_grad_column_t> fast, slow, plain;
    hbk::det_split(n_cols, cols, &fast, &slow, &plain);
    size_t total = hbk::det_layout((int32_t)slow.size(), slow.data()).total;
    if (total != 0) total += 256;
    size_t planned = 0;
    for (const hbk_lookup_grad_column_t& h : fast) planned += hbk::col_workspace(h, true);
    total += planned == 0 ? 0 : planned + 256;
    planned = 0;
    for (const hbk_lookup_grad_column_t& h : plain) planned += hbk::col_workspace(h);
    return total + (planned == 0 ? 0 : planned + 256);
  }
  size_t total = 0;
  for (int32_t c = 0; c < n_cols; ++c) total += hbk::col_workspace(cols[c]);
  return total == 0 ? 0 : total + 256;  // the head (counters, descriptors) is aligned inside
}

extern "C" int hbk_group_lookup_bwd(int32_t n_cols, const hbk_lookup_grad_column_t* cols,
                                    float apply_lr, void* workspace, size_t workspace_bytes,
                                    hbk_stream_t stream_) {
  return hbk_group_lookup_bwd_apply(n_cols, cols, HBK_APPLY_SGD, apply_lr, workspace,
                                    workspace_bytes, stream_);
}

namespace hbk {
namespace {
// Four streams and their fork / join events per device, kept for the life of the process.  The
// mutex is held while a call enqueues on them (host side only: microseconds).
constexpr int kHelperStreams = 4;
struct BwdHelpers {
  hipStream_t s[kHelperStreams];
  hipEvent_t fork, join[kHelperStreams];
  std::mutex mu;
};

// compute units of the current device (cached)
int device_cus() {
  static std::mutex mu;
  static std::map<int, int> cus;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  std::lock_guard<std::mutex> lock(mu);
  auto it = cus.find(dev);
  if (it != cus.end()) return it->second;
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
    n = 256;
  }
  cus[dev] = n;
  return n;
}

BwdHelpers* bwd_helpers(hipStream_t caller) {
  static std::mutex table_mu;
  static std::map<int, BwdHelpers*> table;
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(caller, &capturing);
  if (capturing != hipStreamCaptureStatusNone) return nullptr;   // a captured call stays on its stream
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  BwdHelpers* h = nullptr;
  {
    std::lock_guard<std::mutex> lock(table_mu);
    auto it = table.find(dev);
    if (it == table.end()) {
      h = new BwdHelpers();
      bool ok = true;
      for (int i = 0; i < kHelperStreams; ++i) {
        ok = ok && hipStreamCreateWithFlags(&h->s[i], hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&h->join[i], hipEventDisableTiming) == hipSuccess;
      }
      ok = ok && hipEventCreateWithFlags(&h->fork, hipEventDisableTiming) == hipSuccess;
      if (!ok) {
        delete h;
        h = nullptr;
      }
      table[dev] = h;
    } else {
      h = it->second;
    }
  }
  return h;
}
}  // namespace
}  // namespace hbk

static int bwd_planned(int32_t n_cols, const hbk_lookup_grad_column_t* cols, int32_t apply,
                       float apply_lr, void* workspace, hipStream_t stream, bool det);

extern "C" int hbk_group_lookup_bwd_apply(int32_t n_cols, const hbk_lookup_grad_column_t* cols,
                                          int32_t apply, float apply_lr, void* workspace,
                                          size_t workspace_bytes, hbk_stream_t stream_) {
  using namespace hbk;
  HBK_REQUIRE(apply == HBK_APPLY_SGD || apply == HBK_APPLY_ADAGRAD,
              "group_lookup_bwd: apply must be HBK_APPLY_SGD or HBK_APPLY_ADAGRAD, got %d", apply);
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
    HBK_REQUIRE(h.n_ids == 0 || (h.ids && h.grad_out), "group_lookup_bwd: column %d: NULL buffer", c);
    HBK_REQUIRE(h.n_ids == 0 || (h.unique_rows != nullptr) == (h.grad_rows != nullptr),
                "group_lookup_bwd: column %d: unique_rows and grad_rows go together", c);
    HBK_REQUIRE(h.n_ids == 0 || h.grad_rows != nullptr || apply_lr != 0.0f,
                "group_lookup_bwd: column %d: no output buffers and no optimizer step: nothing "
                "to do (unique_rows / grad_rows may only be NULL with apply_lr != 0)", c);
    HBK_REQUIRE(apply_lr == 0.0f || h.table != nullptr || h.n_ids == 0,
                "group_lookup_bwd: column %d: table is NULL but apply_lr != 0", c);
    HBK_REQUIRE(apply_lr == 0.0f || apply != HBK_APPLY_ADAGRAD || h.accum != nullptr ||
                    h.n_ids == 0,
                "group_lookup_bwd: column %d: Adagrad needs the accumulator table", c);
    HBK_REQUIRE(h.table_pitch == 0 || h.table_pitch >= h.dim,
                "group_lookup_bwd: column %d: table_pitch %d is smaller than dim %d", c, h.table_pitch,
                h.dim);
    HBK_REQUIRE((h.flags & ~HBK_GRAD_DETERMINISTIC) == 0, "group_lookup_bwd: column %d: unknown flags 0x%x", c,
                (unsigned)h.flags);
    HBK_REQUIRE(h.n_runs >= 0, "group_lookup_bwd: column %d: n_runs must be >= 0", c);
    HBK_REQUIRE(h.n_runs == 0 || (h.row_splits == nullptr && h.run_start && h.run_ids &&
                                  h.run_grads),
                "group_lookup_bwd: column %d: segmented inputs need run tables and no "
                "row_splits", c);
  }
  const size_t need = hbk_group_lookup_bwd_workspace_bytes(n_cols, cols);
  HBK_REQUIRE(need == 0 || (workspace != nullptr && workspace_bytes >= need),
              "group_lookup_bwd: workspace too small: need %zu bytes, got %zu", need,
              workspace_bytes);
  HBK_REQUIRE(((uintptr_t)workspace & 7) == 0,
              "group_lookup_bwd: workspace must be 8-byte aligned");
  {
    const int rc = sync_check("group_lookup_bwd", stream);
    if (rc != HBK_OK) return rc;
  }
  // option bwd_deterministic: the in-order forms -- row-sorted jobs where they fit (1), the sort + walk
  // of lookup_bwd_det.h for the other columns (and for all of them under 2)
  if (any_deterministic(n_cols, cols)) {
    std::vector<hbk_lookup_grad_column_t> fast, slow, plain;
    det_split(n_cols, cols, &fast, &slow, &plain);
    char* at = reinterpret_cast<char*>(workspace);
    if (!slow.empty()) {
      const int rc = det_backward((int32_t)slow.size(), slow.data(), apply, apply_lr, at, stream);
      if (rc != HBK_OK) return rc;
      const size_t slow_bytes = det_layout((int32_t)slow.size(), slow.data()).total;
      at += slow_bytes == 0 ? 0 : slow_bytes + 256;
    }
    if (!fast.empty()) {
      const int rc = bwd_planned((int32_t)fast.size(), fast.data(), apply, apply_lr, at, stream, true);
      if (rc != HBK_OK) return rc;
      size_t planned = 0;
      for (const hbk_lookup_grad_column_t& h : fast) planned += col_workspace(h, true);
      at += planned == 0 ? 0 : planned + 256;
    }
    if (plain.empty()) return HBK_OK;
    return bwd_planned((int32_t)plain.size(), plain.data(), apply, apply_lr, at, stream, false);
  }
  return bwd_planned(n_cols, cols, apply, apply_lr, workspace, stream, false);
}

// the bucket plans: grouping, reduce, merge (everything but the sort path of the deterministic mode)
static int bwd_planned(int32_t n_cols, const hbk_lookup_grad_column_t* cols, int32_t apply,
                       float apply_lr, void* workspace, hipStream_t stream, bool det) {
  using namespace hbk;
  // head of the workspace: the job descriptors of all columns (16-byte aligned), then the
  // per-column buffers
  char* cp = reinterpret_cast<char*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  char* dp = cp;
  for (int32_t c = 0; c < n_cols; ++c) {
    if (cols[c].n_ids > 0) dp += 256;       // one claim counter per column, one per line
  }
  char* wp = dp;
  for (int32_t c = 0; c < n_cols; ++c) {
    if (cols[c].n_ids <= 0) continue;
    const ColPlan p = plan_of(cols[c].n_ids, cols[c].dim, cols[c].rows, cols[c].row_splits != nullptr, det);
    wp += ((size_t)p.n_buckets + p.e_max) * sizeof(int4);
  }

  // Per column: plan, row shape, kind.  Everything that can fail is checked here, before any
  // helper stream is forked.
  struct ColInfo {
    ColPlan p;
    RowShape shape;
    int kind;       // 2 * (0 dense lean | 1 dense with the sorted walk | 2 hashed) + (16-byte chunks ?
                    // 0 : 1): columns of a launch group are sorted by it, so every kernel
                    // instantiation runs on ONE range of job slots
    bool onepass;   // small enough for the one-launch grouping (bwd_group_kernel)
  };
  std::vector<ColInfo> info((size_t)n_cols);
  std::vector<int32_t> order;   // live columns: the one-launch ones first, then the large ones
  for (int pass = 0; pass < 2; ++pass) {
    for (int32_t c = 0; c < n_cols; ++c) {
      const hbk_lookup_grad_column_t& h = cols[c];
      if (h.n_ids <= 0) continue;
      ColInfo& ci = info[(size_t)c];
      if (pass == 0) {
        ci.p = plan_of(h.n_ids, h.dim, h.rows, h.row_splits != nullptr, det);
        HBK_REQUIRE(h.grad_stride == 0 || (h.grad_stride >= h.dim && h.n_runs == 0),
                    "group_lookup_bwd: column %d: bad grad_stride %d", c, h.grad_stride);
        HBK_REQUIRE(make_rowshape(h.dim,
                                  (uintptr_t)h.grad_out | (uintptr_t)h.grad_rows |
                                      ((uintptr_t)(uint32_t)h.grad_stride * 4) |
                                      (apply_lr != 0.0f ? (uintptr_t)h.table | (uintptr_t)h.accum |
                                                              ((uintptr_t)(uint32_t)h.table_pitch * 4) : 0),
                                  &ci.shape),
                    "group_lookup_bwd: dim %d needs more than 64 lanes per row", h.dim);
        // bucket kind: 0 dense, 1 dense with sorted duplicates, 2 hashed, 3 hashed with the wide
        // sorted walk (an optimizer step, >= 16 lanes per row, one id per sample: bucket_reduce)
        const bool wide = apply_lr != 0.0f && ci.shape.lpr_log2 >= 4 &&
                          (options().bwd_wide == 2 ||
                           (options().bwd_wide == 1 && h.row_splits == nullptr));
        const int bkind = ci.p.rowsort ? 4 : ci.p.dense_mul != 0 ? (ci.p.dense_sort ? 1 : 0) : wide ? 3 : 2;
        ci.kind = 2 * bkind + (ci.shape.vec4 ? 0 : 1);
        ci.onepass = options().bwd_onepass != 0 && ci.p.n_buckets <= kGroupMaxBuckets &&
                     ci.p.tiles <= 64;
      }
      if ((pass == 0) == (ci.onepass != (options().bwd_large_first != 0))) order.push_back(c);
    }
  }
  for (int32_t c = 0; c < n_cols; ++c) {
    if (cols[c].n_ids == 0) HBK_HIP_OK(hipMemsetAsync(cols[c].n_unique, 0, sizeof(int32_t), stream));
  }

  // More than kMaxCols columns make several launch groups (config 5: 200 columns = 4).  They are
  // independent (own workspace slices), so they rotate over four streams of the library:
  // the tail of one group's reduce kernel -- a few long-lived workgroups on an idle chip -- runs
  // beside the next group's grouping launches instead of in front of them.  A group holds either
  // one-launch columns or large ones (a single large column would otherwise put its whole group
  // on the three grouping launches).
  const int32_t live_cols = (int32_t)order.size();
  int group_cols = options().bwd_group_cols;   // columns per launch group (tuning; 0: kMaxCols)
  if (group_cols <= 0 || group_cols > kMaxCols) group_cols = kMaxCols;
  bool mixed = false;
  for (int32_t q = 1; q < live_cols; ++q) {
    mixed = mixed || info[(size_t)order[q]].onepass != info[(size_t)order[0]].onepass;
  }
  const int n_streams = options().bwd_streams < 0 ? 0 : options().bwd_streams > kHelperStreams
                                                           ? kHelperStreams : options().bwd_streams;
  BwdHelpers* helpers = (live_cols > group_cols || mixed) && n_streams > 0 ? bwd_helpers(stream) : nullptr;
  std::unique_lock<std::mutex> hold;   // (released on every return path)
  if (helpers != nullptr) {
    hold = std::unique_lock<std::mutex>(helpers->mu);
    HBK_HIP_OK(hipEventRecord(helpers->fork, stream));
    for (int i = 0; i < kHelperStreams; ++i) {
      HBK_HIP_OK(hipStreamWaitEvent(helpers->s[i], helpers->fork, 0));
    }
  }
  // after the fork every path joins the helper streams before it returns: work queued on them
  // uses the caller's workspace and outputs
  int status = HBK_OK;
  int group_no = 0;
  int32_t q0 = 0;
  while (q0 < live_cols && status == HBK_OK) {
    hipStream_t ls = helpers != nullptr ? helpers->s[group_no++ % n_streams] : stream;
    GArgs args, seg_args;
    int4* const desc_group = reinterpret_cast<int4*>(dp);
    // the group's columns: up to group_cols of one grouping form, sorted by kind (stable)
    const bool group_onepass = info[(size_t)order[q0]].onepass;
    int32_t members[kMaxCols];
    int32_t k_n = 0;
    while (q0 < live_cols && k_n < group_cols && info[(size_t)order[q0]].onepass == group_onepass) {
      members[k_n++] = order[q0++];
    }
    for (int32_t i = 1; i < k_n; ++i) {   // insertion sort: <= 64 entries
      const int32_t m = members[i];
      int32_t j = i;
      while (j > 0 && info[(size_t)members[j - 1]].kind > info[(size_t)m].kind) {
        members[j] = members[j - 1];
        --j;
      }
      members[j] = m;
    }
    int32_t k = 0, ks = 0;
    int64_t tiles = 0, buckets = 0, segtiles = 0, merges = 0, scans = 0, sync_words = 0;
    size_t lds_hist = 0;
    bool small_scan = true;
    constexpr int kKinds = 10;
    int64_t slot_lo[kKinds] = {0}, slot_hi[kKinds] = {0};     // job slots of every kind
    int64_t merge_lo[kKinds] = {0}, merge_hi[kKinds] = {0};   // merge blocks of every kind
    bool have_kind[kKinds] = {false};
    struct KindCol { int32_t n_buckets, e_max, weight; };   // slots of a column: buckets, then extras
    std::vector<KindCol> kind_cols[kKinds];
    for (; k < k_n; ++k) {
      const int32_t col_index = members[k];
      const hbk_lookup_grad_column_t& h = cols[col_index];
      const ColInfo& ci = info[(size_t)col_index];
      const ColPlan& p = ci.p;
      GCol& d = args.col[k];
      memset(&d, 0, sizeof(d));
      d.ids = h.ids;
      d.grad_out = h.grad_out;
      d.splits = h.row_splits;
      d.unique_rows = h.unique_rows;
      d.grad_rows = h.grad_rows;
      d.n_unique = h.n_unique;
      d.counter = reinterpret_cast<int32_t*>(cp);
      cp += 256;
      d.table = h.table;
      d.accum = h.accum;
      d.hist = reinterpret_cast<int32_t*>(wp);
      wp += align8(((size_t)p.tiles * p.n_buckets) * 4);
      d.bstart = reinterpret_cast<int32_t*>(wp);
      wp += align8(((size_t)p.n_buckets + 1) * 4);
      d.pair_row[0] = reinterpret_cast<int64_t*>(wp);
      wp += (size_t)h.n_ids * 8;
      d.packed = pairs_packed(p) ? 1 : 0;
      d.pair_seg[0] = nullptr;
      if (!d.packed) {
        d.pair_seg[0] = reinterpret_cast<int32_t*>(wp);
        wp += align8((size_t)h.n_ids * 4);
      }
      // ragged columns: the grouping kernels find the segment of an id themselves (0a); option
      // bwd_seg_inline = 0 keeps the seg-of array and the launch that writes it
      d.seg_of = nullptr;
      if (h.row_splits != nullptr && options().bwd_seg_inline == 0) {
        d.seg_of = reinterpret_cast<int32_t*>(wp);
        wp += align8((size_t)h.n_ids * 4);
      }
      float* scaled = nullptr;
      if (h.row_splits != nullptr && h.combiner != HBK_COMBINER_SUM) {
        scaled = reinterpret_cast<float*>(((uintptr_t)wp + 15) & ~(uintptr_t)15);
        wp += align8((size_t)h.n_segments * h.dim * 4) + 16;
      }
      d.desc = reinterpret_cast<int4*>(dp);
      dp += ((size_t)p.n_buckets + p.e_max) * sizeof(int4);
      d.work = reinterpret_cast<int32_t*>(wp);
      wp += align8((size_t)p.e_max * 8);
      d.n_extra = reinterpret_cast<int32_t*>(wp);
      wp += 8;
      d.pcount = reinterpret_cast<int32_t*>(wp);
      wp += align8(((size_t)p.n_buckets) * 4);
      d.part_rows = reinterpret_cast<int64_t*>(wp);
      wp += (size_t)h.n_ids * 8;
      d.part_vals = reinterpret_cast<float*>(((uintptr_t)wp + 15) & ~(uintptr_t)15);
      wp += align8((size_t)h.n_ids * h.dim * 4) + 16;
      d.no_emit = 0;
      if (h.grad_rows == nullptr) {
        d.no_emit = 1;
        d.unique_rows = reinterpret_cast<int64_t*>(wp);
        wp += (size_t)h.n_ids * 8;
        d.grad_rows = reinterpret_cast<float*>(((uintptr_t)wp + 15) & ~(uintptr_t)15);
        wp += align8((size_t)h.n_ids * h.dim * 4) + 16;
      }
      d.split_t = p.split_t;
      d.e_max = p.e_max;
      d.dense_mul = p.dense_mul;
      d.rowsort = p.rowsort ? 1 : 0;
      d.det = det ? 1 : 0;
      d.merge0 = (int32_t)merges;
      const int64_t merge_blocks = p.e_max < kMergeBlocks ? p.e_max : kMergeBlocks;
      d.scan0 = (int32_t)scans;
      scans += ((int64_t)p.n_buckets + kBlock - 1) / kBlock;
      small_scan = small_scan && p.n_buckets <= 4 * kBlock && p.tiles <= 64;
      d.sync0 = (int32_t)sync_words;
      sync_words += (int64_t)p.tiles * p.n_buckets;
      d.run_start = h.run_start;
      d.run_ids = h.run_ids;
      d.run_grads = h.run_grads;
      d.n_runs = h.n_runs;
      d.grad_stride = h.grad_stride > 0 ? h.grad_stride : h.dim;
      d.map = make_idmap(h.bucket, h.divisor, h.rows);
      d.n_ids = h.n_ids;
      d.n_seg = h.n_segments;
      d.dim = h.dim;
      d.tpitch = h.table_pitch > 0 ? h.table_pitch : h.dim;
      d.chunks = ci.shape.chunks;
      d.lpr_log2 = ci.shape.lpr_log2;
      d.vec4 = ci.shape.vec4;
      d.ids64 = h.ids_dtype == HBK_INT64;
      d.combiner = (uint8_t)h.combiner;
      d.n_buckets = p.n_buckets;
      d.tile0 = (int32_t)tiles;
      d.bucket0 = (int32_t)buckets;
      d.segtile0 = 0;
      if (!have_kind[ci.kind]) {
        have_kind[ci.kind] = true;
        slot_lo[ci.kind] = buckets;
        merge_lo[ci.kind] = merges;
      }
      kind_cols[ci.kind].push_back({p.n_buckets, p.e_max, 24 + h.dim});
      tiles += p.tiles;
      buckets += (int64_t)p.n_buckets + p.e_max;
      merges += merge_blocks;
      slot_hi[ci.kind] = buckets;
      merge_hi[ci.kind] = merges;
      if (tiles >= (1ll << 31) || buckets >= (1ll << 31)) {
        status = fail(HBK_INVALID_ARGUMENT, "group_lookup_bwd: grid too large");
        break;
      }
      if ((size_t)4 * p.n_buckets > lds_hist) lds_hist = (size_t)4 * p.n_buckets;
      args.tile0[k] = d.tile0;
      args.bucket0[k] = d.bucket0;
      args.merge0[k] = d.merge0;
      args.scan0[k] = d.scan0;
      args.segtile0[k] = 0;
      // large ragged mean / sqrtn columns without a seg-of array: the histogram launch scales the
      // segments' rows (it knows the caller's gradient through raw_*)
      const bool scale_in_hist = scaled != nullptr && d.seg_of == nullptr && !group_onepass &&
                                 options().bwd_scale_fused != 0 && h.n_segments > 0;
      if (scale_in_hist) {
        d.scaled = scaled;
        d.raw_grad = h.grad_out;
        d.raw_stride = h.grad_stride > 0 ? h.grad_stride : h.dim;
        d.raw_combiner = h.combiner;
        d.grad_out = scaled;
        d.grad_stride = h.dim;
        d.combiner = HBK_COMBINER_SUM;
      }
      // the seg-of launch: columns with a seg-of array, or with rows to scale (mean / sqrtn)
      if (!scale_in_hist && h.row_splits != nullptr && h.n_segments > 0 &&
          (d.seg_of != nullptr || scaled != nullptr)) {
        GCol& sdesc = seg_args.col[ks];
        sdesc = d;
        sdesc.scaled = scaled;
        if (scaled != nullptr) {
          // the seg-of launch scales the segments' gradient rows once; everything behind it reads
          // the scaled rows as the gradient of a SUM column
          d.grad_out = scaled;
          d.grad_stride = h.dim;
          d.combiner = HBK_COMBINER_SUM;
        }
        sdesc.segtile0 = (int32_t)segtiles;
        seg_args.segtile0[ks] = (int32_t)segtiles;
        segtiles += (h.n_segments + kBlock - 1) / kBlock;
        ++ks;
      }
    }
    if (status != HBK_OK) break;
    if (k == 0) continue;
    if (options().bwd_trace != 0) {   // HBK_BWD_TRACE: the composition of every launch group on stderr
      int64_t g_ids = 0, g_bytes = 0;
      int by_kind[kKinds] = {0};
      for (int32_t i = 0; i < k; ++i) {
        g_ids += cols[members[i]].n_ids;
        g_bytes += (int64_t)cols[members[i]].n_ids * cols[members[i]].dim * 4;
        ++by_kind[info[(size_t)members[i]].kind];
      }
      fprintf(stderr, "[hbk bwd] group %d on stream %d: %d columns (%s), %lld ids, %.1f MB of gradient rows, "
              "%lld tiles, %lld job slots; columns by kind", group_no - (helpers != nullptr ? 1 : 0),
              helpers != nullptr ? (group_no - 1) % n_streams : -1, k, group_onepass ? "one-launch" : "large",
              (long long)g_ids, (double)g_bytes / 1e6, (long long)tiles, (long long)buckets);
      for (int q = 0; q < kKinds; ++q) {
        if (by_kind[q] != 0) fprintf(stderr, " %d:%d", q, by_kind[q]);
      }
      fprintf(stderr, "\n");
    }
    args.n_cols = k;
    args.lr = apply_lr;
    args.apply = apply;
    // Reduce jobs to XCDs (xcd_contiguous): whole columns per XCD -- when that leaves the XCDs
    // evenly loaded.  The eight ranges hold equal numbers of job slots, not equal work: a job of
    // a dim-128 column costs several jobs of a dim-4 one, and a launch of mixed columns then waits
    // for its heaviest XCD (config-5 shape: + 3 %, where equal columns gain 4 %).  The host knows
    // the columns: it adds up each range's work (jobs x (24 + dim); the extra slots of a column
    // are mostly unused) and keeps the round-robin placement when the heaviest range is > 8 %
    // above the mean.  Option bwd_xcd: 0 never, 1 this rule, 2 always.
    args.xcd = 0;
    for (int kind = 0; kind < kKinds; ++kind) {
      if (!have_kind[kind] || options().bwd_xcd == 0) continue;
      bool even = options().bwd_xcd == 2;
      if (!even) {
        const int64_t per = kind >= 4 && kind < 8 ? kTeams : 1;
        const int64_t n = (slot_hi[kind] - slot_lo[kind] + per - 1) / per;   // the launch's blocks
        const int64_t q = n / 8, r = n % 8;
        double work[8] = {0}, total = 0;
        int64_t slot = 0;      // walked once: ranges and columns both ascend
        int x = 0;
        int64_t x_end = (r > 0 ? q + 1 : q) * per;
        for (const KindCol& kc : kind_cols[kind]) {
          for (int part = 0; part < 2; ++part) {
            int64_t len = part == 0 ? kc.n_buckets : kc.e_max;
            const double w = part == 0 ? (double)kc.weight : 0.0;
            while (len > 0) {
              while (x < 7 && slot >= x_end) {
                ++x;
                x_end += (x < r ? q + 1 : q) * per;
              }
              const int64_t take = x < 7 && x_end - slot < len ? x_end - slot : len;
              work[x] += w * (double)take;
              total += w * (double)take;
              slot += take;
              len -= take;
            }
          }
        }
        double heaviest = 0;
        for (double w : work) heaviest = w > heaviest ? w : heaviest;
        even = total > 0 && heaviest <= 1.08 * total / 8;
      }
      if (even) args.xcd |= 1 << kind;
    }
    // The launches that pass get ranges of equal WORK rather than of equal slot counts (the extra
    // slots at every column's end are mostly empty): config 2 99.9 -> 98.9 us, + SGD 181 -> 176,
    // step only 144 -> 140.3, toggled in-process.  (bwd_xcd = 4, a probe: such ranges for every
    // launch -- the config-5 shape loses 5 % with them, so its loss above is not the imbalance
    // alone.)
    args.xcd_w = 0;
    args.stage_p = 0;
    int64_t xcd_grid[kKinds] = {0};
    if ((options().bwd_xcd == 4 || options().bwd_xcd == 1 || options().bwd_xcd == 3) && kTeams == 1) {
      for (int kind = 0; kind < kKinds; ++kind) {
        if (!have_kind[kind]) continue;
        if (options().bwd_xcd != 4 && !((args.xcd >> kind) & 1)) continue;
        double total = 0;
        for (const KindCol& kc : kind_cols[kind]) total += (double)kc.weight * kc.n_buckets;
        if (total <= 0) continue;
        int32_t* st = args.xcd_start[kind];
        st[0] = 0;
        int x = 1;
        double acc = 0;
        int64_t slot = 0;
        for (const KindCol& kc : kind_cols[kind]) {
          // the live slots of the column, one by one in blocks: boundary x sits where the work
          // before it first reaches x / 8 of the total
          for (int64_t b = 0; b < kc.n_buckets; ++b) {
            while (x < 8 && acc >= total * x / 8) st[x++] = (int32_t)(slot + b);
            acc += kc.weight;
          }
          slot += (int64_t)kc.n_buckets + kc.e_max;
        }
        while (x <= 8) st[x++] = (int32_t)slot;
        st[8] = (int32_t)(slot_hi[kind] - slot_lo[kind]);
        int64_t longest = 0;
        for (int q = 0; q < 8; ++q) longest = st[q + 1] - st[q] > longest ? st[q + 1] - st[q] : longest;
        xcd_grid[kind] = 8 * longest;
        args.xcd_w |= 1 << kind;
      }
    }
    // (equal tiles: always even; bwd_xcd = 3: the rule above without this -- probes)
    if (options().bwd_xcd == 1 || options().bwd_xcd == 2) args.xcd |= 1 << kXcdScatterBit;
    if (ks > 0) {
      seg_args.n_cols = ks;
      seg_args.lr = 0.f;
      hipLaunchKernelGGL(bwd_segof_kernel, dim3((unsigned)segtiles), dim3(kBlock), 0, ls,
                         seg_args);
    }
    // every column of the group: no segmented inputs, no seg-of array, packed pairs, int64 ids, row-range buckets
    bool simple_group = options().bwd_simple != 0;
    for (int32_t i = 0; i < k; ++i) {
      const GCol& g = args.col[i];
      simple_group = simple_group && g.n_runs == 0 && g.seg_of == nullptr && g.packed != 0 && g.ids64 != 0 &&
                     g.dense_mul != 0;
    }
    GSync sync;
    memset(&sync, 0, sizeof(sync));
    bool onepass = group_onepass && sync_words < (1ll << 30);
    if (onepass) {
      SyncTake take;
      onepass = sync_take(ls, (size_t)sync_words, &take,
                          reinterpret_cast<const void*>(simple_group ? &bwd_group_kernel<true> : &bwd_group_kernel<false>),
                          kBlock, 64, stream);
      if (onepass) {
        sync.hist = take.words;
        sync.zero = take.zero;
        sync.zero_words = take.zero_words;
        sync.wait = sync_wait_of(take);
      }
    }
    if (onepass) {
      // (hist, scan and scatter are this one launch)
      SyncChain chain(ls);   // never beside another kernel whose tiles wait for later tiles
      bool unpacked = false;
      for (int32_t i = 0; i < k; ++i) unpacked = unpacked || args.col[i].packed == 0;
      if (simple_group) {
        hipLaunchKernelGGL(bwd_group_kernel<true>, dim3((unsigned)tiles), dim3(kBlock),
                           (size_t)options().bwd_lds_pad * 1024, ls, args, sync);
      } else {
        hipLaunchKernelGGL(bwd_group_kernel<false>, dim3((unsigned)tiles), dim3(kBlock),
                           (unpacked ? (size_t)kTile * 4 : 0) + (size_t)options().bwd_lds_pad * 1024, ls,
                           args, sync);
      }
    } else {
      if (simple_group) {
        hipLaunchKernelGGL(bwd_hist_kernel<true>, dim3((unsigned)tiles), dim3(kBlock), lds_hist, ls, args);
      } else {
        hipLaunchKernelGGL(bwd_hist_kernel<false>, dim3((unsigned)tiles), dim3(kBlock), lds_hist, ls, args);
      }
      if (small_scan) {
        hipLaunchKernelGGL(bwd_scan_fused_kernel, dim3((unsigned)k), dim3(kBlock), 0, ls, args);
      } else {
        hipLaunchKernelGGL(bwd_scan_tiles_kernel, dim3((unsigned)scans), dim3(kBlock), 0, ls,
                           args);
        hipLaunchKernelGGL(bwd_scan_kernel, dim3((unsigned)k), dim3(kBlock), 0, ls, args);
      }
      if (lds_hist <= (size_t)4 * kStageMaxBuckets && options().bwd_scatter_staged != 0) {
        bool unpacked = false;
        for (int32_t i = 0; i < k; ++i) unpacked = unpacked || args.col[i].packed == 0;
        args.stage_p = (int32_t)(lds_hist / 4);
        if (simple_group) {
          hipLaunchKernelGGL(bwd_scatter_staged_kernel<true>, dim3((unsigned)tiles), dim3(kBlock),
                             2 * lds_hist + (size_t)options().bwd_lds_pad * 1024, ls, args);
        } else {
          hipLaunchKernelGGL(bwd_scatter_staged_kernel<false>, dim3((unsigned)tiles), dim3(kBlock),
                             2 * lds_hist + (unpacked ? (size_t)kTile * 4 : 0) +
                                 (size_t)options().bwd_lds_pad * 1024, ls, args);
        }
      } else {
        hipLaunchKernelGGL(bwd_scatter_pairs_kernel, dim3((unsigned)tiles), dim3(kBlock), lds_hist,
                           ls, args);
      }
    }
    // one instantiation per kind and optimizer (none / SGD / Adagrad), each on its own job slots
    const int step = apply_lr == 0.0f ? 0 : apply == HBK_APPLY_ADAGRAD ? 2 : 1;
    typedef void (*reduce_fn)(const GArgs, const int4*, int, int, const int32_t*);
    typedef void (*merge_fn)(const GArgs, int, const int32_t*);
    const int32_t* poison = onepass ? sync.wait.poison : nullptr;
    static const reduce_fn kReduce[kKinds][3] = {
        {&bwd_dense_kernel<f32x4, 0, false>, &bwd_dense_kernel<f32x4, 1, false>,
         &bwd_dense_kernel<f32x4, 2, false>},
        {&bwd_dense_kernel<float, 0, false>, &bwd_dense_kernel<float, 1, false>,
         &bwd_dense_kernel<float, 2, false>},
        {&bwd_dense_kernel<f32x4, 0, true>, &bwd_dense_kernel<f32x4, 1, true>,
         &bwd_dense_kernel<f32x4, 2, true>},
        {&bwd_dense_kernel<float, 0, true>, &bwd_dense_kernel<float, 1, true>,
         &bwd_dense_kernel<float, 2, true>},
        {&bwd_reduce_kernel<f32x4, 0, false>, &bwd_reduce_kernel<f32x4, 1, false>,
         &bwd_reduce_kernel<f32x4, 2, false>},
        {&bwd_reduce_kernel<float, 0, false>, &bwd_reduce_kernel<float, 1, false>,
         &bwd_reduce_kernel<float, 2, false>},
        {nullptr, &bwd_reduce_kernel<f32x4, 1, true>, &bwd_reduce_kernel<f32x4, 2, true>},
        {nullptr, &bwd_reduce_kernel<float, 1, true>, &bwd_reduce_kernel<float, 2, true>},
        {&bwd_rowsort_kernel<f32x4, 0>, &bwd_rowsort_kernel<f32x4, 1>, &bwd_rowsort_kernel<f32x4, 2>},
        {&bwd_rowsort_kernel<float, 0>, &bwd_rowsort_kernel<float, 1>, &bwd_rowsort_kernel<float, 2>}};
    static const merge_fn kMerge[kKinds][3] = {
        {&bwd_dense_merge_kernel<f32x4, 0>, &bwd_dense_merge_kernel<f32x4, 1>,
         &bwd_dense_merge_kernel<f32x4, 2>},
        {&bwd_dense_merge_kernel<float, 0>, &bwd_dense_merge_kernel<float, 1>,
         &bwd_dense_merge_kernel<float, 2>},
        {&bwd_dense_merge_kernel<f32x4, 0>, &bwd_dense_merge_kernel<f32x4, 1>,
         &bwd_dense_merge_kernel<f32x4, 2>},
        {&bwd_dense_merge_kernel<float, 0>, &bwd_dense_merge_kernel<float, 1>,
         &bwd_dense_merge_kernel<float, 2>},
        {&bwd_merge_kernel<f32x4, 0, false>, &bwd_merge_kernel<f32x4, 1, false>,
         &bwd_merge_kernel<f32x4, 2, false>},
        {&bwd_merge_kernel<float, 0, false>, &bwd_merge_kernel<float, 1, false>,
         &bwd_merge_kernel<float, 2, false>},
        {nullptr, &bwd_merge_kernel<f32x4, 1, true>, &bwd_merge_kernel<f32x4, 2, true>},
        {nullptr, &bwd_merge_kernel<float, 1, true>, &bwd_merge_kernel<float, 2, true>},
        {&bwd_rowsort_merge_kernel<f32x4, 0>, &bwd_rowsort_merge_kernel<f32x4, 1>,
         &bwd_rowsort_merge_kernel<f32x4, 2>},
        {&bwd_rowsort_merge_kernel<float, 0>, &bwd_rowsort_merge_kernel<float, 1>,
         &bwd_rowsort_merge_kernel<float, 2>}};
    static const reduce_fn kReduceDet[2][3] = {
        {&bwd_rowsort_kernel<f32x4, 0, true>, &bwd_rowsort_kernel<f32x4, 1, true>, &bwd_rowsort_kernel<f32x4, 2, true>},
        {&bwd_rowsort_kernel<float, 0, true>, &bwd_rowsort_kernel<float, 1, true>, &bwd_rowsort_kernel<float, 2, true>}};
    if (det) {   // where every bucket's rows begin in its column's output (lookup_bwd_rowsort.h)
      hipLaunchKernelGGL(bwd_rowsort_count_kernel, dim3((unsigned)buckets), dim3(kBlock), 0, ls, args, desc_group,
                         (int)buckets, poison);
    }
    for (int kind = 0; kind < kKinds; ++kind) {
      if (!have_kind[kind]) continue;
      const int64_t n_slots = slot_hi[kind] - slot_lo[kind];
      const int64_t per = kind >= 4 && kind < 8 ? kTeams : 1;   // job slots per workgroup
      const int64_t grid = xcd_grid[kind] > 0 ? xcd_grid[kind] : (n_slots + per - 1) / per;
      if (det) {
        if (kind < 8) {
          status = fail(HBK_INTERNAL, "group_lookup_bwd: a deterministic column without row-sorted buckets");
          break;
        }
        hipLaunchKernelGGL(kReduceDet[kind - 8][step], dim3((unsigned)grid), dim3(kBlock), 0, ls, args,
                           desc_group, (int)slot_lo[kind], (int)slot_hi[kind], poison);
        continue;
      }
      hipLaunchKernelGGL(kReduce[kind][step], dim3((unsigned)grid),
                         dim3(kBlock), 0, ls, args, desc_group, (int)slot_lo[kind],
                         (int)slot_hi[kind], poison);
    }
    for (int kind = 0; kind < kKinds; ++kind) {
      if (!have_kind[kind]) continue;
      hipLaunchKernelGGL(kMerge[kind][step], dim3((unsigned)(merge_hi[kind] - merge_lo[kind])),
                         dim3(kBlock), 0, ls, args, (int)merge_lo[kind], poison);
    }
    const hipError_t launch_err = hipGetLastError();
    if (launch_err != hipSuccess) {
      status = fail(HBK_INTERNAL, "group_lookup_bwd: launch failed: %s",
                    hipGetErrorString(launch_err));
    }
  }
  if (helpers != nullptr) {
    for (int i = 0; i < kHelperStreams; ++i) {
      const hipError_t e1 = hipEventRecord(helpers->join[i], helpers->s[i]);
      const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(stream, helpers->join[i], 0) : e1;
      if (e2 != hipSuccess && status == HBK_OK) {
        status = fail(HBK_INTERNAL, "group_lookup_bwd: joining the helper streams failed: %s",
                      hipGetErrorString(e2));
      }
    }
  }
  return status;
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
      HBK_REQUIRE(h.n_runs >= 0 && (h.n_runs == 0 || (h.run_start && h.run_base)),
                  "group_stitch_bwd: column %d: bad run tables", ci);
      d.run_start = h.run_start;
      d.run_base = h.run_base;
      d.n_runs = h.n_runs;
      HBK_REQUIRE(h.grad_stride == 0 || h.grad_stride >= h.dim,
                  "group_stitch_bwd: column %d: bad grad_stride %d", ci, h.grad_stride);
      d.grad_stride = h.grad_stride > 0 ? h.grad_stride : h.dim;
      d.n_seg = h.n_segments;
      d.dim = h.dim;
      RowShape shape;
      HBK_REQUIRE(make_rowshape(h.dim,
                                (uintptr_t)h.grad_out | (uintptr_t)h.grad_rows |
                                    ((uintptr_t)(uint32_t)h.grad_stride * 4),
                                &shape),
                  "group_stitch_bwd: dim %d needs more than 64 lanes per row", h.dim);
      d.chunks = shape.chunks;
      d.lpr_log2 = shape.lpr_log2;
      d.vec4 = shape.vec4;
      d.combiner = (uint8_t)h.combiner;
      d.pad_ = 0;
      const int64_t rpi = kWave >> d.lpr_log2;
      const int64_t per_block = kWavesPerBlock * kIters * rpi;
      d.tile0 = (int32_t)t_seg;
      args.tile0[k] = (int32_t)t_seg;
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
