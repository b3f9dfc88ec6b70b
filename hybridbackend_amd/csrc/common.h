// Internal helpers shared by the gfx950 kernels of libhbk_core.so.
#ifndef HBK_CSRC_COMMON_H_
#define HBK_CSRC_COMMON_H_

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "hbk.h"

namespace hbk {

// Thread-local message behind hbk_last_error(); returns `code` for `return fail(...)`.
int fail(int code, const char* fmt, ...);

#define HBK_HIP_OK(expr)                                                         \
  do {                                                                           \
    hipError_t e__ = (expr);                                                     \
    if (e__ != hipSuccess) {                                                     \
      return ::hbk::fail(HBK_INTERNAL, "%s failed: %s (%s:%d)", #expr,           \
                         hipGetErrorString(e__), __FILE__, __LINE__);            \
    }                                                                            \
  } while (0)

#define HBK_REQUIRE(cond, ...)                                                   \
  do {                                                                           \
    if (!(cond)) return ::hbk::fail(HBK_INVALID_ARGUMENT, __VA_ARGS__);          \
  } while (0)

inline hipStream_t as_stream(hbk_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int dtype_size(int32_t dtype) {
  switch (dtype) {
    case HBK_INT8:
    case HBK_UINT8:
      return 1;
    case HBK_HALF:
      return 2;
    case HBK_INT32:
    case HBK_UINT32:
    case HBK_FLOAT:
      return 4;
    case HBK_INT64:
    case HBK_UINT64:
    case HBK_DOUBLE:
      return 8;
    default:
      return 0;
  }
}

// ---------------------------------------------------------------------------------
// Division of a 64-bit value by a launch-invariant divisor without the ~100-instruction
// software divide: round-up magic multiply (q = (((n - t) >> 1) + t) >> shift with
// t = mulhi(magic, n)), exact for every 64-bit n.  kind 0: d == 1, kind 1: d = 2^shift.
struct FastDiv {
  uint64_t d;
  uint64_t magic;
  uint32_t shift;
  uint32_t kind;
};

inline FastDiv make_fastdiv(uint64_t d) {
  FastDiv f;
  f.d = d;
  f.magic = 0;
  f.shift = 0;
  f.kind = 0;
  if (d <= 1) return f;
  uint32_t k = 63u - (uint32_t)__builtin_clzll(d);  // floor(log2 d)
  if ((d & (d - 1)) == 0) {
    f.kind = 1;
    f.shift = k;
    return f;
  }
  // magic = floor(2^(65+k) / d) + 1 - 2^64
  unsigned __int128 num = (unsigned __int128)1 << (64 + k);
  uint64_t m = (uint64_t)(num / d);
  uint64_t rem = (uint64_t)(num % d);
  uint64_t m2 = m * 2;
  uint64_t twice_rem = rem * 2;
  if (twice_rem >= d || twice_rem < rem) m2 += 1;
  f.magic = m2 + 1;
  f.shift = k;
  f.kind = 2;
  return f;
}

__host__ __device__ inline uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

__host__ __device__ inline uint64_t fastdiv(uint64_t n, const FastDiv& f) {
  if (f.kind == 0) return n;
  if (f.kind == 1) return n >> f.shift;
  uint64_t t = mulhi64(f.magic, n);
  return (((n - t) >> 1) + t) >> f.shift;
}

__host__ __device__ inline uint64_t fastmod(uint64_t n, const FastDiv& f) {
  if (f.kind == 0) return 0;
  if (f.kind == 1) return n & (f.d - 1);
  return n - fastdiv(n, f) * f.d;
}

// Block b runs on XCD b % 8 (observed, not promised: MI355X_MICROARCH.md "Workgroup dispatch");
// with work item = xcd_contiguous(block) every XCD works through ONE contiguous range of the
// launch's items (whole columns) instead of every eighth: rows of one column that share a
// 128-byte line are then fetched through one L2.  Bijective for any n; a pure speed choice --
// any placement is correct, a different one just finds fewer lines in L2.
__host__ __device__ inline int xcd_contiguous(int b, int n, int on) {
  if (!on) return b;
  const int q = n >> 3, r = n & 7, x = b & 7, i = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}


// Python/TF floor-mod of a signed value by d > 0, result in [0, d).
// For v < 0: floormod(v, d) = d - 1 - ((-v - 1) mod d), and -v - 1 == ~v.
__host__ __device__ inline uint64_t floormod_i64(int64_t v, const FastDiv& f) {
  uint64_t u = v < 0 ? (uint64_t)(~v) : (uint64_t)v;
  uint64_t r = fastmod(u, f);
  return v < 0 ? f.d - 1 - r : r;
}

// Tuning / diagnostic options (hbk_set_option, include/hbk.h).  Read from the environment ONCE,
// when the library is loaded; the entry points read plain ints, never getenv.
struct Options {
  int bwd_buckets_log2 = -1;   // HBK_BWD_LOG2P: force 2^v buckets per column in the backward
  int bwd_bucket_pairs = 0;    // HBK_BWD_TARGET: aimed pairs per bucket (0: default)
  int bwd_split_pairs = 0;     // HBK_BWD_SPLIT: pairs per workgroup of a split bucket (0: default)
  int bwd_onepass = 1;         // HBK_BWD_ONEPASS: 0 = histogram, scan and scatter as three launches
  int fwd_xcd = 1;             // HBK_FWD_XCD: lookup tiles dealt to the XCDs in contiguous ranges (0 never: round robin,
                               // 1 the hot-row kernel, 2 always)
  int bwd_xcd = 1;             // HBK_BWD_XCD: reduce jobs / scatter tiles to the XCDs in contiguous ranges: 0 never (round robin),
                               // 1 by rule (even launches: ranges of equal work), 2 always equal slot ranges, 3 = 1 without the
                               // scatter tiles, 4 always equal work ranges (2-4: probes)
  int bwd_wide = 1;            // HBK_BWD_WIDE: wide sorted walk of the hashed backward (0 never,
                               // 1 columns of one id per sample, 2 ragged columns too)
  int bwd_group_cols = 0;      // HBK_BWD_GROUP_COLS: columns per launch group of the backward (0: 64)
  int bwd_dense = 1;           // HBK_BWD_DENSE: 0 = hashed buckets for every column (no row-range buckets), 1 = by policy,
                               // 2 = bitmap buckets (4b, sorted walk) wherever the row range fits, 3 = row-sorted buckets (4c) wherever it fits
  int bwd_scatter_staged = 1;  // HBK_BWD_SCATTER_STAGED: large columns: the scatter's pairs go through LDS sorted by bucket (0: direct)
  int bwd_rowsort_pos = 64;    // HBK_BWD_ROWSORT_POS: > 0: row-sorted jobs sized for this many sorted positions per lane group (wide rows: smaller jobs)
  int bwd_pairs_packed = 1;    // HBK_BWD_PAIRS_PACKED: row-sorted columns: a pair is ONE 8-byte word (row << 32 | gradient row) instead of an int64 + an int32 array (0: two arrays)
  int bwd_seg_inline = 1;      // HBK_BWD_SEG_INLINE: ragged columns: the segment of an id is found inside the grouping kernels (row splits of the tile in LDS); 0: a seg-of array written by a launch of its own
  int bwd_scale_fused = 1;     // HBK_BWD_SCALE_FUSED: large ragged mean / sqrtn columns: the segments' gradient rows are scaled by the histogram launch (0: by a launch of their own)
  int bwd_deterministic = 0;   // HBK_BWD_DETERMINISTIC: every row's gradient is summed in id order by ONE lane group: bit-identical from run
                               // to run and equal to the in-order fp32 sum, rows ascending.  1 = the row-sorted jobs' in-order form
                               // (lookup_bwd_rowsort.h: DET) for the columns whose row range fits them, the sort for the others;
                               // 2 = a stable sort of the batch's (row, gradient row) pairs + sequential walk for every column (lookup_bwd_det.h)
  int bwd_simple = 1;          // HBK_BWD_SIMPLE: launch groups whose columns are all plain (no segmented inputs, packed pairs, int64 ids, row-range buckets) take the grouping kernels' instantiation with those questions compiled out (0: the general one; A/B)
  int bwd_streams = 4;         // HBK_BWD_STREAMS: launch groups of a backward of > 64 (or mixed) columns rotate over this many library streams (0: all on the caller's stream)
  int bwd_lds_pad = 0;         // HBK_BWD_LDS_PAD: a probe: KB of unused LDS added to the grouping launches (fewer resident tiles)
  int bwd_trace = 0;           // HBK_BWD_TRACE: the composition of every launch group of a backward call on stderr
  int bwd_large_first = 0;     // HBK_BWD_LARGE_FIRST: the launch groups of the large columns (histogram / scan / scatter chain) are enqueued before the one-launch groups
  int bwd_rowsort_ratio = 8;   // HBK_BWD_ROWSORT_RATIO: row-sorted buckets for columns of rows <= ratio x ids (four times that for dim <= 32; 0: never)
  int fwd_d16 = 1;             // HBK_FWD_D16: one-id-per-segment columns of 16 floats with int64 ids take the gather's instantiation with those as constants (0: the general one; A/B)
  int fwd_interleave = 2;      // HBK_FWD_INTERLEAVE: lookup tiles of one dense output block ordered row tile first (lookup_fwd.hip)
  int fwd_hot_rows = 0;        // HBK_FWD_HOT: forward of wide one-id-per-sample columns: 1 = 256-segment tiles with
                               // repeated rows staged in LDS, 2 = the large tiles alone (probe), 0 = per-wave gather
  int unique_buckets_log2 = -1;  // HBK_UNIQUE_LOG2P
  int partition_sub_tiles = 1;   // HBK_PART_SUB
  int partition_fixed_max = 8;   // HBK_PART_FIXED
  int partition_onepass = 1;     // HBK_PART_ONEPASS: 0 = always the three-launch path
  int unique_onepass = 1;        // HBK_UNIQUE_ONEPASS: 0 = always the nine-launch path
  int sharded_groups = 0;        // HBK_SHARDED_GROUPS: column groups a sharded step pipelines (0: 2, or 1 on one rank)
  int sharded_id64 = 0;          // HBK_SHARDED_ID64: keep int64 ids on the wire
  int sharded_copy_self = 0;     // HBK_SHARDED_COPY_SELF: own slice through a device copy
  int sharded_trace = 0;         // HBK_SHARDED_TRACE: host-side phase times on stderr
  int sharded_pack_early = 1;    // HBK_SHARDED_PACK_EARLY: ids packed peer-major behind the partition, offsets from the device's sizes (0: after the host has them)
  int sharded_wire_fused = 1;    // HBK_SHARDED_WIRE_FUSED: fp16 wire: gather writes / stitch reads fp16 rows (0: two cast passes)
  int sharded_inline = 1;        // HBK_SHARDED_INLINE: exchanges enqueued on the compute stream (no event hops); the default since round 5: under a
                                 // modelled wire the column-group pipeline (0) lost to it in every case -- a cross-stream hop costs ~11 us, three per group, more than a group hides
                                 // (profiles/r05_overlap_model.txt)
  int sharded_p2p = 1;           // HBK_SHARDED_P2P: plans whose outputs were registered (hbk_sharded_p2p_bind) run the p2p form of the step (0: never)
  int sharded_p2p_test_refuse = -1;  // HBK_SHARDED_P2P_TEST_REFUSE: test hook, the rank whose hbk_sharded_p2p_bind finds "a peer's memory cannot be mapped"
                                     // (what a driver without hipIpc* support answers): every rank must then fall back to the exchange form
  int sync_wait_ms = 2000;       // HBK_SYNC_WAIT_MS: bound of a wait between the tiles of a one-launch kernel
  int sync_onepass_off = 0;      // HBK_SYNC_ONEPASS_OFF: 1 = multi-launch forms only (set by a wait that ran out)
  int sync_test_withhold = -1;   // HBK_SYNC_TEST_WITHHOLD: test hook, the tile that never publishes its counts
};
Options& options();

// Zeroed words for kernels whose tiles wait for each other (sync.hip).  sync_take: `words` zeroed
// int32 of the stream's buffer for this call; the call's FIRST kernel must clear zero[0,
// zero_words) (what the call before left set).  false while the stream is being captured into a
// graph (or without memory): the caller takes its multi-launch form instead.
struct SyncTake {
  int32_t* words;
  int32_t* zero;
  int64_t zero_words;
  int32_t* status;    // host-visible word of the OWNER stream (the caller's), raised by a wait that ran
                      // out: reported at that stream's next entry call, not at somebody else's
  int32_t* summary;   // host-visible, process-wide: "some status word is raised" (the entries' fast path)
  int32_t* poison;    // device word of THIS call (zero at its start, cleared with the call's words):
                      // set by a wait that ran out; the later kernels of the call read it first and
                      // leave without touching anything (their inputs were never written)
  unsigned long long wait_ticks;   // bound of a wait, 100 MHz ticks (option sync_wait_ms)
  int32_t withhold;   // test hook (option sync_test_withhold): the tile that never publishes, -1
};
// what the waiting kernels need of a SyncTake, by value in their arguments
struct SyncWait {
  int32_t* status;
  int32_t* summary;
  int32_t* poison;
  unsigned long long ticks;
  int32_t withhold;
  int32_t pad_;
};
inline SyncWait sync_wait_of(const SyncTake& t) {
  SyncWait w;
  w.status = t.status;
  w.summary = t.summary;
  w.poison = t.poison;
  w.ticks = t.wait_ticks;
  w.withhold = t.withhold;
  w.pad_ = 0;
  return w;
}
// `words` zeroed int32 (+ the call's poison word) of the stream's buffer.  `kernel` / `block` /
// `max_column_wgs`: the waiting kernel, its workgroup size and the most workgroups of ONE column
// that wait for each other: false when the DEVICE cannot hold them all at once (partitioned
// modes, fewer CUs; a per-STREAM CU mask is not visible to the occupancy query: the bounded wait is
// what protects such a stream,
// partitioned modes), as when the stream is being captured or a wait has run out before.
// `owner`: the stream whose caller is told of a wait that ran out (default: `stream` itself; the
// backward launches on helper streams of the library and the sharded plan on its prefetch stream
// on behalf of the caller's stream).
#define HBK_SYNC_SAME_STREAM (reinterpret_cast<hipStream_t>(~(uintptr_t)0))
bool sync_take(hipStream_t stream, size_t words, SyncTake* out, const void* kernel = nullptr,
               int block = 0, int max_column_wgs = 0, hipStream_t owner = HBK_SYNC_SAME_STREAM);
// Kernels whose tiles wait for tiles launched AFTER them (partition_onepass, unique_group,
// bwd_group) must not run beside each other: two of them on different streams can each fill the
// chip's workgroup slots with waiting tiles while the tiles they wait for find no slot -- a
// deadlock (seen: two launch groups of the config-5 backward on the library's helper streams; the
// bounded wait turned it into an error instead of a hang).  They are therefore chained
// device-wide: a launch waits (stream-side, no host block) for the event recorded behind the
// previous such launch on whatever stream that ran.  Hold the guard around the launch.
class SyncChain {
 public:
  explicit SyncChain(hipStream_t stream);
  ~SyncChain();
  SyncChain(const SyncChain&) = delete;
  SyncChain& operator=(const SyncChain&) = delete;

 private:
  hipStream_t stream_;
  void* state_;
};
// HBK_OK, or -- ONCE per timed-out wait -- HBK_INTERNAL with the story in hbk_last_error(); the
// one-launch forms are then off for the rest of the process (option sync_onepass_off).
// With a stream: only what was launched on behalf of THAT stream of the current device (a wait
// that ran out on stream A is reported to A's next entry call, never consumed by a call on stream
// B: round 5).  sync_check_any: whatever any stream of any device has raised
// (hbk_sync_check(), for callers that synchronise the whole device).
int sync_check(const char* who, hipStream_t stream);
int sync_check_any(const char* who);

constexpr int kWave = 64;  // gfx950 wavefront

// a wait of an earlier kernel of this call ran out: what it should have written is not there
__device__ inline bool poisoned(const int32_t* poison) {
  return poison != nullptr &&
         __hip_atomic_load(poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}
// a wait ran out: poison the call (device word), tell the host (pinned word)
__device__ inline void give_up(const SyncWait& w) {
  __hip_atomic_store(w.poison, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // (the owner's word first, the summary behind it: a host that has seen the summary finds the word)
  __hip_atomic_store(w.status, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(w.summary, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ inline int lane_id() { return (int)(threadIdx.x & (kWave - 1)); }

// number of set bits of `mask` strictly below this lane
__device__ inline int rank_below(unsigned long long mask) {
  return (int)__builtin_amdgcn_mbcnt_hi(
      (unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

}  // namespace hbk

#endif  // HBK_CSRC_COMMON_H_
