// ---- 4c: row-sorted buckets (round 4) -- included by lookup_bwd.hip, inside its namespace --------
// The reduce stage for columns whose batch is DENSE in the table (rows <= ~8 x ids: ragged
// columns, small and medium tables -- most of a 200-column model): a bucket is a contiguous ROW
// RANGE (as in 4b) but holds ~1800 pairs instead of ~450, and the job is built for throughput
// instead of for the common single-pair row:
//   A  every pair sets its row's bit in a bitmap over the range;
//   B  one scan of the bitmap's popcounts ranks the rows: rank = output position (the workgroup's
//      ONE global atomic claims the range) -- the rows leave sorted, their numbers straight from
//      the bitmap;
//   C  every pair takes a ticket on its row's counter (LDS atomic; a row that fills a wave is
//      counted once per wave);
//   D  one scan of the counters gives every row its run [start, start + count) in the sorted order;
//   E  the pairs' gradient rows (their numbers) go to their sorted positions, in LDS;
//   F  the WALK: every lane group takes an EQUAL share of the sorted positions -- whatever rows
//      they belong to -- and streams through it, W gradient rows in flight per lane, sums in
//      registers; a run that ends inside the share it began in leaves at once (with the optimizer
//      step, whose table / accumulator rows are requested for all rows that finished in the
//      batch).  No barrier and no LDS traffic on the way, no float atomics: the hashed and the
//      bitmap paths spend 2 barriers and 2 memory round trips per 48 rows here, this one a round
//      trip per W x groups (= 512 at dim 16) rows;
//   G  a run that crosses shares (a hot row: the Zipf head, a 100-row table) is the sum its first
//      group holds + the "heads" the following groups leave in LDS: one barrier, and the shares
//      stay equal however skewed the ids are.
// A job of several chunks (a bucket above kRsCap pairs: skewed ids, the ranges of a split bucket,
// the merge of their partial entries) ranks its rows over all chunks first (A, B), then runs C-G
// per chunk; a row that an earlier chunk emitted is added to (this workgroup owns it), and the
// optimizer step is taken once per row after the last chunk, from the finished sums.
// (kRsCap, pairs per chunk: lookup_bwd.hip)
// DET (round 6, option bwd_deterministic = 1): the same job with every row's terms added IN ID ORDER by
// one chain of additions -- E' orders the pairs of a run by gradient row, F gives a run to ONE lane
// group (shares snapped to run boundaries; runs of >= kRsDetLong pairs to the whole workgroup,
// rs_long_runs), a job of several chunks cuts them at tile shares of the bucket and continues a row
// from what the earlier chunks left, and the output range comes from bwd_rowsort_count_kernel's
// bucket counts instead of an atomic: bit-equal to the sequential fp32 sum, rows ascending.
constexpr int kRsPT = kRsCap / kBlock;           // pairs per thread and chunk
constexpr int kRsSpan = 16384;                   // rows of a bucket's range (bits of the bitmap)
constexpr int kRsWords = kRsSpan / 32;
constexpr int kRsWPT = kRsWords / kBlock;        // bitmap words per thread in the scan
constexpr int kRsBits = 13;                      // start / count fields of a row's counter word
constexpr uint32_t kRsMask = (1u << kRsBits) - 1u;
constexpr uint16_t kRsNoRow = 0xffff;
constexpr int kRsDetRank = 16;                // deterministic jobs: runs up to this many pairs are ordered by counting (48: the same times)
constexpr int kRsDetLong = 256;               // deterministic jobs: runs from this many pairs on are summed by the whole workgroup
constexpr uint16_t kRsLongBit = 0x8000;       //   (their positions carry this bit in su[]: the lane groups' walk steps over them)
static_assert(kRsCap < kRsLongBit, "row indices of a chunk leave bit 15 free");
static_assert(kRsCap % kBlock == 0 && kRsCap <= (1 << (kRsBits - 1)), "counter fields");
static_assert(kRsWords % kBlock == 0, "whole bitmap words per thread");
static_assert(kRsSpan <= 65536, "16-bit row offsets");

// A gradient row chunk.  Ragged columns with mean / sqrtn hand their rows over already scaled (the
// seg-of launch divides every segment's row once, lookup_bwd.hip): no combiner arithmetic here.
template <typename V, bool F16 = false>
__device__ inline V rs_load_grad(const ReduceJob& job, int32_t seg, int sub, bool live) {
  constexpr int VE = sizeof(V) / 4;
  const uint64_t off = !F16 && job.seg_is_offset ? (uint64_t)(uint32_t)seg
                                         : (uint64_t)(uint32_t)seg * (uint32_t)job.stride;
  V g = zero_v<V>();
  if (live) g = HBK_GRAD_LOAD(reinterpret_cast<const V*>(job.grad + off + (uint64_t)sub * VE));
  return g;
}

// Output rows of a one-chunk job are written once and not read again by this kernel: non-temporal,
// so that they do not push the gradient lines out of L2 -- a ragged column reads every segment's
// gradient row ~8 times, from different jobs (probe builds: -DHBK_RS_OUT_NT=0 plain stores).
#ifndef HBK_RS_OUT_NT
#define HBK_RS_OUT_NT 1
#endif
template <typename V, bool F16 = false>
__device__ inline void rs_store_row(const GCol& c, const ReduceJob& job, int32_t u, int sub, V v) {
  constexpr int VE = sizeof(V) / 4;
  V* o = reinterpret_cast<V*>(job.out_vals + (int64_t)u * (F16 ? 16 : c.dim) + (int64_t)sub * VE);
#if defined(HBK_RS_OUT_ASM)   // probe builds: the cache policy bits of the row store, spelled out
  if constexpr (sizeof(V) == 16) {
    asm volatile("global_store_dwordx4 %0, %1, off " HBK_RS_OUT_ASM : : "v"(o), "v"(v) : "memory");
  } else {
    asm volatile("global_store_dword %0, %1, off " HBK_RS_OUT_ASM : : "v"(o), "v"(v) : "memory");
  }
#elif HBK_RS_OUT_NT
  __builtin_nontemporal_store(v, o);
#else
  *o = v;
#endif
}

// Whole-line stores for 64-byte rows (round 5, tools/store_probe.hip): a non-temporal store of HALF a
// 128-byte line -- what a 4-lane group writes per instruction -- costs 1.33 write requests even when
// the other half follows in the very next instruction (lg_seq nt: 14.1 M requests for 10.6 M rows,
// 208 us; plain stores 10.6 M but they evict the gradient lines), while the two halves written by
// ADJACENT lanes of ONE instruction are one request each (pair_line nt: 10.6 M, 122 us).  Rows that
// finish at neighbouring positions of a lane group's share have consecutive ranks, so an even
// half and the odd half behind it leave together: the two lane groups of an 8-lane octet take
// turns, in turn e the octet stores the pair of its lane group e -- lanes of group e hold the
// first row, the other group's lanes fetch the second row over DPP (row_shl / row_shr by 4).
// (Only in the instantiation without an optimizer step.  Tried in the SGD / Adagrad ones as well, with a
// per-batch guard "some lane group finishes two rows at neighbouring positions": config 2 + SGD 171 -> 166 us on
// one box and 150 -> 152.6 on another, ragged 10 M rows + SGD -5 %, ragged over 100 k rows + SGD + 3 %, dim 4 / 64
// + 1-2 % -- the extra registers spill in kernels that are already at 128: not shipped.)
#ifndef HBK_RS_PAIR_STORES
#define HBK_RS_PAIR_STORES 1
#endif
#ifndef HBK_RS_PAIR_DIST
#define HBK_RS_PAIR_DIST 1   // positions between two rows of one lane group that still leave as one line; 2 and 3: probe builds (round 6, see below)
#endif
template <int CTRL>
__device__ inline int rs_dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
// (__float_as_int on the elements: __builtin_bit_cast(int, v.y) of an ext-vector ELEMENT reads element 0 with this
// compiler -- all four moves came out as one, found in the ISA)
template <int CTRL>
__device__ inline f32x4 rs_dpp_v(f32x4 v) {
  f32x4 r;
  r.x = __int_as_float(rs_dpp_i<CTRL>(__float_as_int(v.x)));
  r.y = __int_as_float(rs_dpp_i<CTRL>(__float_as_int(v.y)));
  r.z = __int_as_float(rs_dpp_i<CTRL>(__float_as_int(v.z)));
  r.w = __int_as_float(rs_dpp_i<CTRL>(__float_as_int(v.w)));
  return r;
}
// lane `src`'s value (any lane of the wave)
template <typename V>
__device__ inline V rs_shfl_v(V v, int src);
template <>
__device__ inline float rs_shfl_v<float>(float v, int src) { return __shfl(v, src, kWave); }
template <>
__device__ inline f32x4 rs_shfl_v<f32x4>(f32x4 v, int src) {
  return f32x4{__shfl(v.x, src, kWave), __shfl(v.y, src, kWave), __shfl(v.z, src, kWave),
               __shfl(v.w, src, kWave)};
}
constexpr int kDppRowShl4 = 0x104;   // lane i reads lane i + 4 of its row of 16
constexpr int kDppRowShr4 = 0x114;   // lane i reads lane i - 4

// (the row NUMBERS stay plain stores: 8-byte pieces per lane that the L2 combines into whole
// sectors; non-temporal they went out one by one -- ragged 1M-row case 650-667 -> 830-920 us)
#ifndef HBK_RS_ROWNUM_NT
#define HBK_RS_ROWNUM_NT 0
#endif
#ifndef HBK_RS_ROWS_FROM_ROFF
#define HBK_RS_ROWS_FROM_ROFF 1
#endif
#ifndef HBK_RS_W0
#define HBK_RS_W0 11   // (8: ragged dim 16 675 us, 12: 616 us; 16 spills.  Round 5: 12 spilled 2 VGPRs into the
                       // walk -- 11 does not: ragged 561-563 -> 523-539 us in-box, 10: 547-551; with the step
                       // W1 = 5 / W2 = 3 also stop their spills but measure the same as 6 / 4: kept)
#endif
#ifndef HBK_RS_W1
#define HBK_RS_W1 6
#endif
#ifndef HBK_RS_W2
#define HBK_RS_W2 4
#endif

// the finished rows of one batch of a lane group (bit w of `mine`: position w ends a run of mine, its
// sum is in g[w], its rank among the job's rows uu[w]); half0: parity of the job's first output row
// in its 128-byte line.  Every lane of the wave calls it.
template <int W, bool F16 = false>
__device__ inline void rs_store_pairs(const GCol& c, const ReduceJob& job, int32_t base_u, int half0,
                                      const uint32_t (&uu)[W], const f32x4 (&g)[W], uint32_t mine,
                                      int lane, int sub) {
  const bool odd_group = (lane >> 2) & 1;
  // (lane groups leave the walk when their share ends: a pair needs the octet's other group present)
  const bool partner_on = (__builtin_amdgcn_ballot_w64(true) >> (lane ^ 4)) & 1ull;
  uint32_t done = 0;
#pragma unroll
  for (int w = 0; w < W; ++w) {
    const int32_t R = base_u + (int32_t)uu[w];
    bool pair = false;
    if (w + 1 < W) {
      // the NEXT row this lane group finishes in the batch sits at the next set bit of `mine`; rows
      // finish in rank order, so it is row R + 1 -- the other half of R's line when R is the even one.
      // Round 5 paired only rows finishing at ADJACENT positions (the second row a single pair):
      // 45 % of the even rows of the ragged case; up to HBK_RS_PAIR_DIST positions apart covers runs of
      // up to that many pairs.
      const uint32_t ahead = ((mine & ~done) >> (w + 1)) & ((1u << HBK_RS_PAIR_DIST) - 1u);
      const int d = ahead != 0u ? __builtin_ctz(ahead) + 1 : 0;
      pair = partner_on && ((mine >> w) & 1u) != 0u && d != 0 && ((R + half0) & 1) == 0 &&
             ((done >> w) & 1u) == 0u;
      f32x4 nxt = g[w + 1];
#pragma unroll
      for (int k = 2; k <= HBK_RS_PAIR_DIST; ++k) {
        if (w + k < W && d == k) nxt = g[w + k];
      }
      // turn e: the octet stores the pair of its lane group e
      // (every DPP move outside any lane-dependent condition: a source lane that is masked off reads as 0)
      const int p_i = pair ? 1 : 0;
      const int p_shr = rs_dpp_i<kDppRowShr4>(p_i);
      const int p_shl = rs_dpp_i<kDppRowShl4>(p_i);
      const int p_from_even = odd_group ? p_shr : p_i;   // group 0's flag
      const int p_from_odd = odd_group ? p_i : p_shl;     // group 1's flag
      if (__builtin_amdgcn_ballot_w64(p_from_even != 0) != 0ull) {   // (wave-uniform branch)
        const f32x4 second = rs_dpp_v<kDppRowShr4>(nxt);          // group 0's next row in group 1's lanes
        const int32_t r_even = rs_dpp_i<kDppRowShr4>(R);
        if (p_from_even != 0) {
          rs_store_row<f32x4, F16>(c, job, odd_group ? r_even + 1 : R, sub, odd_group ? second : g[w]);
        }
      }
      if (__builtin_amdgcn_ballot_w64(p_from_odd != 0) != 0ull) {
        const f32x4 second = rs_dpp_v<kDppRowShl4>(nxt);          // group 1's next row in group 0's lanes
        const int32_t r_odd = rs_dpp_i<kDppRowShl4>(R);
        if (p_from_odd != 0) {
          rs_store_row<f32x4, F16>(c, job, odd_group ? R : r_odd + 1, sub, odd_group ? g[w] : second);
        }
      }
      if (pair) done |= (1u << w) | (1u << (w + d));
    }
    if (((mine & ~done) >> w) & 1u) rs_store_row<f32x4, F16>(c, job, R, sub, g[w]);
  }
}

struct RsLds {
  uint32_t present[kRsWords];   // rows of the job
  uint32_t pre[kRsWords];       // rows before word w
  uint32_t cmap[kRsWords];      // jobs of several chunks: rows of the chunk,
  uint32_t cpre[kRsWords];      //   rows of the chunk before word w,
  uint32_t seen[kRsWords];      //   rows an earlier chunk has emitted
  uint32_t cnt[kRsCap];         // per row of the chunk: tickets, then start | count << 13
  int32_t sseg[kRsCap];         // gradient row of every pair, sorted by row
  uint16_t su[kRsCap + 8];      // the pair's row (its index among the chunk's rows)
  uint16_t roff[kRsCap];        // row - first row of the range, per row of the chunk
  float red[kBlock * 4];        // per lane group: what it holds of a run that began in an earlier group's share
  int32_t wave_tot[kWavesPerBlock];
  int32_t n_sorted, base_u;
  int32_t det_end, det_tile, det_long;   // deterministic jobs: end of the chunk, its last tile + 1, "a run above kRsDetRank"
  int32_t n_long;                        //   runs of >= kRsDetLong pairs in the chunk, their rows
  uint16_t det_list[kRsCap / kRsDetLong + 2];
};

// Deterministic jobs, a chunk with a run above kRsDetRank pairs: one bitonic sort of the chunk's (row,
// gradient row) words, then the runs the whole workgroup will sum are listed and their positions
// marked.  Not inlined (rare; its code stays out of the job every bucket takes).
__device__ __attribute__((noinline)) void rs_det_sort_chunk(RsLds& L, int n_rows) {
  const int tid = (int)threadIdx.x;
  const int n_sorted = L.n_sorted;
  for (int i = n_sorted + tid; i < kRsCap; i += kBlock) {   // behind the pairs: the largest word
    L.su[i] = kRsNoRow;
    L.sseg[i] = 0x7fffffff;
  }
  __syncthreads();
  for (int len = 2; len <= kRsCap; len <<= 1) {
    for (int j = len >> 1; j > 0; j >>= 1) {
#pragma unroll
      for (int e = 0; e < kRsCap / 2 / kBlock; ++e) {
        const int t = e * kBlock + tid;              // compare-exchange number t of this step
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int x = i | j;
        const uint64_t a = ((uint64_t)L.su[i] << 32) | (uint32_t)L.sseg[i];
        const uint64_t b = ((uint64_t)L.su[x] << 32) | (uint32_t)L.sseg[x];
        const bool up = (i & len) == 0;
        if ((a > b) == up) {
          L.su[i] = (uint16_t)(b >> 32);
          L.sseg[i] = (int32_t)(uint32_t)b;
          L.su[x] = (uint16_t)(a >> 32);
          L.sseg[x] = (int32_t)(uint32_t)a;
        }
      }
      __syncthreads();
    }
  }
  // the runs the whole workgroup will sum (below): listed, their positions marked
  if (tid == 0) L.n_long = 0;
  __syncthreads();
  for (int u = tid; u < n_rows; u += kBlock) {
    if ((int)((L.cnt[u] >> kRsBits) & kRsMask) >= kRsDetLong) {
      L.det_list[atomicAdd(&L.n_long, 1)] = (uint16_t)u;   // (any order: the runs are summed one by one)
    }
  }
  for (int q = tid; q < n_sorted; q += kBlock) {
    const uint16_t u = L.su[q];
    if ((int)((L.cnt[u] >> kRsBits) & kRsMask) >= kRsDetLong) L.su[q] = u | kRsLongBit;
  }
}

// Deterministic jobs: the LONG runs of a chunk (rowsort_reduce, below), summed one after the other by
// the whole workgroup.  A function of its own, not inlined: its registers (WB rows in flight per lane)
// stay out of the budget of the walk every job takes.  (Its arguments by value, in registers: a
// reference to a column or a job would make the caller keep the kernel's whole argument block in
// scratch memory -- 24 KB per lane.)
struct RsLongArgs {
  const float* grad;
  float* out_vals;
  float* table;
  float* accum;
  int32_t stride, dim, tpitch, base_u;
  uint32_t base;
  float lr;
  int32_t sub, lpr_log2;
  bool seg_is_offset, one_chunk, emit, live;
};
template <typename V, int STEP>
__device__ __attribute__((noinline)) void rs_long_runs(RsLds& L, const RsLongArgs a) {
  constexpr int VE = sizeof(V) / 4;
  constexpr bool adagrad = STEP == 2;
  constexpr int WB = 8;   // terms a lane holds of a round (twice that in flight; 4: Zipf / one hot row at dim 128 1178 / 5795 us)
  const int tid = (int)threadIdx.x;
  const int lane = tid & (kWave - 1), wave = tid >> 6;
  const int lpr_log2 = a.lpr_log2, sub = a.sub;
  const bool stepping = STEP && a.lr != 0.0f;
  const int R = kWave >> lpr_log2;               // rows of one wave load
  const int grp = lane >> lpr_log2;
  const int T = WB * R;                          // terms of a wave per round
  V* carry = reinterpret_cast<V*>(L.red);        // [lanes of a row]
  const int n_long = L.n_long;
  for (int li = 0; li < n_long; ++li) {
    const uint32_t u = L.det_list[li];
    const uint32_t cv = L.cnt[u];
    const int start = (int)(cv & kRsMask), end = start + (int)((cv >> kRsBits) & kRsMask);
    // the row's place in the output; (several chunks) what the earlier chunks left there
    int32_t oi = a.base_u + (int32_t)u;
    bool first = true;
    if (!a.one_chunk) {
      const uint32_t off = L.roff[u];
      const int w = (int)(off >> 5);
      oi = a.base_u + (int32_t)L.pre[w] + __builtin_popcount(L.present[w] & ((1u << (off & 31u)) - 1u));
      first = ((L.seen[w] >> (off & 31u)) & 1u) == 0u;
    }
    V* o = reinterpret_cast<V*>(a.out_vals + (int64_t)oi * a.dim + (int64_t)sub * VE);
    __syncthreads();   // (the run before is done with carry)
    if (wave == 0 && grp == 0) {
      V a0 = zero_v<V>();
      if (a.live && !first) a0 = __builtin_nontemporal_load(o);
      carry[sub] = a0;
    }
    __syncthreads();
    // lane group j of a wave holds WB CONSECUTIVE terms: it adds them to the sum that reaches it without
    // a lane permute per term; the sum goes from lane group to lane group (one permute per WB terms)
    // and from wave to wave (LDS).  The next round's terms are requested before this round's turns.
    auto request = [&](V (&g)[WB], int p) {
#pragma unroll
      for (int b = 0; b < WB; ++b) {
        const int q = p + wave * T + grp * WB + b;
        const int32_t seg = L.sseg[q < end ? q : end - 1];
        const uint64_t off = a.seg_is_offset ? (uint64_t)(uint32_t)seg : (uint64_t)(uint32_t)seg * (uint32_t)a.stride;
        g[b] = zero_v<V>();
        if (a.live && q < end) g[b] = HBK_GRAD_LOAD(reinterpret_cast<const V*>(a.grad + off + (uint64_t)sub * VE));
      }
    };
    V g[WB], gn[WB];
    request(g, start);
    for (int p = start; p < end; p += kWavesPerBlock * T) {
      const int p_w = p + wave * T;
      if (p + kWavesPerBlock * T < end) {   // (uniform)
        request(gn, p + kWavesPerBlock * T);
      }
      for (int turn = 0; turn < kWavesPerBlock; ++turn) {
        if (wave == turn && p_w < end) {   // (wave-uniform)
          V acc = carry[sub];
          for (int j = 0; j < R && p_w + j * WB < end; ++j) {
            V mine = acc;
#pragma unroll
            for (int b = 0; b < WB; ++b) {
              if (p_w + j * WB + b < end) mine = mine + g[b];   // (right in lane group j, whose terms these are)
            }
            acc = rs_shfl_v<V>(mine, (j << lpr_log2) + sub);
          }
          if (grp == 0) carry[sub] = acc;
        }
        __syncthreads();
      }
#pragma unroll
      for (int b = 0; b < WB; ++b) g[b] = gn[b];
    }
    if (wave == 0 && grp == 0 && a.live) {
      const V acc = carry[sub];
      if (a.one_chunk && (a.emit || !stepping)) {   // (written once, not read again by this kernel: rs_store_row)
#if HBK_RS_OUT_NT
        __builtin_nontemporal_store(acc, o);
#else
        *o = acc;
#endif
      } else if (!a.one_chunk) {
        *o = acc;   // (the step of a row of several chunks is taken once, behind the last chunk)
      }
      if (a.one_chunk && stepping) {
        const int64_t toff = (int64_t)(a.base + L.roff[u]) * a.tpitch + (int64_t)sub * VE;
        const V tv = HBK_STEP_LOAD(reinterpret_cast<const V*>(a.table + toff));
        V av = zero_v<V>();
        if (adagrad) av = HBK_STEP_LOAD(reinterpret_cast<const V*>(a.accum + toff));
        // (step_row, lookup_bwd.hip)
        if (adagrad) {
          const V acc2 = av + acc * acc;
          *reinterpret_cast<V*>(a.accum + toff) = acc2;
          *reinterpret_cast<V*>(a.table + toff) = tv - (a.lr * acc) * rsqrt_v<V>(acc2);
        } else {
          *reinterpret_cast<V*>(a.table + toff) = tv - a.lr * acc;
        }
      }
    }
  }
}

template <typename V, int STEP, bool DET = false, int OC = -1, bool F16 = false>
__device__ inline void rowsort_reduce(const GCol& c, const ReduceJob& job, RsLds& L, int bucket) {
  constexpr int VE = sizeof(V) / 4;
  constexpr int PT = kRsPT;
  constexpr bool adagrad = STEP == 2;
  // gradient rows a lane keeps in flight (with the step, the table / accumulator rows of the rows
  // that finish in a batch travel together: register budget of 128)
  constexpr int W = STEP == 2 ? HBK_RS_W2 : STEP == 1 ? HBK_RS_W1 : HBK_RS_W0;
  const int tid = (int)threadIdx.x;
  const int lane = tid & (kWave - 1), wave = tid >> 6;
  // (F16: the caller has looked -- rows of 16 floats in four 16-byte chunks, packed pairs, gradient rows by number)
  const int lpr_log2 = F16 ? 2 : c.lpr_log2;
  const int sub = lane & ((1 << lpr_log2) - 1);
  const bool live = F16 ? true : sub < c.chunks;
  const int groups = kBlock >> lpr_log2;
  const int my_group = tid >> lpr_log2;
  // rows of 64 bytes (dim 16, four 16-byte chunks) leave as whole 128-byte lines where they can
  const bool pair_rows = HBK_RS_PAIR_STORES != 0 && sizeof(V) == 16 && (F16 || (lpr_log2 == 2 && c.chunks == 4 &&
                         c.dim == 16)) && ((uintptr_t)job.out_vals & 63u) == 0u;
  const int half0 = (int)(((uintptr_t)job.out_vals >> 6) & 1u);
  const int32_t n_pairs = job.n_pairs;
  if (n_pairs <= 0) return;
  const int64_t* prow = job.prow;
  const int32_t* pseg = job.pseg;
  const bool one_chunk = OC < 0 ? n_pairs <= kRsCap : OC != 0;   // (OC >= 0: the caller has looked)
  const float lr = STEP ? job.lr : 0.0f;
  const bool stepping = STEP && lr != 0.0f;
  const bool emit = !(STEP && job.no_emit);    // (no_emit only for jobs of one chunk: decode_job)
  // (no combiner arithmetic: a ragged mean / sqrtn column arrives as a SUM column over scaled rows)
  const uint32_t M = c.dense_mul;
  const uint32_t base = (uint32_t)dense_first_row(M, bucket);
  uint64_t lim = dense_first_row(M, bucket + 1);
  if (lim > c.map.rows) lim = c.map.rows;
  const int words = (int)((lim - base + 31) >> 5);   // <= kRsWords (host: plan_of)

  const bool packed = F16 ? true : job.packed;   // uniform: one word per pair, row << 32 | gradient row
  const bool pairs_nt = c.splits != nullptr && (packed || pseg != nullptr);   // ragged column, not a merge job
  uint32_t off_[PT];   // row - base of my pairs of the chunk, ~0u: none
  int32_t seg_[PT];
  auto load_pairs = [&](int32_t cb, int32_t ce) {   // the pairs [cb, ce) (<= kRsCap of them)
    // all loads first, in one straight line (indices behind the end are clamped and masked
    // afterwards): with the range check and the sign test inside the loop the compiler waited for
    // every pair before it requested the next -- 8 memory round trips, the job's first 8 us
    // Ragged columns read every gradient row ~8 times, from different jobs: their pairs -- read
    // once -- are loaded non-temporally so that they do not push gradient lines out of the XCD's
    // L2 (ragged 1M-row case 611-631 -> 599 us); columns of one id per segment have no reuse to
    // protect and lose 2 % with the hint (probe builds, profiles/r04_variants.txt): plain loads.
    int64_t r_[PT];
    if (pairs_nt) {   // uniform
#pragma unroll
      for (int k = 0; k < PT; ++k) {
        const int32_t e = cb + k * kBlock + tid;
        r_[k] = __builtin_nontemporal_load(prow + (e < ce ? e : ce - 1));
      }
    } else {
#pragma unroll
      for (int k = 0; k < PT; ++k) {
        const int32_t e = cb + k * kBlock + tid;
        r_[k] = HBK_PAIR_LOAD(prow + (e < ce ? e : ce - 1));
      }
    }
    if (packed) {
#pragma unroll
      for (int k = 0; k < PT; ++k) seg_[k] = (int32_t)(uint32_t)(uint64_t)r_[k];
    } else if (pseg != nullptr && pairs_nt) {   // uniform
#pragma unroll
      for (int k = 0; k < PT; ++k) {
        const int32_t e = cb + k * kBlock + tid;
        seg_[k] = __builtin_nontemporal_load(pseg + (e < ce ? e : ce - 1));
      }
    } else if (pseg != nullptr) {
#pragma unroll
      for (int k = 0; k < PT; ++k) {
        const int32_t e = cb + k * kBlock + tid;
        seg_[k] = HBK_PAIR_LOAD(pseg + (e < ce ? e : ce - 1));
      }
    } else {
#pragma unroll
      for (int k = 0; k < PT; ++k) seg_[k] = cb + k * kBlock + tid;
    }
#pragma unroll
    for (int k = 0; k < PT; ++k) {
      const int32_t e = cb + k * kBlock + tid;
      if (packed) {
        off_[k] = e < ce ? (uint32_t)((uint64_t)r_[k] >> 32) - base : ~0u;
      } else {
        off_[k] = e < ce && r_[k] >= 0 ? (uint32_t)r_[k] - base : ~0u;
      }
    }
  };
  // A: rows -> bitmap (a bit that is set is not set again: a hot row's pairs would serialise)
  auto mark = [&](uint32_t* bm) {
#pragma unroll
    for (int k = 0; k < PT; ++k) {
      if (off_[k] != ~0u) {
        const int w = (int)(off_[k] >> 5);
        const uint32_t bit = 1u << (off_[k] & 31u);
        if ((bm[w] & bit) == 0u) atomicOr(&bm[w], bit);
      }
    }
  };
  // B: rows before every word; returns the number of rows.  One barrier inside; the caller's
  // next barrier makes pre[] visible.
  auto scan_bitmap = [&](const uint32_t* bm, uint32_t* pre) -> int32_t {
    uint32_t cn[kRsWPT], sum = 0;
#pragma unroll
    for (int q = 0; q < kRsWPT; ++q) {
      const int w = tid * kRsWPT + q;
      cn[q] = w < words ? (uint32_t)__builtin_popcount(bm[w]) : 0u;
      sum += cn[q];
    }
    uint32_t incl = sum;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const uint32_t y = (uint32_t)__shfl_up((int)incl, o, kWave);
      if (lane >= o) incl += y;
    }
    if (lane == kWave - 1) L.wave_tot[wave] = (int32_t)incl;
    __syncthreads();
    uint32_t run = incl - sum, total = 0;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) {
      const uint32_t t = (uint32_t)L.wave_tot[w];
      if (w < wave) run += t;
      total += t;
    }
#pragma unroll
    for (int q = 0; q < kRsWPT; ++q) {
      const int w = tid * kRsWPT + q;
      if (w < words) pre[w] = run;
      run += cn[q];
    }
    return (int32_t)total;
  };

  load_pairs(0, n_pairs);       // they travel while the tables are cleared
  __syncthreads();     // a workgroup may run several jobs (merge): the previous one is done with L
  for (int w = tid; w < words; w += kBlock) {
    L.present[w] = 0u;
    L.seen[w] = 0u;
  }
  for (int i = tid; i < kRsCap; i += kBlock) L.cnt[i] = 0u;
  __syncthreads();
  HBK_STAMP(2);

  // A + B over the whole job: its rows, their ranks, the output range
  if (one_chunk) {
    mark(L.present);
  } else {
    for (int32_t cb = 0; cb < n_pairs; cb += kRsCap) {
      if (cb > 0) load_pairs(cb, n_pairs);
      mark(L.present);
    }
  }
  __syncthreads();
#ifdef HBK_RS_PROBE_ARRIVAL
  const int32_t n_rows_scan = scan_bitmap(L.present, L.pre);
  const int32_t n_rows_job = one_chunk ? n_pairs : n_rows_scan;
#else
  const int32_t n_rows_job = scan_bitmap(L.present, L.pre);
#endif
  // One global atomic per job claims the output range; a returning device-scope atomic takes
  // microseconds under load: its round trip runs beside C-E.  Step only: just the count is wanted.
  int32_t claimed = 0;
  if (DET) {
    // deterministic: the output ranges follow the buckets -- buckets are row ranges in row order and a
    // job's rows leave sorted, so the column's rows leave ASCENDING, at the same positions on every
    // run.  bwd_rowsort_count_kernel (below) has left every bucket's row count in pcount[]: the rows
    // in front of this bucket are the sum of the counts in front of it (the last wave adds them up
    // beside C; no atomic, nobody waits for another job).
    if (wave == kWavesPerBlock - 1) {
      int32_t before = 0;
      for (int p = lane; p < bucket; p += kWave) before += c.pcount[p];
#pragma unroll
      for (int o = kWave / 2; o > 0; o >>= 1) before += __shfl_xor(before, o, kWave);
      claimed = before;
    }
  } else if (tid == kBlock - 1) {
    if (!emit) {
      __hip_atomic_fetch_add(job.out_counter, n_rows_job, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      claimed = atomicAdd(job.out_counter, n_rows_job);
    }
  }
  int32_t base_u = 0;
  HBK_STAMP(3);

  // Deterministic jobs of several chunks: a chunk is a run of WHOLE tile shares of the bucket.  The
  // grouping stage lays the tiles' shares of a bucket one behind the other in the pair arrays (tile
  // order = id order; INSIDE a share the order comes from LDS tickets), so the gradient rows of such
  // a chunk all lie behind every earlier chunk's: sorting a row's pairs inside each chunk and taking
  // the chunks in turn visits the row's terms in id order.  hist[t][bucket] = where tile t's share
  // begins (left by the scan over the tiles / by the one-launch grouping).  A share is <= kTile <=
  // kRsCap pairs, so every chunk takes at least one.
  static_assert(kTile <= kRsCap, "a tile's share of a bucket fits one chunk");
  int det_tile = 0;   // the tile whose share begins the next chunk
  auto det_chunk_end = [&](int32_t cb) -> int32_t {
    const int P = c.n_buckets;
    const int n_tiles = (int)((c.n_ids + kTile - 1) / kTile);
    const int32_t* tp = c.hist + bucket;
    int32_t end = cb;
    for (;;) {
      const int tt = det_tile + 1 + tid;   // the chunk may end where this tile's share begins
      const int32_t v = tt < n_tiles ? tp[(int64_t)tt * P] : n_pairs;
      const unsigned long long fits = __ballot(v - cb <= kRsCap);   // (monotone over the threads)
      if (lane == 0) L.wave_tot[wave] = (int32_t)__builtin_popcountll(fits);
      __syncthreads();
      int n_fit = 0;
#pragma unroll
      for (int w = 0; w < kWavesPerBlock; ++w) n_fit += L.wave_tot[w];
      if (tid == n_fit - 1) L.det_end = v;
      __syncthreads();
      if (n_fit > 0) end = L.det_end;
      det_tile += n_fit;
      if (end > cb || det_tile >= n_tiles || n_fit == 0) break;   // (shares without a pair of this bucket: on)
    }
    if (end <= cb) end = cb + kRsCap < n_pairs ? cb + kRsCap : n_pairs;   // (unreachable: keeps the loop finite)
    return end;
  };

  int32_t ce = n_pairs;
  for (int32_t cb = 0; cb < n_pairs; cb = ce) {
    const uint32_t* bm = L.present;
    const uint32_t* pr = L.pre;
    int32_t n_rows = n_rows_job;
    if (!one_chunk) {
      if (DET) {
        ce = det_chunk_end(cb);
      } else {
        ce = cb + kRsCap < n_pairs ? cb + kRsCap : n_pairs;
      }
      __syncthreads();   // the chunk before is done with cnt / sseg / su / cmap (pre[] is visible)
      load_pairs(cb, ce);
      for (int w = tid; w < words; w += kBlock) L.cmap[w] = 0u;
      if (cb > 0) {
        for (int i = tid; i < kRsCap; i += kBlock) L.cnt[i] = 0u;
            }
      __syncthreads();
      mark(L.cmap);
      __syncthreads();
      n_rows = scan_bitmap(L.cmap, L.cpre);
      bm = L.cmap;
      pr = L.cpre;
    }
    __syncthreads();     // pre[] / cpre[] are in

    // C: the row of every pair (its index among the chunk's rows) and a ticket on its counter
    uint32_t u_[PT];
    int32_t tk_[PT];
#pragma unroll
    for (int k = 0; k < PT; ++k) {
      const bool valid = off_[k] != ~0u;
      uint32_t u = ~0u;
      if (valid) {
        const int w = (int)(off_[k] >> 5);
        u = pr[w] + (uint32_t)__builtin_popcount(bm[w] & ((1u << (off_[k] & 31u)) - 1u));
        if (HBK_RS_ROWS_FROM_ROFF || stepping || !one_chunk) {
          L.roff[u] = (uint16_t)off_[k];   // (every pair of the row: same value)
        }
      }
      int32_t tk = 0;
      bool done = !valid;
      // a row that holds much of the wave is counted once: same-address LDS atomics serialise
      const unsigned long long vm = __ballot(valid);
      if (vm != 0ull) {
        const int leader = __builtin_ctzll(vm);
        const uint32_t ul = (uint32_t)__builtin_amdgcn_readlane((int)u, leader);
        const unsigned long long same = __ballot(valid && u == ul);
        const int n_same = (int)__builtin_popcountll(same);
        if (n_same >= 8) {   // wave-uniform
          int32_t first = 0;
          if (lane == leader) first = (int32_t)atomicAdd(&L.cnt[ul], (uint32_t)n_same);
          first = __builtin_amdgcn_readlane(first, leader);
          if ((same >> lane) & 1ull) {
            tk = first + rank_below(same);
            done = true;
          }
        }
      }
      if (!done) tk = (int32_t)atomicAdd(&L.cnt[u], 1u);
      u_[k] = u;
      tk_[k] = tk;
    }
    __syncthreads();
    if (cb == 0) HBK_STAMP(4);

    // D: counters -> runs.  Thread t owns rows [t * PT, t * PT + PT).
    {
      uint32_t cn[PT], sum = 0;
#pragma unroll
      for (int k = 0; k < PT; ++k) {
        const int u = tid * PT + k;
        cn[k] = u < n_rows ? L.cnt[u] : 0u;
        sum += cn[k];
      }
      uint32_t incl = sum;
#pragma unroll
      for (int o = 1; o < kWave; o <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)incl, o, kWave);
        if (lane >= o) incl += y;
      }
      if (lane == kWave - 1) L.wave_tot[wave] = (int32_t)incl;
      __syncthreads();
      uint32_t run = incl - sum, total = 0;
#pragma unroll
      for (int w = 0; w < kWavesPerBlock; ++w) {
        const uint32_t t = (uint32_t)L.wave_tot[w];
        if (w < wave) run += t;
        total += t;
      }
#pragma unroll
      for (int k = 0; k < PT; ++k) {
        const int u = tid * PT + k;
        if (u < n_rows) L.cnt[u] = run | (cn[k] << kRsBits);
        run += cn[k];
      }
      if (tid == 0) {
        L.n_sorted = (int32_t)total;
        L.su[total] = kRsNoRow;   // behind the last run
      }
    }
    __syncthreads();

    // E: gradient rows to their sorted positions
#pragma unroll
    for (int k = 0; k < PT; ++k) {
      if (u_[k] != ~0u) {
#ifdef HBK_RS_PROBE_ARRIVAL   // probe builds (results wrong): every pair its own row, walked in ARRIVAL order --
                              // what the walk costs when the jobs of a column read the gradient block in step
        const int pos = k * kBlock + tid;
        L.sseg[pos] = seg_[k];
        L.su[pos] = (uint16_t)pos;
#else
        const int pos = (int)(L.cnt[u_[k]] & kRsMask) + tk_[k];
        L.sseg[pos] = seg_[k];
        L.su[pos] = (uint16_t)u_[k];
#endif
      }
    }
    if (cb == 0 && emit && tid == kBlock - 1) L.base_u = job.out_base + claimed;
    if (DET && tid == 0) L.det_long = 0;
    __syncthreads();
    if (cb == 0 && emit) base_u = L.base_u;
    if (cb == 0) HBK_STAMP(5);

    // E' (deterministic): the pairs of a row stand in ticket order -- put them in GRADIENT-ROW order,
    // which is id order (a segment's number grows with the id's position; two pairs of one row with
    // the same gradient row are the same term).  Short runs (nearly all): every pair counts the
    // pairs of its run that go in front of it; a chunk with a longer run: one bitonic sort of the
    // chunk's (row, gradient row) words.
    if (DET) {
      int32_t np_[PT];
      bool long_run = false;
#pragma unroll
      for (int k = 0; k < PT; ++k) {
        np_[k] = -1;
        if (u_[k] != ~0u) {
          const uint32_t cv = L.cnt[u_[k]];
          const int start = (int)(cv & kRsMask), n = (int)((cv >> kRsBits) & kRsMask);
          if (n > kRsDetRank) {
            long_run = true;
          } else if (n > 1) {   // (also runs of two: a + b == b + a, but a job of several chunks continues
                                // the row from what the earlier chunks left -- (e + a) + b != (e + b) + a;
                                // skipping them in one-chunk jobs measured no gain)
            const int pos = start + tk_[k];
            const int32_t mine = seg_[k];
            int before = 0;
            for (int q = start; q < start + n; ++q) {
              const int32_t other = L.sseg[q];
              before += ((uint32_t)other < (uint32_t)mine || (other == mine && q < pos)) ? 1 : 0;
            }
            np_[k] = start + before;
          }
        }
      }
      if (long_run) L.det_long = 1;   // (benign race: every writer stores 1)
      __syncthreads();                 // every count is taken
      if (L.det_long == 0) {           // uniform
#pragma unroll
        for (int k = 0; k < PT; ++k) {
          if (np_[k] >= 0) L.sseg[np_[k]] = seg_[k];
        }
      } else {
        rs_det_sort_chunk(L, n_rows);
      }
      __syncthreads();
    }
    const bool has_long = DET && L.det_long != 0 && L.n_long > 0;   // (uniform)
    // Deterministic, segmented inputs (the owner side of the sharded backward: ids and gradient rows
    // as runs inside larger buffers): the pairs carry the id's POSITION -- what E' has just ordered
    // them by, the gradient rows of two runs need not ascend -- and become float offsets of the
    // gradient rows here, for the walks (job.seg_is_offset).
    if (DET && c.n_runs > 0) {
      const int n_sorted = L.n_sorted;
      for (int q = tid; q < n_sorted; q += kBlock) {
        const int64_t j = (int64_t)(uint32_t)L.sseg[q];
        int k = 0;
        for (int r = 1; r < c.n_runs; ++r) k = j >= c.run_start[r] ? r : k;
        L.sseg[q] = (int32_t)(uint32_t)(c.run_grads[k] + (j - c.run_start[k]) * c.dim);
      }
      __syncthreads();
    }

    // a finished row leaves: (one chunk) straight to its output row, with the optimizer step;
    // (several chunks) into its output row, which an earlier chunk may have started
    auto out_index = [&](uint32_t u) -> int32_t {
      if (one_chunk) return base_u + (int32_t)u;
      const uint32_t off = L.roff[u];
      const int w = (int)(off >> 5);
      return base_u + (int32_t)L.pre[w] +
             __builtin_popcount(L.present[w] & ((1u << (off & 31u)) - 1u));
    };
    auto is_first = [&](uint32_t u) -> bool {
      if (one_chunk) return true;
      const uint32_t off = L.roff[u];
      return ((L.seen[off >> 5] >> (off & 31u)) & 1u) == 0u;
    };

    // F: the walk.  Lane group g takes the positions [g * per, (g + 1) * per) of the sorted order,
    // whatever rows they belong to: equal shares for every lane group, also when one row owns
    // most of the chunk.  A run that began in an earlier group's share is summed into `head` (left
    // in LDS for the group where the run began); runs that begin in my share are mine: the ones
    // that end there leave at once, the last one -- if it goes on -- waits in `acc` for the heads
    // of the groups it goes on in (G).
    if (DET && !one_chunk) {
      // Deterministic, several chunks: every run of the chunk front to back by ONE lane group, on top
      // of what the earlier chunks left in the row's output (the chunks visit a row's terms in id
      // order, see det_chunk_end): acc = ((earlier + t1) + t2) + ...  -- the sequential sum.  Jobs like
      // this are the hot buckets of skewed ids; the walk is short on loads in flight, not on exactness.
      constexpr int WD = 4;
      for (int u = my_group; u < n_rows; u += groups) {
        const uint32_t cv = L.cnt[u];
        const int start = (int)(cv & kRsMask), n = (int)((cv >> kRsBits) & kRsMask);
        if (has_long && n >= kRsDetLong) continue;   // (the whole workgroup's, below)
        const int32_t oi = out_index((uint32_t)u);
        V* o = reinterpret_cast<V*>(job.out_vals + (int64_t)oi * c.dim + (int64_t)sub * VE);
        V acc = zero_v<V>();
        if (live && !is_first((uint32_t)u)) acc = __builtin_nontemporal_load(o);
        for (int p = start; p < start + n; p += WD) {
          V g[WD];
#pragma unroll
          for (int w = 0; w < WD; ++w) {
            const int q = p + w < start + n ? p + w : start + n - 1;
            g[w] = rs_load_grad<V, F16>(job, L.sseg[q], sub, live);
          }
#pragma unroll
          for (int w = 0; w < WD; ++w) {
            if (p + w < start + n) acc = acc + g[w];
          }
        }
        if (live) *o = acc;
      }
    } else {
      const int n_sorted = L.n_sorted;
      const int per = (n_sorted + groups - 1) / groups;
      int lo = my_group * per < n_sorted ? my_group * per : n_sorted;
      int hi = lo + per < n_sorted ? lo + per : n_sorted;
      if (DET && lo < hi) {
        // Deterministic: a run is walked by ONE lane group, front to back -- the group in whose share
        // it begins takes all of it, the groups it runs through begin behind it.  (Shares are then
        // as unequal as the ids are skewed; no heads, no tails: in_head and tail below stay false.)
        if (lo > 0 && L.su[lo - 1] == L.su[lo]) {
          const uint32_t cv = L.cnt[L.su[lo] & ~kRsLongBit];
          lo = (int)(cv & kRsMask) + (int)((cv >> kRsBits) & kRsMask);
        }
        if (hi < n_sorted && L.su[hi - 1] == L.su[hi]) {
          const uint32_t cv = L.cnt[L.su[hi] & ~kRsLongBit];
          hi = (int)(cv & kRsMask) + (int)((cv >> kRsBits) & kRsMask);
        }
        if (lo > hi) lo = hi;
      }
      bool in_head = lo < hi && lo > 0 && L.su[lo - 1] == L.su[lo];
      V acc = zero_v<V>();
      for (int p = lo; p < hi; p += W) {
        if (has_long && (L.su[p] & kRsLongBit) != 0) {   // a run of the whole workgroup's (below): on behind it
          const uint32_t cv = L.cnt[L.su[p] & ~kRsLongBit];
          p = (int)(cv & kRsMask) + (int)((cv >> kRsBits) & kRsMask) - W;
          continue;
        }
        V g[W];
        int32_t sg[W];
        uint32_t uu[W];
        uint32_t val = 0, fin = 0;   // bit w: position p + w is mine / is the last of its run
        // (positions behind the share's end are clamped to its last one and masked out below: the
        // W loads leave in one straight line -- behind a branch per position the compiler waited
        // for every load before it issued the next)
#pragma unroll
        for (int w = 0; w < W; ++w) {
          const int q = p + w < hi ? p + w : hi - 1;
          uu[w] = L.su[q];
          sg[w] = L.sseg[q];
          if (p + w < hi && !(has_long && (uu[w] & kRsLongBit) != 0)) {
            val |= 1u << w;
            if (L.su[q + 1] != (uint16_t)uu[w]) fin |= 1u << w;
          }
        }
        // the runs that END here and are mine: all but the end of an earlier group's run
        uint32_t mine = fin;
        if (in_head && fin != 0u) mine &= fin - 1u;   // (its end is the lowest bit)
        // ALL loads of the batch in one go: W gradient rows and -- with the optimizer step -- the
        // table / accumulator rows of the rows that finish in it (their numbers are known before
        // anything arrives).  (Requested after the sums, the step's rows were a second memory round
        // trip per batch: config-5 shape + Adagrad, W = 4: two round trips per four positions.)
        V tv[STEP ? W : 1], av[STEP == 2 ? W : 1];
#pragma unroll
        for (int w = 0; w < W; ++w) g[w] = rs_load_grad<V, F16>(job, sg[w], sub, live);
        if (STEP && one_chunk && stepping) {
#pragma unroll
          for (int w = 0; w < W; ++w) {
            tv[STEP ? w : 0] = zero_v<V>();
            if (STEP == 2) av[STEP == 2 ? w : 0] = zero_v<V>();
            if ((mine >> w & 1u) && live) {
              const int64_t toff = (int64_t)(base + L.roff[uu[w]]) * c.tpitch + (int64_t)sub * VE;
              tv[STEP ? w : 0] = HBK_STEP_LOAD(reinterpret_cast<const V*>(c.table + toff));
              if (STEP == 2) {
                av[STEP == 2 ? w : 0] = HBK_STEP_LOAD(reinterpret_cast<const V*>(c.accum + toff));
              }
            }
          }
        }
        // the sum of a run of mine ends up in the registers its last gradient row arrived in
#pragma unroll
        for (int w = 0; w < W; ++w) {
          if (val >> w & 1u) {
            acc = acc + g[w];
            if (fin >> w & 1u) {
              if (mine >> w & 1u) {
                g[w] = acc;
              } else {           // (uniform in the lane group) the end of an earlier group's run
                *reinterpret_cast<V*>(&L.red[(size_t)tid * VE]) = acc;
                in_head = false;
              }
              acc = zero_v<V>();
            }
          }
        }
        if constexpr (HBK_RS_PAIR_STORES && sizeof(V) == 16 && STEP == 0) {
          // (the paired stores are cooperative: a lane group with nothing to emit lends its lanes)
          if (pair_rows && one_chunk && !stepping) {
            rs_store_pairs<W, F16>(c, job, base_u, half0, uu, g, mine, lane, sub);
            continue;
          }
        }
        if (mine == 0u || !live) continue;
        if (one_chunk && stepping) {
#pragma unroll
          for (int w = 0; w < W; ++w) {
            if (mine >> w & 1u) {
              if (emit) rs_store_row<V, F16>(c, job, base_u + (int32_t)uu[w], sub, g[w]);
              const int64_t toff = (int64_t)(base + L.roff[uu[w]]) * c.tpitch + (int64_t)sub * VE;
              step_row<V>(c, adagrad, lr, toff, g[w], tv[STEP ? w : 0],
                          STEP == 2 ? av[STEP == 2 ? w : 0] : zero_v<V>());
            }
          }
        } else if (one_chunk) {
#pragma unroll
          for (int w = 0; w < W; ++w) {
            if (mine >> w & 1u) rs_store_row<V, F16>(c, job, base_u + (int32_t)uu[w], sub, g[w]);
          }
        } else {
#pragma unroll
          for (int w = 0; w < W; ++w) {
            if (mine >> w & 1u) {
              emit_row<V>(c, job, out_index(uu[w]), is_first(uu[w]), sub, g[w]);
            }
          }
        }
      }
      // my whole share inside ONE run that began earlier: all of it is that run's head
      if (in_head) {
        *reinterpret_cast<V*>(&L.red[(size_t)tid * VE]) = acc;
        acc = zero_v<V>();
      }
      // G: the run that goes on behind my share (it began in it): + the heads of the groups it
      // goes on in.  (A run that owns 200 groups' shares is 200 LDS reads for its owner.)
      const bool tail = lo < hi && !in_head && hi < n_sorted && L.su[hi - 1] == L.su[hi];
      __syncthreads();   // (every lane group gets here: the heads are in)
      if (tail && live) {
        const uint32_t u = L.su[hi - 1];
        const uint32_t cv = L.cnt[u];
        const int end = (int)(cv & kRsMask) + (int)((cv >> kRsBits) & kRsMask);
        const int g_last = (end - 1) / per;
        for (int gp = my_group + 1; gp <= g_last; ++gp) {
          acc = acc + *reinterpret_cast<const V*>(&L.red[(((size_t)gp << lpr_log2) + sub) * VE]);
        }
        if (one_chunk && stepping) {
          if (emit) rs_store_row<V, F16>(c, job, base_u + (int32_t)u, sub, acc);
          const int64_t toff = (int64_t)(base + L.roff[u]) * c.tpitch + (int64_t)sub * VE;
          const V tv = HBK_STEP_LOAD(reinterpret_cast<const V*>(c.table + toff));
          V av = zero_v<V>();
          if (adagrad) av = HBK_STEP_LOAD(reinterpret_cast<const V*>(c.accum + toff));
          step_row<V>(c, adagrad, lr, toff, acc, tv, av);
        } else if (one_chunk) {
          rs_store_row<V, F16>(c, job, base_u + (int32_t)u, sub, acc);
        } else {
          emit_row<V>(c, job, out_index(u), is_first(u), sub, acc);
        }
      }
    }

    // Deterministic: the LONG runs of the chunk (a hot row: the Zipf head, a table of a few rows), one
    // after the other, by the whole workgroup.  A sequential sum has one chain of additions, but its
    // terms can arrive from everywhere: every wave requests WB x (rows per wave load) terms of the
    // run per round -- all lanes loading, as in the equal shares of the default walk -- and the waves
    // take turns adding theirs, in order, to the running sum they hand on through LDS (a term
    // reaches every lane of its wave's rows by a lane permute; the lane groups of a wave all hold the
    // same sum).  acc = ((carry + t_i) + t_i+1) + ...: the same chain as one lane group walking the
    // run, at a workgroup's loads in flight.
    if (has_long) {
      RsLongArgs la;
      la.grad = job.grad;
      la.out_vals = job.out_vals;
      la.table = c.table;
      la.accum = c.accum;
      la.stride = job.stride;
      la.dim = c.dim;
      la.tpitch = c.tpitch;
      la.base_u = base_u;
      la.base = base;
      la.lr = lr;
      la.sub = sub;
      la.lpr_log2 = lpr_log2;
      la.seg_is_offset = job.seg_is_offset;
      la.one_chunk = one_chunk;
      la.emit = emit;
      la.live = live;
      rs_long_runs<V, STEP>(L, la);
    }

    if (cb == 0) HBK_STAMP(6);
    if (!one_chunk) {
      __syncthreads();   // every row of the chunk has left: they count as seen from here on
      for (int w = tid; w < words; w += kBlock) L.seen[w] |= L.cmap[w];
    }
  }

  // the row numbers, sorted.  One chunk: the chunk's row index IS the output rank, and roff[] holds
  // every row's offset: consecutive lanes write consecutive entries (whole lines per store
  // instruction).  Several chunks: straight from the job's bitmap.
#ifdef HBK_RS_NO_ROWNUM   // probe builds: what the row-number stores cost (results are then wrong)
  if (false) {
#else
  if (emit && one_chunk && HBK_RS_ROWS_FROM_ROFF) {
#endif
    for (int u = tid; u < n_rows_job; u += kBlock) {
      job.out_rows[base_u + u] = (int64_t)base + (int64_t)L.roff[u];
    }
  } else if (emit) {
    for (int w = tid; w < words; w += kBlock) {
      uint32_t m = L.present[w];
      int64_t* o = job.out_rows + base_u + (int32_t)L.pre[w];
      const int64_t r0 = (int64_t)base + 32 * (int64_t)w;
      while (m != 0u) {
#if HBK_RS_ROWNUM_NT
        __builtin_nontemporal_store(r0 + __builtin_ctz(m), o++);   // (written once, like the rows)
#else
        *o++ = r0 + __builtin_ctz(m);
#endif
        m &= m - 1u;
      }
    }
  }

  // several chunks: ONE optimizer step per row, from its finished sum (stepping chunk by chunk
  // would round differently from table -= lr * grad_row and is wrong for Adagrad).  Loads of a
  // round first, then the stores.
  if (!one_chunk && stepping) {
    __syncthreads();   // the rows and their numbers are written (this workgroup's own stores)
    constexpr int kAp = STEP == 2 ? 2 : 4;
    for (int i0 = 0; i0 < n_rows_job; i0 += kAp * groups) {
      int64_t toff[kAp];
      V g[kAp], tv[kAp], av[kAp];
#pragma unroll
      for (int k = 0; k < kAp; ++k) {
        const int i = i0 + k * groups + my_group;
        toff[k] = -1;
        g[k] = tv[k] = av[k] = zero_v<V>();
        if (i < n_rows_job && live) {
          const int64_t row = __builtin_nontemporal_load(job.out_rows + base_u + i);
          toff[k] = row * c.tpitch + (int64_t)sub * VE;
          g[k] = __builtin_nontemporal_load(reinterpret_cast<const V*>(
              job.out_vals + (int64_t)(base_u + i) * c.dim + (int64_t)sub * VE));
          tv[k] = HBK_STEP_LOAD(reinterpret_cast<const V*>(c.table + toff[k]));
          if (adagrad) av[k] = HBK_STEP_LOAD(reinterpret_cast<const V*>(c.accum + toff[k]));
        }
      }
#pragma unroll
      for (int k = 0; k < kAp; ++k) {
        if (toff[k] >= 0) step_row<V>(c, adagrad, lr, toff[k], g[k], tv[k], av[k]);
      }
    }
  }
}

#ifndef HBK_RS_LB2
#define HBK_RS_LB2 4   // workgroups per CU the Adagrad instantiation is compiled for
#endif
template <typename V, int STEP, bool DET = false>
__global__ __launch_bounds__(kBlock, STEP == 2 ? HBK_RS_LB2 : 4) void bwd_rowsort_kernel(const GArgs a, const int4* desc,
                                                               int slot0, int total,
                                                               const int32_t* poison) {
  __shared__ RsLds lds;
  if (poisoned(poison)) return;   // the grouping launch gave up: no descriptors, no pairs
  HBK_STAMP_BEGIN()
  const int lane = (int)threadIdx.x & (kWave - 1);
  constexpr int kKind = 8 + (sizeof(V) == 4 ? 1 : 0);   // (the host's ColInfo.kind)
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
  if (!decode_job<V, 2>(a, my_b0, vb, d, a.lr, &ci, &job)) return;
  HBK_STAMP(1);
  // Two copies of the job (round 6): buckets of ONE chunk -- nearly all -- run a copy in which everything
  // that serves jobs of several chunks (per-chunk bitmaps, rows seen earlier, read-modify-write
  // emission, the deferred optimizer step) is compiled out; the others run the general copy.  Same
  // source, `one_chunk` a constant in each: ragged 495 -> 478 us, config 2 emit / + SGD / step only
  // 86.5 / 154 / 128.5 -> 83.9 / 151.5 / 126.5, the deterministic forms 106 / 175 / 149 / 557 -> 99 /
  // 164 / 141.5 / 526 (probe build alternating with the one-copy build in one visit,
  // profiles/r06_split_one_chunk.txt).
  if (job.n_pairs <= kRsCap) {
    // ... and of those, without an optimizer step, a third copy for rows of 16 floats (four 16-byte chunks, packed
    // pairs, gradient rows by number: config 2, the ragged benchmark columns) with lanes per row, row size and pair
    // layout as constants: config 2 emit 84.2 -> 80.6 us, ragged 479 - 509 -> 464 - 502, deterministic emit 99.8 ->
    // 96.5 (with the step: +- 1 %, not instantiated; profiles/r06_split_one_chunk.txt)
    if (STEP == 0 && sizeof(V) == 16 && a.col[ci].lpr_log2 == 2 && a.col[ci].chunks == 4 && a.col[ci].dim == 16 &&
        job.packed && !job.seg_is_offset) {
      rowsort_reduce<V, STEP, DET, 1, STEP == 0 && sizeof(V) == 16>(a.col[ci], job, lds, d.z);
    } else {
      rowsort_reduce<V, STEP, DET, 1>(a.col[ci], job, lds, d.z);
    }
  } else {
    rowsort_reduce<V, STEP, DET, 0>(a.col[ci], job, lds, d.z);
  }
  HBK_STAMP(7);
}

// Deterministic mode (option bwd_deterministic = 1), in front of the reduce launch: the DISTINCT ROWS
// of every bucket -- one workgroup per bucket marks its pairs' rows in an LDS bitmap and counts the
// bits.  The reduce jobs then take their output ranges from the counts in front of them, without an
// atomic and without waiting for each other, and the merge launch adds up the column's row count.
// (Tried first: a chained scan with look-back inside the reduce launch -- the jobs' polling of each
// other's status words cost the ragged case 85 us and forbade dealing the jobs to the XCDs, another
// 50 us; then this kernel with the prefix taken by the column's last workgroup behind an acq_rel
// atomic at agent scope -- every workgroup's release wrote the L2 back: 41 / 249 us for config 2 / the
// ragged case instead of the few microseconds the counting takes.)
__global__ __launch_bounds__(kBlock) void bwd_rowsort_count_kernel(const GArgs a, const int4* desc, int total,
                                                                   const int32_t* poison) {
  __shared__ uint32_t bm[kRsWords];
  __shared__ int32_t wave_tot[kWavesPerBlock];
  if (poisoned(poison)) return;
  const int tid = (int)threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  const int vb = (int)blockIdx.x;
  if (vb >= total) return;
  const int4 d = desc[vb];
  if (d.z < 0) return;   // (uniform) a spare slot
  const int my_b0 = lane < a.n_cols ? a.bucket0[lane] : 0x7fffffff;
  int ci = (int)__builtin_popcountll(__ballot(my_b0 <= vb)) - 1;
  ci = __builtin_amdgcn_readfirstlane(ci);
  const GCol& c = a.col[ci];
  const int bucket = d.z;
  const int32_t n_pairs = d.y;
  const int64_t* prow = c.pair_row[0] + d.x;
  const bool packed = c.packed != 0;
  const uint32_t M = c.dense_mul;
  const uint32_t base = (uint32_t)dense_first_row(M, bucket);
  uint64_t lim = dense_first_row(M, bucket + 1);
  if (lim > c.map.rows) lim = c.map.rows;
  const int words = (int)((lim - base + 31) >> 5);
  for (int w = tid; w < words; w += kBlock) bm[w] = 0u;
  __syncthreads();
  constexpr int kIn = 8;   // pairs in flight per thread
  for (int32_t e0 = 0; e0 < n_pairs; e0 += kIn * kBlock) {
    int64_t r_[kIn];
#pragma unroll
    for (int k = 0; k < kIn; ++k) {
      const int32_t e = e0 + k * kBlock + tid;
      r_[k] = __builtin_nontemporal_load(prow + (e < n_pairs ? e : n_pairs - 1));
    }
#pragma unroll
    for (int k = 0; k < kIn; ++k) {
      const int32_t e = e0 + k * kBlock + tid;
      uint32_t off = ~0u;
      if (packed) {
        if (e < n_pairs) off = (uint32_t)((uint64_t)r_[k] >> 32) - base;
      } else if (e < n_pairs && r_[k] >= 0) {
        off = (uint32_t)r_[k] - base;
      }
      if (off != ~0u) {
        const uint32_t bit = 1u << (off & 31u);
        if ((bm[off >> 5] & bit) == 0u) atomicOr(&bm[off >> 5], bit);
      }
    }
  }
  __syncthreads();
  int32_t n = 0;
  for (int w = tid; w < words; w += kBlock) n += __builtin_popcount(bm[w]);
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) n += __shfl_xor(n, o, kWave);
  if (lane == 0) wave_tot[wave] = n;
  __syncthreads();
  if (tid == 0) {
    int32_t rows = 0;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) rows += wave_tot[w];
    c.pcount[bucket] = rows;
  }
}
