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
//   D  one scan of the counters gives every row its run [start, start + count) in the sorted
//      order; rows above kRsHotMin / 4 x the average share of a lane group are listed as HOT and
//      placed behind the others;
//   E  the pairs' gradient rows (their numbers) go to their sorted positions, in LDS;
//   F  the WALK: every lane group takes an equal, row-aligned share of the sorted positions and
//      streams through it, W gradient rows in flight per lane, sums in registers, a row leaves
//      (with the optimizer step, whose table / accumulator rows are requested for all rows that
//      finished in the batch) when its run ends.  No barrier, no LDS traffic between lane groups,
//      no float atomics: the hashed and the bitmap paths spend 2 barriers and 2 memory round trips
//      per 48 rows here, this one a round trip per W x groups (= 512 at dim 16) rows;
//   G  a hot row is summed by ALL lane groups (strided slices, partial sums through LDS).
// A job of several chunks (a bucket above kRsCap pairs: skewed ids, the ranges of a split bucket,
// the merge of their partial entries) ranks its rows over all chunks first (A, B), then runs C-G
// per chunk; a row that an earlier chunk emitted is added to (this workgroup owns it), and the
// optimizer step is taken once per row after the last chunk, from the finished sums.
// (kRsCap, pairs per chunk: lookup_bwd.hip)
constexpr int kRsPT = kRsCap / kBlock;           // pairs per thread and chunk
constexpr int kRsSpan = 16384;                   // rows of a bucket's range (bits of the bitmap)
constexpr int kRsWords = kRsSpan / 32;
constexpr int kRsWPT = kRsWords / kBlock;        // bitmap words per thread in the scan
constexpr int kRsBits = 13;                      // start / count fields of a row's counter word
constexpr uint32_t kRsMask = (1u << kRsBits) - 1u;
constexpr uint32_t kRsHotBit = 1u << 31;
constexpr int kRsHotMin = 128;                   // a row is hot above max(this, 4 x pairs / groups)
constexpr int kRsMaxHot = 32;
constexpr uint16_t kRsNoRow = 0xffff;
static_assert(kRsCap % kBlock == 0 && kRsCap <= (1 << (kRsBits - 1)), "counter fields");
static_assert(kRsCap / kRsHotMin <= kRsMaxHot, "hot list");
static_assert(kRsWords % kBlock == 0, "whole bitmap words per thread");
static_assert(kRsSpan <= 65536, "16-bit row offsets");

struct RsLds {
  uint32_t present[kRsWords];   // rows of the job
  uint32_t pre[kRsWords];       // rows before word w
  uint32_t cmap[kRsWords];      // jobs of several chunks: rows of the chunk,
  uint32_t cpre[kRsWords];      //   rows of the chunk before word w,
  uint32_t seen[kRsWords];      //   rows an earlier chunk has emitted
  uint32_t cnt[kRsCap];         // per row of the chunk: tickets, then start | count << 13 | hot
  int32_t sseg[kRsCap];         // gradient row of every pair, sorted by row
  uint16_t su[kRsCap + 8];      // the pair's row (its index among the chunk's rows); hot: kRsNoRow
  uint16_t roff[kRsCap];        // row - first row of the range, per row of the chunk
  float red[kBlock * 4];        // hot rows: the lane groups' partial sums
  uint16_t hot[kRsMaxHot];      // rows of the chunk summed by the whole workgroup
  int32_t wave_tot[kWavesPerBlock];
  int32_t n_main, n_hot, base_u;
};

template <typename V, int STEP>
__device__ inline void rowsort_reduce(const GCol& c, const ReduceJob& job, RsLds& L, int bucket) {
  constexpr int VE = sizeof(V) / 4;
  constexpr int PT = kRsPT;
  constexpr bool adagrad = STEP == 2;
  // gradient rows a lane keeps in flight (with the step, the table / accumulator rows of the rows
  // that finish in a batch travel together: register budget of 128)
  constexpr int W = STEP == 2 ? 4 : STEP == 1 ? 6 : 8;
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
  const bool one_chunk = n_pairs <= kRsCap;
  const float lr = STEP ? job.lr : 0.0f;
  const bool stepping = STEP && lr != 0.0f;
  const bool emit = !(STEP && job.no_emit);    // (no_emit only for jobs of one chunk: decode_job)
  const bool scaled = job.scale && c.combiner != HBK_COMBINER_SUM && c.splits != nullptr;   // uniform
  const uint32_t M = c.dense_mul;
  const uint32_t base = (uint32_t)dense_first_row(M, bucket);
  uint64_t lim = dense_first_row(M, bucket + 1);
  if (lim > c.map.rows) lim = c.map.rows;
  const int words = (int)((lim - base + 31) >> 5);   // <= kRsWords (host: plan_of)

  uint32_t off_[PT];   // row - base of my pairs of the chunk, ~0u: none
  int32_t seg_[PT];
  auto load_pairs = [&](int32_t cb) {
#pragma unroll
    for (int k = 0; k < PT; ++k) {
      const int32_t e = cb + k * kBlock + tid;
      off_[k] = ~0u;
      seg_[k] = e;
      if (e < n_pairs) {
        const int64_t r = HBK_PAIR_LOAD(prow + e);
        if (pseg != nullptr) seg_[k] = HBK_PAIR_LOAD(pseg + e);
        if (r >= 0) off_[k] = (uint32_t)r - base;
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

  load_pairs(0);       // they travel while the tables are cleared
  __syncthreads();     // a workgroup may run several jobs (merge): the previous one is done with L
  for (int w = tid; w < words; w += kBlock) {
    L.present[w] = 0u;
    L.seen[w] = 0u;
  }
  for (int i = tid; i < kRsCap; i += kBlock) L.cnt[i] = 0u;
  if (tid == 0) L.n_hot = 0;
  __syncthreads();

  // A + B over the whole job: its rows, their ranks, the output range
  if (one_chunk) {
    mark(L.present);
  } else {
    for (int32_t cb = 0; cb < n_pairs; cb += kRsCap) {
      if (cb > 0) load_pairs(cb);
      mark(L.present);
    }
  }
  __syncthreads();
  const int32_t n_rows_job = scan_bitmap(L.present, L.pre);
  // One global atomic per job claims the output range; a returning device-scope atomic takes
  // microseconds under load: its round trip runs beside C-E.  Step only: just the count is wanted.
  int32_t claimed = 0;
  if (tid == kBlock - 1) {
    if (!emit) {
      __hip_atomic_fetch_add(job.out_counter, n_rows_job, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      claimed = atomicAdd(job.out_counter, n_rows_job);
    }
  }
  int32_t base_u = 0;

  for (int32_t cb = 0; cb < n_pairs; cb += kRsCap) {
    const uint32_t* bm = L.present;
    const uint32_t* pr = L.pre;
    int32_t n_rows = n_rows_job;
    if (!one_chunk) {
      __syncthreads();   // the chunk before is done with cnt / sseg / su / cmap (pre[] is visible)
      load_pairs(cb);
      for (int w = tid; w < words; w += kBlock) L.cmap[w] = 0u;
      if (cb > 0) {
        for (int i = tid; i < kRsCap; i += kBlock) L.cnt[i] = 0u;
        if (tid == 0) L.n_hot = 0;
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
        if (stepping || !one_chunk) L.roff[u] = (uint16_t)off_[k];   // (every pair of the row: same value)
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

    // D: counters -> runs.  Thread t owns rows [t * PT, t * PT + PT); ordinary rows in the low
    // half of the packed sums, hot rows (placed behind all others) in the high half.
    {
      int32_t n_chunk = n_pairs - cb < kRsCap ? n_pairs - cb : kRsCap;
      int32_t t_hot = 4 * ((n_chunk + groups - 1) / groups);
      if (t_hot < kRsHotMin) t_hot = kRsHotMin;
      uint32_t cn[PT], sum = 0;
#pragma unroll
      for (int k = 0; k < PT; ++k) {
        const int u = tid * PT + k;
        cn[k] = u < n_rows ? L.cnt[u] : 0u;
        sum += (int32_t)cn[k] > t_hot ? cn[k] << 16 : cn[k];
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
      const uint32_t n_main = total & 0xffffu;
#pragma unroll
      for (int k = 0; k < PT; ++k) {
        const int u = tid * PT + k;
        if (u < n_rows) {
          if ((int32_t)cn[k] > t_hot) {
            L.cnt[u] = kRsHotBit | (n_main + (run >> 16)) | (cn[k] << kRsBits);
            L.hot[atomicAdd(&L.n_hot, 1)] = (uint16_t)u;
            run += cn[k] << 16;
          } else {
            L.cnt[u] = (run & 0xffffu) | (cn[k] << kRsBits);
            run += cn[k];
          }
        }
      }
      if (tid == 0) {
        L.n_main = (int32_t)n_main;
        L.su[n_main] = kRsNoRow;   // behind the last ordinary run (hot pairs write the same)
      }
    }
    __syncthreads();

    // E: gradient rows to their sorted positions
#pragma unroll
    for (int k = 0; k < PT; ++k) {
      if (u_[k] != ~0u) {
        const uint32_t cv = L.cnt[u_[k]];
        const int pos = (int)(cv & kRsMask) + tk_[k];
        L.sseg[pos] = seg_[k];
        L.su[pos] = (cv & kRsHotBit) ? kRsNoRow : (uint16_t)u_[k];
      }
    }
    if (cb == 0 && emit && tid == kBlock - 1) L.base_u = job.out_base + claimed;
    __syncthreads();
    if (cb == 0 && emit) base_u = L.base_u;

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

    // F: the walk.  Lane group g owns the runs that START in [g * per, (g + 1) * per) of the
    // ordinary positions.
    {
      const int n_main = L.n_main;
      const int per = (n_main + groups - 1) / groups;
      auto align = [&](int x) -> int {
        if (x <= 0) return 0;
        if (x >= n_main) return n_main;
        const uint32_t cv = L.cnt[L.su[x]];
        const int s = (int)(cv & kRsMask);
        return s == x ? x : s + (int)((cv >> kRsBits) & kRsMask);
      };
      const int p_lo = align(my_group * per), p_hi = align(my_group * per + per);
      V acc = zero_v<V>();
      for (int p = p_lo; p < p_hi; p += W) {
        V g[W];
        int32_t n_[W];
        uint32_t uu[W];
        uint32_t val = 0, fin = 0;   // bit w: position p + w is mine / is the last of its run
#pragma unroll
        for (int w = 0; w < W; ++w) {
          const int q = p + w;
          g[w] = zero_v<V>();
          n_[w] = 0;
          uu[w] = 0;
          if (q < p_hi) {
            val |= 1u << w;
            uu[w] = L.su[q];
            if (L.su[q + 1] != (uint16_t)uu[w]) fin |= 1u << w;
            if (live) g[w] = load_grad_raw<V>(c, job, L.sseg[q], sub, &n_[w]);
          }
        }
        if (scaled) {
#pragma unroll
          for (int w = 0; w < W; ++w) g[w] = scale_grad<V>(c, g[w], n_[w]);
        }
        // the sum of a run ends up in the registers its last gradient row arrived in
#pragma unroll
        for (int w = 0; w < W; ++w) {
          if (val >> w & 1u) {
            acc = acc + g[w];
            if (fin >> w & 1u) {
              g[w] = acc;
              acc = zero_v<V>();
            }
          }
        }
        if (fin == 0u || !live) continue;
        if (one_chunk && stepping) {
          V tv[STEP ? W : 1], av[STEP == 2 ? W : 1];
#pragma unroll
          for (int w = 0; w < W; ++w) {
            if (fin >> w & 1u) {
              const int64_t toff = (int64_t)(base + L.roff[uu[w]]) * c.dim + (int64_t)sub * VE;
              tv[STEP ? w : 0] = HBK_STEP_LOAD(reinterpret_cast<const V*>(c.table + toff));
              if (STEP == 2) {
                av[STEP == 2 ? w : 0] = HBK_STEP_LOAD(reinterpret_cast<const V*>(c.accum + toff));
              }
            }
          }
#pragma unroll
          for (int w = 0; w < W; ++w) {
            if (fin >> w & 1u) {
              if (emit) emit_row<V>(c, job, base_u + (int32_t)uu[w], true, sub, g[w]);
              const int64_t toff = (int64_t)(base + L.roff[uu[w]]) * c.dim + (int64_t)sub * VE;
              step_row<V>(c, adagrad, lr, toff, g[w], tv[STEP ? w : 0],
                          STEP == 2 ? av[STEP == 2 ? w : 0] : zero_v<V>());
            }
          }
        } else {
#pragma unroll
          for (int w = 0; w < W; ++w) {
            if (fin >> w & 1u) {
              emit_row<V>(c, job, out_index(uu[w]), is_first(uu[w]), sub, g[w]);
            }
          }
        }
      }
    }

    // G: hot rows, one after the other, every lane group a strided slice of the run
    const int n_hot = L.n_hot;   // uniform (written before the last barrier)
    for (int h = 0; h < n_hot; ++h) {
      const uint32_t u = L.hot[h];
      const uint32_t cv = L.cnt[u];
      const int b = (int)(cv & kRsMask), e = b + (int)((cv >> kRsBits) & kRsMask);
      V part = zero_v<V>();
      for (int q0 = b + my_group * W; q0 < e; q0 += groups * W) {
        V g[W];
        int32_t n_[W];
#pragma unroll
        for (int w = 0; w < W; ++w) {
          g[w] = zero_v<V>();
          n_[w] = 0;
          if (q0 + w < e && live) g[w] = load_grad_raw<V>(c, job, L.sseg[q0 + w], sub, &n_[w]);
        }
#pragma unroll
        for (int w = 0; w < W; ++w) {
          if (scaled) g[w] = scale_grad<V>(c, g[w], n_[w]);
          part = part + g[w];
        }
      }
      *reinterpret_cast<V*>(&L.red[(size_t)tid * VE]) = part;
      __syncthreads();
      if (my_group == 0 && live) {
        V tot = zero_v<V>();
        for (int gp = 0; gp < groups; ++gp) {
          tot = tot + *reinterpret_cast<const V*>(&L.red[(((size_t)gp << lpr_log2) + sub) * VE]);
        }
        if (one_chunk && stepping) {
          if (emit) emit_row<V>(c, job, base_u + (int32_t)u, true, sub, tot);
          const int64_t toff = (int64_t)(base + L.roff[u]) * c.dim + (int64_t)sub * VE;
          const V tv = HBK_STEP_LOAD(reinterpret_cast<const V*>(c.table + toff));
          V av = zero_v<V>();
          if (adagrad) av = HBK_STEP_LOAD(reinterpret_cast<const V*>(c.accum + toff));
          step_row<V>(c, adagrad, lr, toff, tot, tv, av);
        } else {
          emit_row<V>(c, job, out_index(u), is_first(u), sub, tot);
        }
      }
      __syncthreads();
    }

    if (!one_chunk) {
      __syncthreads();   // every row of the chunk has left: they count as seen from here on
      for (int w = tid; w < words; w += kBlock) L.seen[w] |= L.cmap[w];
    }
  }

  // the row numbers, sorted, straight from the bitmap
  if (emit) {
    for (int w = tid; w < words; w += kBlock) {
      uint32_t m = L.present[w];
      int64_t* o = job.out_rows + base_u + (int32_t)L.pre[w];
      const int64_t r0 = (int64_t)base + 32 * (int64_t)w;
      while (m != 0u) {
        *o++ = r0 + __builtin_ctz(m);
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
          toff[k] = row * c.dim + (int64_t)sub * VE;
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

template <typename V, int STEP>
__global__ __launch_bounds__(kBlock, 4) void bwd_rowsort_kernel(const GArgs a, const int4* desc,
                                                               int slot0, int total,
                                                               const int32_t* poison) {
  __shared__ RsLds lds;
  if (poisoned(poison)) return;   // the grouping launch gave up: no descriptors, no pairs
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
  rowsort_reduce<V, STEP>(a.col[ci], job, lds, d.z);
}
