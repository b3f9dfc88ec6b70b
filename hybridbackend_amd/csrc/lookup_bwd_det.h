// ---- 5: the DETERMINISTIC backward by sorting (round 6; option bwd_deterministic = 2, and under = 1 the
// columns the row-sorted jobs' in-order form does not fit: lookup_bwd_rowsort.h DET) -- included by
// lookup_bwd.hip, inside its namespace -------------------------------------------------------------
// The default backward takes pair slots by LDS atomics, splits hot buckets over workgroups and
// joins partial sums: every row's sum is right to rounding, but the ORDER of its terms -- hence
// its last bits -- changes from run to run (the reference's TF path, unsorted_segment_sum on a
// GPU, behaves the same).  This form
//   1  turns every id into a key (column, row) and a value (its gradient row),      [det_keys_kernel]
//   2  sorts the pairs by key with a STABLE radix sort (det_prims.hip): a row's terms end up
//      side by side IN ID ORDER,
//   3  flags the first pair of every row and ranks the flags (an exclusive scan):   [det_heads_kernel]
//      rank = output position, rows leave sorted by (column, row),
//   4  lets ONE lane group walk each row's run front to back, acc = acc + term in fp32,
//      and take the optimizer step from the finished sum.                           [det_reduce_kernel]
// The sum of a row is then the same bits on every run and EQUAL to the sequential fp32 sum in id
// order -- oracle.unsorted_segment_sum, TF's CPU kernel -- whatever the column holds (hot rows,
// ragged segments, ids outside the table, segmented inputs, tables of any size).  Nothing here
// depends on the bucket plans of the default backward: the general form, and the second
// implementation the in-order row-sorted jobs are tested against.  Cost: a full sort of the batch's
// pairs and a walk that is as long as the hottest row (3.5 - 4.4 x the default, 50 x under heavy
// skew).
constexpr int kDetTile = 2048;          // pairs per workgroup in the key / head kernels
constexpr int kDetLanes = 16;           // most lanes that walk one row
constexpr int kDetMaxE = 16;            // most row chunks per lane: dim <= 256 (make_rowshape's bound)
constexpr int kDetW = 4;                // pairs of a run requested together

struct DCol {
  const void* ids;
  const float* grad;
  const int32_t* splits;
  int64_t* unique_rows;      // NULL: step only
  float* grad_rows;
  float* table;
  float* accum;
  const int64_t* run_start;
  const int64_t* run_ids;
  const int64_t* run_grads;
  IdMap map;
  int64_t n_ids;
  int64_t n_seg;
  int32_t dim;
  int32_t grad_stride;
  int32_t n_runs;
  int32_t tpitch;            // floats between rows of table / accum
  uint8_t ids64, combiner, pad_[2];
};

struct DArgs {
  int32_t n_cols;
  int32_t row_bits;          // key = column << row_bits | row; row field all ones: no row
  int32_t apply;             // HBK_APPLY_SGD | HBK_APPLY_ADAGRAD
  float lr;
  int64_t total;             // pairs of the launch group
  uint64_t* keys;            // [total] as the ids come (1) / sorted (3, 4)
  uint32_t* vals;
  int32_t* heads;            // [total] 1: the first pair of a row
  const int32_t* ranks;      // [total] exclusive scan of heads
  int32_t tile0[kMaxCols];   // first workgroup of every column in the key kernel's grid
  int32_t base[kMaxCols + 1];   // first pair of every column
  int32_t* n_unique[kMaxCols];
  DCol col[kMaxCols];
};
static_assert(sizeof(DArgs) <= 24576, "kernarg budget");

// 1: (column, row) keys and gradient-row values, in id order
__global__ __launch_bounds__(kBlock) void det_keys_kernel(const DArgs a) {
  HBK_FIND_COL(a, tile0)
  const int tid = (int)threadIdx.x;
  const int64_t j0 = (int64_t)((int)blockIdx.x - a.tile0[ci]) * kDetTile;
  const uint64_t none = (1ull << a.row_bits) - 1ull;
  int k_run = -1;
  int64_t run_next = 0, id_delta = 0, grad_delta = 0;
  for (int k = 0; k < kDetTile / kBlock; ++k) {
    const int64_t j = j0 + (int64_t)k * kBlock + tid;
    if (j >= c.n_ids) break;
    uint32_t seg = (uint32_t)j;
    if (c.n_runs > 0) {          // segmented inputs: the value is the gradient row's float offset
      while (j >= run_next) {
        ++k_run;
        const int64_t start = c.run_start[k_run];
        run_next = k_run + 1 < c.n_runs ? c.run_start[k_run + 1] : (int64_t)1 << 62;
        id_delta = c.run_ids[k_run] - start;
        grad_delta = c.run_grads[k_run] - start * c.dim;
      }
      seg = (uint32_t)(grad_delta + j * c.dim);
    } else if (c.splits != nullptr) {   // ragged: the segment that holds position j
      int64_t lo = 0, hi = c.n_seg;     // last s with splits[s] <= j
      while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)c.splits[mid] <= j) {
          lo = mid;
        } else {
          hi = mid;
        }
      }
      seg = (uint32_t)lo;
    }
    const int64_t id = load_id(c.ids, c.ids64 != 0, j + id_delta);
    const uint64_t r = id_to_row(c.map, id);
    const uint64_t key = ((uint64_t)ci << a.row_bits) | (r == kNoRow ? none : r);
    a.keys[a.base[ci] + j] = key;
    a.vals[a.base[ci] + j] = seg;
  }
}

// 3: the first pair of every (column, row) run of the sorted keys
__global__ __launch_bounds__(kBlock) void det_heads_kernel(const DArgs a) {
  const uint64_t none = (1ull << a.row_bits) - 1ull;
  const int64_t p0 = (int64_t)blockIdx.x * kDetTile;
  for (int k = 0; k < kDetTile / kBlock; ++k) {
    const int64_t p = p0 + (int64_t)k * kBlock + (int)threadIdx.x;
    if (p >= a.total) break;
    const uint64_t key = a.keys[p];
    const bool head = (key & none) != none && (p == 0 || a.keys[p - 1] != key);
    a.heads[p] = head ? 1 : 0;
  }
}

// the distinct rows of every column: heads inside the column's range of the sorted pairs
__global__ void det_counts_kernel(const DArgs a) {
  const int c = (int)threadIdx.x;
  if (c >= a.n_cols) return;
  const int64_t b = a.base[c], e = a.base[c + 1];
  int32_t n = 0;
  if (e > b) {
    const int32_t upto = e < a.total ? a.ranks[e] : a.ranks[a.total - 1] + a.heads[a.total - 1];
    n = upto - a.ranks[b];
  }
  *a.n_unique[c] = n;
}

// 4: one lane group per sorted position; the group of a row's FIRST pair walks the row's run in
// order.  A row is split into chunks of V (16 bytes when every column of the launch group has
// 16-byte rows, else 4) over L = 2^L_LOG2 lanes, E chunks per lane (chunk sub + e * L): the host
// picks the smallest (L, E) that covers the group's widest row.
template <typename V>
__device__ inline V det_div(V g, float by);
template <>
__device__ inline float det_div<float>(float g, float by) { return g / by; }
template <>
__device__ inline f32x4 det_div<f32x4>(f32x4 g, float by) {
  return f32x4{g.x / by, g.y / by, g.z / by, g.w / by};
}

template <typename V, int L_LOG2, int E>
__global__ __launch_bounds__(kBlock) void det_reduce_kernel(const DArgs a) {
  constexpr int L = 1 << L_LOG2;
  constexpr int VE = sizeof(V) / 4;
  constexpr int kGroups = kBlock / L;
  const int tid = (int)threadIdx.x;
  const int sub = tid & (L - 1);
  const int64_t p = (int64_t)blockIdx.x * kGroups + (tid >> L_LOG2);
  if (p >= a.total || a.heads[p] == 0) return;
  const uint64_t key = a.keys[p];
  const uint64_t none = (1ull << a.row_bits) - 1ull;
  const int ci = (int)(key >> a.row_bits);
  const int64_t row = (int64_t)(key & none);
  const DCol& c = a.col[ci];
  const int64_t end = a.base[ci + 1];
  const int chunks = c.dim / VE;
  const bool offsets = c.n_runs > 0;
  const bool scaled = c.splits != nullptr && c.combiner != HBK_COMBINER_SUM;
  // the gradient row (chunk e of this lane) of the pair at sorted position q, divided as the
  // combiner asks -- the term of the sum
  auto term = [&](uint32_t seg, int e) -> V {
    const int64_t off = offsets ? (int64_t)seg : (int64_t)seg * c.grad_stride;
    return *reinterpret_cast<const V*>(c.grad + off + (int64_t)(sub + e * L) * VE);
  };
  auto divisor = [&](uint32_t seg) -> float {
    const int32_t n = c.splits[seg + 1] - c.splits[seg];
    return c.combiner == HBK_COMBINER_MEAN ? (float)n : sqrtf((float)n);
  };
  V acc[E];
#pragma unroll
  for (int e = 0; e < E; ++e) acc[e] = zero_v<V>();
  // the run goes on while the next pair has the same key (the ids outside the table sit behind the
  // column's last row with a key of their own and no head flag: the flags cannot tell)
  const bool single = p + 1 >= end || a.keys[p + 1] != key;
  if (single) {   // (most rows of most batches: one term, no loop)
    const uint32_t seg = a.vals[p];
    const float by = scaled ? divisor(seg) : 1.0f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      if (sub + e * L < chunks) {
        const V g = term(seg, e);
        acc[e] = acc[e] + (scaled ? det_div<V>(g, by) : g);
      }
    }
  } else {
    // kDetW pairs per round; the flags / values of the NEXT round are requested together with the
    // gradient rows of this one: one memory round trip per kDetW terms of a long run
    uint32_t seg[kDetW];
    bool same[kDetW];
    auto fetch = [&](int64_t q, uint32_t (&sg)[kDetW], bool (&sm)[kDetW]) {
#pragma unroll
      for (int w = 0; w < kDetW; ++w) {
        const int64_t qq = q + w < end ? q + w : end - 1;
        sm[w] = q + w < end && a.keys[qq] == key;   // sorted: the row's pairs are a prefix
        sg[w] = a.vals[qq];
      }
    };
    fetch(p, seg, same);
    for (int64_t q = p;; q += kDetW) {
      uint32_t seg_n[kDetW];
      bool same_n[kDetW];
      fetch(q + kDetW, seg_n, same_n);
      V g[kDetW][E];
      float by[kDetW];
#pragma unroll
      for (int w = 0; w < kDetW; ++w) {
        by[w] = same[w] && scaled ? divisor(seg[w]) : 1.0f;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          g[w][e] = same[w] && sub + e * L < chunks ? term(seg[w], e) : zero_v<V>();
        }
      }
      // the terms join the sum one after the other, in id order: the order IS the contract
#pragma unroll
      for (int w = 0; w < kDetW; ++w) {
        if (!same[w]) continue;
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = acc[e] + (scaled ? det_div<V>(g[w][e], by[w]) : g[w][e]);
      }
      if (!same[kDetW - 1] || !same_n[0]) break;
#pragma unroll
      for (int w = 0; w < kDetW; ++w) {
        seg[w] = seg_n[w];
        same[w] = same_n[w];
      }
    }
  }
  const int32_t u = a.ranks[p] - a.ranks[a.base[ci]];
  if (c.unique_rows != nullptr) {
    if (sub == 0) c.unique_rows[u] = row;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int ch = sub + e * L;
      if (ch < chunks) *reinterpret_cast<V*>(c.grad_rows + (int64_t)u * c.dim + (int64_t)ch * VE) = acc[e];
    }
  }
  if (a.lr != 0.0f) {
    // the sparse optimizer step from the finished sum (the arithmetic of step_row)
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int ch = sub + e * L;
      if (ch >= chunks) continue;
      const int64_t t = row * c.tpitch + (int64_t)ch * VE;
      V* tp = reinterpret_cast<V*>(c.table + t);
      if (a.apply == HBK_APPLY_ADAGRAD) {
        V* ap = reinterpret_cast<V*>(c.accum + t);
        const V ac = *ap + acc[e] * acc[e];
        *ap = ac;
        *tp = *tp - (a.lr * acc[e]) * rsqrt_v<V>(ac);
      } else {
        *tp = *tp - a.lr * acc[e];
      }
    }
  }
}

// (V, L, E) for a launch group whose widest row has `chunks` chunks of V
template <typename V>
inline void det_launch_reduce(const DArgs& a, int chunks, hipStream_t stream) {
  int l_log2 = 0;
  while (l_log2 < 4 && (1 << l_log2) < chunks) ++l_log2;
  const int e = (chunks + (1 << l_log2) - 1) >> l_log2;
  const auto grid = [&](int groups) { return dim3((unsigned)((a.total + groups - 1) / groups)); };
#define HBK_DET_CASE(LL, EE)                                                                      \
  hipLaunchKernelGGL((det_reduce_kernel<V, LL, EE>), grid(kBlock >> LL), dim3(kBlock), 0, stream, a)
  if (l_log2 == 0) {
    HBK_DET_CASE(0, 1);
  } else if (l_log2 == 1) {
    HBK_DET_CASE(1, 1);
  } else if (l_log2 == 2) {
    HBK_DET_CASE(2, 1);
  } else if (l_log2 == 3) {
    HBK_DET_CASE(3, 1);
  } else if (e <= 1) {
    HBK_DET_CASE(4, 1);
  } else if (e <= 2) {
    HBK_DET_CASE(4, 2);
  } else if (e <= 4) {
    HBK_DET_CASE(4, 4);
  } else if (e <= 8) {
    HBK_DET_CASE(4, 8);
  } else {
    HBK_DET_CASE(4, 16);
  }
#undef HBK_DET_CASE
}
