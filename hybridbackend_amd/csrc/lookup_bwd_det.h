// ---- 5: the DETERMINISTIC backward (round 6; option bwd_deterministic) -- included by
// lookup_bwd.hip, inside its namespace -------------------------------------------------------------
// The default backward takes pair slots by LDS atomics, splits hot buckets over workgroups and
// joins partial sums: every row's sum is right to rounding, but the ORDER of its terms -- hence
// its last bits -- changes from run to run (the reference's TF path, unsorted_segment_sum on a
// GPU, behaves the same).  With bwd_deterministic = 1 a call instead
//   1  turns every id into a key (column, row) and a value (its gradient row),      [det_keys_kernel]
//   2  sorts the pairs by key with a STABLE radix sort (det_prims.hip): a row's terms end up
//      side by side IN ID ORDER,
//   3  flags the first pair of every row and ranks the flags (an exclusive scan):   [det_heads_kernel]
//      rank = output position, rows leave sorted by (column, row),
//   4  lets ONE lane group walk each row's run front to back, acc = acc + term in fp32,
//      and take the optimizer step from the finished sum.                           [det_reduce_kernel]
// The sum of a row is then the same bits on every run and EQUAL to the sequential fp32 sum in id
// order -- oracle.unsorted_segment_sum, TF's CPU kernel -- whatever the column holds (hot rows,
// ragged segments, ids outside the table, segmented inputs).  One path for every column: nothing
// here depends on the bucket plans of the default backward.  Cost: a full sort of the batch's pairs
// and a walk that is as long as the hottest row (measured in DESIGN.md 4.4): a reproducibility
// tool -- TF_DETERMINISTIC_OPS' analogue -- not the fast path.
constexpr int kDetTile = 2048;          // pairs per workgroup in the key / head kernels
constexpr int kDetLanes = 16;           // lanes that walk one row (element d of a row: lane d % 16)
constexpr int kDetMaxE = 16;            // row elements per lane: dim <= 256 (make_rowshape's bound)
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
// order.  Element d of the row belongs to lane d % 16 of the group (4-byte loads: 64-byte pieces of a
// row per group and instruction), so any dim, stride and alignment takes the same path.
__global__ __launch_bounds__(kBlock) void det_reduce_kernel(const DArgs a) {
  constexpr int kGroups = kBlock / kDetLanes;
  const int tid = (int)threadIdx.x;
  const int sub = tid & (kDetLanes - 1);
  const int64_t p = (int64_t)blockIdx.x * kGroups + (tid >> 4);
  if (p >= a.total || a.heads[p] == 0) return;
  const uint64_t key = a.keys[p];
  const uint64_t none = (1ull << a.row_bits) - 1ull;
  const int ci = (int)(key >> a.row_bits);
  const int64_t row = (int64_t)(key & none);
  const DCol& c = a.col[ci];
  const int64_t end = a.base[ci + 1];
  const int dim = c.dim;
  const bool offsets = c.n_runs > 0;
  const bool scaled = c.splits != nullptr && c.combiner != HBK_COMBINER_SUM;
  float acc[kDetMaxE];
#pragma unroll
  for (int e = 0; e < kDetMaxE; ++e) acc[e] = 0.0f;
  // (the keys / values of the NEXT kDetW pairs are requested together with the gradient rows of the
  // current ones: one memory round trip per kDetW terms of a long run instead of two)
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
    float g[kDetW][kDetMaxE];
    float div[kDetW];
#pragma unroll
    for (int w = 0; w < kDetW; ++w) {
      div[w] = 1.0f;
      const int64_t off = offsets ? (int64_t)seg[w] : (int64_t)seg[w] * c.grad_stride;
#pragma unroll
      for (int e = 0; e < kDetMaxE; ++e) {
        const int d = sub + e * kDetLanes;
        g[w][e] = same[w] && d < dim ? c.grad[off + d] : 0.0f;
      }
      if (same[w] && scaled) {
        const int32_t n = c.splits[seg[w] + 1] - c.splits[seg[w]];
        div[w] = c.combiner == HBK_COMBINER_MEAN ? (float)n : sqrtf((float)n);
      }
    }
    // the terms join the sum one after the other, in id order: the order IS the contract
#pragma unroll
    for (int w = 0; w < kDetW; ++w) {
      if (!same[w]) continue;
#pragma unroll
      for (int e = 0; e < kDetMaxE; ++e) {
        const float term = scaled ? g[w][e] / div[w] : g[w][e];
        acc[e] = acc[e] + term;
      }
    }
    if (!same[kDetW - 1]) break;
#pragma unroll
    for (int w = 0; w < kDetW; ++w) {
      seg[w] = seg_n[w];
      same[w] = same_n[w];
    }
  }
  const int32_t u = a.ranks[p] - a.ranks[a.base[ci]];
  if (c.unique_rows != nullptr) {
    if (sub == 0) c.unique_rows[u] = row;
#pragma unroll
    for (int e = 0; e < kDetMaxE; ++e) {
      const int d = sub + e * kDetLanes;
      if (d < dim) c.grad_rows[(int64_t)u * dim + d] = acc[e];
    }
  }
  if (a.lr != 0.0f) {
    // the sparse optimizer step from the finished sum (the arithmetic of step_row)
#pragma unroll
    for (int e = 0; e < kDetMaxE; ++e) {
      const int d = sub + e * kDetLanes;
      if (d >= dim) continue;
      const int64_t t = row * dim + d;
      if (a.apply == HBK_APPLY_ADAGRAD) {
        const float ac = c.accum[t] + acc[e] * acc[e];
        c.accum[t] = ac;
        c.table[t] = c.table[t] - (a.lr * acc[e]) * (1.0f / sqrtf(ac));
      } else {
        c.table[t] = c.table[t] - a.lr * acc[e];
      }
    }
  }
}
