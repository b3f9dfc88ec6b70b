// HbGroupLookupGrad: backward of the fused group lookup for gfx950 (R10) -- replaces the TF
// autodiff chain SparseSegment*Grad -> GatherV2 grad -> UnsortedSegmentSum (duplicate-id
// reduction) -> IndexedSlices(values, indices = row) of SURVEY 3.4 / hbtf/embedding/
// sharding.py:186-200 in reverse, optionally fused with the sparse SGD apply on the shard
// (sharded variables are not aggregated across ranks, hbtf/training/gradient.py:193-217).
//
//   1 rows      row(j) for every id (bucketize + `// W`), all columns in one launch
//   2 unique    first-occurrence unique of the rows (unique.hip) -> unique_rows, inverse
//               index, multiplicity of every distinct row
//   3 zero      only the grad rows that will receive more than one contribution
//   4 scatter   per segment: scaled grad chunk -> grad_rows[inverse(j)].  A row touched once
//               (the common case for uniform ids) takes a plain 16-byte store -- no atomics,
//               no prior zeroing; duplicated rows take fp32 atomic adds (order not fixed:
//               1e-5 relative tolerance)
//   5 apply     (apply_lr != 0) table[unique_rows[u]] -= lr * grad_rows[u]
#include <alloca.h>
#include <string.h>

#include <vector>

#include "lookup_common.h"
#include "unique.h"

namespace hbk {
namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kMaxCols = 22;
constexpr int kRowsTile = kBlock * 4;  // ids per block in the rows kernel
constexpr int kIters = 4;              // segments (scatter) / rows (zero, apply) per lane group

struct BCol {
  const void* ids;
  int64_t* rows_tmp;         // [n_ids] row(j), -1 when out of range
  const float* grad_out;     // [n_seg, dim]
  const int32_t* splits;
  const int32_t* inv;        // [n_ids] position of row(j) in unique_rows
  const int32_t* mult;       // [n_ids] multiplicity per unique row
  const int32_t* n_unique;
  const int64_t* unique_rows;
  float* grad_rows;
  float* table;
  IdMap map;
  int64_t n_ids;
  int64_t n_seg;
  int32_t dim;
  int32_t chunks;
  uint8_t lpr_log2, ids64, combiner, vec4;
  int32_t tile_ids;   // first tile of this column in the rows kernel (kRowsTile ids per tile)
  int32_t tile_seg;   // ... in the scatter kernel (4 * kIters * rpi segments per tile)
  int32_t tile_urow;  // ... in the zero / apply kernels (4 * kIters * rpi unique rows per tile)
};

struct BArgs {
  int32_t n_cols;
  float lr;
  BCol col[kMaxCols];
};
static_assert(sizeof(BArgs) <= 4096, "kernarg budget");

#define HBK_FIND_COL(FIELD)                                                  \
  int ci = 0;                                                                \
  while (ci + 1 < a.n_cols && a.col[ci + 1].FIELD <= (int)blockIdx.x) ++ci;  \
  const BCol& c = a.col[ci];                                                 \
  const int64_t tile = (int)blockIdx.x - c.FIELD;

__global__ __launch_bounds__(kBlock) void bwd_rows_kernel(const BArgs a) {
  HBK_FIND_COL(tile_ids)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t j = tile * kRowsTile + (int64_t)k * kBlock + threadIdx.x;
    if (j < c.n_ids) {
      const uint64_t r = id_to_row(c.map, load_id(c.ids, c.ids64, j));
      c.rows_tmp[j] = r == kNoRow ? -1 : (int64_t)r;
    }
  }
}

template <typename V>
__device__ inline void atomic_add_v(float* p, V v);
template <>
__device__ inline void atomic_add_v<f32x4>(float* p, f32x4 v) {
  unsafeAtomicAdd(p + 0, v.x);
  unsafeAtomicAdd(p + 1, v.y);
  unsafeAtomicAdd(p + 2, v.z);
  unsafeAtomicAdd(p + 3, v.w);
}
template <>
__device__ inline void atomic_add_v<float>(float* p, float v) { unsafeAtomicAdd(p, v); }

// zero the grad rows that will be accumulated into (multiplicity > 1)
template <typename V>
__device__ inline void zero_rows(const BCol& c, int64_t urow0) {
  constexpr int VE = sizeof(V) / 4;
  const int lane = lane_id();
  const int rpi = kWave >> c.lpr_log2;
  const int sub = lane & ((1 << c.lpr_log2) - 1);
  const int grp = lane >> c.lpr_log2;
  const int32_t n_u = *c.n_unique;
  for (int it = 0; it < kIters; ++it) {
    const int64_t u = urow0 + (int64_t)it * rpi + grp;
    if (u < n_u && sub < c.chunks && c.mult[u] > 1) {
      *reinterpret_cast<V*>(c.grad_rows + u * (int64_t)c.dim + (int64_t)sub * VE) = zero_v<V>();
    }
  }
}

__global__ __launch_bounds__(kBlock) void bwd_zero_kernel(const BArgs a) {
  HBK_FIND_COL(tile_urow)
  const int rpi = kWave >> c.lpr_log2;
  const int64_t urow0 = (tile * kWavesPerBlock + (threadIdx.x >> 6)) * (int64_t)(kIters * rpi);
  if (urow0 >= *c.n_unique) return;
  if (c.vec4) {
    zero_rows<f32x4>(c, urow0);
  } else {
    zero_rows<float>(c, urow0);
  }
}

template <typename V>
__device__ inline void scatter_segments(const BCol& c, int64_t seg0) {
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
        reinterpret_cast<const V*>(c.grad_out + s * (int64_t)c.dim + (int64_t)sub * VE));
    const int32_t n = end - beg;
    if (c.combiner == HBK_COMBINER_MEAN) {
      g = g / (float)n;
    } else if (c.combiner == HBK_COMBINER_SQRTN) {
      g = g / sqrtf((float)n);
    }
    for (int32_t j = beg; j < end; ++j) {
      const int32_t u = c.inv[j];
      float* dst = c.grad_rows + (int64_t)u * c.dim + (int64_t)sub * VE;
      if (c.mult == nullptr || c.mult[u] == 1) {
        *reinterpret_cast<V*>(dst) = g;
      } else {
        atomic_add_v<V>(dst, g);
      }
    }
  }
}

__global__ __launch_bounds__(kBlock) void bwd_scatter_kernel(const BArgs a) {
  HBK_FIND_COL(tile_seg)
  const int rpi = kWave >> c.lpr_log2;
  const int64_t seg0 = (tile * kWavesPerBlock + (threadIdx.x >> 6)) * (int64_t)(kIters * rpi);
  if (seg0 >= c.n_seg) return;
  if (c.vec4) {
    scatter_segments<f32x4>(c, seg0);
  } else {
    scatter_segments<float>(c, seg0);
  }
}

template <typename V>
__device__ inline void apply_rows(const BCol& c, float lr, int64_t urow0) {
  constexpr int VE = sizeof(V) / 4;
  const int lane = lane_id();
  const int rpi = kWave >> c.lpr_log2;
  const int sub = lane & ((1 << c.lpr_log2) - 1);
  const int grp = lane >> c.lpr_log2;
  const int32_t n_u = *c.n_unique;
  for (int it = 0; it < kIters; ++it) {
    const int64_t u = urow0 + (int64_t)it * rpi + grp;
    if (u >= n_u || sub >= c.chunks) continue;
    const int64_t r = c.unique_rows[u];
    if (r < 0) continue;
    const V g = *reinterpret_cast<const V*>(c.grad_rows + u * (int64_t)c.dim + (int64_t)sub * VE);
    V* t = reinterpret_cast<V*>(c.table + r * (int64_t)c.dim + (int64_t)sub * VE);
    *t = *t - lr * g;
  }
}

__global__ __launch_bounds__(kBlock) void bwd_apply_kernel(const BArgs a) {
  HBK_FIND_COL(tile_urow)
  const int rpi = kWave >> c.lpr_log2;
  const int64_t urow0 = (tile * kWavesPerBlock + (threadIdx.x >> 6)) * (int64_t)(kIters * rpi);
  if (urow0 >= *c.n_unique) return;
  if (c.vec4) {
    apply_rows<f32x4>(c, a.lr, urow0);
  } else {
    apply_rows<float>(c, a.lr, urow0);
  }
}

inline size_t align8(size_t v) { return (v + 7) & ~(size_t)7; }

struct BwdLayout {
  size_t per_col_bytes;  // rows_tmp + inv + mult of every column
  size_t unique_bytes;
  size_t total() const { return per_col_bytes + unique_bytes; }
};

BwdLayout layout_of(int32_t n_cols, const hbk_lookup_grad_column_t* cols) {
  BwdLayout l = {0, 0};
  int64_t* lens = (int64_t*)alloca(sizeof(int64_t) * (size_t)(n_cols > 0 ? n_cols : 1));
  for (int32_t c = 0; c < n_cols; ++c) {
    const int64_t n = cols[c].n_ids > 0 ? cols[c].n_ids : 0;
    lens[c] = n;
    l.per_col_bytes += (size_t)n * 8 + 2 * align8((size_t)n * 4);
  }
  l.unique_bytes = unique_workspace_bytes(n_cols, lens);
  return l;
}

}  // namespace
}  // namespace hbk

extern "C" size_t hbk_group_lookup_bwd_workspace_bytes(int32_t n_cols,
                                                       const hbk_lookup_grad_column_t* cols) {
  if (n_cols <= 0 || cols == nullptr) return 0;
  return hbk::layout_of(n_cols, cols).total();
}

extern "C" int hbk_group_lookup_bwd(int32_t n_cols, const hbk_lookup_grad_column_t* cols,
                                    float apply_lr, void* workspace, size_t workspace_bytes,
                                    hbk_stream_t stream_) {
  using namespace hbk;
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
    HBK_REQUIRE(h.n_ids == 0 || (h.ids && h.grad_out && h.unique_rows && h.grad_rows),
                "group_lookup_bwd: column %d: NULL buffer", c);
    HBK_REQUIRE(apply_lr == 0.0f || h.table != nullptr || h.n_ids == 0,
                "group_lookup_bwd: column %d: table is NULL but apply_lr != 0", c);
  }
  const BwdLayout l = layout_of(n_cols, cols);
  HBK_REQUIRE(l.total() == 0 || (workspace != nullptr && workspace_bytes >= l.total()),
              "group_lookup_bwd: workspace too small: need %zu bytes, got %zu", l.total(),
              workspace_bytes);
  HBK_REQUIRE(((uintptr_t)workspace & 7) == 0,
              "group_lookup_bwd: workspace must be 8-byte aligned");
  char* wp = reinterpret_cast<char*>(workspace);
  void* unique_ws = wp + l.per_col_bytes;

  UniqueColumn* ucols = (UniqueColumn*)alloca(sizeof(UniqueColumn) * (size_t)n_cols);
  int32_t c0 = 0;
  int32_t u0 = 0;  // columns already described in ucols
  // pass 1: rows for every column group
  struct Group { BArgs args; int64_t t_ids, t_seg, t_urow; };
  std::vector<Group> groups;
  while (c0 < n_cols) {
    groups.emplace_back();
    Group& gr = groups.back();
    BArgs& args = gr.args;
    int32_t k = 0;
    int64_t t_ids = 0, t_seg = 0, t_urow = 0;
    while (c0 < n_cols && k < kMaxCols) {
      const hbk_lookup_grad_column_t& h = cols[c0++];
      UniqueColumn& uc = ucols[u0++];
      const int64_t n = h.n_ids;
      int64_t* rows_tmp = reinterpret_cast<int64_t*>(wp);
      wp += (size_t)n * 8;
      int32_t* inv = reinterpret_cast<int32_t*>(wp);
      wp += align8((size_t)n * 4);
      int32_t* mult = reinterpret_cast<int32_t*>(wp);
      wp += align8((size_t)n * 4);
      uc.in = rows_tmp;
      uc.len = n;
      uc.unique_out = h.unique_rows;
      uc.index_out = inv;
      uc.n_unique = h.n_unique;
      uc.multiplicity = mult;
      if (n == 0) continue;
      BCol& d = args.col[k];
      d.ids = h.ids;
      d.rows_tmp = rows_tmp;
      d.grad_out = h.grad_out;
      d.splits = h.row_splits;
      d.inv = inv;
      d.mult = mult;
      d.n_unique = h.n_unique;
      d.unique_rows = h.unique_rows;
      d.grad_rows = h.grad_rows;
      d.table = h.table;
      d.map = make_idmap(h.bucket, h.divisor, h.rows);
      d.n_ids = n;
      d.n_seg = h.n_segments;
      d.dim = h.dim;
      RowShape shape;
      HBK_REQUIRE(make_rowshape(h.dim,
                                (uintptr_t)h.grad_out | (uintptr_t)h.grad_rows |
                                    (apply_lr != 0.0f ? (uintptr_t)h.table : 0),
                                &shape),
                  "group_lookup_bwd: dim %d needs more than 64 lanes per row", h.dim);
      d.chunks = shape.chunks;
      d.lpr_log2 = shape.lpr_log2;
      d.vec4 = shape.vec4;
      d.ids64 = h.ids_dtype == HBK_INT64;
      d.combiner = (uint8_t)h.combiner;
      const int64_t rpi = kWave >> d.lpr_log2;
      const int64_t per_block = kWavesPerBlock * kIters * rpi;
      d.tile_ids = (int32_t)t_ids;
      d.tile_seg = (int32_t)t_seg;
      d.tile_urow = (int32_t)t_urow;
      t_ids += (n + kRowsTile - 1) / kRowsTile;
      t_seg += (h.n_segments + per_block - 1) / per_block;
      t_urow += (n + per_block - 1) / per_block;
      HBK_REQUIRE(t_ids < (1ll << 31) && t_seg < (1ll << 31) && t_urow < (1ll << 31),
                  "group_lookup_bwd: grid too large");
      ++k;
    }
    args.n_cols = k;
    args.lr = apply_lr;
    gr.t_ids = t_ids;
    gr.t_seg = t_seg;
    gr.t_urow = t_urow;
    if (k > 0 && t_ids > 0) {
      hipLaunchKernelGGL(bwd_rows_kernel, dim3((unsigned)t_ids), dim3(kBlock), 0, stream, args);
      HBK_HIP_OK(hipGetLastError());
    }
  }
  // pass 2: unique over the rows of all columns
  int rc = unique_n_impl(n_cols, ucols, unique_ws, l.unique_bytes, stream);
  if (rc != HBK_OK) return rc;
  // pass 3-5
  for (Group& gr : groups) {
    if (gr.args.n_cols == 0) continue;
    if (gr.t_urow > 0) {
      hipLaunchKernelGGL(bwd_zero_kernel, dim3((unsigned)gr.t_urow), dim3(kBlock), 0, stream,
                         gr.args);
    }
    if (gr.t_seg > 0) {
      hipLaunchKernelGGL(bwd_scatter_kernel, dim3((unsigned)gr.t_seg), dim3(kBlock), 0, stream,
                         gr.args);
    }
    if (apply_lr != 0.0f && gr.t_urow > 0) {
      hipLaunchKernelGGL(bwd_apply_kernel, dim3((unsigned)gr.t_urow), dim3(kBlock), 0, stream,
                         gr.args);
    }
    HBK_HIP_OK(hipGetLastError());
  }
  return HBK_OK;
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
    BArgs args;
    int32_t k = 0;
    int64_t t_seg = 0;
    while (c0 < n_cols && k < kMaxCols) {
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
      BCol& d = args.col[k];
      memset(&d, 0, sizeof(d));
      d.grad_out = h.grad_out;
      d.splits = h.row_splits;
      d.inv = h.index;
      d.mult = nullptr;
      d.grad_rows = h.grad_rows;
      d.n_ids = h.n_ids;
      d.n_seg = h.n_segments;
      d.dim = h.dim;
      RowShape shape;
      HBK_REQUIRE(make_rowshape(h.dim, (uintptr_t)h.grad_out | (uintptr_t)h.grad_rows, &shape),
                  "group_stitch_bwd: dim %d needs more than 64 lanes per row", h.dim);
      d.chunks = shape.chunks;
      d.lpr_log2 = shape.lpr_log2;
      d.vec4 = shape.vec4;
      d.combiner = (uint8_t)h.combiner;
      const int64_t rpi = kWave >> d.lpr_log2;
      const int64_t per_block = kWavesPerBlock * kIters * rpi;
      d.tile_seg = (int32_t)t_seg;
      t_seg += (h.n_segments + per_block - 1) / per_block;
      HBK_REQUIRE(t_seg < (1ll << 31), "group_stitch_bwd: grid too large");
      ++k;
    }
    if (k == 0) continue;
    args.n_cols = k;
    args.lr = 0.0f;
    hipLaunchKernelGGL(bwd_scatter_kernel, dim3((unsigned)t_seg), dim3(kBlock), 0, stream, args);
    HBK_HIP_OK(hipGetLastError());
  }
  return HBK_OK;
}
