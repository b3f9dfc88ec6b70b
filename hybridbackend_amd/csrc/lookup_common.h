// Pieces shared by the forward and backward group-lookup kernels.
#ifndef HBK_CSRC_LOOKUP_COMMON_H_
#define HBK_CSRC_LOOKUP_COMMON_H_

#include "common.h"

namespace hbk {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr uint64_t kNoRow = ~0ull;

// id -> table row: optional bucketize (R1, TF FloorMod), optional owner-side `// W`
// (hbtf/embedding/sharding.py:189), range check against the (local) table.
struct IdMap {
  FastDiv bucket;  // bucket.d == 0: ids are row numbers already
  FastDiv div;
  uint64_t rows;
};

inline IdMap make_idmap(int64_t bucket, int32_t divisor, int64_t rows) {
  IdMap m;
  m.bucket = make_fastdiv((uint64_t)bucket);
  m.bucket.d = (uint64_t)bucket;
  m.div = make_fastdiv((uint64_t)divisor);
  m.rows = (uint64_t)rows;
  return m;
}

__device__ inline uint64_t id_to_row(const IdMap& m, int64_t id) {
  uint64_t r;
  if (m.bucket.d != 0) {
    r = floormod_i64(id, m.bucket);
  } else {
    if (id < 0) return kNoRow;
    r = (uint64_t)id;
  }
  r = fastdiv(r, m.div);
  return r < m.rows ? r : kNoRow;
}

// ids are read once per kernel, in order: non-temporal (probe builds: -DHBK_IDS_NT=0)
#ifndef HBK_IDS_NT
#define HBK_IDS_NT 1
#endif
__device__ inline int64_t load_id(const void* ids, bool ids64, int64_t j) {
#if HBK_IDS_NT
  if (ids64) return __builtin_nontemporal_load(reinterpret_cast<const int64_t*>(ids) + j);
  return (int64_t)__builtin_nontemporal_load(reinterpret_cast<const int32_t*>(ids) + j);
#else
  if (ids64) return reinterpret_cast<const int64_t*>(ids)[j];
  return (int64_t)reinterpret_cast<const int32_t*>(ids)[j];
#endif
}

__device__ inline uint64_t shfl_u64(uint64_t v, int src_lane) {
  int lo = __shfl((int)(uint32_t)v, src_lane, kWave);
  int hi = __shfl((int)(uint32_t)(v >> 32), src_lane, kWave);
  return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}

template <typename V>
__device__ inline V zero_v();
template <>
__device__ inline f32x4 zero_v<f32x4>() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
template <>
__device__ inline float zero_v<float>() { return 0.f; }

inline int ceil_log2_u32(uint32_t v) {
  int l = 0;
  while ((1u << l) < v) ++l;
  return l;
}

// How a row of `dim` floats is split over lanes: 16-byte chunks when dim % 4 == 0 and the
// buffers are 16-byte aligned, else 4-byte chunks; lanes per row = pow2 >= chunks.
struct RowShape {
  int32_t chunks;
  uint8_t lpr_log2;
  uint8_t vec4;
};

inline bool make_rowshape(int32_t dim, uintptr_t align_bits, RowShape* s) {
  const bool vec4 = (dim % 4 == 0) && (align_bits % 16 == 0);
  s->vec4 = vec4 ? 1 : 0;
  s->chunks = vec4 ? dim / 4 : dim;
  if (s->chunks > kWave) return false;
  s->lpr_log2 = (uint8_t)ceil_log2_u32((uint32_t)s->chunks);
  return true;
}

}  // namespace hbk

#endif  // HBK_CSRC_LOOKUP_COMMON_H_
