// N-ary elementwise kernels of the path for gfx950:
//   hbk_floormod_n  R1 bucketize `feature % embedding_size` (docs/tutorial/ranking/data.py:179,186)
//   hbk_cast_n      R6 fp32 <-> fp16 wire casts (hbtf/common/cast.cu.cc:37-42,60-65,84-95,287)
// One launch covers all N tensors: a block finds its tensor with a wave-uniform scan of the
// tile prefix in the kernel-argument segment instead of the reference's
// `max_len x N` thread grid + H2D pointer tables (cast.cu.cc:97-285).
#include <hip/hip_fp16.h>

#include "common.h"

namespace hbk {
namespace {

constexpr int kBlock = 256;
constexpr int kPerThread = 8;
constexpr int kTile = kBlock * kPerThread;
constexpr int kMaxCols = 128;

struct EwCol {
  const void* in;
  void* out;
  int64_t len;
  FastDiv div;
  int32_t tile_start;
  int32_t pad_;
};

struct EwArgs {
  int32_t n_cols;
  int32_t pad_;
  EwCol col[kMaxCols];
};
static_assert(sizeof(EwArgs) <= 16384, "kernarg budget");

__device__ inline int find_col(const EwArgs& a, int tile) {
  int lo = 0, hi = a.n_cols;  // wave-uniform binary search over the kernarg descriptors
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (a.col[mid].tile_start <= tile) {
      lo = mid;
    } else {
      hi = mid;
    }
  }
  return lo;
}

struct FloorModI64 {
  static constexpr int kCast = 0;
  typedef int64_t In;
  typedef int64_t Out;
  __device__ static Out apply(In v, const FastDiv& f) { return (Out)floormod_i64(v, f); }
};
struct FloorModI32 {
  static constexpr int kCast = 0;
  typedef int32_t In;
  typedef int32_t Out;
  __device__ static Out apply(In v, const FastDiv& f) { return (Out)floormod_i64((int64_t)v, f); }
};
struct F32ToF16 {
  static constexpr int kCast = 1;
  typedef float In;
  typedef __half Out;
  __device__ static Out apply(In v, const FastDiv&) { return __float2half_rn(v); }
};
struct F16ToF32 {
  static constexpr int kCast = 2;
  typedef __half In;
  typedef float Out;
  __device__ static Out apply(In v, const FastDiv&) { return __half2float(v); }
};

// streams: non-temporal accesses (the builtin does not take __half)
template <typename T>
__device__ inline T stream_load(const T* p) { return __builtin_nontemporal_load(p); }
__device__ inline __half stream_load(const __half* p) { return *p; }
template <typename T>
__device__ inline void stream_store(T v, T* p) { __builtin_nontemporal_store(v, p); }
__device__ inline void stream_store(__half v, __half* p) { *p = v; }

template <typename Op>
__global__ __launch_bounds__(kBlock) void ew_kernel(const EwArgs a) {
  typedef typename Op::In In;
  typedef typename Op::Out Out;
  const int tile = (int)blockIdx.x;
  const EwCol& c = a.col[find_col(a, tile)];
  const In* in = reinterpret_cast<const In*>(c.in);
  Out* out = reinterpret_cast<Out*>(c.out);
  const int64_t base = (int64_t)(tile - c.tile_start) * kTile;
  In v[kPerThread];
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const int64_t i = base + (int64_t)k * kBlock + threadIdx.x;
    if (i < c.len) v[k] = stream_load(in + i);
  }
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const int64_t i = base + (int64_t)k * kBlock + threadIdx.x;
    if (i < c.len) stream_store(Op::apply(v[k], c.div), out + i);
  }
}

// The wire casts move 16 bytes per lane on the wide side: 8 elements per thread, contiguous
// (two 16-byte fp32 accesses against one 16-byte fp16 access); the scalar kernel above handles
// tensors that are not 16-byte aligned.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <bool TO_HALF>
__global__ __launch_bounds__(kBlock) void cast_vec_kernel(const EwArgs a) {
  const int tile = (int)blockIdx.x;
  const EwCol& c = a.col[find_col(a, tile)];
  const int64_t i = (int64_t)(tile - c.tile_start) * kTile + (int64_t)threadIdx.x * 8;
  if (i >= c.len) return;
  if (i + 8 <= c.len) {
    if (TO_HALF) {
      const f32x4* in = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(c.in) + i);
      const f32x4 lo = __builtin_nontemporal_load(in), hi = __builtin_nontemporal_load(in + 1);
      f16x8 h;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        h[k] = (_Float16)lo[k];       // round to nearest even (v_cvt_f16_f32)
        h[4 + k] = (_Float16)hi[k];
      }
      __builtin_nontemporal_store(h, reinterpret_cast<f16x8*>(reinterpret_cast<__half*>(c.out) + i));
    } else {
      const f16x8 h = __builtin_nontemporal_load(
          reinterpret_cast<const f16x8*>(reinterpret_cast<const __half*>(c.in) + i));
      f32x4 lo, hi;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        lo[k] = (float)h[k];
        hi[k] = (float)h[4 + k];
      }
      f32x4* out = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(c.out) + i);
      __builtin_nontemporal_store(lo, out);
      __builtin_nontemporal_store(hi, out + 1);
    }
    return;
  }
  for (int64_t j = i; j < c.len; ++j) {  // the tensor's last, partial group of 8
    if (TO_HALF) {
      reinterpret_cast<__half*>(c.out)[j] = __float2half_rn(reinterpret_cast<const float*>(c.in)[j]);
    } else {
      reinterpret_cast<float*>(c.out)[j] = __half2float(reinterpret_cast<const __half*>(c.in)[j]);
    }
  }
}

template <typename Op>
int launch_n(const char* what, int32_t n, const void* const* inputs, const int64_t* lens,
             const int64_t* divisors, void* const* outputs, hipStream_t stream) {
  HBK_REQUIRE(n >= 0, "%s: n must be >= 0", what);
  if (n == 0) return HBK_OK;
  HBK_REQUIRE(inputs && lens && outputs, "%s: NULL argument array", what);
  int32_t c0 = 0;
  while (c0 < n) {
    EwArgs args;
    int32_t k = 0;
    int64_t tiles = 0;
    while (c0 < n && k < kMaxCols) {
      const int32_t c = c0++;
      HBK_REQUIRE(lens[c] >= 0, "%s: negative length for input %d", what, c);
      if (lens[c] == 0) continue;
      HBK_REQUIRE(inputs[c] && outputs[c], "%s: NULL buffer for input %d", what, c);
      EwCol& d = args.col[k];
      d.in = inputs[c];
      d.out = outputs[c];
      d.len = lens[c];
      if (divisors) {
        HBK_REQUIRE(divisors[c] > 0, "%s: divisor for input %d must be > 0, got %lld", what, c,
                    (long long)divisors[c]);
        d.div = make_fastdiv((uint64_t)divisors[c]);
        d.div.d = (uint64_t)divisors[c];
      } else {
        d.div = make_fastdiv(1);
      }
      d.tile_start = (int32_t)tiles;
      d.pad_ = 0;
      tiles += (lens[c] + kTile - 1) / kTile;
      HBK_REQUIRE(tiles < (1ll << 31), "%s: grid too large", what);
      ++k;
    }
    if (k == 0) continue;
    args.n_cols = k;
    args.pad_ = 0;
    if (Op::kCast != 0) {
      bool aligned = true;
      for (int32_t q = 0; q < k; ++q) {
        aligned &= (((uintptr_t)args.col[q].in | (uintptr_t)args.col[q].out) & 15) == 0;
      }
      if (aligned) {
        if (Op::kCast == 1) {
          hipLaunchKernelGGL(cast_vec_kernel<true>, dim3((unsigned)tiles), dim3(kBlock), 0,
                             stream, args);
        } else {
          hipLaunchKernelGGL(cast_vec_kernel<false>, dim3((unsigned)tiles), dim3(kBlock), 0,
                             stream, args);
        }
        HBK_HIP_OK(hipGetLastError());
        continue;
      }
    }
    hipLaunchKernelGGL(ew_kernel<Op>, dim3((unsigned)tiles), dim3(kBlock), 0, stream, args);
    HBK_HIP_OK(hipGetLastError());
  }
  return HBK_OK;
}

}  // namespace

// used by comm.hip for the fp16 wire format
int cast_n_impl(int32_t n, int32_t src_dtype, int32_t dst_dtype, const void* const* inputs,
                const int64_t* lens, void* const* outputs, hipStream_t stream) {
  if (src_dtype == HBK_FLOAT && dst_dtype == HBK_HALF) {
    return launch_n<F32ToF16>("cast_n", n, inputs, lens, nullptr, outputs, stream);
  }
  if (src_dtype == HBK_HALF && dst_dtype == HBK_FLOAT) {
    return launch_n<F16ToF32>("cast_n", n, inputs, lens, nullptr, outputs, stream);
  }
  return fail(HBK_INVALID_ARGUMENT,
              "cast_n: only float->half and half->float wire casts exist (got %d -> %d)",
              src_dtype, dst_dtype);
}

}  // namespace hbk

extern "C" int hbk_floormod_n(int32_t n_cols, int32_t dtype, const void* const* inputs,
                              const int64_t* lens, const int64_t* buckets,
                              void* const* outputs, hbk_stream_t stream) {
  using namespace hbk;
  HBK_REQUIRE(buckets != nullptr || n_cols == 0, "floormod_n: buckets is NULL");
  if (dtype == HBK_INT64) {
    return launch_n<FloorModI64>("floormod_n", n_cols, inputs, lens, buckets, outputs,
                                 as_stream(stream));
  }
  if (dtype == HBK_INT32) {
    for (int32_t c = 0; c < n_cols; ++c) {
      HBK_REQUIRE(buckets[c] < (1ll << 31), "floormod_n: bucket %lld does not fit int32",
                  (long long)buckets[c]);
    }
    return launch_n<FloorModI32>("floormod_n", n_cols, inputs, lens, buckets, outputs,
                                 as_stream(stream));
  }
  return fail(HBK_INVALID_ARGUMENT, "floormod_n: dtype must be int32 or int64");
}

extern "C" int hbk_cast_n(int32_t n, int32_t src_dtype, int32_t dst_dtype,
                          const void* const* inputs, const int64_t* lens,
                          void* const* outputs, hbk_stream_t stream) {
  return hbk::cast_n_impl(n, src_dtype, dst_dtype, inputs, lens, outputs,
                          hbk::as_stream(stream));
}
