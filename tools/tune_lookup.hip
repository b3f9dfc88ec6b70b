// Standalone tuning probe for the dominant kernel (fused multi-table row gather, one id per
// segment, dim 16 fp32): runs template variants of the gather back to back on one MI355X and
// prints achieved algorithmic GB/s for each (136 B per lookup).  Not part of the library;
// results feed the constants in hybridbackend_amd/csrc/lookup_fwd.hip and DESIGN.md.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tune_lookup.hip -o tools/bin/tune_lookup
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../include/hbk.h"

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                       \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kCols = 26;
constexpr int kDim = 16;

struct Args {
  const float* table[kCols];
  const int64_t* ids;  // [kCols * B]
  float* out;          // [kCols * B, 16]
  int B;
  uint64_t rows;
  uint64_t magic;
  uint32_t shift;
};

__device__ inline uint64_t to_row(int64_t id, const Args& a) {
  uint64_t u = id < 0 ? (uint64_t)(~id) : (uint64_t)id;
  uint64_t t = __umul64hi(a.magic, u);
  uint64_t q = (((u - t) >> 1) + t) >> a.shift;
  uint64_t r = u - q * a.rows;
  return id < 0 ? a.rows - 1 - r : r;
}

// MODE 0: ids one per lane + shuffle (library scheme); MODE 1: every lane of the row group
// loads its own id (redundant, TA-coalesced).
template <int U, int BLOCK, bool NT_ROW, bool NT_OUT, int MODE>
__global__ __launch_bounds__(BLOCK) void gather_variant(const Args a) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  constexpr int WAVES = BLOCK / 64;
  const int sub = lane & 3, grp = lane >> 2;
  const int64_t total = (int64_t)kCols * a.B;
  const int64_t row0 = ((int64_t)blockIdx.x * WAVES + wave) * (16 * U);
  if (row0 >= total) return;
  f32x4 v[U];
  if (MODE == 0) {
    static_assert(U <= 4 || MODE != 0, "one id register per lane covers 64 rows");
    uint64_t myrow = 0;
    int mycol = 0;
    const int64_t s = row0 + lane;
    if (lane < 16 * U && s < total) {
      mycol = (int)(s / a.B);
      myrow = to_row(__builtin_nontemporal_load(a.ids + s), a);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int src = u * 16 + grp;
      const uint32_t lo = __shfl((int)(uint32_t)myrow, src, 64);
      const uint32_t hi = __shfl((int)(uint32_t)(myrow >> 32), src, 64);
      const int col = __shfl(mycol, src, 64);
      const uint64_t r = ((uint64_t)hi << 32) | lo;
      const f32x4* p = reinterpret_cast<const f32x4*>(a.table[col] + r * kDim + sub * 4);
      v[u] = (row0 + src < total) ? (NT_ROW ? __builtin_nontemporal_load(p) : *p)
                                  : f32x4{0, 0, 0, 0};
    }
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t s = row0 + u * 16 + grp;
      v[u] = f32x4{0, 0, 0, 0};
      if (s < total) {
        const int col = (int)(s / a.B);
        const uint64_t r = to_row(__builtin_nontemporal_load(a.ids + s), a);
        const f32x4* p = reinterpret_cast<const f32x4*>(a.table[col] + r * kDim + sub * 4);
        v[u] = NT_ROW ? __builtin_nontemporal_load(p) : *p;
      }
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t s = row0 + u * 16 + grp;
    if (s < total) {
      f32x4* q = reinterpret_cast<f32x4*>(a.out + s * kDim + sub * 4);
      if (NT_OUT) {
        __builtin_nontemporal_store(v[u], q);
      } else {
        *q = v[u];
      }
    }
  }
}

// reference points: same bytes moved as a pure stream (ids + sequential rows + outputs)
__global__ __launch_bounds__(256) void stream_copy(const f32x4* in, f32x4* out, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; i < n; i += stride) out[i] = in[i];
}

// 16-byte global load with an explicit cache policy (FLAVOR bits: 1 sc0, 2 sc1, 4 nt).
// The asm loads are not tracked by the compiler: the caller waits vmcnt(0) before any use.
template <int FLAVOR>
__device__ inline f32x4 load16(const void* p) {
  f32x4 v;
  if (FLAVOR == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  if (FLAVOR == 1) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
  if (FLAVOR == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  if (FLAVOR == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  if (FLAVOR == 4) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
  if (FLAVOR == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 nt" : "=v"(v) : "v"(p) : "memory");
  if (FLAVOR == 6) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
  if (FLAVOR == 7) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <int U, int FLAVOR>
__global__ __launch_bounds__(256) void gather_flavor(const float* table, const uint32_t* rowidx,
                                                     int64_t n, float* sink) {
  const int lane = threadIdx.x & 63, sub = lane & 3, grp = lane >> 2;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (16 * U);
  f32x4 v[U];
  uint32_t r[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t s = row0 + u * 16 + grp;
    r[u] = s < n ? rowidx[s] : 0;
  }
#pragma unroll
  for (int u = 0; u < U; ++u) v[u] = load16<FLAVOR>(table + (uint64_t)r[u] * 16 + sub * 4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  f32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < U; ++u) acc += v[u];
  if (acc.x == 12345.678f) sink[0] = acc.y;
}

// gather only (no id math, no output): rows named by a precomputed int32 index
template <int U, int RB>
__global__ __launch_bounds__(256) void gather_only(const float* table, const uint32_t* rowidx,
                                                   int64_t n, float* sink) {
  // RB = row bytes (64 / 128 / 256): RB/16 lanes per row
  constexpr int LPR = RB / 16;
  const int lane = threadIdx.x & 63, sub = lane % LPR, grp = lane / LPR;
  constexpr int RPI = 64 / LPR;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (RPI * U);
  f32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t s = row0 + u * RPI + grp;
    if (s < n) {
      const uint64_t r = rowidx[s];
      acc += *reinterpret_cast<const f32x4*>(table + r * (RB / 4) + sub * 4);
    }
  }
  if (acc.x == 12345.678f) sink[0] = acc.y;  // keep the loads alive
}


// granularity probe: the same 64-byte rows fetched with SCALAR loads (s_load_dwordx16 through the
// scalar data cache, 64-byte lines) instead of vector loads (vector L1: 128-byte lines)
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
template <int U>
__global__ __launch_bounds__(256) void gather_scalar(const float* table, const uint32_t* rowidx,
                                                     int64_t n, float* sink) {
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int64_t base = wave * 64;
  const uint32_t myr = base + lane < n ? rowidx[base + lane] : 0;
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 64; k += U) {
    u32x16 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)myr, k + u);
      const float* p = table + (uint64_t)r * 16;
      asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(v[u]) : "s"(p) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u][0] ^ v[u][7] ^ v[u][15];
  }
  if (acc == 0x12345678u) sink[0] = (float)acc;
}

static bool quick = false;  // --quick: few passes of every probe (for rocprofv3 --pmc)

struct Ctx {
  Args a;
  std::vector<int64_t*> id_batches;
  hipEvent_t e0, e1;
};

template <typename F>
float time_us(Ctx& c, int iters, F launch) {
  if (quick) iters = 2;
  for (int i = 0; i < (quick ? 1 : 3); ++i) launch(i);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(c.e0, 0));
  for (int i = 0; i < iters; ++i) launch(i + 3);
  CK(hipEventRecord(c.e1, 0));
  CK(hipEventSynchronize(c.e1));
  float ms;
  CK(hipEventElapsedTime(&ms, c.e0, c.e1));
  return ms * 1000.f / iters;
}

template <int U, int BLOCK, bool NT_ROW, bool NT_OUT, int MODE>
void run_variant(Ctx& c, const char* name) {
  const int64_t total = (int64_t)kCols * c.a.B;
  const int64_t rows_per_block = (BLOCK / 64) * 16 * U;
  const unsigned grid = (unsigned)((total + rows_per_block - 1) / rows_per_block);
  float us = time_us(c, 30, [&](int i) {
    Args a = c.a;
    a.ids = c.id_batches[i % c.id_batches.size()];
    hipLaunchKernelGGL((gather_variant<U, BLOCK, NT_ROW, NT_OUT, MODE>), dim3(grid), dim3(BLOCK),
                       0, 0, a);
  });
  const double gbs = total * 136.0 / us / 1e3;
  printf("%-44s B=%7d  %8.2f us  %8.1f GB/s  %5.1f%% of 8TB/s  %8.1f Mlookups/s\n", name, c.a.B,
         us, gbs, gbs / 80.0, total / us);
  fflush(stdout);
}

  // --quick: one pass of every probe (for rocprofv3 --pmc runs)

int main(int argc, char** argv) {
  quick = argc > 1 && !strcmp(argv[1], "--quick");
  const uint64_t rows = 1000000;
  Ctx c;
  CK(hipEventCreate(&c.e0));
  CK(hipEventCreate(&c.e1));
  // tables: one allocation, 26 x 1M x 16 fp32
  float* tab;
  CK(hipMalloc(&tab, (size_t)kCols * rows * kDim * 4));
  CK(hipMemset(tab, 0x3c, (size_t)kCols * rows * kDim * 4));
  for (int k = 0; k < kCols; ++k) c.a.table[k] = tab + (size_t)k * rows * kDim;
  c.a.rows = rows;
  {  // magic for 1e6 (same derivation as common.h make_fastdiv)
    uint32_t k = 63u - (uint32_t)__builtin_clzll(rows);
    unsigned __int128 num = (unsigned __int128)1 << (64 + k);
    uint64_t m = (uint64_t)(num / rows), rem = (uint64_t)(num % rows);
    uint64_t m2 = m * 2, tr = rem * 2;
    if (tr >= rows || tr < rem) m2 += 1;
    c.a.magic = m2 + 1;
    c.a.shift = k;
  }
  const int maxB = 262144;
  const int n_batches = 8;
  std::vector<int64_t> h((size_t)kCols * maxB);
  uint64_t s = 88172645463325252ull;
  for (int b = 0; b < n_batches; ++b) {
    for (auto& v : h) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      v = (int64_t)(s & ((1ull << 40) - 1));
    }
    int64_t* d;
    CK(hipMalloc(&d, h.size() * 8));
    CK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    c.id_batches.push_back(d);
  }
  float* out;
  CK(hipMalloc(&out, (size_t)kCols * maxB * kDim * 4));
  c.a.out = out;

  // --- stream reference: copy of the same number of bytes a step moves (232 MB at B=65536)
  {
    const int64_t n16 = (int64_t)kCols * maxB * 4;  // f32x4 elements of the output = 436 MB
    float us = time_us(c, 20, [&](int) {
      hipLaunchKernelGGL(stream_copy, dim3(2048), dim3(256), 0, 0,
                         reinterpret_cast<const f32x4*>(tab), reinterpret_cast<f32x4*>(out), n16);
    });
    printf("stream copy %lld MB read + write: %.2f us, %.1f GB/s (read+write)\n",
           (long long)(n16 * 16 >> 20), us, 2.0 * n16 * 16 / us / 1e3);
  }
  // --- random row gather only (no output), by row size: HBM/L2 granularity probe
  {
    const int64_t n = 4 * 1024 * 1024;
    std::vector<uint32_t> hidx(n);
    uint32_t* didx;
    float* sink;
    CK(hipMalloc(&didx, n * 4));
    CK(hipMalloc(&sink, 64));
    for (int rb : {64, 128, 256}) {
      const uint64_t nrows = (uint64_t)kCols * rows * 64 / rb;
      for (auto& v : hidx) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        v = (uint32_t)(s % nrows);
      }
      CK(hipMemcpy(didx, hidx.data(), n * 4, hipMemcpyHostToDevice));
      float us = 0;
      auto go = [&](auto kern, int rpi_u) {
        const unsigned grid = (unsigned)((n + 4 * rpi_u - 1) / (4 * rpi_u));
        us = time_us(c, 10, [&](int) {
          hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, tab, didx, n, sink);
        });
      };
      if (rb == 64) go(gather_only<4, 64>, 16 * 4);
      if (rb == 128) go(gather_only<4, 128>, 8 * 4);
      if (rb == 256) go(gather_only<4, 256>, 4 * 4);
      printf("gather-only %3d-B rows x %lld: %.2f us, %.1f GB/s useful\n", rb, (long long)n, us,
             (double)n * rb / us / 1e3);
      if (rb == 64) {
        const unsigned grid = (unsigned)((n + 255) / 256);
        auto fl = [&](auto kern, const char* nm) {
          float t = time_us(c, quick ? 1 : 10, [&](int) {
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, tab, didx, n, sink);
          });
          printf("gather-only  64-B rows, policy %-10s: %.2f us, %.1f GB/s useful\n", nm, t,
                 (double)n * 64 / t / 1e3);
        };
        {
          const unsigned sgrid = (unsigned)((n + 255) / 256);
          float t = time_us(c, quick ? 1 : 10, [&](int) {
            hipLaunchKernelGGL(gather_scalar<4>, dim3(sgrid), dim3(256), 0, 0, tab, didx, n, sink);
          });
          printf("gather-only  64-B rows, scalar loads (s_load_dwordx16): %.2f us, %.1f GB/s useful\n",
                 t, (double)n * 64 / t / 1e3);
        }
        fl(gather_flavor<4, 0>, "plain");
        fl(gather_flavor<4, 1>, "sc0");
        fl(gather_flavor<4, 2>, "sc1");
        fl(gather_flavor<4, 3>, "sc0 sc1");
        fl(gather_flavor<4, 4>, "nt");
        fl(gather_flavor<4, 5>, "sc0 nt");
        fl(gather_flavor<4, 6>, "sc1 nt");
        fl(gather_flavor<4, 7>, "sc0 sc1 nt");
      }
    }
  }
  if (quick) {
    c.a.B = 65536;
    run_variant<4, 256, false, true, 0>(c, "U4 b256 shfl  row:ld  out:nt  (library)");
    return 0;
  }

  // 26 separately allocated tables (as a framework allocator would hand them out)
  std::vector<float*> sep(kCols);
  for (int k = 0; k < kCols; ++k) {
    CK(hipMalloc(&sep[k], rows * kDim * 4));
    CK(hipMemset(sep[k], 0x3c, rows * kDim * 4));
  }
  for (int B : {65536, 262144}) {
    c.a.B = B;
    for (int separate = 0; separate < 2; ++separate) {
      // the shipped kernel through the C ABI, on the same buffers
      std::vector<hbk_lookup_column_t> cols(kCols);
      float us = time_us(c, 30, [&](int i) {
        const int64_t* ids = c.id_batches[i % c.id_batches.size()];
        for (int k = 0; k < kCols; ++k) {
          hbk_lookup_column_t& h = cols[k];
          h.table = separate ? sep[k] : c.a.table[k];
          h.rows = (int64_t)rows;
          h.dim = kDim;
          h.ids_dtype = HBK_INT64;
          h.ids = ids + (size_t)k * B;
          h.n_ids = B;
          h.row_splits = nullptr;
          h.n_segments = B;
          h.bucket = (int64_t)rows;
          h.divisor = 1;
          h.combiner = HBK_COMBINER_SUM;
          h.out = c.a.out + (size_t)k * B * kDim;
        }
        if (hbk_group_lookup_fwd(kCols, cols.data(), nullptr) != 0) {
          fprintf(stderr, "hbk_group_lookup_fwd: %s\n", hbk_last_error());
          exit(1);
        }
      });
      const double total = (double)kCols * B;
      printf("%-44s B=%7d  %8.2f us  %8.1f GB/s  %5.1f%% of 8TB/s  %8.1f Mlookups/s\n",
             separate ? "libhbk_core hbk_group_lookup_fwd (26 allocs)"
                      : "libhbk_core hbk_group_lookup_fwd (1 alloc)",
             B, us, total * 136.0 / us / 1e3, total * 136.0 / us / 1e3 / 80.0, total / us);
    }
    run_variant<4, 256, false, true, 0>(c, "U4 b256 shfl  row:ld  out:nt  (library)");
    run_variant<4, 256, true, true, 0>(c, "U4 b256 shfl  row:nt  out:nt");
    run_variant<4, 256, false, false, 0>(c, "U4 b256 shfl  row:ld  out:st");
    run_variant<2, 256, false, true, 0>(c, "U2 b256 shfl  row:ld  out:nt");
    run_variant<1, 256, false, true, 0>(c, "U1 b256 shfl  row:ld  out:nt");
    run_variant<4, 64, false, true, 0>(c, "U4 b64  shfl  row:ld  out:nt");
    run_variant<4, 128, false, true, 0>(c, "U4 b128 shfl  row:ld  out:nt");
    run_variant<4, 512, false, true, 0>(c, "U4 b512 shfl  row:ld  out:nt");
    run_variant<4, 1024, false, true, 0>(c, "U4 b1024 shfl row:ld  out:nt");
    run_variant<4, 256, false, true, 1>(c, "U4 b256 direct row:ld  out:nt");
    run_variant<8, 256, false, true, 1>(c, "U8 b256 direct row:ld  out:nt");
    run_variant<16, 256, false, true, 1>(c, "U16 b256 direct row:ld out:nt");
    run_variant<2, 256, false, true, 1>(c, "U2 b256 direct row:ld  out:nt");
    run_variant<8, 256, true, true, 1>(c, "U8 b256 direct row:nt  out:nt");
    run_variant<2, 64, false, true, 0>(c, "U2 b64  shfl  row:ld  out:nt");
    run_variant<2, 128, false, true, 0>(c, "U2 b128 shfl  row:ld  out:nt");
    run_variant<1, 64, false, true, 0>(c, "U1 b64  shfl  row:ld  out:nt");
    run_variant<3, 256, false, true, 1>(c, "U3 b256 direct row:ld  out:nt");
  }
  return 0;
}
