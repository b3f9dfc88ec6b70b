// Round-2 probes behind DESIGN.md 4.1 ("what bounds the dim-16 forward"): built by tools/Makefile
// into tools/bin/ceiling_probe, run on the GPU box alone and under rocprofv3 --pmc with the TCC
// request-size counters of gfx950 (TCC_EA0_RDREQ_{32B,64B,128B}_sum, TCC_HIT/MISS_sum, WRREQ).
//
//   gather   n random rows of RB bytes out of a 1.66 GB table, nothing written
//   store    n rows of RB bytes written back to back (plain / non-temporal), nothing read
//   both     the forward's shape without id arithmetic: gather n rows, store them in order
//   uncached the gather on memory from hipExtMallocWithFlags(hipDeviceMallocUncached)
//   sorted   the gather with the row numbers sorted (DRAM page locality)
// Kernel names carry the case so the PMC rows can be told apart.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <vector>

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e = (x);                                                                \
    if (e != hipSuccess) {                                                             \
      fprintf(stderr, "%s failed: %s (%s:%d)\n", #x, hipGetErrorString(e), __FILE__,   \
              __LINE__);                                                               \
      exit(1);                                                                         \
    }                                                                                  \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// RB = row bytes; RB/16 lanes per row, U rows in flight per lane group
template <int RB, int U, bool STORE, bool NT>
__device__ inline void gather_body(const float* table, const uint32_t* rowidx, int64_t n,
                                   float* out, float* sink) {
  constexpr int LPR = RB / 16;
  constexpr int RPI = 64 / LPR;
  const int lane = threadIdx.x & 63, sub = lane % LPR, grp = lane / LPR;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (RPI * U);
  f32x4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t s = row0 + u * RPI + grp;
    v[u] = f32x4{0, 0, 0, 0};
    if (s < n) {
      const uint64_t r = rowidx[s];
      v[u] = *reinterpret_cast<const f32x4*>(table + r * (RB / 4) + sub * 4);
    }
  }
  if (STORE) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t s = row0 + u * RPI + grp;
      if (s < n) {
        f32x4* q = reinterpret_cast<f32x4*>(out + s * (RB / 4) + sub * 4);
        if (NT) {
          __builtin_nontemporal_store(v[u], q);
        } else {
          *q = v[u];
        }
      }
    }
  } else {
    f32x4 acc = v[0];
#pragma unroll
    for (int u = 1; u < U; ++u) acc += v[u];
    if (acc.x == 12345.678f) sink[0] = acc.y;
  }
}

#define GATHER_KERNEL(NAME, RB, U, STORE, NT)                                                   \
  __global__ __launch_bounds__(256) void NAME(const float* table, const uint32_t* rowidx,      \
                                              int64_t n, float* out, float* sink) {            \
    gather_body<RB, U, STORE, NT>(table, rowidx, n, out, sink);                                 \
  }
GATHER_KERNEL(gather_64B, 64, 2, false, false)
GATHER_KERNEL(gather_128B, 128, 2, false, false)
GATHER_KERNEL(gather_256B, 256, 2, false, false)
GATHER_KERNEL(gather_512B, 512, 2, false, false)
GATHER_KERNEL(gather_64B_sorted, 64, 2, false, false)
GATHER_KERNEL(gather_64B_uncached, 64, 2, false, false)
GATHER_KERNEL(gather_128B_uncached, 128, 2, false, false)
GATHER_KERNEL(gather_store_64B_nt, 64, 2, true, true)
GATHER_KERNEL(gather_store_64B_plain, 64, 2, true, false)
GATHER_KERNEL(gather_store_512B_nt, 512, 2, true, true)

template <bool NT>
__device__ inline void store_body(float* out, int64_t n16) {
  // every lane writes 16 bytes, a wave 1 KB back to back, 2 per wave like the forward
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x);
  const int64_t stride = (int64_t)gridDim.x * 256;
  const f32x4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
  for (int64_t i = i0; i < n16; i += stride) {
    f32x4* q = reinterpret_cast<f32x4*>(out) + i;
    if (NT) {
      __builtin_nontemporal_store(v, q);
    } else {
      *q = v;
    }
  }
}
__global__ __launch_bounds__(256) void store_nt(float* out, int64_t n16) { store_body<true>(out, n16); }
__global__ __launch_bounds__(256) void store_plain(float* out, int64_t n16) { store_body<false>(out, n16); }
__global__ __launch_bounds__(256) void read_stream(const float* in, int64_t n16, float* sink) {
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x);
  const int64_t stride = (int64_t)gridDim.x * 256;
  f32x4 acc = {0, 0, 0, 0};
  for (int64_t i = i0; i < n16; i += stride) {
    acc += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(in) + i);
  }
  if (acc.x == 12345.678f) sink[0] = acc.y;
}

static bool quick = false;
static hipEvent_t e0, e1;

template <typename F>
float time_us(int iters, F launch) {
  if (quick) iters = 2;
  for (int i = 0; i < (quick ? 1 : 3); ++i) launch(i);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) launch(i + 3);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / iters;
}

int main(int argc, char** argv) {
  quick = argc > 1 && !strcmp(argv[1], "--quick");
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const size_t table_bytes = (size_t)26 * 1000000 * 64;   // config 2: 26 x 1M x dim16 fp32
  float *tab, *tab_uc = nullptr, *out, *sink;
  CK(hipMalloc(&tab, table_bytes));
  CK(hipMemset(tab, 0x3c, table_bytes));
  if (hipExtMallocWithFlags(reinterpret_cast<void**>(&tab_uc), table_bytes,
                            hipDeviceMallocUncached) != hipSuccess) {
    tab_uc = nullptr;
    (void)hipGetLastError();
    printf("hipDeviceMallocUncached: not available\n");
  } else {
    CK(hipMemset(tab_uc, 0x3c, table_bytes));
  }
  const size_t out_bytes = (size_t)26 * 65536 * 512;       // config 4 output: 872 MB
  CK(hipMalloc(&out, out_bytes));
  CK(hipMalloc(&sink, 64));
  const int kBatches = 8;                                  // fresh row numbers every launch
  uint64_t s = 88172645463325252ull;
  auto rnd = [&] {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return s;
  };

  for (int64_t n : {(int64_t)26 * 65536, (int64_t)4 * 1024 * 1024}) {
    printf("---- n = %lld rows per launch\n", (long long)n);
    std::vector<uint32_t*> idx(kBatches), idx_sorted(kBatches);
    std::vector<uint32_t> h(n);
    auto fill = [&](int rb, std::vector<uint32_t*>& dst, bool sorted) {
      const uint64_t nrows = table_bytes / rb;
      for (int b = 0; b < kBatches; ++b) {
        if (sorted) {
          // per "column" (26 equal slices of the launch) rows in increasing order, like ids of
          // one table sorted by row
          const int64_t per = n / 26;
          for (int c = 0; c < 26; ++c) {
            const uint64_t r0 = nrows / 26 * c, span = nrows / 26;
            const int64_t lo = c * per, hi = c == 25 ? n : lo + per;
            for (int64_t i = lo; i < hi; ++i) h[i] = (uint32_t)(r0 + rnd() % span);
            std::sort(h.begin() + lo, h.begin() + hi);
          }
        } else {
          for (auto& v : h) v = (uint32_t)(rnd() % nrows);
        }
        if (dst[b] == nullptr) CK(hipMalloc(&dst[b], n * 4));
        CK(hipMemcpy(dst[b], h.data(), n * 4, hipMemcpyHostToDevice));
      }
    };
    auto run = [&](const char* name, auto kern, int rb, const float* table,
                   std::vector<uint32_t*>& ix, bool store) {
      const int rpi = 64 / (rb / 16);
      const unsigned grid = (unsigned)((n + 4 * rpi * 2 - 1) / (4 * rpi * 2));
      const float us = time_us(20, [&](int i) {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, table, ix[i % kBatches], n, out, sink);
      });
      const double useful = (double)n * rb * (store ? 2 : 1);
      printf("%-26s %4d-B rows: %8.2f us  %6.2f G rows/s  %7.1f GB/s useful\n", name, rb, us,
             n / us / 1e3, useful / us / 1e3);
      fflush(stdout);
    };
    std::fill(idx.begin(), idx.end(), nullptr);
    std::fill(idx_sorted.begin(), idx_sorted.end(), nullptr);
    fill(64, idx, false);
    run("gather", gather_64B, 64, tab, idx, false);
    if (tab_uc) run("gather uncached", gather_64B_uncached, 64, tab_uc, idx, false);
    run("gather + nt store", gather_store_64B_nt, 64, tab, idx, true);
    run("gather + plain store", gather_store_64B_plain, 64, tab, idx, true);
    fill(64, idx_sorted, true);
    run("gather sorted rows", gather_64B_sorted, 64, tab, idx_sorted, false);
    fill(128, idx, false);
    run("gather", gather_128B, 128, tab, idx, false);
    if (tab_uc) run("gather uncached", gather_128B_uncached, 128, tab_uc, idx, false);
    fill(256, idx, false);
    run("gather", gather_256B, 256, tab, idx, false);
    fill(512, idx, false);
    run("gather", gather_512B, 512, tab, idx, false);
    if ((size_t)n * 512 <= out_bytes) run("gather + nt store", gather_store_512B_nt, 512, tab, idx, true);
    for (auto p : idx) CK(hipFree(p));
    for (auto p : idx_sorted) CK(hipFree(p));
  }
  printf("---- pure streams\n");
  for (size_t bytes : {(size_t)26 * 65536 * 64, out_bytes}) {
    const int64_t n16 = (int64_t)(bytes / 16);
    for (unsigned grid : {2048u, 8192u}) {
      float us = time_us(20, [&](int) { hipLaunchKernelGGL(store_nt, dim3(grid), dim3(256), 0, 0, out, n16); });
      printf("store nt    %7.1f MB grid %5u: %8.2f us  %7.1f GB/s\n", bytes / 1e6, grid, us, bytes / us / 1e3);
      us = time_us(20, [&](int) { hipLaunchKernelGGL(store_plain, dim3(grid), dim3(256), 0, 0, out, n16); });
      printf("store plain %7.1f MB grid %5u: %8.2f us  %7.1f GB/s\n", bytes / 1e6, grid, us, bytes / us / 1e3);
    }
  }
  {
    const int64_t n16 = (int64_t)(table_bytes / 16);
    const float us = time_us(10, [&](int) {
      hipLaunchKernelGGL(read_stream, dim3(8192), dim3(256), 0, 0, tab, n16, sink);
    });
    printf("read stream %7.1f MB grid  8192: %8.2f us  %7.1f GB/s\n", table_bytes / 1e6, us,
           table_bytes / us / 1e3);
  }
  return 0;
}
