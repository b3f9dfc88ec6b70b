// Round-5 probe behind BASELINE.md 2 / DESIGN.md 4.1: does ANY load path of gfx950 fetch a random
// 64-byte table row with a 64-byte fabric request instead of a 128-byte one?
//
// Same access pattern for every variant: n = 26 x 65536 random 64-byte rows out of a 1.66 GB table
// (config 2's tables), nothing written.  Every variant is its own kernel name, so one
//   rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum
//             TCC_EA0_RDREQ_128B_sum -- tools/bin/fetch_probe --quick
// pass gives the request mix per variant; without rocprofv3 the binary prints the times.
//
//   vec4         global_load_dwordx4, 4 lanes per row, 2 rows in flight per lane  (the shipped path)
//   vec4_u8      the same with 8 rows in flight per lane
//   vec4_nt      non-temporal loads
//   vec1         global_load_dword, 16 lanes per row;  vec1_sc: the same, system scope (sc0 sc1)
//   half_row     only the first 32 bytes of every row are read (what does the fabric fetch?)
//   sload_x16    s_load_dwordx16, one row per scalar load, 4 rows in flight per wave
//   sload_x8     2 x s_load_dwordx8 per row, 8 loads (4 rows) in flight per wave
//   lds_dma      global_load_lds_dwordx4 (LDS-DMA), 4 pieces of 16 rows in flight per wave
//   buf_lds      buffer_load_dwordx4 ... lds, the same through a buffer resource
//   memory kinds (vec4 kernel): hipMalloc | hipExtMallocWithFlags fine-grained | uncached |
//                hipMallocManaged (prefetched to the device, coarse-grain advice unset)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

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
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

enum { kPlain = 0, kNt = 1, kSc = 2 };

template <int MODE>
__device__ inline f32x4 load16(const float* p) {
  if (MODE == kNt) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return *reinterpret_cast<const f32x4*>(p);
}

// LPR lanes per row, each CB bytes (LPR * CB = bytes read of the 64-byte row), U rows in flight
template <int LPR, int U, int MODE>
__device__ inline void vec_body(const float* table, const uint32_t* rowidx, int64_t n, float* sink) {
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
      if (LPR == 16 && MODE == kSc) {
        // system-scope load (global_load_dword sc0 sc1), through the compiler's own atomics
        v[u].x = __hip_atomic_load(table + r * 16 + sub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      } else if (LPR == 16) {
        v[u].x = table[r * 16 + sub];
      } else {
        v[u] = load16<MODE>(table + r * 16 + sub * 4);   // LPR 4: whole row, LPR 2: first half
      }
    }
  }
  f32x4 acc = v[0];
#pragma unroll
  for (int u = 1; u < U; ++u) acc += v[u];
  if (acc.x == 12345.678f) sink[0] = acc.y;
}

#define VEC_KERNEL(NAME, LPR, U, MODE)                                                        \
  __global__ __launch_bounds__(256) void NAME(const float* table, const uint32_t* rowidx,     \
                                              int64_t n, float* sink) {                       \
    vec_body<LPR, U, MODE>(table, rowidx, n, sink);                                            \
  }
VEC_KERNEL(fetch_vec4, 4, 2, kPlain)
VEC_KERNEL(fetch_vec4_u8, 4, 8, kPlain)
VEC_KERNEL(fetch_vec4_nt, 4, 2, kNt)
VEC_KERNEL(fetch_vec1_sc, 16, 2, kSc)
VEC_KERNEL(fetch_vec1, 16, 2, kPlain)
VEC_KERNEL(fetch_half_row, 2, 2, kPlain)
VEC_KERNEL(fetch_vec4_finegrained, 4, 2, kPlain)
VEC_KERNEL(fetch_vec4_uncached, 4, 2, kPlain)
VEC_KERNEL(fetch_vec4_managed, 4, 2, kPlain)

// ---- scalar loads: a wave walks its share of the rows, 4 rows per round ------------------
__global__ __launch_bounds__(256) void fetch_sload_x16(const char* table, const uint32_t* rowidx,
                                                       int64_t n, int rows_per_wave, int* sink) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  int64_t s = (int64_t)wave * rows_per_wave;
  int64_t end = s + rows_per_wave;
  if (end > n) end = n;
  int acc = 0;
  // the NEXT round's row numbers are requested before this round's rows (one latency per round)
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 nxt = s + 4 <= end ? *reinterpret_cast<const u32x4*>(rowidx + s) : u32x4{0, 0, 0, 0};
  for (; s + 4 <= end; s += 4) {
    const u32x4 cur = nxt;
    if (s + 8 <= end) nxt = *reinterpret_cast<const u32x4*>(rowidx + s + 4);
    const uint32_t r0 = __builtin_amdgcn_readfirstlane(cur.x);
    const uint32_t r1 = __builtin_amdgcn_readfirstlane(cur.y);
    const uint32_t r2 = __builtin_amdgcn_readfirstlane(cur.z);
    const uint32_t r3 = __builtin_amdgcn_readfirstlane(cur.w);
    const char* p0 = table + (uint64_t)r0 * 64;
    const char* p1 = table + (uint64_t)r1 * 64;
    const char* p2 = table + (uint64_t)r2 * 64;
    const char* p3 = table + (uint64_t)r3 * 64;
    i32x16 a, b, c, d;
    asm volatile(
        "s_load_dwordx16 %0, %4, 0x0\n s_load_dwordx16 %1, %5, 0x0\n"
        "s_load_dwordx16 %2, %6, 0x0\n s_load_dwordx16 %3, %7, 0x0\n s_waitcnt lgkmcnt(0)"
        : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d)
        : "s"(p0), "s"(p1), "s"(p2), "s"(p3)
        : "memory");
    acc += a[0] + b[5] + c[10] + d[15];
  }
  if (acc == 12345) sink[0] = acc;
}
__global__ __launch_bounds__(256) void fetch_sload_x8(const char* table, const uint32_t* rowidx,
                                                      int64_t n, int rows_per_wave, int* sink) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  int64_t s = (int64_t)wave * rows_per_wave;
  int64_t end = s + rows_per_wave;
  if (end > n) end = n;
  int acc = 0;
  // the NEXT round's row numbers are requested before this round's rows (one latency per round)
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 nxt = s + 4 <= end ? *reinterpret_cast<const u32x4*>(rowidx + s) : u32x4{0, 0, 0, 0};
  for (; s + 4 <= end; s += 4) {
    const u32x4 cur = nxt;
    if (s + 8 <= end) nxt = *reinterpret_cast<const u32x4*>(rowidx + s + 4);
    const uint32_t r0 = __builtin_amdgcn_readfirstlane(cur.x);
    const uint32_t r1 = __builtin_amdgcn_readfirstlane(cur.y);
    const uint32_t r2 = __builtin_amdgcn_readfirstlane(cur.z);
    const uint32_t r3 = __builtin_amdgcn_readfirstlane(cur.w);
    const char* p0 = table + (uint64_t)r0 * 64;
    const char* p1 = table + (uint64_t)r1 * 64;
    const char* p2 = table + (uint64_t)r2 * 64;
    const char* p3 = table + (uint64_t)r3 * 64;
    i32x8 a0, a1, b0, b1, c0, c1, d0, d1;
    asm volatile(
        "s_load_dwordx8 %0, %8, 0x0\n s_load_dwordx8 %1, %8, 0x20\n"
        "s_load_dwordx8 %2, %9, 0x0\n s_load_dwordx8 %3, %9, 0x20\n"
        "s_load_dwordx8 %4, %10, 0x0\n s_load_dwordx8 %5, %10, 0x20\n"
        "s_load_dwordx8 %6, %11, 0x0\n s_load_dwordx8 %7, %11, 0x20\n s_waitcnt lgkmcnt(0)"
        : "=&s"(a0), "=&s"(a1), "=&s"(b0), "=&s"(b1), "=&s"(c0), "=&s"(c1), "=&s"(d0), "=&s"(d1)
        : "s"(p0), "s"(p1), "s"(p2), "s"(p3)
        : "memory");
    acc += a0[0] + a1[7] + b0[1] + b1[6] + c0[2] + c1[5] + d0[3] + d1[4];
  }
  if (acc == 12345) sink[0] = acc;
}

// ---- hybrid: a wave fetches 32 rows with vector loads AND KS rows with scalar loads per round -----
// (does the memory side move more rows per second when part of them are 64-byte requests?)
template <int KS>
__device__ inline void hybrid_body(const char* table, const uint32_t* rowidx, int64_t n,
                                   int rows_per_wave, float* sink) {
  constexpr int PER = 32 + KS;
  const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int lane = threadIdx.x & 63, sub = lane & 3, grp = lane >> 2;
  int64_t s = (int64_t)wave * rows_per_wave;
  int64_t end = s + rows_per_wave;
  if (end > n) end = n;
  f32x4 acc = {0, 0, 0, 0};
  int sacc = 0;
  for (; s + PER <= end; s += PER) {
    // one row number per lane (PER <= 64), coalesced
    const uint32_t mine = lane < PER ? rowidx[s + lane] : 0u;
    const uint32_t r0 = __builtin_amdgcn_readlane(mine, 32);
    const uint32_t r1 = __builtin_amdgcn_readlane(mine, 33);
    const char* p0 = table + (uint64_t)r0 * 64;
    const char* p1 = table + (uint64_t)r1 * 64;
    i32x16 a, b, c, d;
    if (KS == 4) {
      const uint32_t r2 = __builtin_amdgcn_readlane(mine, 34);
      const uint32_t r3 = __builtin_amdgcn_readlane(mine, 35);
      const char* p2 = table + (uint64_t)r2 * 64;
      const char* p3 = table + (uint64_t)r3 * 64;
      asm volatile("s_load_dwordx16 %0, %4, 0x0\n s_load_dwordx16 %1, %5, 0x0\n"
                   "s_load_dwordx16 %2, %6, 0x0\n s_load_dwordx16 %3, %7, 0x0"
                   : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d)
                   : "s"(p0), "s"(p1), "s"(p2), "s"(p3)
                   : "memory");
    } else {
      asm volatile("s_load_dwordx16 %0, %2, 0x0\n s_load_dwordx16 %1, %3, 0x0"
                   : "=&s"(a), "=&s"(b)
                   : "s"(p0), "s"(p1)
                   : "memory");
    }
    // the 32 vector rows: 2 per lane group, in flight beside the scalar loads
    const uint32_t v0 = __shfl(mine, grp, 64), v1 = __shfl(mine, 16 + grp, 64);
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(table + (uint64_t)v0 * 64 + sub * 16);
    const f32x4 x1 = *reinterpret_cast<const f32x4*>(table + (uint64_t)v1 * 64 + sub * 16);
    acc += x0 + x1;
    if (KS == 4) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b), "+s"(c), "+s"(d)::"memory");
      sacc += a[0] + b[5] + c[10] + d[15];
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b)::"memory");
      sacc += a[0] + b[5];
    }
  }
  if (acc.x == 12345.678f || sacc == 12345) sink[0] = acc.y;
}
__global__ __launch_bounds__(256) void fetch_hybrid_s4(const char* table, const uint32_t* rowidx,
                                                       int64_t n, int rows_per_wave, float* sink) {
  hybrid_body<4>(table, rowidx, n, rows_per_wave, sink);
}
__global__ __launch_bounds__(256) void fetch_hybrid_s2(const char* table, const uint32_t* rowidx,
                                                       int64_t n, int rows_per_wave, float* sink) {
  hybrid_body<2>(table, rowidx, n, rows_per_wave, sink);
}
// the same loop without the scalar rows (what the loop structure itself costs)
__global__ __launch_bounds__(256) void fetch_hybrid_s0(const char* table, const uint32_t* rowidx,
                                                       int64_t n, int rows_per_wave, float* sink) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int lane = threadIdx.x & 63, sub = lane & 3, grp = lane >> 2;
  int64_t s = (int64_t)wave * rows_per_wave;
  int64_t end = s + rows_per_wave;
  if (end > n) end = n;
  f32x4 acc = {0, 0, 0, 0};
  for (; s + 32 <= end; s += 32) {
    const uint32_t mine = lane < 32 ? rowidx[s + lane] : 0u;
    const uint32_t v0 = __shfl(mine, grp, 64), v1 = __shfl(mine, 16 + grp, 64);
    acc += *reinterpret_cast<const f32x4*>(table + (uint64_t)v0 * 64 + sub * 16) +
           *reinterpret_cast<const f32x4*>(table + (uint64_t)v1 * 64 + sub * 16);
  }
  if (acc.x == 12345.678f) sink[0] = acc.y;
}

// ---- LDS-DMA: every lane names 16 bytes of global memory, the wave's 1 KB lands in LDS ------
constexpr int kDmaU = 4;
template <bool BUFFER>
__device__ inline void dma_body(const float* table, uint32_t table_bytes, const uint32_t* rowidx,
                                int64_t n, float* sink) {
  __shared__ __attribute__((aligned(16))) char buf[4][kDmaU][1024];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, sub = lane & 3, grp = lane >> 2;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + w) * (16 * kDmaU);
  uint32_t r[kDmaU];
#pragma unroll
  for (int u = 0; u < kDmaU; ++u) {
    const int64_t s = row0 + u * 16 + grp;
    r[u] = rowidx[s < n ? s : n - 1];
  }
  // every row number is in before the first DMA leaves (the compiler's waits count the DMAs too:
  // a row number asked for late would wait for the pieces issued before it)
#pragma unroll
  for (int u = 0; u < kDmaU; ++u) asm volatile("" : "+v"(r[u]));
  if (BUFFER) {
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(table), 0, (int)table_bytes,
                                                  0x00020000);
#pragma unroll
    for (int u = 0; u < kDmaU; ++u) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LDS_AS void*)&buf[w][u][0], 16,
                                               (int)(r[u] * 64u + sub * 16u), 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int u = 0; u < kDmaU; ++u) {
      const float* p = table + (uint64_t)r[u] * 16 + sub * 4;
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)p, (LDS_AS void*)&buf[w][u][0], 16, 0,
                                       0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float acc = 0.f;
#pragma unroll
  for (int u = 0; u < kDmaU; ++u) acc += *reinterpret_cast<const float*>(&buf[w][u][lane * 16]);
  if (acc == 12345.678f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void fetch_lds_dma(const float* table, uint32_t table_bytes,
                                                     const uint32_t* rowidx, int64_t n, float* sink) {
  dma_body<false>(table, table_bytes, rowidx, n, sink);
}
__global__ __launch_bounds__(256) void fetch_buf_lds(const float* table, uint32_t table_bytes,
                                                     const uint32_t* rowidx, int64_t n, float* sink) {
  dma_body<true>(table, table_bytes, rowidx, n, sink);
}

static bool quick = false;
static hipEvent_t e0, e1;

template <typename F>
float time_us(int iters, F launch) {
  if (quick) iters = 3;
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
  const size_t table_bytes = (size_t)26 * 1000000 * 64;
  const uint64_t nrows = table_bytes / 64;
  const int64_t n = (int64_t)26 * 65536;
  float *tab, *sink;
  CK(hipMalloc(&tab, table_bytes));
  CK(hipMemset(tab, 0x3c, table_bytes));
  CK(hipMalloc(&sink, 4096));
  const int kBatches = 8;   // other row numbers every launch: 1.66 GB >> the 256 MB Infinity Cache
  uint64_t s = 88172645463325252ull;
  auto rnd = [&] {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return s;
  };
  std::vector<uint32_t*> idx(kBatches);
  {
    std::vector<uint32_t> h(n);
    for (int b = 0; b < kBatches; ++b) {
      for (auto& v : h) v = (uint32_t)(rnd() % nrows);
      CK(hipMalloc(&idx[b], n * 4));
      CK(hipMemcpy(idx[b], h.data(), n * 4, hipMemcpyHostToDevice));
    }
  }
  auto report = [&](const char* name, float us) {
    printf("%-24s %8.2f us  %6.2f G rows/s  %7.1f GB/s of 64-byte rows\n", name, us, n / us / 1e3,
           (double)n * 64 / us / 1e3);
    fflush(stdout);
  };
  auto run_vec = [&](const char* name, auto kern, int lpr, int u, const float* table) {
    const int rpi = 64 / lpr;
    const unsigned grid = (unsigned)((n + 4 * rpi * u - 1) / (4 * rpi * u));
    report(name, time_us(20, [&](int i) {
             hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, table, idx[i % kBatches], n, sink);
           }));
  };
  run_vec("vec4", fetch_vec4, 4, 2, tab);
  run_vec("vec4_u8", fetch_vec4_u8, 4, 8, tab);
  run_vec("vec4_nt", fetch_vec4_nt, 4, 2, tab);
  run_vec("vec1", fetch_vec1, 16, 2, tab);
  run_vec("vec1_sc (sc0 sc1)", fetch_vec1_sc, 16, 2, tab);
  run_vec("half_row (32 of 64 B)", fetch_half_row, 2, 2, tab);
  {
    // persistent waves: 256 CUs x 4 SIMDs x 8 waves
    for (unsigned waves : {8192u, 16384u}) {
      const int rpw = (int)(((n + waves - 1) / waves + 3) / 4 * 4);
      char name[64];
      snprintf(name, sizeof name, "sload_x16 (%u waves)", waves);
      report(name, time_us(10, [&](int i) {
               hipLaunchKernelGGL(fetch_sload_x16, dim3(waves / 4), dim3(256), 0, 0,
                                  reinterpret_cast<const char*>(tab), idx[i % kBatches], n, rpw,
                                  reinterpret_cast<int*>(sink));
             }));
      snprintf(name, sizeof name, "sload_x8 (%u waves)", waves);
      report(name, time_us(10, [&](int i) {
               hipLaunchKernelGGL(fetch_sload_x8, dim3(waves / 4), dim3(256), 0, 0,
                                  reinterpret_cast<const char*>(tab), idx[i % kBatches], n, rpw,
                                  reinterpret_cast<int*>(sink));
             }));
      if (quick) break;
    }
  }
  for (unsigned waves : {8192u, 16384u, 32768u}) {
    char name[64];
    auto run_h = [&](const char* what, auto kern, int per) {
      const int rpw = (int)(((n + waves - 1) / waves + per - 1) / per * per);
      snprintf(name, sizeof name, "%s (%u waves)", what, waves);
      report(name, time_us(10, [&](int i) {
               hipLaunchKernelGGL(kern, dim3(waves / 4), dim3(256), 0, 0,
                                  reinterpret_cast<const char*>(tab), idx[i % kBatches], n, rpw, sink);
             }));
    };
    run_h("hybrid 32v+0s", fetch_hybrid_s0, 32);
    run_h("hybrid 32v+2s", fetch_hybrid_s2, 34);
    run_h("hybrid 32v+4s", fetch_hybrid_s4, 36);
    if (quick) break;
  }
  {
    const unsigned grid = (unsigned)((n + 4 * 16 * kDmaU - 1) / (4 * 16 * kDmaU));
    report("lds_dma", time_us(20, [&](int i) {
             hipLaunchKernelGGL(fetch_lds_dma, dim3(grid), dim3(256), 0, 0, tab,
                                (uint32_t)table_bytes, idx[i % kBatches], n, sink);
           }));
    report("buf_lds", time_us(20, [&](int i) {
             hipLaunchKernelGGL(fetch_buf_lds, dim3(grid), dim3(256), 0, 0, tab,
                                (uint32_t)table_bytes, idx[i % kBatches], n, sink);
           }));
  }
  CK(hipFree(tab));
  // ---- other memory kinds under the shipped access path
  {
    float* t = nullptr;
    if (hipExtMallocWithFlags(reinterpret_cast<void**>(&t), table_bytes, hipDeviceMallocFinegrained) ==
        hipSuccess) {
      CK(hipMemset(t, 0x3c, table_bytes));
      run_vec("vec4 fine-grained", fetch_vec4_finegrained, 4, 2, t);
      CK(hipFree(t));
    } else {
      (void)hipGetLastError();
      printf("hipDeviceMallocFinegrained: not available\n");
    }
    t = nullptr;
    if (hipExtMallocWithFlags(reinterpret_cast<void**>(&t), table_bytes, hipDeviceMallocUncached) ==
        hipSuccess) {
      CK(hipMemset(t, 0x3c, table_bytes));
      run_vec("vec4 uncached", fetch_vec4_uncached, 4, 2, t);
      CK(hipFree(t));
    } else {
      (void)hipGetLastError();
      printf("hipDeviceMallocUncached: not available\n");
    }
    t = nullptr;
    if (hipMallocManaged(reinterpret_cast<void**>(&t), table_bytes) == hipSuccess) {
      int dev = 0;
      CK(hipGetDevice(&dev));
      (void)hipMemAdvise(t, table_bytes, hipMemAdviseUnsetCoarseGrain, dev);
      (void)hipMemPrefetchAsync(t, table_bytes, dev, 0);
      (void)hipGetLastError();
      CK(hipMemset(t, 0x3c, table_bytes));
      CK(hipDeviceSynchronize());
      run_vec("vec4 managed", fetch_vec4_managed, 4, 2, t);
      CK(hipFree(t));
    } else {
      (void)hipGetLastError();
      printf("hipMallocManaged: not available\n");
    }
  }
  return 0;
}
