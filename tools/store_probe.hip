// Round 5: what does a NON-TEMPORAL store of HALF a 128-byte line cost at the memory side?
// The row-sorted reduce of the backward writes 10.6 M output rows of 64 bytes (dim 16) per ragged
// launch and the counters show 15.2 M write requests where 11.9 M are expected
// (profiles/r05_rowsort_counters.txt).  Its lane groups (4 lanes x 16 bytes = one row) each own a
// contiguous range of output rows: one store instruction of a wave writes 16 rows that lie far
// apart, and the two halves of a line leave in two instructions.  This probe writes the same bytes in
// the patterns below and is run under  --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum :
//   wave_seq    a wave's 64 lanes write 1 KB contiguous per instruction (whole lines; the forward's pattern)
//   lg_seq      every lane group walks ITS OWN range: the halves of a line in consecutive instructions
//   lg_gap      the same, with the second half of every line 8 instructions after the first
//   pair_line   adjacent lane groups write the two halves of one line in ONE instruction
// each with non-temporal and with plain stores.     make -C tools bin/store_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e__ = (x);                                                             \
    if (e__ != hipSuccess) {                                                          \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e__));      \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ inline void st(f32x4* p, f32x4 v) {
  if (NT) {
    __builtin_nontemporal_store(v, p);
  } else {
    *p = v;
  }
}

// rows [0, n): row r = 4 chunks of 16 bytes.  kPer rows per lane group.
constexpr int kPer = 32;

template <bool NT>
__global__ __launch_bounds__(256) void wave_seq(f32x4* out, long n) {
  const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const long r0 = wave * 16 * kPer;   // a wave owns 16 * kPer consecutive rows
  const f32x4 v = {1.f, 2.f, 3.f, (float)lane};
  for (int i = 0; i < kPer; ++i) {
    const long r = r0 + (long)i * 16 + (lane >> 2);
    if (r < n) st<NT>(out + r * 4 + (lane & 3), v);
  }
}

template <bool NT, int GAP>
__global__ __launch_bounds__(256) void lg_seq(f32x4* out, long n) {
  const long g = ((long)blockIdx.x * 256 + threadIdx.x) >> 2;   // lane group
  const int sub = threadIdx.x & 3;
  const long r0 = g * kPer;
  const f32x4 v = {1.f, 2.f, 3.f, (float)sub};
  if (GAP == 0) {
#pragma unroll 8
    for (int i = 0; i < kPer; ++i) {
      const long r = r0 + i;
      if (r < n) st<NT>(out + r * 4 + sub, v);
    }
  } else {   // batches of 2 * GAP rows: all even rows first, then all odd rows
    for (int b = 0; b < kPer; b += 2 * GAP) {
#pragma unroll
      for (int i = 0; i < GAP; ++i) {
        const long r = r0 + b + 2 * i;
        if (r < n) st<NT>(out + r * 4 + sub, v);
      }
#pragma unroll
      for (int i = 0; i < GAP; ++i) {
        const long r = r0 + b + 2 * i + 1;
        if (r < n) st<NT>(out + r * 4 + sub, v);
      }
    }
  }
}

template <bool NT>
__global__ __launch_bounds__(256) void pair_line(f32x4* out, long n) {
  const long gp = ((long)blockIdx.x * 256 + threadIdx.x) >> 3;   // pair of lane groups
  const int sub8 = threadIdx.x & 7;
  const long r0 = gp * 2 * kPer;   // the pair owns 2 * kPer consecutive rows
  const f32x4 v = {1.f, 2.f, 3.f, (float)sub8};
#pragma unroll 8
  for (int i = 0; i < kPer; ++i) {
    const long r = r0 + 2 * i + (sub8 >> 2);
    if (r < n) st<NT>(out + r * 4 + (sub8 & 3), v);
  }
}

int main(int argc, char** argv) {
  const long n = 10600000;   // rows of 64 bytes
  f32x4* out;
  CK(hipMalloc(&out, (size_t)n * 64 * 4));   // four regions: consecutive launches write other memory
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = argc > 1 ? atoi(argv[1]) : 8;
  auto run = [&](const char* name, auto kern, long threads) {
    const unsigned grid = (unsigned)((threads + 255) / 256);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out + (size_t)(i % 4) * n * 4, n);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out + (size_t)(i % 4) * n * 4, n);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const float us = ms * 1000.f / iters;
    printf("%-22s %8.2f us  %7.1f GB/s\n", name, us, (double)n * 64 / us / 1e3);
    fflush(stdout);
  };
  const long lgs = (n + kPer - 1) / kPer;
  run("wave_seq nt", wave_seq<true>, (n + 16 * kPer - 1) / (16 * kPer) * 64);
  run("wave_seq plain", wave_seq<false>, (n + 16 * kPer - 1) / (16 * kPer) * 64);
  run("lg_seq nt", lg_seq<true, 0>, lgs * 4);
  run("lg_seq plain", lg_seq<false, 0>, lgs * 4);
  run("lg_gap nt", lg_seq<true, 8>, lgs * 4);
  run("lg_gap plain", lg_seq<false, 8>, lgs * 4);
  run("pair_line nt", pair_line<true>, (lgs + 1) / 2 * 8);
  run("pair_line plain", pair_line<false>, (lgs + 1) / 2 * 8);
  return 0;
}
