// How fast can a short streaming kernel over 13.6 MB (1.7 M int64 ids) be, as a function of the
// workgroup size and of the ids per thread?  Each thread reads K ids (coalesced across the block),
// folds them, and one thread per block writes a word.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)

template <int K>
__global__ void probe(const int64_t* in, int64_t n, int64_t* out) {
  const int64_t base = (int64_t)blockIdx.x * blockDim.x * K;
  int64_t v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int64_t i = base + (int64_t)k * blockDim.x + threadIdx.x;
    v[k] = i < n ? in[i] : 0;
  }
  int64_t acc = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) acc ^= v[k] * 0x9e3779b97f4a7c15ll;
  if (acc == 0x1234567) out[blockIdx.x] = acc;
}

template <int K>
float run(const int64_t* in, int64_t n, int64_t* out, int block) {
  const unsigned grid = (unsigned)((n + (int64_t)block * K - 1) / ((int64_t)block * K));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe<K>, dim3(grid), dim3(block), 0, 0, in, n, out);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(probe<K>, dim3(grid), dim3(block), 0, 0, in, n, out);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("block %4d  ids/thread %3d  grid %6u  waves %6u : %7.2f us per launch (back to back)\n", block, K,
         grid, grid * (block / 64), ms * 1000.f / 50);
  return ms;
}

int main() {
  for (int64_t n : {(int64_t)26 * 65536, (int64_t)26 * 1048576}) {
    int64_t *in, *out;
    CK(hipMalloc(&in, n * 8));
    CK(hipMalloc(&out, 1 << 22));
    CK(hipMemset(in, 1, n * 8));
    printf("n = %lld ids (%.1f MB)\n", (long long)n, n * 8 / 1e6);
    for (int block : {64, 256, 1024}) {
      run<1>(in, n, out, block);
      run<4>(in, n, out, block);
      run<16>(in, n, out, block);
      run<64>(in, n, out, block);
    }
    hipFree(in);
    hipFree(out);
  }
  return 0;
}
