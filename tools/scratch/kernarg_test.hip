// Does a HIP kernel on gfx950 accept a by-value argument larger than 4 KB?
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int N> struct Big { int v[N]; };
template <int N> __global__ void k(Big<N> b, int* out) { out[0] = b.v[N - 1] + b.v[blockIdx.x]; }
template <int N> void run(int* d) {
  Big<N> b; for (int i = 0; i < N; ++i) b.v[i] = i;
  hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, b, d);
  hipError_t e = hipDeviceSynchronize(); int h = -1; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
  printf("kernarg %6zu bytes: %s, result %d (expect %d)\n", sizeof(b), hipGetErrorString(e), h, N - 1);
  (void)hipGetLastError();
}
int main() { int* d; hipMalloc(&d, 4); run<1000>(d); run<2048>(d); run<4096>(d); run<16000>(d); run<65000>(d); return 0; }
