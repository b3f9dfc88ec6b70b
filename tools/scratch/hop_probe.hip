// What a cross-stream hop costs on this chip: kernels chained on one stream vs. ping-ponged between two streams
// through events (the pattern of the sharded step's exchanges and of the backward's side-by-side launch groups).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin(int* p, int n) {
  int v = threadIdx.x;
  for (int i = 0; i < n; ++i) v = v * 1664525 + 1013904223;
  if (v == 42) *p = v;
}
int main() {
  int* d;
  CK(hipMalloc(&d, 4));
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  hipEvent_t e1, e2, t0, t1;
  CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
  CK(hipEventCreate(&t0));
  CK(hipEventCreate(&t1));
  const int iters = 200;
  for (int work : {100, 3000}) {           // ~2 us and ~20 us kernels
    for (int mode = 0; mode < 3; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {  // first rep warms up
        CK(hipEventRecord(t0, a));
        for (int i = 0; i < iters; ++i) {
          hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, a, d, work);
          if (mode == 0) {
            hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, a, d, work);
          } else if (mode == 1) {
            CK(hipEventRecord(e1, a));
            CK(hipStreamWaitEvent(b, e1, 0));
            hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, b, d, work);
            CK(hipEventRecord(e2, b));
            CK(hipStreamWaitEvent(a, e2, 0));
          } else {
            // record / wait on the SAME stream (no hop): what the event packets alone cost
            CK(hipEventRecord(e1, a));
            CK(hipStreamWaitEvent(a, e1, 0));
            hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, a, d, work);
            CK(hipEventRecord(e2, a));
            CK(hipStreamWaitEvent(a, e2, 0));
          }
        }
        CK(hipEventRecord(t1, a));
        CK(hipEventSynchronize(t1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, t0, t1));
        if (rep == 1) {
          printf("work %5d  %-34s %8.2f us per pair of kernels\n", work,
                 mode == 0 ? "one stream" : mode == 1 ? "ping-pong over two streams" : "events on the same stream",
                 ms * 1000.0f / iters);
        }
      }
    }
  }
  return 0;
}
