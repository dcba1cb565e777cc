// Microbenchmark: cycles per s_barrier for 4 / 8 / 16 waves per workgroup (one workgroup per CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int WORK>
__global__ void k(unsigned long long* out, int iters, float* sink) {
  unsigned long long t0, t1;
  float x = threadIdx.x;
  asm volatile("s_barrier" ::: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < WORK; ++j) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
    asm volatile("s_barrier" ::: "memory");
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
  if (x == 1.2345f) sink[0] = x;
}
template <int WORK>
void run(int waves, unsigned long long* out, float* sink) {
  const int iters = 1000;
  k<WORK><<<256, waves * 64>>>(out, iters, sink); hipDeviceSynchronize();
  k<WORK><<<256, waves * 64>>>(out, iters, sink); hipDeviceSynchronize();
  unsigned long long h[4096]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) s += h[b * 16 + w];
  printf("waves %2d, %3d VALU between barriers: %.1f cycles per iteration\n", waves, WORK, s / (256.0 * waves * iters));
}
int main() {
  unsigned long long* out; float* sink; hipMalloc(&out, 4096 * 8); hipMalloc(&sink, 4);
  for (int w : {1, 4, 8, 16}) { run<0>(w, out, sink); run<32>(w, out, sink); }
  return 0;
}
