// Reference point for the activation-quantize kernel: plain device copy of 16 MiB -> 16 MiB with 16-byte accesses,
// in the same shapes the quantizer uses (one workgroup per 32 rows x 256 columns of fp16 vs grid-stride).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k_copy_tile(const uint4* __restrict__ in, uint4* __restrict__ out, int K16) {
  // tile: 32 rows x 256 fp16 = 32 x 32 uint4; 256 threads x 4
  const int nchunk = K16 / 32;
  const int rg = blockIdx.x / nchunk, c = blockIdx.x % nchunk;
#pragma unroll
  for (int s2 = 0; s2 < 4; ++s2) {
    const int s = threadIdx.x + s2 * 256, row = s >> 5, col = s & 31;
    const size_t idx = (size_t)(rg * 32 + row) * K16 + c * 32 + col;
    out[idx] = in[idx];
  }
}
__global__ __launch_bounds__(256) void k_copy_stride(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
int main() {
  const int M = 2048, K = 4096;
  const size_t n16 = (size_t)M * K * 2 / 16;
  uint4 *a, *b;
  hipMalloc(&a, n16 * 16); hipMalloc(&b, n16 * 16);
  hipMemset(a, 1, n16 * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int variant = 0; variant < 3; ++variant) {
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      for (int it = 0; it < 20; ++it) {
        if (variant == 0) k_copy_tile<<<(M / 32) * (K / 256), 256>>>(a, b, K * 2 / 16);
        if (variant == 1) k_copy_stride<<<1024, 256>>>(a, b, n16);
        if (variant == 2) k_copy_stride<<<4096, 256>>>(a, b, n16);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms / 20 < best) best = ms / 20;
    }
    printf("variant %d: %.2f us per 16+16 MiB copy = %.2f TB/s\n", variant, best * 1e3, 2.0 * n16 * 16 / best / 1e9);
  }
  return 0;
}
