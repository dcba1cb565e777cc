// What a streaming kernel can get from HBM on this device when the working set exceeds the 256 MB Infinity Cache:
// grid-stride copy (read + write) and read-only sum over buffers of 32 MiB .. 1 GiB, 16-byte accesses.  Reference
// point for the per-token quantizer and the side GEMM of the INT configurations (168 MB activations: NOTEBOOK.md §7).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ in, unsigned* __restrict__ out, size_t n) {
  unsigned s = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const uint4 v = in[i];
    s += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (s == 0x12345678u) out[0] = s;
}
int main() {
  const size_t maxb = (size_t)1 << 30;
  uint4 *a, *b;
  unsigned* o;
  hipMalloc(&a, maxb), hipMalloc(&b, maxb), hipMalloc(&o, 4);
  hipMemset(a, 1, maxb);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (size_t bytes = (size_t)32 << 20; bytes <= maxb; bytes *= 2) {
    if (bytes == ((size_t)128 << 20)) bytes = (size_t)168 << 20;  // the C4 activation
    const size_t n = bytes / 16;
    float best[2] = {1e9f, 1e9f};
    for (int kind = 0; kind < 2; ++kind)
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        for (int it = 0; it < 5; ++it) {
          if (kind == 0) k_copy<<<8192, 256>>>(a, b, n);
          else k_read<<<8192, 256>>>(a, o, n);
        }
        hipEventRecord(e1), hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms / 5 < best[kind]) best[kind] = ms / 5;
      }
    printf("%5zu MiB: copy %7.1f us = %.2f TB/s (read + write)   read-only %7.1f us = %.2f TB/s\n", bytes >> 20, best[0] * 1e3,
           2.0 * bytes / best[0] / 1e9, best[1] * 1e3, 1.0 * bytes / best[1] / 1e9);
    if (bytes == ((size_t)168 << 20)) bytes = (size_t)128 << 20;
  }
  return 0;
}
