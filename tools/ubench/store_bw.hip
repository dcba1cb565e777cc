// How long does the GEMM's output store pattern take by itself?  256 workgroups x 8 waves, each wave writes its
// 128 rows x 32 fp16 columns as the kernel does (per instruction: 32 rows x 32 contiguous bytes), vs the same bytes
// written as full 512-byte row segments per wave-instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512) void k_tile_pattern(uint4* y, int ldy16) {  // ldy16 = row stride in uint4
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lh = lane >> 5;
  const int tm = blockIdx.x / 16, tn = blockIdx.x % 16;
  for (int i = 0; i < 4; ++i)
    for (int p = 0; p < 2; ++p) {
      const int m = tm * 128 + i * 32 + l31;
      const int n16 = (tn * 256 + wave * 32 + 16 * p + 8 * lh) / 8;
      y[(size_t)m * ldy16 + n16] = make_uint4(m, n16, i, p);
    }
}
__global__ __launch_bounds__(512) void k_row_pattern(uint4* y, int ldy16) {
  // the same tile, but a wave-instruction writes 2 rows x 512 B (32 lanes x 16 B per row)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tm = blockIdx.x / 16, tn = blockIdx.x % 16;
  for (int it = 0; it < 8; ++it) {
    const int m = tm * 128 + wave * 16 + it * 2 + (lane >> 5);
    const int n16 = tn * 32 + (lane & 31);
    y[(size_t)m * ldy16 + n16] = make_uint4(m, n16, it, 0);
  }
}
int main() {
  const int M = 2048, N = 4096;
  uint4* y; hipMalloc(&y, (size_t)M * N * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int v = 0; v < 2; ++v) {
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      for (int it = 0; it < 20; ++it) {
        if (v == 0) k_tile_pattern<<<256, 512>>>(y, N * 2 / 16);
        else k_row_pattern<<<256, 512>>>(y, N * 2 / 16);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms / 20 < best) best = ms / 20;
    }
    printf("%s: %.2f us for 16 MiB\n", v == 0 ? "kernel's pattern (32 rows x 32 B per instruction)" : "row pattern (2 rows x 512 B per instruction)", best * 1e3);
  }
  return 0;
}
