// Issue rate of the MFMA forms discussed in NOTEBOOK.md §4.1, one wave per SIMD, 4 independent accumulators:
//   v_mfma_f32_32x32x16_bf16 (the kernels' instruction), v_mfma_i32_32x32x16_i8 (one 16-k block per instruction),
//   v_mfma_i32_32x32x32_i8 (full-rate int8), v_mfma_f32_32x32x16_f16.
// build: hipcc --offload-arch=gfx950 -O2 -o mfma_rate mfma_rate.hip ; run on one MI355X.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) short s8;
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(4))) int i4;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(16))) int i16v;

template <int KIND>
__global__ __launch_bounds__(512) void k(int iters, unsigned long long* cyc, float* sink) {
  s8 a = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, b = {1, 1, 2, 2, 3, 3, 4, 4};
  f16v f[4];
  i16v q[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) f[i][j] = 0.f, q[i][j] = 0;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (KIND == 0) f[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, f[i], 0, 0, 0);
      if constexpr (KIND == 1) q[i] = __builtin_amdgcn_mfma_i32_32x32x16_i8(__builtin_bit_cast(long, (__attribute__((ext_vector_type(4))) short){a[0], a[1], a[2], a[3]}),
                                                                           __builtin_bit_cast(long, (__attribute__((ext_vector_type(4))) short){b[0], b[1], b[2], b[3]}), q[i], 0, 0, 0);
      if constexpr (KIND == 2) q[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i4, a), __builtin_bit_cast(i4, b), q[i], 0, 0, 0);
      if constexpr (KIND == 3) f[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), f[i], 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) s += f[i][j] + (float)q[i][j];
  sink[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  unsigned long long* c;
  float* sink;
  hipMalloc(&c, 8);
  hipMalloc(&sink, 256 * 512 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const char* names[4] = {"v_mfma_f32_32x32x16_bf16", "v_mfma_i32_32x32x16_i8 ", "v_mfma_i32_32x32x32_i8 ", "v_mfma_f32_32x32x16_f16 "};
  for (int threads = 256; threads <= 512; threads += 256)
    for (int kind = 0; kind < 4; ++kind) {
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (kind == 0) k<0><<<256, threads>>>(iters, c, sink);
        if (kind == 1) k<1><<<256, threads>>>(iters, c, sink);
        if (kind == 2) k<2><<<256, threads>>>(iters, c, sink);
        if (kind == 3) k<3><<<256, threads>>>(iters, c, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      unsigned long long h;
      hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
      const double ops = 2.0 * 32 * 32 * (kind == 2 ? 32 : 16) * 4.0 * iters * (threads / 64) * 256;
      printf("%s  %5.1f cycles per instruction and wave, %d wave(s) per SIMD; whole kernel %.3f ms = %.0f T(FL)OP/s\n", names[kind],
             (double)h / (4.0 * iters), threads / 256, ms, ops / (ms * 1e-3) / 1e12);
    }
  return 0;
}
