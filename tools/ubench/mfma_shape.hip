// Which MFMA shape does the chip run faster under sustained load?  The per-clock rates of the 32x32 and 16x16 forms are equal
// (tools/ubench/mfma_rate.hip), but the chip lowers its clock under MFMA load and the clock it holds depends on the shape
// (MI355X_MICROARCH.md, DVFS give-back item 7).  Bare loops, operands in registers, RANDOM operand bits (zeros raise the
// clock), the same 128 x 32 output tile per wave (64 accumulator registers), two waves per SIMD on every CU, >= 1.5 s per
// arm, arms interleaved; reports T(FL)OP/s and the in-kernel clock (s_memtime / s_memrealtime).
// build: hipcc --offload-arch=gfx950 -O2 -o mfma_shape mfma_shape.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) int i4;
typedef __attribute__((ext_vector_type(8))) short s8;
typedef __attribute__((ext_vector_type(16))) int i16v;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(4))) float f4v;

// KIND 0: bf16 32x32x16, 1: bf16 16x16x32, 2: i8 32x32x32, 3: i8 16x16x64, 4 / 5: the two i8 forms with the vector work of
// the int8 kernel's weight-block-128 main loop beside the MFMAs (a group tile per 128 k folded into the running tile with one
// v_lshl_add_u32 per element, 12 instructions of nibble expand).  One "step" = 128 x 32 x 128 MACs per wave.
template <int KIND>
__global__ __launch_bounds__(512) void k(int steps, const uint32_t* __restrict__ rnd, unsigned long long* stamps, float* sink) {
  // operand fragments: 4 "weight" fragments and 8 "activation" fragments of random bits (bf16: exponent bits masked so that
  // every value is a normal number of moderate size - no NaN / inf arithmetic)
  i4 wf[4], xf[8];
  const uint32_t mask = KIND < 2 ? 0xBF7FBF7Fu : 0xFFFFFFFFu, orr = KIND < 2 ? 0x3C003C00u : 0u;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) wf[i][j] = (int)((rnd[(threadIdx.x * 16 + i * 4 + j) & 4095] & mask) | orr);
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 4; ++j) xf[i][j] = (int)((rnd[(threadIdx.x * 32 + 2048 + i * 4 + j) & 4095] & mask) | orr);
  f16v a32[4];
  f4v a16[16];
  i16v b32[4];
  i4 b16[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) a32[i][j] = 0.f, b32[i][j] = 0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 4; ++j) a16[i][j] = 0.f, b16[i][j] = 0;
  unsigned long long c0, r0, c1, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
  for (int st = 0; st < steps; ++st) {
    if constexpr (KIND == 0) {  // 128 k = 8 slices of 16: 8 x 4 MFMAs of 32 cycles
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          a32[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s8, wf[s & 3]), __builtin_bit_cast(s8, xf[(i + s) & 7]), a32[i], 0, 0, 0);
    } else if constexpr (KIND == 1) {  // 4 slices of 32: 4 x 16 MFMAs of 16 cycles (8 token tiles x 2 column tiles)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          a16[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(s8, wf[(s + (i >> 3)) & 3]), __builtin_bit_cast(s8, xf[(i + s) & 7]), a16[i], 0, 0, 0);
    } else if constexpr (KIND == 2) {  // 4 slices of 32: 4 x 4 MFMAs of 32 cycles
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) b32[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[s], xf[(i + s) & 7], b32[i], 0, 0, 0);
    } else if constexpr (KIND == 3) {  // 2 slices of 64: 2 x 16 MFMAs of 16 cycles
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 16; ++i) b16[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[(s + (i >> 3)) & 3], xf[(i + s) & 7], b16[i], 0, 0, 0);
    } else if constexpr (KIND == 4) {  // 32x32x32 with group tiles and folds
      const uint32_t sh = (uint32_t)(st & 3);
#pragma unroll
      for (int s = 0; s < 4; ++s) {  // expand: 3 vector instructions per 8 weights
        const uint32_t w0 = (uint32_t)wf[s][0] + st, w1 = (uint32_t)wf[s][1] + st;
        wf[s] = (i4){(int)((w0 << 4) & 0xF0F0F0F0u), (int)(w0 & 0xF0F0F0F0u), (int)((w1 << 4) & 0xF0F0F0F0u), (int)(w1 & 0xF0F0F0F0u)};
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const i16v z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        i16v G = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[0], xf[i], z, 0, 0, 0);
#pragma unroll
        for (int s = 1; s < 4; ++s) G = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[s], xf[(i + s) & 7], G, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 16; ++j) b32[i][j] = (int)(((uint32_t)G[j] << sh) + (uint32_t)b32[i][j]);
      }
    } else {  // 16x16x64 with group tiles and folds
      const uint32_t sh = (uint32_t)(st & 3);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const uint32_t w0 = (uint32_t)wf[s][0] + st, w1 = (uint32_t)wf[s][1] + st;
        wf[s] = (i4){(int)((w0 << 4) & 0xF0F0F0F0u), (int)(w0 & 0xF0F0F0F0u), (int)((w1 << 4) & 0xF0F0F0F0u), (int)(w1 & 0xF0F0F0F0u)};
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const i4 z = {0, 0, 0, 0};
        i4 G = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[(i >> 3)], xf[i & 7], z, 0, 0, 0);
        G = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[2 + (i >> 3)], xf[(i + 1) & 7], G, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) b16[i][j] = (int)(((uint32_t)G[j] << sh) + (uint32_t)b16[i][j]);
      }
    }
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
  float sum = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) sum += a32[i][j] + (float)b32[i][j];
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 4; ++j) sum += a16[i][j] + (float)b16[i][j];
  sink[blockIdx.x * 512 + threadIdx.x] = sum;
  if ((threadIdx.x & 63) == 0) {
    stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = c1 - c0;
    stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0;
  }
}

static int cmp(const void* a, const void* b) { return *(const double*)a < *(const double*)b ? -1 : 1; }

int main() {
  unsigned long long* stamps;
  float* sink;
  uint32_t* rnd;
  hipMalloc(&stamps, 256 * 8 * 2 * 8), hipMalloc(&sink, 256 * 512 * 4), hipMalloc(&rnd, 4096 * 4);
  uint32_t h[4096];
  srand(1);
  for (int i = 0; i < 4096; ++i) h[i] = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
  hipMemcpy(rnd, h, sizeof(h), hipMemcpyHostToDevice);
  const char* names[6] = {"bf16 v_mfma_f32_32x32x16_bf16", "bf16 v_mfma_f32_16x16x32_bf16", "i8   v_mfma_i32_32x32x32_i8  ", "i8   v_mfma_i32_16x16x64_i8  ",
                          "i8   32x32x32 + expand + fold", "i8   16x16x64 + expand + fold"};
  const int steps = 20000;  // per launch: 20000 x 128 x 32 x 128 MACs per wave
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const double ops = 2.0 * 128 * 32 * 128 * (double)steps * 256 * 8;
  for (int round = 0; round < 2; ++round)
    for (int kind = 0; kind < 6; ++kind) {
      float ms = 0, tot = 0;
      int n = 0;
      while (tot < 1200.f) {  // sustained: >= 1.5 s of back-to-back launches, the last launch is the sample
        hipEventRecord(e0);
        if (kind == 0) k<0><<<256, 512>>>(steps, rnd, stamps, sink);
        if (kind == 1) k<1><<<256, 512>>>(steps, rnd, stamps, sink);
        if (kind == 2) k<2><<<256, 512>>>(steps, rnd, stamps, sink);
        if (kind == 3) k<3><<<256, 512>>>(steps, rnd, stamps, sink);
        if (kind == 4) k<4><<<256, 512>>>(steps, rnd, stamps, sink);
        if (kind == 5) k<5><<<256, 512>>>(steps, rnd, stamps, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        tot += ms, ++n;
      }
      unsigned long long hs[256 * 8 * 2];
      hipMemcpy(hs, stamps, sizeof(hs), hipMemcpyDeviceToHost);
      double clk[2048];
      for (int i = 0; i < 2048; ++i) clk[i] = (double)hs[2 * i] / (double)hs[2 * i + 1] * 100e6;
      qsort(clk, 2048, sizeof(double), cmp);
      printf("round %d  %s  %8.3f ms per launch  %7.1f T(FL)OP/s  in-kernel clock %.3f GHz  (%d launches)\n", round, names[kind], ms,
             ops / (ms * 1e-3) / 1e12, clk[1024] / 1e9, n);
    }
  return 0;
}
