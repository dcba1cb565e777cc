// Compute-side ceiling of the int8 main loop of gemm_w4a8_i8.hip against the 256-row bf16 loop it replaces, for the
// per-token INT configurations.  Geometry of the 256 x 256 kernels: a wave owns 256 token rows x 32 columns = 8 tiles
// of 32 x 32, two waves per SIMD.  Per 128 k:
//   KIND 0  int8, two's-complement nibbles ((w << 4) & 0xF0F0F0F0, w & 0xF0F0F0F0: 3 VALU per 8 weights), 32
//           v_mfma_i32_32x32x32_i8 accumulating straight into the running i32 tile (uniform weight scale per row)
//   KIND 1  the same, tile-major with a group accumulator per tile that is folded into the running i32 tile with one
//           v_lshl_add_u32 per element (weight blocks of 128: a per-column shift per 128-k group)
//   KIND 2  the bf16 loop: 64 v_mfma_f32_32x32x16_bf16 + 8 sign-magnitude -> fp8 -> bf16 expands (14 VALU per 8 weights)
// No memory traffic: operands are register values - this measures whether the VALU work hides under the MFMAs.
// build: hipcc --offload-arch=gfx950 -O3 -o int8_loop2 int8_loop2.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) int i4;
typedef __attribute__((ext_vector_type(8))) short s8;
typedef __attribute__((ext_vector_type(16))) int i16v;
typedef __attribute__((ext_vector_type(16))) float f16v;

__device__ __forceinline__ i4 expand_tc(uint32_t w0, uint32_t w1) {
  return (i4){(int)((w0 << 4) & 0xF0F0F0F0u), (int)(w0 & 0xF0F0F0F0u), (int)((w1 << 4) & 0xF0F0F0F0u), (int)(w1 & 0xF0F0F0F0u)};
}
__device__ __forceinline__ s8 expand_bf16(uint32_t word, float scale) {
  constexpr uint32_t LUT_LO = 0x44403800u, LUT_HI = 0x4E4C4A48u;
  const uint32_t t = word >> 4;
  uint32_t fe = __builtin_amdgcn_perm(LUT_HI, LUT_LO, word & 0x07070707u), fo = __builtin_amdgcn_perm(LUT_HI, LUT_LO, t & 0x07070707u);
  fe |= (word << 4) & 0x80808080u;
  fo |= word & 0x80808080u;
  typedef __attribute__((ext_vector_type(4))) uint32_t u4;
  u4 r;
  r[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(fe, scale, false));
  r[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(fe, scale, true));
  r[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(fo, scale, false));
  r[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(fo, scale, true));
  return __builtin_bit_cast(s8, r);
}

template <int KIND>
__global__ __launch_bounds__(512) void k(int steps, const uint32_t* __restrict__ codes, float* sink) {
  uint32_t w[8];
  for (int i = 0; i < 8; ++i) w[i] = codes[(threadIdx.x * 8 + i) & 1023];
  constexpr int NT = 8;
  i4 xf[NT];  // one activation fragment per tile (distinct, so that no two MFMA chains are common subexpressions)
  for (int i = 0; i < NT; ++i) xf[i] = (i4){(int)threadIdx.x * 77 + 17 * i, 0x01020304 + i, 0x7f80ff01 - i, 0x10203040 ^ i};
  float sum = 0.f;
  if constexpr (KIND < 2) {
    i16v R[NT];
    for (int i = 0; i < NT; ++i)
      for (int j = 0; j < 16; ++j) R[i][j] = 0;
    int sv = threadIdx.x & 3;
    for (int st = 0; st < steps; ++st) {
      i4 wf[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) wf[s] = expand_tc(w[2 * s] + st, w[2 * s + 1] ^ st);
      if constexpr (KIND == 0) {
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int s = 0; s < 4; ++s) R[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xf[i], wf[s], R[i], 0, 0, 0);
      } else {
        const i16v z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          i16v G = z;
#pragma unroll
          for (int s = 0; s < 4; ++s) G = __builtin_amdgcn_mfma_i32_32x32x32_i8(xf[i], wf[s], s == 0 ? z : G, 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 16; ++j) R[i][j] = (int)(((unsigned)G[j] << sv) + (unsigned)R[i][j]);
        }
        sv = (sv + 1) & 3;
      }
    }
    for (int i = 0; i < NT; ++i)
      for (int j = 0; j < 16; ++j) sum += (float)R[i][j];
  } else {
    f16v acc[NT];
    for (int i = 0; i < NT; ++i)
      for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float scale = 1.0f + (float)(threadIdx.x & 3);
    for (int st = 0; st < steps; ++st) {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const s8 wf = expand_bf16(w[s] + st, scale);
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, __builtin_bit_cast(s8, xf[i]), acc[i], 0, 0, 0);
      }
    }
    for (int i = 0; i < NT; ++i)
      for (int j = 0; j < 16; ++j) sum += acc[i][j];
  }
  sink[blockIdx.x * 512 + threadIdx.x] = sum;
}

int main() {
  float* sink;
  uint32_t* codes;
  hipMalloc(&sink, 256 * 512 * 4), hipMalloc(&codes, 4096);
  hipMemset(codes, 0x5a, 4096);
  const int steps = 2000;
  const char* names[3] = {"int8 32x32x32, straight accumulation (uniform row scale)     ",
                          "int8 32x32x32, tile-major + v_lshl_add_u32 fold per 128 k     ",
                          "bf16 32x32x16 loop of the 256-row kernel (same 128 k)         "};
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  float ms_ref = 0;
  for (int kind = 2; kind >= 0; --kind) {
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (kind == 0) k<0><<<256, 512>>>(steps, codes, sink);
      if (kind == 1) k<1><<<256, 512>>>(steps, codes, sink);
      if (kind == 2) k<2><<<256, 512>>>(steps, codes, sink);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    if (kind == 2) ms_ref = ms;
    const double ops = 2.0 * 256 * 8 * 256.0 * 32 * 128 * steps;  // per launch: CUs x waves x tile x k
    printf("%s  %.3f ms for %d steps of 128 k (two waves per SIMD) = %.2fx the bf16 loop, %.2f POP/s\n", names[kind], ms, steps,
           ms_ref / ms, ops / (ms * 1e-3) / 1e15);
  }
  return 0;
}
