// Compute-side ceiling of an int8 main loop for the per-token INT configurations (NOTEBOOK.md §7): per 128-k step
// a wave that owns a 128 x 32 tile (the 128-row kernel's geometry: i32 + fp32 accumulators = 128 registers) expands 16
// words of 4-bit sign-magnitude codes to int8, issues 16 v_mfma_i32_32x32x32_i8 (4 k-slices x 4 m-tiles) and folds the
// i32 tile into fp32 accumulators with one scale per output column (weight blocks of 128).  No memory traffic: operands are register constants - this measures only
// whether the VALU work hides under the MFMAs with two waves per SIMD, against the bf16 loop of the shipped kernel
// (32 v_mfma_f32_32x32x16_bf16 + 16 fp8->bf16 expands for the same 128 k).
// build: hipcc --offload-arch=gfx950 -O2 -o int8_loop int8_loop.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) int i4;
typedef __attribute__((ext_vector_type(8))) short s8;
typedef __attribute__((ext_vector_type(16))) int i16v;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(2))) float f2;

// 8 sign-magnitude nibbles (nibble p: k = p/2 even, 4 + p/2 odd) -> two dwords of int8 (k 0..3, k 4..7)
__device__ __forceinline__ void expand_i8(uint32_t w, uint32_t& lo, uint32_t& hi) {
  constexpr uint32_t POS_LO = 0x03020100u, POS_HI = 0x07060504u, NEG_LO = 0xFDFEFF00u, NEG_HI = 0xF9FAFBFCu;
  const uint32_t a = w & 0x0F0F0F0Fu, b = (w >> 4) & 0x0F0F0F0Fu;
  const uint32_t ma = a & 0x07070707u, mb = b & 0x07070707u;
  const uint32_t sa = ((a & 0x08080808u) >> 3) * 255u, sb = ((b & 0x08080808u) >> 3) * 255u;
  const uint32_t pa = __builtin_amdgcn_perm(POS_HI, POS_LO, ma), na = __builtin_amdgcn_perm(NEG_HI, NEG_LO, ma);
  const uint32_t pb = __builtin_amdgcn_perm(POS_HI, POS_LO, mb), nb = __builtin_amdgcn_perm(NEG_HI, NEG_LO, mb);
  lo = (na & sa) | (pa & ~sa);
  hi = (nb & sb) | (pb & ~sb);
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

template <int KIND>  // 0: int8 with a rescale per 128 k, 1: int8 without (whole-row weight blocks), 2: the bf16 loop
__global__ __launch_bounds__(512) void k(int steps, const uint32_t* __restrict__ codes, unsigned long long* cyc, float* sink) {
  uint32_t w[16];
  for (int i = 0; i < 16; ++i) w[i] = codes[(threadIdx.x * 16 + i) & 1023];
  constexpr int NT = 4;
  i4 xf[NT];  // activation fragments, one per m tile (constant here; distinct, so that no MFMA is a common subexpression)
  for (int i = 0; i < NT; ++i) xf[i] = (i4){(int)threadIdx.x + 17 * i, 0x01020304 + i, 0x7f80ff01 - i, 0x10203040 ^ i};
  f16v accf[NT];
  for (int i = 0; i < NT; ++i)
    for (int j = 0; j < 16; ++j) accf[i][j] = 0.f;
  float scale = 1.0f + (float)(threadIdx.x & 3);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int st = 0; st < steps; ++st) {
    if constexpr (KIND < 2) {
      i16v acc[NT];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        uint32_t d[4];
        expand_i8(w[4 * s] + st, d[0], d[1]);
        expand_i8(w[4 * s + 1] + st, d[2], d[3]);
        const i4 wf = {(int)d[0], (int)d[1], (int)d[2], (int)d[3]};
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          if (s == 0 && (KIND == 0 || st == 0)) {
            const i16v z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, xf[i], z, 0, 0, 0);
          } else {
            acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, xf[i], acc[i], 0, 0, 0);
          }
        }
      }
      if (KIND == 0 || st == steps - 1) {
        const f2 sc = {scale, scale};
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int j = 0; j < 16; j += 2) {
            const f2 v = {(float)acc[i][j], (float)acc[i][j + 1]};
            const f2 a = {accf[i][j], accf[i][j + 1]};
            const f2 r = __builtin_elementwise_fma(v, sc, a);
            accf[i][j] = r[0], accf[i][j + 1] = r[1];
          }
        scale += 1.0f;
      }
    } else {
#pragma unroll
      for (int s = 0; s < 8; ++s) {  // 8 slices of 16 k
        const s8 wf0 = expand_bf16(w[2 * s] + st, scale), wf1 = expand_bf16(w[2 * s + 1] + st, scale);
        (void)wf1;
#pragma unroll
        for (int i = 0; i < NT; ++i)
          accf[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s & 1 ? wf1 : wf0, __builtin_bit_cast(s8, xf[i]), accf[i], 0, 0, 0);
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float sum = 0;
  for (int i = 0; i < NT; ++i)
    for (int j = 0; j < 16; ++j) sum += accf[i][j];
  sink[blockIdx.x * 512 + threadIdx.x] = sum;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  unsigned long long* c;
  float* sink;
  uint32_t* codes;
  hipMalloc(&c, 8), hipMalloc(&sink, 256 * 512 * 4), hipMalloc(&codes, 4096);
  hipMemset(codes, 0x5a, 4096);
  const int steps = 2000;
  const char* names[3] = {"int8, i32 tile folded into fp32 every 128 k (weight blocks of 128)", "int8, folded once (whole-row weight blocks)              ",
                          "bf16 loop of the shipped kernel (same 128 k)                   "};
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  float ms_ref = 0;
  for (int kind = 2; kind >= 0; --kind) {  // bf16 first: the reference for the ratios
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (kind == 0) k<0><<<256, 512>>>(steps, codes, c, sink);
      if (kind == 1) k<1><<<256, 512>>>(steps, codes, c, sink);
      if (kind == 2) k<2><<<256, 512>>>(steps, codes, c, sink);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    if (kind == 2) ms_ref = ms;
    // whole-kernel time: a wave's own cycle counter is misleading here (the two waves of a SIMD are not served evenly)
    printf("%s  %.3f ms for %d steps of 128 k on every SIMD (two waves each) = %.2fx the bf16 loop's speed\n", names[kind], ms, steps,
           ms_ref / ms);
  }
  return 0;
}
