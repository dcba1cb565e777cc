// Shared device helpers for the LQER HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "../../include/lqer_hip.h"

namespace lqer {

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4;    // 16x16 accumulator

// Resolved MXINT parameters handed to kernels by value.
struct QP {
  int kind;   // LQER_Q_*
  int mbits;  // width - 1
  int block;  // elements per shared exponent; <=0 = whole row
  int emin, emax;
  float mmax;  // 2^mbits - 1 (integer kind: the upper clamp)
  // what distinguishes "integer" (fixed point, LQER_Q_INT) from block_fp in the shared element routine: no +1e-9, no
  // pass-through of |x| <= 1e-8, a two's-complement range (the negative clamp is one step larger), a fixed exponent
  float mneg;  // magnitude of the negative clamp (block_fp: = mmax)
  float eps;   // added to |x| before scaling (block_fp.py:57: 1e-9; integer: 0)
  float tiny;  // |x| <= tiny is not quantized (block_fp.py:79-80: 1e-8; integer: -1 = never)
  int width;   // bits per element incl. sign
};

__host__ inline QP make_qp(const lqer_qfmt_t& f) {
  QP q;
  q.kind = f.kind;
  q.mbits = f.width - 1;
  q.block = f.block;
  q.emin = -f.exp_bias;
  q.emax = (1 << f.exp_width) - 1 - f.exp_bias;
  q.mmax = (float)((1 << (f.width - 1)) - 1);
  q.mneg = q.mmax, q.eps = 1e-9f, q.tiny = 1e-8f, q.width = f.width;
  if (f.kind == LQER_Q_INT) {  // value = m 2^-frac: "exponent" pinned to 0, mbits = frac_width
    const bool is_signed = f.exp_width != 0;
    q.mbits = f.exp_bias;
    q.block = -1;
    q.emin = q.emax = 0;
    q.mmax = is_signed ? (float)((1 << (f.width - 1)) - 1) : (float)((1u << f.width) - 1);
    q.mneg = is_signed ? (float)(1 << (f.width - 1)) : 0.0f;
    q.eps = 0.0f, q.tiny = -1.0f;
  }
  return q;
}

// ---- element loads of the three caller dtypes -------------------------------------------------
template <int DT>
__device__ __forceinline__ float load_elem(const void* p, int64_t i);
template <>
__device__ __forceinline__ float load_elem<LQER_F32>(const void* p, int64_t i) {
  return ((const float*)p)[i];
}
template <>
__device__ __forceinline__ float load_elem<LQER_F16>(const void* p, int64_t i) {
  return (float)((const _Float16*)p)[i];
}
template <>
__device__ __forceinline__ float load_elem<LQER_BF16>(const void* p, int64_t i) {
  return __uint_as_float(((uint32_t)((const bf16_t*)p)[i]) << 16);
}

// Internal fourth element type of the GEMM kernels: fp16 output AND fp16 activations multiplied natively
// (x_quantizer = LQER_Q_PASSTHROUGH_F16: the activation image holds fp16 bits, the weights are expanded to fp16 and
// the main loop runs v_mfma_*_f16).  Everything else about it is LQER_F16.
constexpr int LQER_F16X = 3;

template <int DT>
__device__ __forceinline__ void store_elem(void* p, int64_t i, float v);
template <>
__device__ __forceinline__ void store_elem<LQER_F16X>(void* p, int64_t i, float v) {
  ((_Float16*)p)[i] = (_Float16)v;
}
template <>
__device__ __forceinline__ void store_elem<LQER_F32>(void* p, int64_t i, float v) {
  ((float*)p)[i] = v;
}
template <>
__device__ __forceinline__ void store_elem<LQER_F16>(void* p, int64_t i, float v) {
  ((_Float16*)p)[i] = (_Float16)v;  // v_cvt_f16_f32: round to nearest even
}
__device__ __forceinline__ bf16_t f32_to_bf16_rne(float v) {
  uint32_t u = __float_as_uint(v);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // NaN stays NaN
  return (bf16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
template <>
__device__ __forceinline__ void store_elem<LQER_BF16>(void* p, int64_t i, float v) {
  ((bf16_t*)p)[i] = f32_to_bf16_rne(v);
}

// four consecutive outputs of one row, elements [at, at + 4) of y with columns n .. n + 3 of N: one 8- / 16-byte store when the
// four exist and are aligned, else element by element
template <int DT>
__device__ __forceinline__ void store_row4(void* y, int64_t at, int n, int N, const float (&out)[4]) {
  if (n + 3 < N && (at & 3) == 0 && ((uintptr_t)y & 15) == 0) {
    if constexpr (DT == LQER_F32) {
      *(f32x4*)((float*)y + at) = (f32x4){out[0], out[1], out[2], out[3]};
    } else if constexpr (DT == LQER_BF16) {
      *(uint2*)((bf16_t*)y + at) = make_uint2((uint32_t)f32_to_bf16_rne(out[0]) | ((uint32_t)f32_to_bf16_rne(out[1]) << 16),
                                              (uint32_t)f32_to_bf16_rne(out[2]) | ((uint32_t)f32_to_bf16_rne(out[3]) << 16));
    } else {
      typedef __attribute__((ext_vector_type(4))) _Float16 h4;
      *(h4*)((_Float16*)y + at) = (h4){(_Float16)out[0], (_Float16)out[1], (_Float16)out[2], (_Float16)out[3]};
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (n + j < N) store_elem<DT>(y, at + j, out[j]);
  }
}

// ---- the reference's exponent rule --------------------------------------------------------------
// e = torch.ceil(torch.log2(amax)) evaluated in fp32 (reference block_fp.py:58).  torch's log2 is
// correctly rounded, so for amax = 2^k (1 + j 2^-23) the sum k + log2(1 + j 2^-23) rounds back to k
// while j <= J(k), J = floor(2^(p-1) ln 2) with p the binade of the reals just above k
// (oracle/lqer_oracle.py::ceil_log2_f32; table pinned by tests/golden "log2_rule").
__device__ __forceinline__ int ceil_log2_rule(float amax) {
  // branch-free: slack(p) = {0, 0, 1, 2, 5, 11, 22, 22}[p], p = floor(log2 |k|) (one less for k = -2^j); |k| < 4 gives p <= 1
  const uint32_t bits = __float_as_uint(amax);
  const int k = (int)((bits >> 23) & 0xff) - 127;
  const int mant = (int)(bits & 0x7fffffu);
  const int ak = k < 0 ? -k : k;
  int p = 31 - __clz(ak | 1);
  p -= (k < 0 && (ak & (ak - 1)) == 0) ? 1 : 0;
  p = p < 0 ? 0 : p;
  const uint32_t tw = p < 4 ? 0x02010000u : 0x16160B05u;
  const int slack = (int)((tw >> ((p & 3) * 8)) & 0xffu);
  return mant > slack ? k + 1 : k;
}

__device__ __forceinline__ int block_exponent(float amax, const QP& q) {
  int e = ceil_log2_rule(amax);
  return e < q.emin ? q.emin : (e > q.emax ? q.emax : e);
}

// The same exponent behind ONE wave-uniform branch: the largest slack of the rule is 22, so a maximum more than 22 ulps above its
// power of two takes k + 1 whatever the binade - five instructions instead of ~30; the rule itself runs only when some lane of the wave
// sits that close to a power of two (or on it, or at zero).  For the quantizer kernels (a block per lane or lane pair: the rule was a
// third of their vector work); NOT for code that shares a scheduling region with MFMAs (a branch ends the region).
__device__ __forceinline__ int block_exponent_u(float amax, const QP& q) {
  const uint32_t bits = __float_as_uint(amax);
  const int mant = (int)(bits & 0x7fffffu);
  int e;
  if (__builtin_amdgcn_ballot_w64(mant <= 22) == 0) e = (int)((bits >> 23) & 0xff) - 126;
  else e = ceil_log2_rule(amax);
  return e < q.emin ? q.emin : (e > q.emax ? q.emax : e);
}

// Signed mantissa of one element given its block exponent (block_fp.py:55-65):
//   m = min(rne((|x| + 1e-9) / 2^e * 2^mbits), 2^mbits - 1), sign from x.
// |x| <= 1e-8 is the reference's pass-through (block_fp.py:79-80); packed images flush it to 0.
// (integer kind, quantizers/integer.py:37-40: clamp(rne(x 2^frac), lo, hi) - the same routine with eps = 0, tiny = -1 and
// the two's-complement negative clamp)
__device__ __forceinline__ float mxint_mantissa(float x, int e, const QP& q) {
  const float v = fabsf(x) + q.eps;
  const float t = ldexpf(v, q.mbits - e);
  // (written as mmax + (x < 0 ? mneg - mmax : 0): a select between the two struct fields themselves becomes, when q lives in
  // the kernel-argument segment, a per-lane LOAD from a selected address - a memory round trip per element)
  const float m = fminf(rintf(t), q.mmax + (x < 0.0f ? q.mneg - q.mmax : 0.0f));
  return fabsf(x) <= q.tiny ? 0.0f : copysignf(m, x);
}

// The bf16 image of 16 values that share the block exponent e, in packed-fp32 arithmetic (v_pk_add/fma/mul_f32: two
// elements per instruction) - the same results as mxint_mantissa + ldexpf element by element (block_fp.py:55-65):
//   t = |x| + 1e-9;  m = min(rne(t * 2^(mbits-e)), mmax);  value = sign(x) * m * 2^(e-mbits);  |x| <= 1e-8 -> 0.
// rne(t s) = fma(t, s, 1.5 * 2^23) - 1.5 * 2^23: t s is exact (s is a power of two), the fma rounds once to an integer
// (t s < 2^22 because t <= 2^e + 1e-9).  Needs s and 1/s to be normal floats: mxint16_fast_ok(e, q).
__device__ __forceinline__ bool mxint16_fast_ok(int e, const QP& q) {
  const int d = q.mbits - e;  // both 2^d and 2^-d normal
  return d <= 126 && d >= -126;
}

// One value: sign(x) min(rne((|x| + eps) 2^(mbits-e)), clamp) 2^(e-mbits) - mxint_mantissa + ldexpf without the two ldexpf
// calls (power-of-two factors built from exponent bits, round-to-nearest-even by the 1.5 * 2^23 trick, as in
// mxint16_bf16_fast below) when mxint16_fast_ok(e, q); the |x| <= tiny case is the caller's.
__device__ __forceinline__ float mxint_value(float x, int e, const QP& q) {
  if (mxint16_fast_ok(e, q) && q.mmax <= 4194304.0f) {  // (the magic-number rounding is exact up to 2^22)
    const float s = __uint_as_float((uint32_t)(127 + q.mbits - e) << 23), inv = __uint_as_float((uint32_t)(127 + e - q.mbits) << 23);
    const float r = __builtin_fmaf(fabsf(x) + q.eps, s, 12582912.0f) - 12582912.0f;
    return copysignf(fminf(r, q.mmax + (x < 0.0f ? q.mneg - q.mmax : 0.0f)) * inv, x);
  }
  const float t = ldexpf(fabsf(x) + q.eps, q.mbits - e);
  return copysignf(ldexpf(fminf(rintf(t), q.mmax + (x < 0.0f ? q.mneg - q.mmax : 0.0f)), e - q.mbits), x);
}

// Written on the SIGNED value (no |x|, no sign transplant): u = fma(x, s, copysign(1e-9 s, x)) is sign(x) fl((|x| + 1e-9) s)
// - scaling by a power of two commutes with the rounding of the sum -, the 1.5 * 2^23 trick rounds a signed u to nearest even
// like an unsigned one (|u| <= 2^mbits), v_med3_f32 clamps both ends, and the bf16 image is the high half of r 2^(e-mbits)
// with its own sign (a negative x that rounds to zero gives -0, as the sign transplant did).  Nine vector instructions per
// pair of elements (two v_bfi, three packed fp32 ops, two v_med3, one packed multiply, one v_perm) instead of fourteen.
template <bool FLUSH_TINY, int N = 16>  // FLUSH_TINY false when the input type cannot hold a non-zero |x| <= 1e-8 (fp16); N values (a block of
                                       // 16, or the 8 of it that one lane of act16_fused.hip holds)
__device__ __forceinline__ void mxint16_bf16_fast(const float (&v)[N], int e, const QP& q, uint32_t (&w)[N / 2]) {
  typedef __attribute__((ext_vector_type(2))) float f2;
  const float s = __uint_as_float((uint32_t)(127 + q.mbits - e) << 23);
  const float inv = __uint_as_float((uint32_t)(127 + e - q.mbits) << 23);
  const float es = 1e-9f * s, lo = -q.mneg, hi = q.mmax;
  const f2 magic = {12582912.0f, 12582912.0f};
#pragma unroll
  for (int i = 0; i < N / 2; ++i) {
    const f2 x = {v[2 * i], v[2 * i + 1]};
    const f2 c = {copysignf(es, x[0]), copysignf(es, x[1])};
    f2 r = (__builtin_elementwise_fma(x, (f2){s, s}, c) + magic) - magic;
    r[0] = __builtin_amdgcn_fmed3f(r[0], lo, hi);
    r[1] = __builtin_amdgcn_fmed3f(r[1], lo, hi);
    const f2 val = r * (f2){inv, inv};
    uint32_t b0 = __float_as_uint(val[0]), b1 = __float_as_uint(val[1]);
    if constexpr (FLUSH_TINY) {
      b0 = fabsf(x[0]) <= 1e-8f ? 0u : b0;
      b1 = fabsf(x[1]) <= 1e-8f ? 0u : b1;
    }
    w[i] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);  // the two high halves
  }
}

// In place: N (even) fp32 values that share the block exponent e -> their quantizer images, two per packed fp32 instruction -
// the B_out re-quantization of the tile kernels' side product.  block_fp (block_fp.py:55-65, 79-80):
//   q = sign(v) min(rne((|v| + 1e-9) 2^(mbits-e)), mmax) 2^(e-mbits);  |v| <= 1e-8 keeps v;
// integer (quantizers/integer.py:37-40; the QP carries eps = 0, tiny = -1, the two's-complement negative clamp, e = 0):
//   q = clamp(rne(v 2^frac), lo, hi) 2^-frac.
// Same arithmetic as mxint16_bf16_fast (fma on the signed value, 1.5 * 2^23 rounding, v_med3 clamp): the results of
// mxint_mantissa + ldexpf element by element; exponents whose scale factors are not normal floats, and clamps beyond 2^22
// (where the magic-number rounding stops being exact), take that route.
template <int N>
__device__ __forceinline__ void mxint_requant_fast(float (&v)[N], int e, const QP& q) {
  typedef __attribute__((ext_vector_type(2))) float f2;
  // (the 1.5 * 2^23 rounding is exact for |v s| <= 2^22 only: formats with a clamp beyond that - B_out widths 24 - take the
  // element routine)
  if (mxint16_fast_ok(e, q) && q.mmax <= 4194304.0f && q.mneg <= 4194304.0f) {
    const float s = __uint_as_float((uint32_t)(127 + q.mbits - e) << 23);
    const float inv = __uint_as_float((uint32_t)(127 + e - q.mbits) << 23);
    const float es = q.eps * s, hi = q.mmax, lo = -q.mneg, tiny = q.tiny;
    const f2 magic = {12582912.0f, 12582912.0f};
#pragma unroll
    for (int i = 0; i < N; i += 2) {
      const f2 x = {v[i], v[i + 1]};
      const f2 c = {copysignf(es, x[0]), copysignf(es, x[1])};
      f2 r = (__builtin_elementwise_fma(x, (f2){s, s}, c) + magic) - magic;
      r[0] = __builtin_amdgcn_fmed3f(r[0], lo, hi);
      r[1] = __builtin_amdgcn_fmed3f(r[1], lo, hi);
      const f2 val = r * (f2){inv, inv};
      // (a negative v that rounds to zero: -0 here, copysign(0, v) = -0 in the element routine as well)
      v[i] = fabsf(x[0]) <= tiny ? x[0] : val[0];
      v[i + 1] = fabsf(x[1]) <= tiny ? x[1] : val[1];
    }
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const float t = v[i];
      v[i] = fabsf(t) <= q.tiny ? t : ldexpf(mxint_mantissa(t, e, q), e - q.mbits);
    }
  }
}

// The int8 image of 16 values that share the exponent e: the signed mantissas themselves (two's complement bytes), same
// arithmetic as mxint16_bf16_fast without the final scaling.  Needs mbits <= 7 and mxint16_fast_ok(e, q).
template <bool FLUSH_TINY>
__device__ __forceinline__ void mxint16_i8_fast(const float (&v)[16], int e, const QP& q, uint32_t (&w)[4]) {
  typedef __attribute__((ext_vector_type(2))) float f2;
  const float s = __uint_as_float((uint32_t)(127 + q.mbits - e) << 23);
  const float es = 1e-9f * s, lo = -q.mneg, hi = q.mmax;
  const f2 magic = {12582912.0f, 12582912.0f};
  uint32_t h[8];  // pairs of int16
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const f2 x = {v[2 * i], v[2 * i + 1]};
    const f2 c = {copysignf(es, x[0]), copysignf(es, x[1])};
    f2 r = (__builtin_elementwise_fma(x, (f2){s, s}, c) + magic) - magic;
    r[0] = __builtin_amdgcn_fmed3f(r[0], lo, hi);
    r[1] = __builtin_amdgcn_fmed3f(r[1], lo, hi);
    if constexpr (FLUSH_TINY) {
      r[0] = fabsf(x[0]) <= 1e-8f ? 0.0f : r[0];
      r[1] = fabsf(x[1]) <= 1e-8f ? 0.0f : r[1];
    }
    h[i] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16((int)r[0], (int)r[1]));
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = __builtin_amdgcn_perm(h[2 * i + 1], h[2 * i], 0x06040200u);  // low bytes of the four int16
}

// bf16 bits of an exactly representable fp32 value (low 16 bits are zero by construction).
__device__ __forceinline__ uint32_t exact_bf16_bits(float v) { return __float_as_uint(v) >> 16; }

// max over the 16 lanes of a DPP row (lanes 16g..16g+15); every lane of the row gets the result.
__device__ __forceinline__ float row16_max(float v) {
  int t;
  t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
  v = fmaxf(v, __int_as_float(t));
  t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true);  // quad_perm [2,3,0,1]
  v = fmaxf(v, __int_as_float(t));
  t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true);  // row_half_mirror
  v = fmaxf(v, __int_as_float(t));
  t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true);  // row_mirror
  v = fmaxf(v, __int_as_float(t));
  return v;
}


// Expand 8 sign-magnitude 4-bit codes (one 32-bit word = 8 consecutive k of one weight row; nibble p
// holds k = p/2 for even p, 4 + p/2 for odd p) times the block scale into one MFMA operand fragment:
// magnitude -> fp8 (e4m3) byte through a v_perm_b32 table, sign bit OR-ed in, then
// v_cvt_scalef32_pk_bf16_fp8 converts two elements per instruction and applies the scale.
// 14 VALU ops per 8 weights; every step is exact (integers 0..7 and powers of two).
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
__device__ __forceinline__ bf16x8 expand_frag(uint32_t word, uint32_t scale_bits) {
  const float scale = __uint_as_float(scale_bits);
  constexpr uint32_t LUT_LO = 0x44403800u, LUT_HI = 0x4E4C4A48u;  // e4m3 bytes of 0,1,2,3 | 4,5,6,7
  const uint32_t t = word >> 4;
  uint32_t fe = __builtin_amdgcn_perm(LUT_HI, LUT_LO, word & 0x07070707u);  // k 0..3
  uint32_t fo = __builtin_amdgcn_perm(LUT_HI, LUT_LO, t & 0x07070707u);     // k 4..7
  fe |= (word << 4) & 0x80808080u;
  fo |= word & 0x80808080u;
  u32x4 r;
  r[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(fe, scale, false));
  r[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(fe, scale, true));
  r[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(fo, scale, false));
  r[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(fo, scale, true));
  return __builtin_bit_cast(bf16x8, r);
}


// Two's-complement nibbles (the `integer` weight quantizer: codes -8 .. 7, reference quantizers/integer.py:37-40): the e4m3 byte
// comes from one of two 8-entry tables indexed by the low three bits - values 0..7, or -8..-1 with the sign baked in - chosen per
// byte by a mask of the nibbles' sign bits (v_perm_b32's sign selectors: 0x00 / 0xFF per odd byte of its sources).  20 vector
// instructions per 8 weights instead of 14; every step exact.
__device__ __forceinline__ bf16x8 expand_frag_twos(uint32_t word, uint32_t scale_bits) {
  const float scale = __uint_as_float(scale_bits);
  constexpr uint32_t LUT_LO = 0x44403800u, LUT_HI = 0x4E4C4A48u;    // e4m3 of 0,1,2,3 | 4,5,6,7
  constexpr uint32_t LUTN_LO = 0xCACCCED0u, LUTN_HI = 0xB8C0C4C8u;  // e4m3 of -8,-7,-6,-5 | -4,-3,-2,-1
  const uint32_t ie = word & 0x07070707u, io = (word >> 4) & 0x07070707u, t = word << 4;
  const uint32_t me = __builtin_amdgcn_perm(t, word << 12, 0x0B090A08u);    // low nibbles (k 0..3): 0xFF where negative
  const uint32_t mo = __builtin_amdgcn_perm(word, word << 8, 0x0B090A08u);  // high nibbles (k 4..7)
  const uint32_t pe = __builtin_amdgcn_perm(LUT_HI, LUT_LO, ie), ne = __builtin_amdgcn_perm(LUTN_HI, LUTN_LO, ie);
  const uint32_t po = __builtin_amdgcn_perm(LUT_HI, LUT_LO, io), no = __builtin_amdgcn_perm(LUTN_HI, LUTN_LO, io);
  const uint32_t fe = (me & ne) | (~me & pe), fo = (mo & no) | (~mo & po);
  u32x4 r;
  r[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(fe, scale, false));
  r[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(fe, scale, true));
  r[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(fo, scale, false));
  r[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(fo, scale, true));
  return __builtin_bit_cast(bf16x8, r);
}

// The same with fp16 results (v_cvt_scalef32_pk_f16_fp8): exact while code * 2^scale stays inside the fp16 range,
// subnormals included (checked once per weight image, pack.hip::weight_f16_ok).
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
template <bool F16>
__device__ __forceinline__ bf16x8 expand_frag_t(uint32_t word, uint32_t scale_bits) {
  if constexpr (!F16) {
    return expand_frag(word, scale_bits);
  } else {
    const float scale = __uint_as_float(scale_bits);
    constexpr uint32_t LUT_LO = 0x44403800u, LUT_HI = 0x4E4C4A48u;
    const uint32_t t = word >> 4;
    uint32_t fe = __builtin_amdgcn_perm(LUT_HI, LUT_LO, word & 0x07070707u);
    uint32_t fo = __builtin_amdgcn_perm(LUT_HI, LUT_LO, t & 0x07070707u);
    fe |= (word << 4) & 0x80808080u;
    fo |= word & 0x80808080u;
    u32x4 r;
    r[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(fe, scale, false));
    r[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(fe, scale, true));
    r[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(fo, scale, false));
    r[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(fo, scale, true));
    return __builtin_bit_cast(bf16x8, r);
  }
}

// MFMA on 16-bit fragments held as raw bits: bf16 (every MXINT image) or fp16 (LQER_F16X main loops)
template <bool F16>
__device__ __forceinline__ f32x16 mfma_32x32x16(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ f32x4 mfma_16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}


// ---- the int8 route (gemm_w4a8_i8.hip): per-token 8-bit activations x 4-bit weights with blocks of 128 k or more ----------
// Activation image: int8 mantissas [Mp][Kp8] (Kp8 = K padded to 128) followed, 256-byte aligned, by the row scales
// fp32 [Mp] - together never larger than the bf16 image [Mp][Kp] they replace in the workspace (K >= 128).
constexpr int I8_BK = 128;                  // k per main-loop step = per weight-shift group
constexpr int I8_WBLOCK = 256 * 64 + 256;   // bytes of one (256-row n tile, 128-k step) block: nibbles + one shift byte per row
__host__ __device__ inline int64_t padded_k8(int64_t K) { return (K + I8_BK - 1) / I8_BK * I8_BK; }
__host__ inline size_t i8_act_image_bytes(int64_t M, int64_t K) {
  return ((size_t)lqer_padded_m(M) * padded_k8(K) + 255) / 256 * 256;
}
__host__ inline const float* i8_row_scales(const void* xq8, int64_t M, int64_t K) {
  return (const float*)((const unsigned char*)xq8 + i8_act_image_bytes(M, K));
}
// Weight image: per (n tile of 256 rows, step of 128 k) one block of I8_WBLOCK bytes - 256 rows x 64 B of
// two's-complement nibbles, then 256 shift bytes -, then the row scales fp32 [Np] (gemm_w4a8_i8.hip has the details).
// ... and one MODE byte per n tile (256-byte padded): how the tile's per-(row, group) exponents are carried (I8_MODE_*).
constexpr int I8_MODE_NONE = 0;      // every group of every row has the row's exponent: plain accumulation
constexpr int I8_MODE_FOLD = 1;      // shift bytes s = e[n,g] - emin[n]: the group sums are folded with v_lshl_add_u32
constexpr int I8_MODE_PRESHIFT = 2;  // shift bytes q = emax[n] - e[n,g] <= 4: the int8 lane is cw << (4 - q), no folds
constexpr int I8_MODE_PRESHIFT1 = 3; // the same with q <= 1 everywhere in the tile: the sign fill is the sign bit itself (7 instead of 11
                                     // vector instructions per 8 weights)
__host__ __device__ inline size_t i8_weight_mode_offset(int64_t Np, int64_t nk8) {
  return (size_t)(Np / 256) * nk8 * I8_WBLOCK + (size_t)Np * sizeof(float);
}
__host__ inline size_t i8_weight_image_bytes(int64_t N, int64_t K) {
  const int64_t Np = lqer_padded_n(N);
  return i8_weight_mode_offset(Np, padded_k8(K) / I8_BK) + (size_t)((Np / 256 + 255) / 256 * 256);
}
// ... of 8-bit codes (weights of 5..8 bits): per (n tile, 64-k half-step) 256 rows x 64 B, then the row scales fp32 [Np]
__host__ inline size_t i8_weight8_image_bytes(int64_t N, int64_t K) {
  const int64_t Np = lqer_padded_n(N);
  return (size_t)(Np / 256) * (padded_k8(K) / 64) * (256 * 64) + (size_t)Np * sizeof(float);
}

#ifndef LQER_AMAX_NSEG
#define LQER_AMAX_NSEG 16  // column-segment partials per row of a one-block-per-row B_out (int8 route: k_bout_amax -> k_lqer_gemm_i8)
#define LQER_AMAX_NSEG_WIDE 32  // ... beyond N = 4096 (more than 8 column tiles per segment otherwise): twice the cells, the same scratch as the exchange's granules
#endif

// ---- the int8 image of per-token activations: one row's 16-byte chunks of 8 sixteen-bit elements (k_quant_row8 in quantize.hip and
// the fused int8-route activation kernel in act8_fused.hip share this arithmetic: same codes, same scales) ---------------------------
// the largest |element| of the MAXCH chunks a lane holds (packed 16-bit maxima for fp16)
template <int DT, int MAXCH>
__device__ __forceinline__ float row8_amax(const u32x4 (&raw)[MAXCH]) {
  float amax = 0.f;
  if constexpr (DT == LQER_F16) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    h2 m = {(_Float16)0.f, (_Float16)0.f};
#pragma unroll
    for (int u = 0; u < MAXCH; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) m = __builtin_elementwise_max(m, __builtin_bit_cast(h2, raw[u][j] & 0x7fff7fffu));
    amax = fmaxf((float)m[0], (float)m[1]);
  } else {
#pragma unroll
    for (int u = 0; u < MAXCH; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        amax = fmaxf(amax, fmaxf(__uint_as_float((raw[u][j] << 16) & 0x7fffffffu), __uint_as_float(raw[u][j] & 0x7fff0000u)));
  }
  return amax;
}
// one chunk -> its 8 int8 mantissas (live = false: zeros).  FAST: the packed signed arithmetic of mxint16_i8_fast with
// s = 2^(mbits - e), es = 1e-9 s (needs mxint16_fast_ok(e, q)); else element by element through mxint_mantissa.
template <int DT, bool FAST>
__device__ __forceinline__ u32x2 row8_chunk(const u32x4 raw, bool live, int e, const QP& q, float s, float es) {
  typedef __attribute__((ext_vector_type(2))) float f2;
  const float lo = -q.mneg, hi = q.mmax;
  const f2 magic = {12582912.0f, 12582912.0f};
  uint32_t h[4] = {0, 0, 0, 0};  // pairs of int16 mantissas
  if (live) {
    // (plain words first: indexing raw[j] with the unrolled j directly made hipcc reuse word 0 for all four pairs)
    const uint32_t wd[4] = {raw[0], raw[1], raw[2], raw[3]};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f2 xv;
      if constexpr (DT == LQER_F16) {
        typedef __attribute__((ext_vector_type(2))) _Float16 h2;
        const h2 hv = __builtin_bit_cast(h2, wd[j]);
        xv = (f2){(float)hv[0], (float)hv[1]};
      } else {
        xv = (f2){__uint_as_float(wd[j] << 16), __uint_as_float(wd[j] & 0xffff0000u)};
      }
      f2 r;
      if constexpr (FAST) {
        const f2 cc = {copysignf(es, xv[0]), copysignf(es, xv[1])};
        r = (__builtin_elementwise_fma(xv, (f2){s, s}, cc) + magic) - magic;
        r[0] = __builtin_amdgcn_fmed3f(r[0], lo, hi);
        r[1] = __builtin_amdgcn_fmed3f(r[1], lo, hi);
        if constexpr (DT != LQER_F16) {  // (fp16 cannot hold a non-zero |x| <= 1e-8)
          r[0] = fabsf(xv[0]) <= 1e-8f ? 0.0f : r[0];
          r[1] = fabsf(xv[1]) <= 1e-8f ? 0.0f : r[1];
        }
      } else {
        r = (f2){mxint_mantissa(xv[0], e, q), mxint_mantissa(xv[1], e, q)};
      }
      h[j] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16((int)r[0], (int)r[1]));
    }
  }
  return (u32x2){__builtin_amdgcn_perm(h[1], h[0], 0x06040200u), __builtin_amdgcn_perm(h[3], h[2], 0x06040200u)};
}

// fp16 rows whose exponent allows it (row8_h16_ok): the same 8 mantissas in PACKED HALF arithmetic, 2.25 vector instructions per element
// instead of 6.25 - t = x 2^(mbits - e) is exact in fp16 (a power-of-two scale inside the half range; what underflows rounds to 0 either
// way), the clamp comes first (t >= mmax + 0.5 would round to 2^mbits), and t + 1536 rounds to nearest even at ulp 1 and leaves the
// two's-complement mantissa in the low byte of the half (1536 + r = 0x6600 + r).  Where is this the fp32 path's result bit for bit?  That
// path rounds fl32(t +- 1e-9 2^(mbits - e)): the epsilon only ever matters for an exact tie t = k + 1/2 whose fp32 half-ulp it reaches -
// ties below 2^(mbits - 5 - e), i.e. none when e >= mbits - 4 (a tie is at least 1/2); fp16 holds no non-zero |x| <= 1e-8.
__device__ __forceinline__ bool row8_h16_ok(int e, const QP& q) {
  return q.kind == LQER_Q_MXINT && q.mbits <= 7 && e >= q.mbits - 4 && e <= 16 && q.mneg == q.mmax;  // (2^(mbits - e) in [2^-9, 2^4])
}
__device__ __forceinline__ u32x2 row8_chunk_h16(const u32x4 raw, bool live, int e, const QP& q) {
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  const _Float16 s1 = __builtin_bit_cast(_Float16, (unsigned short)((15 + q.mbits - e) << 10));  // 2^(mbits - e)
  const _Float16 m1 = (_Float16)q.mmax;
  const h2 s2 = {s1, s1}, hi = {m1, m1}, lo = {-m1, -m1}, magic = {(_Float16)1536.0f, (_Float16)1536.0f};
  uint32_t u[4] = {0, 0, 0, 0};
  if (live) {
    const uint32_t wd[4] = {raw[0], raw[1], raw[2], raw[3]};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      h2 t = __builtin_bit_cast(h2, wd[j]) * s2;
      t = __builtin_elementwise_max(__builtin_elementwise_min(t, hi), lo);
      u[j] = __builtin_bit_cast(uint32_t, t + magic);
    }
  }
  return (u32x2){__builtin_amdgcn_perm(u[1], u[0], 0x06040200u), __builtin_amdgcn_perm(u[3], u[2], 0x06040200u)};
}

// ---- cross-file declarations ----------------------------------------------------------------------
struct QuantOut {
  float* deq;      // [rows, cols] or null
  int8_t* codes;   // [rows, cols] or null
  int8_t* exps;    // [rows, nblk] or null
  bf16_t* xq;      // [rows_p, cols_p] exact bf16 image or null
  int64_t cols_p;  // row stride of xq (zero-filled beyond cols)
  int64_t nblk;    // blocks per row (exps row stride)
  // int8 route (one exponent per row, width <= 8): two's-complement mantissas [rows_p][cols_p8] (cols_p8 a multiple of
  // 128, zero-filled beyond cols) and the row's scale 2^(e - mbits) - value = code * scale
  int8_t* xq8 = nullptr;
  int64_t cols_p8 = 0;
  float* xscale = nullptr;
};

struct GemmArgs {
  const bf16_t* xq;    // [Mp][Kp]
  const uint8_t* wp;   // packed panels
  const bf16_t* xaq;   // [Mp][xaq_ld] (the Linear's rp columns start at the pointer) or null
  int xaq_ld;          // row stride of xaq in elements: rp, or the total padded rank of a group sharing one input
  const bf16_t* bt;    // [limbs][Np][rp]
  const float* bias;   // [Np] or null
  void* y;
  int64_t ldy;
  int M, N, Np, Kp, rp, b_limbs;
  int x_f16;           // xq holds fp16 bits (pass-through fp16 activations): LQER_F16X kernels
  // decode sizes (small-M kernel only): instead of xaq, the split-K partial tiles of x A left by the fused quantize
  // kernel - part[c][m][rp] fp32, c < xa_nchunk, chunk stride xa_cstride floats - summed in ascending c and
  // re-quantized (A_out, blocks of 16) by the GEMM itself: one launch less on a launch-bound path
  const float* xa_part;
  int xa_nchunk;
  int64_t xa_cstride;
  QP aout;
  int w_mbits;
  QP bout;
  int tiles_m, tiles_n;
  int tiles_m_rows;     // rows of a tile of the 128-row kernel family: 128, or 64 for token counts that leave its grid thin
  int xcd_bm;           // > 0: an XCD's tiles form a block of xcd_bm token tiles x (tiles / 8 / xcd_bm) weight tiles (128-row kernel)
  float* bout_amax;     // [Mp][bout_nblk] row-block maxima of xAq @ B (B_out blocks other than 16), else null
  int bout_L, bout_nblk;
  int bout_nseg;        // > 0 (int8 route, one block per row): bout_amax holds [bout_nseg][Mp] column-segment partials (no atomics)
  int amax_zeroed;      // host side: the atomicMax cells of the pre-pass are zero already (gemm_amax_zero_bytes) - no memset launch
  // int8 route: xq holds the int8 activation image (+ row scales), w8 the two's-complement weight image
  const uint8_t* w8;
  const float* xscale;  // [Mp] row scales 2^(e - mbits) of the int8 activation image
  int i8_shift;         // some weight group carries a non-zero shift (blocks of 128 with differing exponents)
  int tuning;           // lqer_linear_desc_t.tuning of the call (LQER_TUNE_*: kernel-variant knobs of tests, same bits)
  int w_twos;           // the packed weight holds two's-complement nibbles (w_quantizer = integer): the 128-row tile kernel only
  int w_i8codes;        // int8 route: w8 holds 8-bit CODES (weights of 5..8 bits, one exponent per row), fragment-major layout
  // int8 route, one round of 128-row tiles, one B_out block per row: NO pre-pass launch - every workgroup publishes the row maxima of
  // its own tile's side product as {value, tag} granules [tiles_n][Mp] in bout_amax and gathers its row band's at the epilogue
  // (gemm_w4a8_i8.hip "exchange"); xch_nonce = the host part of the tag
  int bout_xch;
  uint32_t xch_nonce;
};

int quantize_dispatch(const void* x, int dtype, int64_t rows, int64_t cols, int64_t ld, const QP& q,
                      const QuantOut& o, hipStream_t st);
int quantize_tiles_dispatch(const void* x, int dtype, int64_t batches, int64_t rows, int64_t cols, int64_t R, int64_t L, const QP& q,
                            float* out, float* amax, hipStream_t st);
int pack_weight_dispatch(const void* W, int dtype, int64_t N, int64_t K, int64_t ld, const QP& q, int64_t block_rows, void* out,
                         void* scratch, hipStream_t st);
int unpack_weight_dispatch(const void* in, int64_t N, int64_t K, int mbits, bool twos, float* out, hipStream_t st);
int pack_lowrank_dispatch(const void* A, const void* B, int dtype, int64_t K, int64_t N, int64_t r, void* a_t,
                          void* b_t, int32_t* flags, hipStream_t st);
int bias_passthrough_dispatch(const void* b, int dtype, int64_t N, float* out, hipStream_t st);
int lowrank_xa_dispatch(const bf16_t* xq, int64_t M, int64_t K, int x_limbs, const bf16_t* a_t, int a_limbs, int64_t r,
                        const QP& q, int xa_limbs, bf16_t* xaq, float* scratch, size_t scratch_bytes, hipStream_t st);
int copy_act_f16_dispatch(const void* x, int64_t M, int64_t K, int64_t ldx, bf16_t* xq, hipStream_t st);
// the int8 route's activation side in one launch (act8_fused.hip); LQER_E_UNSUPPORTED = not its case
#ifndef LQER_ACT8_FUSED_MAX_M
#define LQER_ACT8_FUSED_MAX_M 4096
#endif
#ifndef LQER_ACT8_FUSED_MIN_M
#define LQER_ACT8_FUSED_MIN_M 1024
#endif
// the block-16 MXINT activation side in one launch (act16_fused.hip): a_b16 = the bf16 image [rp][Kp] + its fragment-major copy
int act16_fused_dispatch(const void* x, int dtype, int64_t M, int64_t K, int64_t ldx, const QP& qx, bf16_t* xq, const void* a_b16, int64_t r,
                         const QP& qa, bf16_t* xaq, int tuning, hipStream_t st);
size_t a_b16_image_bytes(int64_t K, int64_t r);
int a_b16_prepare_dispatch(const void* a_t_limbs, int64_t K, int64_t r, void* out, hipStream_t st);
int act8_fused_dispatch(const void* x, int dtype, int64_t M, int64_t K, int64_t ldx, const QP& qx, void* xq_i8, const void* a_f16, int64_t r,
                        const QP& qa, bf16_t* xaq, int tuning, hipStream_t st, float* zero_p = nullptr, size_t zero_bytes = 0,
                        bool* zeroed = nullptr);  // (zero_p: gemm_amax_zero_bytes at the head of the scratch the GEMM call will be handed)
size_t a_f16_image_bytes(int64_t K, int64_t r);
int a_frag_dispatch(void* a_f16, int64_t K, int64_t r, hipStream_t st);
int f16_prepare_dispatch(const void* w_packed, int64_t N, int64_t K, const void* a_limbs_img, int a_limbs, int64_t r, void* a_f16,
                         int32_t* flags, hipStream_t st);
int split_act_dispatch(const void* x, int dtype, int64_t M, int64_t K, int64_t ldx, int limbs, bf16_t* xq, hipStream_t st);
size_t xa_scratch_bytes(int64_t m_max, int64_t K, int64_t rp);
int quant_xa_fused_dispatch(const void* x, int dtype, int64_t M, int64_t K, int64_t ldx, const QP& qx, bf16_t* xq,
                            const bf16_t* a_t, int a_limbs, int64_t r, const QP& qa, bf16_t* xaq, float* scratch,
                            size_t scratch_bytes, hipStream_t st);  // xaq == nullptr: leave the partial tiles in scratch
bool xa_fused_partials_ok(const QP& qx, const QP& qa, int64_t r);  // formats the GEMM can reduce itself (decode sizes)
void xa_fused_plan(int64_t M, int64_t K, int64_t r, int* nchunk, int64_t* cstride);
int gemm_dispatch(GemmArgs g, int dtype, bool lowrank, void* scratch, size_t scratch_bytes, hipStream_t st);
size_t gemm_scratch_bytes(int64_t m_max, int64_t N, const QP& bout);
size_t gemm_amax_zero_bytes(GemmArgs g, bool lowrank);
int gemm_route(const GemmArgs& g, bool lowrank);  // LQER_ROUTE_* the dispatch would take (or an error code)
int gemm_tile_rows(const GemmArgs& g);  // 128, or 64 for token counts that leave the 128-row grid thin (LQER_ROUTE_TILE128 family)
bool m256_eligible(const GemmArgs& g);  // gemm_w4a8_m256.hip: fewer (weighted) rounds with 256 x 256 tiles
int m256_dispatch(const GemmArgs& g, int dtype, bool lowrank, int bout, hipStream_t st);
bool i8_eligible(const GemmArgs& g, int bout);     // gemm_w4a8_i8.hip: the int8 MFMA main loop (g.w8 set, large M)
int i8_dispatch(const GemmArgs& g, int dtype, bool lowrank, int bout, hipStream_t st);
int i8_tile_rows(const GemmArgs& g);  // 256, or 128 where that takes fewer (weighted) rounds of one tile per CU
bool i8_amax_exchange_ok(const GemmArgs& g, bool lowrank, int bout);  // the int8 kernel can exchange the B_out row maxima itself (one round)
bool i8_amax_mrx_ok(const GemmArgs& g, bool lowrank, int bout);       // ... or compute and exchange them over several rounds (MRX)
int i8_prepare_dispatch(const void* w_packed, int64_t N, int64_t K, int mbits, void* w_i8, int32_t* flags, hipStream_t st);
int i8_unpack_dispatch(const void* w_i8, int64_t N, int64_t K, float* out, hipStream_t st, bool codes8 = false);
bool smallm_eligible(const GemmArgs& g, int bout);  // gemm_smallm.hip: M <= 64, B_out pass-through or blocks of 16
int smallm_dispatch(const GemmArgs& g, int dtype, bool lowrank, int bout, hipStream_t st);

// decode1.hip: the whole forward of M <= 8 tokens in one launch (LQER_E_UNSUPPORTED without launching when outside its shapes)
size_t decode1_scratch_bytes(int64_t Kp, int rp);
struct DecodeMember {  // one Linear of a one-launch decode forward (a group shares the tokens and the launch)
  const uint8_t* wp;
  const bf16_t* bt;
  const float* bias;
  void* y;
  int64_t ldy;
  int N, Np, rp, b_limbs;
};
int decode1_dispatch(GemmArgs g, int dtype, const void* x, int64_t ldx, int K, const QP& qx, const bf16_t* a_t, int bout,
                     const DecodeMember* mem, int nmem, void* scratch, size_t scratch_bytes, hipStream_t st);

size_t qmatmul_workspace_bytes(int64_t batch, int64_t K, int64_t S2);  // matmul_q.hip
size_t qmatmul_workspace_bytes_ex(int64_t batch, int64_t S1, int64_t K, int64_t S2, bool x_pre, bool y_pre);
int qmatmul_dispatch(const void* x, const void* y, void* out, int dtype, int64_t batch, int64_t S1, int64_t K, int64_t S2, int64_t x_bs,
                     int64_t x_rs, int64_t y_bs, int64_t y_ks, int64_t y_js, const QP& qx, const QP& qy, void* workspace, hipStream_t st);

// ---- kernel attributes (host) -----------------------------------------------------------------------
// Raising a kernel's dynamic-LDS limit is idempotent but not free: done once per kernel instantiation and device,
// thread-safely (the only process-wide state of the library besides the thread-local error text).
struct LdsLimitOnce {
  static constexpr int MAX_DEV = 64;
  std::once_flag flag[MAX_DEV];
  void set(const void* kernel, int bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::call_once(flag[dev & (MAX_DEV - 1)],
                   [&] { (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); });
  }
};

// ---- error plumbing (host) -------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

}  // namespace lqer
