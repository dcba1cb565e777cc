// The int8 route's activation side in ONE launch (round 5): x_quantizer with one exponent per token (linear.py:148), x A
// (linear.py:154) and A_out_quantizer - what used to be k_quant_row8 + k_xa_partial(_lds) + k_xa_reduce4, three launches of
// 5 - 9 us each at M = 2048 beside a 33-us GEMM.
//
// One workgroup = 16 token rows x ALL of K (K <= 8192), so nothing is split along K across workgroups: the row's exponent, the
// side product's sums and the A_out blocks are all complete inside the workgroup - no partial tiles in HBM, no reduce launch.
//   * wave w owns the k range [512 w, 512 w + 512) of the 16 rows.  Lane (r = lane & 15, g = lane >> 4) holds, for each of its 8
//     64-k pairs p, the two 8-element chunks 8 p + 2 g and 8 p + 2 g + 1 of row r: 32 contiguous bytes of the fp16 / bf16 row
//     (two 16-byte loads), all 16 loads issued up front - the row never leaves the registers between the maximum and the
//     quantization;
//   * row maximum: lanes r, r + 16, r + 32, r + 48, then the waves through LDS; every lane derives its row's exponent itself;
//   * quantization in fp32 (common.h arithmetic: fma on the signed value, 1.5 * 2^23 rounding, v_med3 clamp); the mantissas go
//     to the int8 image as ONE 16-byte store per pair (64 contiguous bytes per row and instruction) and, as exact fp16 (or
//     bf16) integers, ARE the lane's A-operand fragments of v_mfma_f32_16x16x32_f16 / _bf16: chunk c of row r in the k slots of
//     group g - the B operand (A^T, from L2) uses the same chunk in the same slots, so the k order inside a 64-k pair is a
//     permutation both operands share;
//   * the waves' partial [16 x rp] tiles are summed through LDS in ascending wave order (ascending k), scaled by the row's power
//     of two (exact) and re-quantized by A_out (blocks of 4 * 2^j rank entries; k_xa_reduce4's tail), bf16 image to xaq.
// Summation order of x A: within a wave ascending k-pair, limb-minor; then ascending wave.  (The three-launch path sums split-K
// chunks of another size: xAq agrees within the summation-order envelope of tests/_envelope.py, like every route of x A.)
#include "common.h"

namespace lqer {

namespace qxr {

constexpr int RB = 16;    // token rows per workgroup
constexpr int MAXP = 8;   // 64-k pairs per wave: 512 k
constexpr int WK = 64 * MAXP;

typedef __attribute__((ext_vector_type(2))) float f2;
typedef __attribute__((ext_vector_type(8))) _Float16 h8;

// A_out tail of one float4 of a row (k_xa_reduce4's arithmetic): the G lanes of a block share their maximum by xor-shuffles
template <int G>
__device__ __forceinline__ void finish4(float4 s, bool live, const QP& q, bf16_t* dst) {
  float amax = fmaxf(fmaxf(fabsf(s.x), fabsf(s.y)), fmaxf(fabsf(s.z), fabsf(s.w)));
#pragma unroll
  for (int d = 1; d < G; d <<= 1) amax = fmaxf(amax, __shfl_xor(amax, d, 64));
  if (!live) return;
  const bool any = amax > 0.f;
  const int e = any ? block_exponent(amax, q) : 0;
  const float v[4] = {s.x, s.y, s.z, s.w};
  uint32_t w[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float m0v = any ? mxint_mantissa(v[2 * i], e, q) : 0.f;
    const float m1v = any ? mxint_mantissa(v[2 * i + 1], e, q) : 0.f;
    w[i] = exact_bf16_bits(ldexpf(m0v, e - q.mbits)) | (exact_bf16_bits(ldexpf(m1v, e - q.mbits)) << 16);
  }
  *(uint2*)dst = make_uint2(w[0], w[1]);
}

template <int DT, bool AF16, int NRT, int MAXW>  // NRT rank tiles of 16 (padded rank <= 16 NRT); at most MAXW waves
__global__ __launch_bounds__(64 * MAXW) void k_quant_rows_xa(const void* __restrict__ x, int64_t M, int64_t K, int64_t ldx, QP qx,
                                                              int8_t* __restrict__ xq8, int64_t Kp8, float* __restrict__ xscale,
                                                              const bf16_t* __restrict__ a_img, int a_limbs, int64_t a_ld,
                                                              int64_t a_limb_stride, int rp, QP qa, int G4, bf16_t* __restrict__ xaq,
                                                              int64_t xaq_ld) {
  static_assert(DT != LQER_F32, "16-bit inputs");
  // A^T fragments in flight ahead of the pair being multiplied (registers: 8 NRT per pair): the L2 round trip of a pair's
  // fragments passes under the quantization of the pairs before it
  constexpr int AH = MAXW <= 8 ? (NRT <= 2 ? 4 : 2) : (MAXW <= 12 ? (NRT <= 2 ? 2 : 1) : 1);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_q[];
  float* const s_amax = (float*)smem_q;            // [nw][16]
  float* const s_part = (float*)(smem_q + 1024);   // [nw][16][rp]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = (int)(blockDim.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int64_t row = (int64_t)blockIdx.x * RB + r;
  const bool row_ok = row < M;
  const int nch = (int)(K / 8);
  const int c0 = wave * (8 * MAXP) + 2 * g;  // chunk of (pair p, half j): c0 + 8 p + j
  const u32x4* const xrow = (const u32x4*)((const bf16_t*)x + (row_ok ? row : 0) * ldx);
  u32x4 raw[MAXP][2];
#pragma unroll
  for (int p = 0; p < MAXP; ++p)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = c0 + 8 * p + j;
#ifdef QXR_NO_X  // (diagnostic build: no loads of x)
      raw[p][j] = (u32x4){(uint32_t)(0x3c003800u + c), 0x34003000u, 0xb800b400u, (uint32_t)lane};
#else
      raw[p][j] = (row_ok && c < nch) ? xrow[c] : (u32x4){0, 0, 0, 0};
#endif
    }
  // the A^T fragments of pair p (limb 0; further limbs - bf16 images of an fp16 A - are fetched on the spot below)
  const int nkc = (int)(a_ld / 8);  // chunks of an A^T row that exist (Kp / 8)
  const bf16_t* a_col[NRT];
#pragma unroll
  for (int t = 0; t < NRT; ++t) {
    const int col = t * 16 + r < rp ? t * 16 + r : rp - 1;  // (columns past the padded rank: a duplicate, never stored)
    a_col[t] = a_img + (int64_t)col * a_ld;
  }
  u32x4 af[AH][2][NRT];
  auto load_a = [&](int p, u32x4 (&dst)[2][NRT], int64_t limb_off) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = c0 + 8 * p + j;
#pragma unroll
#ifdef QXR_NO_A  // (diagnostic build: no loads of A^T)
      for (int t = 0; t < NRT; ++t) dst[j][t] = (u32x4){(uint32_t)c, 0x3c003c00u, (uint32_t)t, (uint32_t)limb_off};
#else
      for (int t = 0; t < NRT; ++t) dst[j][t] = c < nkc ? *(const u32x4*)(a_col[t] + limb_off + (int64_t)c * 8) : (u32x4){0, 0, 0, 0};
#endif
    }
  };
#pragma unroll
  for (int p = 0; p < AH; ++p) load_a(p, af[p], 0);
  // ---- the row's maximum -> exponent
  float am = 0.f;
  if constexpr (DT == LQER_F16) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    h2 m = {(_Float16)0.f, (_Float16)0.f};
#pragma unroll
    for (int p = 0; p < MAXP; ++p)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) m = __builtin_elementwise_max(m, __builtin_bit_cast(h2, raw[p][j][k] & 0x7fff7fffu));
    am = fmaxf((float)m[0], (float)m[1]);
  } else {
#pragma unroll
    for (int p = 0; p < MAXP; ++p)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k)
          am = fmaxf(am, fmaxf(__uint_as_float((raw[p][j][k] << 16) & 0x7fffffffu), __uint_as_float(raw[p][j][k] & 0x7fff0000u)));
  }
  am = fmaxf(am, __shfl_xor(am, 16, 64));
  am = fmaxf(am, __shfl_xor(am, 32, 64));
  if (g == 0) s_amax[wave * 16 + r] = am;
  __syncthreads();
  float amax = 0.f;
  for (int w = 0; w < nw; ++w) amax = fmaxf(amax, s_amax[w * 16 + r]);
  const bool any = amax > 0.f;
  const int e = any ? block_exponent(amax, qx) : 0;
  const float rscale = any ? ldexpf(1.0f, e - qx.mbits) : 1.0f;
  if (wave == 0 && g == 0 && row_ok) xscale[row] = rscale;
  // ---- per pair: quantize (fp32, common.h arithmetic), int8 image, MFMA.
  // (Per lane: the rows of a wave have their own exponents.  A row whose scale 2^(mbits - e) is not a normal float has
  // e < mbits - 126, i.e. every |x| < 2^-119 <= 1e-8: the element routine flushes all of it to 0 in a packed image - so does this.)
  // t = fma(x, 2^(mbits-e), +-1e-9 2^(mbits-e)) + 1.5 2^23 holds rne(.) in its low mantissa bits; clamped there (v_med3 against
  // 1.5 2^23 -+ mmax), its LOW BYTE is the two's-complement mantissa - one v_perm per pair of elements instead of two conversions
  // and a pack -, and t - 1.5 2^23 the integer the MFMA multiplies (exact in fp16 / bf16: |m| <= 127).
  const bool fast = mxint16_fast_ok(e, qx);
  const bool live_row = any && fast;
  const float s = live_row ? __uint_as_float((uint32_t)(127 + qx.mbits - e) << 23) : 0.0f;  // (dead rows: every product 0)
  const float es = 1e-9f * s;
  const float MG = 12582912.0f, tlo = MG - qx.mneg, thi = MG + qx.mmax;
  const f2 magic = {MG, MG};
  int8_t* const qrow = xq8 + (row_ok ? row : 0) * Kp8;
  const int nch_p = (int)(Kp8 / 8);
  f32x4 acc[NRT];
#pragma unroll
  for (int t = 0; t < NRT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    uint32_t bytes[4];
    u32x4 frag[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t wd[4] = {raw[p][j][0], raw[p][j][1], raw[p][j][2], raw[p][j][3]};
      uint32_t tb[4][2];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f2 xv;
        if constexpr (DT == LQER_F16) {
          typedef __attribute__((ext_vector_type(2))) _Float16 h2;
          const h2 hv = __builtin_bit_cast(h2, wd[k]);
          xv = (f2){(float)hv[0], (float)hv[1]};
        } else {
          xv = (f2){__uint_as_float(wd[k] << 16), __uint_as_float(wd[k] & 0xffff0000u)};
          // (bf16 holds non-zero |x| <= 1e-8: flushed in a packed image - s = 0 for them)
          xv[0] = fabsf(xv[0]) <= 1e-8f ? 0.0f : xv[0];
          xv[1] = fabsf(xv[1]) <= 1e-8f ? 0.0f : xv[1];
        }
        const f2 cc = {copysignf(es, xv[0]), copysignf(es, xv[1])};
        f2 tt = __builtin_elementwise_fma(xv, (f2){s, s}, cc) + magic;
        tt[0] = __builtin_amdgcn_fmed3f(tt[0], tlo, thi);
        tt[1] = __builtin_amdgcn_fmed3f(tt[1], tlo, thi);
        tb[k][0] = __float_as_uint(tt[0]), tb[k][1] = __float_as_uint(tt[1]);
        const f2 rr = tt - magic;
        if constexpr (AF16) frag[j][k] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(rr[0], rr[1]));
        else frag[j][k] = __builtin_amdgcn_perm(__float_as_uint(rr[1]), __float_as_uint(rr[0]), 0x07060302u);  // bf16: the high halves
      }
      // low bytes of the eight clamped t: elements 0..3 -> word 2 j, 4..7 -> word 2 j + 1
      const uint32_t b01 = __builtin_amdgcn_perm(tb[0][1], tb[0][0], 0x0c0c0400u), b23 = __builtin_amdgcn_perm(tb[1][1], tb[1][0], 0x0c0c0400u);
      const uint32_t b45 = __builtin_amdgcn_perm(tb[2][1], tb[2][0], 0x0c0c0400u), b67 = __builtin_amdgcn_perm(tb[3][1], tb[3][0], 0x0c0c0400u);
      bytes[2 * j] = b01 | (b23 << 16);
      bytes[2 * j + 1] = b45 | (b67 << 16);
    }
    const int c = c0 + 8 * p;  // (Kp8 is a multiple of 128: a pair lies inside the padded row or outside it)
    if (row_ok && c < nch_p) *(u32x4*)(qrow + (int64_t)c * 8) = (u32x4){bytes[0], bytes[1], bytes[2], bytes[3]};
    // x A: this pair's two 32-k steps, NRT rank tiles (one image of fp16, or up to three bf16 limbs: products exact either way)
    for (int l = 0; l < a_limbs; ++l) {
      u32x4 al[2][NRT];
      if (l > 0) load_a(p, al, (int64_t)l * a_limb_stride);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int t = 0; t < NRT; ++t) {
          const u32x4 bfrag = l == 0 ? af[p % AH][j][t] : al[j][t];
          if constexpr (AF16)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, frag[j]), __builtin_bit_cast(h8, bfrag), acc[t], 0, 0, 0);
          else
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, frag[j]), __builtin_bit_cast(bf16x8, bfrag), acc[t], 0,
                                                             0, 0);
        }
    }
    if (p + AH < MAXP) load_a(p + AH, af[p % AH], 0);
    __builtin_amdgcn_sched_barrier(0);  // (pair by pair: interleaved, every pair's temporaries would be live at once)
  }
  // ---- the waves' partial tiles -> LDS; lane: column t 16 + r of rows 4 g + i
#pragma unroll
  for (int t = 0; t < NRT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int col = t * 16 + r;
      if (col < rp) s_part[((int64_t)wave * 16 + 4 * g + i) * rp + col] = acc[t][i];
    }
  __syncthreads();
  // ---- ascending-wave sum, the row's scale, A_out, bf16 image.  One thread per 4 consecutive rank entries of a row
  const int rq = rp / 4, items = 16 * rq;
  for (int it = tid; it < ((items + 63) / 64) * 64; it += (int)blockDim.x) {  // (whole waves: the block maximum is shared by shuffles)
    const bool live_it = it < items;
    const int rr = live_it ? it / rq : 0, q4 = live_it ? it - rr * rq : 0;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int w = 0; w < nw; ++w) {
      const float4 v = *(const float4*)(s_part + ((int64_t)w * 16 + rr) * rp + 4 * q4);
      sum.x += v.x, sum.y += v.y, sum.z += v.z, sum.w += v.w;
    }
    // (the row's scale from its maximum, as above - recomputed for THIS row)
    float am_r = 0.f;
    for (int w = 0; w < nw; ++w) am_r = fmaxf(am_r, s_amax[w * 16 + rr]);
    const float sc = am_r > 0.f ? ldexpf(1.0f, block_exponent(am_r, qx) - qx.mbits) : 1.0f;
    sum.x *= sc, sum.y *= sc, sum.z *= sc, sum.w *= sc;
    const int64_t orow = (int64_t)blockIdx.x * RB + rr;
    const bool live = live_it && orow < M;
    bf16_t* const dst = xaq + orow * xaq_ld + 4 * q4;
    switch (G4) {
      case 1: finish4<1>(sum, live, qa, dst); break;
      case 2: finish4<2>(sum, live, qa, dst); break;
      case 4: finish4<4>(sum, live, qa, dst); break;
      case 8: finish4<8>(sum, live, qa, dst); break;
      default: finish4<16>(sum, live, qa, dst); break;
    }
  }
}

template <int DT, bool AF16, int NRT, int MAXW>
static void launch(const void* x, int64_t M, int64_t K, int64_t ldx, const QP& qx, int8_t* xq8, int64_t Kp8, float* xscale,
                   const bf16_t* a_img, int a_limbs, int64_t a_ld, int64_t a_limb_stride, int rp, const QP& qa, int G4, bf16_t* xaq,
                   int64_t xaq_ld, int nw, hipStream_t st) {
  const size_t lds = 1024 + (size_t)nw * 16 * rp * sizeof(float);
  static LdsLimitOnce lds_once;
  lds_once.set((const void*)k_quant_rows_xa<DT, AF16, NRT, MAXW>, 1024 + 16 * 16 * 64 * 4);
  k_quant_rows_xa<DT, AF16, NRT, MAXW><<<(unsigned)((M + RB - 1) / RB), 64 * nw, lds, st>>>(x, M, K, ldx, qx, xq8, Kp8, xscale, a_img,
                                                                                           a_limbs, a_ld, a_limb_stride, rp, qa, G4, xaq,
                                                                                           xaq_ld);
}

}  // namespace qxr

// The int8 route's quantize_act_xa in one launch.  LQER_E_UNSUPPORTED (nothing launched) outside: 16-bit tensors with 16-byte
// aligned rows, K a multiple of 8 and <= 8192, padded rank <= 64, A_out block_fp (<= 9 bits) in blocks of 4 * 2^j entries that tile
// the padded rank.  a_limbs = -1: a_img is ONE fp16 image [rp][Kp]; 1..3: bf16 limbs [limb][rp][Kp].
int quant_rows_xa_dispatch(const void* x, int dtype, int64_t M, int64_t K, int64_t ldx, const QP& qx, void* xq8_img, const bf16_t* a_img,
                           int a_limbs, int64_t r, const QP& qa, bf16_t* xaq, int64_t xaq_ld, hipStream_t st) {
  const int rp = (int)lqer_padded_r(r);
  if (dtype == LQER_F32 || K % 8 != 0 || K > 16 * qxr::WK || K < 8 || rp > 64 || r <= 0 || !a_img || !xaq) return LQER_E_UNSUPPORTED;
  if (((uintptr_t)x & 15) || (ldx * 2) % 16 != 0 || qx.mbits > 7 || qx.kind != LQER_Q_MXINT) return LQER_E_UNSUPPORTED;
  if (qa.kind != LQER_Q_MXINT || qa.mbits > 8) return LQER_E_UNSUPPORTED;
  if (!(a_limbs == -1 || (a_limbs >= 1 && a_limbs <= 3))) return LQER_E_UNSUPPORTED;
  const int La = (qa.block <= 0 || qa.block >= rp) ? rp : qa.block;
  const int G4 = La / 4;
  if (rp % La != 0 || La % 4 != 0 || (G4 & (G4 - 1)) != 0 || G4 > 16) return LQER_E_UNSUPPORTED;
  if (M == 0) return LQER_OK;
  const int64_t Kp = lqer_padded_k(K), Kp8 = padded_k8(K);
  int8_t* const xq8 = (int8_t*)xq8_img;
  float* const xscale = const_cast<float*>(i8_row_scales(xq8_img, M, K));
  const int nw = (int)((Kp8 + qxr::WK - 1) / qxr::WK);  // (covers the zero padding of the image up to Kp8)
  const bool f16 = a_limbs == -1;
  const int limbs = f16 ? 1 : a_limbs;
  const int64_t lstride = (int64_t)rp * Kp;
  const int nrt = rp <= 32 ? 2 : 4;
#define QXR(DT, AF, NRT, MW) qxr::launch<DT, AF, NRT, MW>(x, M, K, ldx, qx, xq8, Kp8, xscale, a_img, limbs, Kp, lstride, rp, qa, G4, xaq, xaq_ld, nw, st)
#define QXR_W(DT, AF, NRT) (nw <= 8 ? QXR(DT, AF, NRT, 8) : (nw <= 12 ? QXR(DT, AF, NRT, 12) : QXR(DT, AF, NRT, 16)))
#define QXR_N(DT, AF) (nrt == 2 ? QXR_W(DT, AF, 2) : QXR_W(DT, AF, 4))
  if (dtype == LQER_F16) {
    if (f16) QXR_N(LQER_F16, true); else QXR_N(LQER_F16, false);
  } else {
    if (f16) QXR_N(LQER_BF16, true); else QXR_N(LQER_BF16, false);
  }
#undef QXR_N
#undef QXR_W
#undef QXR
  return check_launch("quantize_rows_xa");
}

}  // namespace lqer
