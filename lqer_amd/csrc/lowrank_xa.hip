// Rank-r side path, first half:  xAq = A_out_quantizer( Q_x(x) @ A )   (reference
// quantized_layers/linear.py:154).  [M,K] x [K,r] with r << K: a skinny GEMM that streams the
// quantized activation once (HBM/L2 bound), so one workgroup owns 16 token rows and the whole
// rank; its 4 waves interleave 32-deep k-steps on v_mfma_f32_16x16x32_bf16 and combine through LDS.
// A comes as exact bf16 limbs (pack.hip), x as the exact bf16 image of the activation quantizer,
// so every product is exact in fp32 and only the fp32 accumulation order differs from the
// reference's torch.matmul.
#include "common.h"

namespace lqer {

constexpr int XA_ROWS = 16;
constexpr int XA_MAX_TILES = 16;  // rp <= 256

template <int NT>
__global__ __launch_bounds__(256) void k_lowrank_xa(const bf16_t* __restrict__ xq, int64_t Kp,
                                                    const bf16_t* __restrict__ a_t, int a_limbs, int rp, QP q,
                                                    bf16_t* __restrict__ xaq) {
  extern __shared__ __attribute__((aligned(16))) float part[];  // [4][16][rp]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t m0 = (int64_t)blockIdx.x * XA_ROWS;
  const int fr = lane & 15, fq = lane >> 4;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const bf16_t* xrow = xq + (m0 + fr) * Kp + 8 * fq;
  const int steps = (int)(Kp / 32);
  for (int s = wave; s < steps; s += 4) {
    const bf16x8 a = *(const bf16x8*)(xrow + s * 32);
    for (int l = 0; l < a_limbs; ++l) {
      const bf16_t* at = a_t + ((int64_t)l * rp + fr) * Kp + s * 32 + 8 * fq;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const bf16x8 b = *(const bf16x8*)(at + (int64_t)t * 16 * Kp);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);
      }
    }
  }
  // C layout: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) part[(wave * XA_ROWS + 4 * fq + j) * rp + t * 16 + fr] = acc[t][j];
  __syncthreads();
  for (int i = threadIdx.x; i < XA_ROWS * rp; i += 256)
    part[i] = (part[i] + part[XA_ROWS * rp + i]) + (part[2 * XA_ROWS * rp + i] + part[3 * XA_ROWS * rp + i]);
  __syncthreads();
  // A_out quantizer: one lane per (row, block of L along r)
  const int L = (q.block <= 0 || q.block >= rp) ? rp : q.block;
  const int nb = rp / L;
  for (int i = threadIdx.x; i < XA_ROWS * nb; i += 256) {
    const int row = i / nb, b0 = (i - row * nb) * L;
    const float* src = part + row * rp + b0;
    float amax = 0.f;
    for (int k = 0; k < L; ++k) amax = fmaxf(amax, fabsf(src[k]));
    const bool any = amax > 0.f;
    const int e = any ? block_exponent(amax, q) : 0;
    bf16_t* dst = xaq + (m0 + row) * rp + b0;
    for (int k = 0; k < L; k += 2) {
      const float m0v = any ? mxint_mantissa(src[k], e, q) : 0.f;
      const float m1v = any ? mxint_mantissa(src[k + 1], e, q) : 0.f;
      *(uint32_t*)(dst + k) =
          exact_bf16_bits(ldexpf(m0v, e - q.mbits)) | (exact_bf16_bits(ldexpf(m1v, e - q.mbits)) << 16);
    }
  }
}

int lowrank_xa_dispatch(const bf16_t* xq, int64_t M, int64_t K, const bf16_t* a_t, int a_limbs, int64_t r,
                        const QP& q, bf16_t* xaq, hipStream_t st) {
  const int64_t Kp = lqer_padded_k(K);
  const int rp = (int)lqer_padded_r(r);
  if (q.kind != LQER_Q_MXINT || q.mbits > 8) {
    set_error("A_out_quantizer must be block_fp with width <= 9 on the HIP path (got kind %d width %d)", q.kind,
              q.mbits + 1);
    return LQER_E_UNSUPPORTED;
  }
  const int L = (q.block <= 0 || q.block >= rp) ? rp : q.block;
  if (rp % L != 0 || L % 2 != 0) {
    set_error("A_out_quantizer block %d does not tile the padded rank %d", q.block, rp);
    return LQER_E_UNSUPPORTED;
  }
  if (rp > 16 * XA_MAX_TILES) {
    set_error("rank %d > %d not supported", (int)r, 16 * XA_MAX_TILES);
    return LQER_E_UNSUPPORTED;
  }
  if (M == 0) return LQER_OK;
  const unsigned grid = (unsigned)((M + XA_ROWS - 1) / XA_ROWS);
  const size_t lds = (size_t)4 * XA_ROWS * rp * sizeof(float);
#define XA_CASE(NT)                                                                         \
  case NT:                                                                                  \
    k_lowrank_xa<NT><<<grid, 256, lds, st>>>(xq, Kp, a_t, a_limbs, rp, q, xaq);             \
    break;
  switch (rp / 16) {
    XA_CASE(1) XA_CASE(2) XA_CASE(3) XA_CASE(4) XA_CASE(5) XA_CASE(6) XA_CASE(7) XA_CASE(8)
    XA_CASE(9) XA_CASE(10) XA_CASE(11) XA_CASE(12) XA_CASE(13) XA_CASE(14) XA_CASE(15) XA_CASE(16)
  }
#undef XA_CASE
  return check_launch("lowrank_xa");
}

}  // namespace lqer
