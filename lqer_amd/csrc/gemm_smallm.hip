// Small-M (decode) variant of the fused W4 x A8 Linear kernel, M <= 64 tokens:
//
//   y[m,n] = sum_k xq[m,k] * Wq[n,k]  +  bq[n]  +  Q_Bout( sum_j xAq[m,j] * B[j,n] )
//
// (reference quantized_layers/linear.py:155-156, same arithmetic as gemm_w4a8.hip).  At these sizes the
// packed weight (0.56 B per element) is the only large operand and the kernel is HBM-bound on it, so the
// tile kernel's 128-row tiles would leave all but N/256 CUs idle.  Here ONE workgroup owns one panel row =
// 16 output columns x all of K: nk panels of 576 B that are contiguous in the packed image, streamed straight
// into registers (no LDS staging: nothing is reused across waves) by 8 waves that take every 8th panel;
// N/16 workgroups (256 for N = 4096) put every CU on the stream.
//
// Arithmetic: v_mfma_f32_16x16x32_bf16, A operand = 16 weight rows x 32 k (lane: row l & 15, k 8 (l >> 4) ..+8 =
// exactly one 32-bit word of 4-bit codes -> one expand per MFMA), B operand = 16 tokens x 32 k from the bf16
// activation image (L2-resident: M x K x 2 B), D = 16 (n) x 16 (token) fp32, lane: token l & 15, n 4 (l >> 4) + j.
// A B_out block (16 consecutive n of a token) is the 4 registers of the 4 lanes l, l^16, l^32, l^48.
// The 8 waves' partial sums are added in a fixed order through LDS (bit-reproducible); wave 0 adds the side
// path (MFMA straight from global memory, re-quantized in registers), the bias, and stores.
#include "common.h"

namespace lqer {

constexpr int SM_MAX_M = 64;   // tokens handled by this kernel (MT = ceil(M / 16) <= 4 token tiles)
constexpr int SM_NW = 8;       // waves per workgroup: wave w takes panels w, w + 8, ...
template <int MT> struct SmUnr { static constexpr int v = MT <= 2 ? 4 : 2; };  // panels per register buffer (two buffers)

__device__ __forceinline__ float quad16_max(float v) {  // max over lanes l, l^16, l^32, l^48 (no LDS crossbar: the swaps)
  auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
  auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
}

struct SmPanel {
  u32x4 cw;      // the lane's 16-byte half of its row's codes: chunks {h, h+2, h+4, h+6}, h = (lane >> 4) & 1
  uint32_t ex;   // the row's 4 biased block exponents
};

template <int DT, bool LOWRANK, int BOUT, int MT>
__global__ __launch_bounds__(64 * SM_NW) void k_lqer_gemm_smallm(GemmArgs g) {
  constexpr bool XF16 = DT == LQER_F16X;  // fp16 activation image and fp16 weight fragments in the main loop (gemm_w4a8.hip)
  __shared__ __attribute__((aligned(16))) float red[(SM_NW - 1) * MT * 4 * 64];
  __shared__ __attribute__((aligned(16))) bf16_t xaq_l[LOWRANK ? MT * 16 * 64 : 8];  // [token][rp <= 64]: x A after A_out (partials mode)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row = lane & 15, q = lane >> 4;  // weight row / token within the tile; k group
  const int n0 = blockIdx.x * 16;
  const int nk = g.Kp / 64;
  const uint8_t* prow = g.wp + (int64_t)blockIdx.x * nk * LQER_PANEL_BYTES;
  // the lane's words for the two 32-deep halves of a panel: chunks q and q + 4 = words q >> 1 and (q >> 1) + 2 of
  // the half-row it loads; their block exponents are bytes q >> 1 and 2 + (q >> 1)
  const int codes_off = row * 32 + (q & 1) * 16;
  const int exps_off = 512 + row * 4;
  const bool hi = (q >> 1) != 0;
  const int sh0 = 8 * (q >> 1), sh1 = 16 + 8 * (q >> 1);

  f32x4 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // wave 0 fetches the first side-path fragments (limb 0, rank entries 0..31) now, under the weight stream
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  bf16x8 sp_b = zero8, sp_x[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) sp_x[t] = zero8;
  const bool from_partials = LOWRANK && g.xa_part != nullptr;
  if (LOWRANK && wave == 0 && g.b_limbs > 0 && 8 * q < g.rp) {
    sp_b = *(const bf16x8*)(g.bt + (int64_t)(n0 + row) * g.rp + 8 * q);
    if (!from_partials) {
#pragma unroll
      for (int t = 0; t < MT; ++t)
        if (t * 16 + row < g.M) sp_x[t] = *(const bf16x8*)(g.xaq + (int64_t)(t * 16 + row) * g.xaq_ld + 8 * q);
    }
  }
  auto load_panel = [&](int kt, SmPanel& p, bf16x8 (&x)[MT][2]) {
    const uint8_t* pp = prow + (int64_t)kt * LQER_PANEL_BYTES;
    p.cw = *(const u32x4*)(pp + codes_off);
    p.ex = *(const uint32_t*)(pp + exps_off);
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      // token rows beyond M load nothing: the activation image (M x K x 2 B, read by every workgroup from L2) is
      // the larger stream of this kernel; their output columns are never stored
      x[t][0] = x[t][1] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (t * 16 + row < g.M) {
        const bf16_t* xr = g.xq + (int64_t)(t * 16 + row) * g.Kp + kt * 64 + 8 * q;
        x[t][0] = *(const bf16x8*)xr;
        x[t][1] = *(const bf16x8*)(xr + 32);
      }
    }
  };
  auto compute_panel = [&](const SmPanel& p, const bf16x8 (&x)[MT][2]) {
    const uint32_t w0 = hi ? p.cw[1] : p.cw[0], w1 = hi ? p.cw[3] : p.cw[2];
    const bf16x8 wb0 = expand_frag_t<XF16>(w0, ((p.ex >> sh0) & 0xffu) << 23);
    const bf16x8 wb1 = expand_frag_t<XF16>(w1, ((p.ex >> sh1) & 0xffu) << 23);
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      acc[t] = mfma_16x16x32<XF16>(wb0, x[t][0], acc[t]);
      acc[t] = mfma_16x16x32<XF16>(wb1, x[t][1], acc[t]);
    }
  };

  // wave w takes panels w, w + SM_NW, ...; two register buffers of SM_UNR panels keep 2 x SM_UNR panels of loads in flight
  constexpr int SM_UNR = SmUnr<MT>::v;
  SmPanel pa[SM_UNR], pb[SM_UNR];
  bf16x8 xa[SM_UNR][MT][2], xb[SM_UNR][MT][2];
  const int per_wave = (nk - wave + SM_NW - 1) / SM_NW;  // panels of this wave
  auto kt_of = [&](int i) { return wave + SM_NW * i; };
  int i = 0;
#pragma unroll
  for (int u = 0; u < SM_UNR; ++u)
    if (u < per_wave) load_panel(kt_of(u), pa[u], xa[u]);
  if constexpr (LOWRANK) {
    if (from_partials) {
      // (after the first weight panels have been requested: the partial tiles' round trip passes under the weight stream's)
      // x A = sum over the split-K chunks, ascending (the order of k_xa_reduce4: same bits as the three-launch route),
      // then A_out in blocks of 16 = 4 consecutive lanes; the bf16 image goes to LDS for wave 0's epilogue (it is read
      // after the combine barrier below).  One lane per 4 rank entries; M * rp / 4 <= 1024 items on 512 lanes.
      const int total = g.M * g.rp / 4;
      for (int base = 0; base < total; base += 64 * SM_NW) {
        const int item = base + (int)threadIdx.x;
        const bool live = item < total;
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) {
          const float* src = g.xa_part + (int64_t)item * 4;
          int c = 0;
          for (; c + 8 <= g.xa_nchunk; c += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *(const float4*)(src + (c + u) * g.xa_cstride);
#pragma unroll
            for (int u = 0; u < 8; ++u) sum.x += v[u].x, sum.y += v[u].y, sum.z += v[u].z, sum.w += v[u].w;
          }
          for (; c < g.xa_nchunk; ++c) {
            const float4 v = *(const float4*)(src + c * g.xa_cstride);
            sum.x += v.x, sum.y += v.y, sum.z += v.z, sum.w += v.w;
          }
        }
        float amax = fmaxf(fmaxf(fabsf(sum.x), fabsf(sum.y)), fmaxf(fabsf(sum.z), fabsf(sum.w)));
        amax = fmaxf(amax, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(amax), 0xB1, 0xf, 0xf, true)));  // quad_perm [1,0,3,2]
        amax = fmaxf(amax, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(amax), 0x4E, 0xf, 0xf, true)));  // quad_perm [2,3,0,1]
        if (live) {
          const bool any = amax > 0.f;
          const int e = any ? block_exponent(amax, g.aout) : 0;
          const float v[4] = {sum.x, sum.y, sum.z, sum.w};
          uint32_t w[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const float q0 = any && fabsf(v[2 * i]) > g.aout.tiny ? mxint_value(v[2 * i], e, g.aout) : 0.f;
            const float q1 = any && fabsf(v[2 * i + 1]) > g.aout.tiny ? mxint_value(v[2 * i + 1], e, g.aout) : 0.f;
            w[i] = exact_bf16_bits(q0) | (exact_bf16_bits(q1) << 16);
          }
          *(uint2*)(xaq_l + item * 4) = make_uint2(w[0], w[1]);
        }
      }
    }
  }

  for (; i < per_wave; i += 2 * SM_UNR) {
#pragma unroll
    for (int u = 0; u < SM_UNR; ++u)
      if (i + SM_UNR + u < per_wave) load_panel(kt_of(i + SM_UNR + u), pb[u], xb[u]);
#pragma unroll
    for (int u = 0; u < SM_UNR; ++u)
      if (i + u < per_wave) compute_panel(pa[u], xa[u]);
#pragma unroll
    for (int u = 0; u < SM_UNR; ++u)
      if (i + 2 * SM_UNR + u < per_wave) load_panel(kt_of(i + 2 * SM_UNR + u), pa[u], xa[u]);
#pragma unroll
    for (int u = 0; u < SM_UNR; ++u)
      if (i + SM_UNR + u < per_wave) compute_panel(pb[u], xb[u]);
  }

  // fixed-order combine: (((w0 + w1) + w2) + ...) + w7
  if (wave > 0) {
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[(((wave - 1) * MT + t) * 4 + j) * 64 + lane] = acc[t][j];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int w2 = 0; w2 < SM_NW - 1; ++w2) acc[t][j] += red[((w2 * MT + t) * 4 + j) * 64 + lane];

  // ---- side path + bias + store (wave 0): lane = token 16 t + row, output columns n0 + 4 q + j
  const int nq = n0 + 4 * q;
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if constexpr (LOWRANK) {
      const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
      if (from_partials && 8 * q < g.rp && t * 16 + row < g.M) sp_x[t] = *(const bf16x8*)(xaq_l + (t * 16 + row) * g.rp + 8 * q);
      if (g.b_limbs > 0) s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sp_b, sp_x[t], s, 0, 0, 0);  // prefetched
      for (int l = 0; l < g.b_limbs; ++l)
        for (int ks = (l == 0 ? 1 : 0); ks * 32 < g.rp; ++ks) {
          const int j0 = ks * 32 + 8 * q;  // rp is a multiple of 16: the upper half of the last 32 may lie beyond it
          bf16x8 bb = zero, xv = zero;
          if (j0 < g.rp) {
            bb = *(const bf16x8*)(g.bt + ((int64_t)l * g.Np + n0 + row) * g.rp + j0);
            if (t * 16 + row < g.M)
              xv = from_partials ? *(const bf16x8*)(xaq_l + (t * 16 + row) * g.rp + j0)
                                 : *(const bf16x8*)(g.xaq + (int64_t)(t * 16 + row) * g.xaq_ld + j0);
          }
          s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb, xv, s, 0, 0, 0);
        }
      if constexpr (BOUT == 1) {
        float amax = fmaxf(fmaxf(fabsf(s[0]), fabsf(s[1])), fmaxf(fabsf(s[2]), fabsf(s[3])));
        amax = quad16_max(amax);
        const int e = block_exponent(amax, g.bout);  // amax = 0: every element takes the pass-through
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] = fabsf(s[j]) <= 1e-8f ? s[j] : mxint_value(s[j], e, g.bout);
      }
    }
    // same association as the tile kernel: the side path and the bias form the initial accumulator
    float out[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = (s[j] + (g.bias ? g.bias[nq + j] : 0.f)) + acc[t][j];
    const int m = t * 16 + row;
    if (m < g.M) store_row4<DT>(g.y, (int64_t)m * g.ldy + nq, nq, g.N, out);
  }
}

bool smallm_eligible(const GemmArgs& g, int bout) { return g.M <= SM_MAX_M && bout <= 1; }

template <int DT>
static int launch_smallm(const GemmArgs& g, bool lowrank, int bout, hipStream_t st) {
  const unsigned grid = (unsigned)(g.Np / 16);
  const int mt = (g.M + 15) / 16;
#define SM_LAUNCH(LR, BO)                                                           \
  switch (mt) {                                                                     \
    case 1: k_lqer_gemm_smallm<DT, LR, BO, 1><<<grid, 64 * SM_NW, 0, st>>>(g); break;      \
    case 2: k_lqer_gemm_smallm<DT, LR, BO, 2><<<grid, 64 * SM_NW, 0, st>>>(g); break;      \
    case 3: k_lqer_gemm_smallm<DT, LR, BO, 3><<<grid, 64 * SM_NW, 0, st>>>(g); break;      \
    default: k_lqer_gemm_smallm<DT, LR, BO, 4><<<grid, 64 * SM_NW, 0, st>>>(g); break;     \
  }
  if (!lowrank) {
    SM_LAUNCH(false, 0)
  } else if (bout == 1) {
    SM_LAUNCH(true, 1)
  } else {
    SM_LAUNCH(true, 0)
  }
#undef SM_LAUNCH
  return check_launch("lqer_gemm_smallm");
}

int smallm_dispatch(const GemmArgs& g, int dtype, bool lowrank, int bout, hipStream_t st) {
  switch (dtype) {
    case LQER_F32: return launch_smallm<LQER_F32>(g, lowrank, bout, st);
    case LQER_F16: return g.x_f16 ? launch_smallm<LQER_F16X>(g, lowrank, bout, st) : launch_smallm<LQER_F16>(g, lowrank, bout, st);
    case LQER_BF16: return launch_smallm<LQER_BF16>(g, lowrank, bout, st);
  }
  set_error("unknown dtype %d", dtype);
  return LQER_E_INVALID;
}

}  // namespace lqer
