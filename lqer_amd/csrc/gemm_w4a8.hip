// The fused W4 x A8 Linear kernel:
//
//   y[m,n] = sum_k xq[m,k] * Wq[n,k]  +  bq[n]  +  Q_Bout( sum_j xAq[m,j] * B[j,n] )
//
// replaces reference quantized_layers/linear.py:155-156 (torch.matmul(xA, B), B_out_quantizer,
// F.linear, add).  One workgroup = one 256(m) x 128(n) output tile, 8 waves as 4(m) x 2(n), each
// wave 64 x 64 = 2 x 2 tiles of v_mfma_f32_32x32x16_bf16 (bf16 holds every MXINT value
// m * 2^e, |m| < 256, exactly; fp32 accumulation - SURVEY.md §7 H1 strategy S1).
//
//  * prologue: the rank-r product xAq @ B runs on the MFMA straight from global memory, is
//    re-quantized in registers (16-lane DPP max = one B_out block of 16 output columns), the bias is
//    added, and the result is the INITIAL accumulator of the main loop - no epilogue pass.
//  * main loop, BK = 64, two LDS stages: the activation tile goes global -> LDS with
//    global_load_lds (16 B/lane, XOR swizzle applied on the source address); the weight tile is
//    loaded as 4-bit codes + block exponents (8 B + 1 B per lane), expanded to bf16 in registers and
//    written to LDS with the same swizzle.
//  * tiles are numbered so that the 8 XCDs each work on a contiguous run of tiles (same x rows ->
//    the activation slab stays in that XCD's L2).
#include "common.h"

namespace lqer {

constexpr int BM = 256, BN = 128, BK = 64;
constexpr int A_STAGE = BM * BK * 2;  // 32 KiB
constexpr int B_STAGE = BN * BK * 2;  // 16 KiB
constexpr int STAGE = A_STAGE + B_STAGE;
constexpr int GEMM_LDS = 2 * STAGE;  // 96 KiB

// byte offset of 16-byte chunk `c` (8 bf16 along k) of tile row `r`; rows are 128 B.
// chunk ^ ((row >> 1) & 7): the 16 lanes of a ds_read_b128 group then hit 16 distinct 16-B slots.
__device__ __forceinline__ int swz(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

// Expand 16 4-bit two's-complement codes (lo = k 0..7, hi = k 8..15) times 2^(e - mbits) to bf16.
__device__ __forceinline__ void expand16(uint32_t lo, uint32_t hi, int e, int mbits, uint32_t (&w)[8]) {
  int ef = e - mbits + 127;
  ef = ef < 1 ? 1 : ef;  // codes of such a block are all zero (|w| <= 1e-8 is flushed)
  const float scale = __uint_as_float((uint32_t)ef << 23);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t word = i < 4 ? lo : hi;
    const int sh = (2 * i) & 7;
    const int c0 = ((int)(word << (28 - 4 * sh))) >> 28;
    const int c1 = ((int)(word << (24 - 4 * sh))) >> 28;
    const uint32_t b0 = __float_as_uint((float)c0 * scale);
    const uint32_t b1 = __float_as_uint((float)c1 * scale);
    w[i] = (b0 >> 16) | (b1 & 0xffff0000u);
  }
}

template <int DT, bool LOWRANK, bool BOUT16>
__global__ __launch_bounds__(512) void k_lqer_gemm(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  // XCD-aware tile order: blocks b, b+8, ... share an XCD; give each XCD a contiguous tile range.
  const int nt = g.tiles_m * g.tiles_n;
  int tile;
  {
    const int b = blockIdx.x, xcd = b & 7, q8 = nt >> 3, r8 = nt & 7;
    tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
  }
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;

  // ---- staging helpers ------------------------------------------------------------------
  // A: wave w stages tile rows [32w, 32w+32): 4 x global_load_lds of 8 rows x 128 B.
  auto stage_a = [&](int kt, int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r0 = wave * 32 + i * 8;
      const int row = r0 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      const bf16_t* src = g.xq + (int64_t)(m0 + row) * g.Kp + kt * BK + chunk * 8;
      unsigned char* dst = smem + buf * STAGE + r0 * 128;
      __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)dst, 16, 0, 0);
    }
  };
  // W: wave w owns panel w of the tile (16 rows x 64 k): lane -> (row = lane/4, segment = lane%4).
  const uint8_t* wpanel0 = g.wp + ((int64_t)(n0 / 16 + wave) * (g.Kp / 64)) * LQER_PANEL_BYTES;
  auto load_w = [&](int kt, uint2& codes, int& e) {
    const uint8_t* p = wpanel0 + (int64_t)kt * LQER_PANEL_BYTES;
    codes = *(const uint2*)(p + lane * 8);
    e = (int)(int8_t)p[512 + lane];
  };
  auto store_w = [&](int buf, uint2 codes, int e) {
    uint32_t w[8];
    expand16(codes.x, codes.y, e, g.w_mbits, w);
    const int row = wave * 16 + (lane >> 2), seg = lane & 3;
    unsigned char* base = smem + buf * STAGE + A_STAGE;
    *(uint4*)(base + swz(row, 2 * seg)) = make_uint4(w[0], w[1], w[2], w[3]);
    *(uint4*)(base + swz(row, 2 * seg + 1)) = make_uint4(w[4], w[5], w[6], w[7]);
  };

  const int nk = g.Kp / BK;
  uint2 wc;
  int we;
  stage_a(0, 0);
  load_w(0, wc, we);

  // ---- low-rank prologue: acc = Q_Bout(xAq @ B) + bias ----------------------------------------
  if constexpr (LOWRANK) {
    for (int l = 0; l < g.b_limbs; ++l) {
      for (int ks = 0; ks < g.rp / 16; ++ks) {
        bf16x8 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[i] = *(const bf16x8*)(g.xaq + (int64_t)(m0 + wm * 64 + i * 32 + l31) * g.rp + ks * 16 + 8 * lh);
          b[i] = *(const bf16x8*)(g.bt + ((int64_t)l * g.Np + n0 + wn * 64 + i * 32 + l31) * g.rp + ks * 16 + 8 * lh);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    if constexpr (BOUT16) {
      // each DPP row of 16 lanes holds 16 consecutive output columns of one token row
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const float t = acc[i][j][k];
            const float amax = row16_max(fabsf(t));
            float qv = 0.f;
            if (amax > 0.f) {
              const int e = block_exponent(amax, g.bout);
              const float v = fabsf(t) + 1e-9f;
              const float m = fminf(rintf(ldexpf(v, g.bout.mbits - e)), g.bout.mmax);
              qv = fabsf(t) <= 1e-8f ? t : copysignf(ldexpf(m, e - g.bout.mbits), t);
            }
            acc[i][j][k] = qv;
          }
    }
  }
  if (g.bias) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float bv = g.bias[n0 + wn * 64 + j * 32 + l31];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[i][j][k] += bv;
    }
  }

  store_w(0, wc, we);
  __syncthreads();  // drains the global_load_lds of stage 0 (vmcnt) and publishes the W writes

  // ---- main loop ------------------------------------------------------------------------
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) {
      stage_a(kt + 1, cur ^ 1);
      load_w(kt + 1, wc, we);
    }
    const unsigned char* sa = smem + cur * STAGE;
    const unsigned char* sb = sa + A_STAGE;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = *(const bf16x8*)(sa + swz(wm * 64 + i * 32 + l31, 2 * ks + lh));
        b[i] = *(const bf16x8*)(sb + swz(wn * 64 + i * 32 + l31, 2 * ks + lh));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_w(cur ^ 1, wc, we);
    __syncthreads();
  }

  // ---- store: C layout col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) ----------
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int m = m0 + wm * 64 + i * 32 + (k & 3) + 8 * (k >> 2) + 4 * lh;
        if (m < g.M && n < g.N) store_elem<DT>(g.y, (int64_t)m * g.ldy + n, acc[i][j][k]);
      }
    }
}

template <int DT>
static int launch_gemm(const GemmArgs& g, bool lowrank, bool bout16, hipStream_t st) {
  const unsigned grid = (unsigned)(g.tiles_m * g.tiles_n);
#define LQER_GEMM_LAUNCH(LR, BO)                                                                          \
  do {                                                                                                    \
    static bool attr_done = false;                                                                        \
    if (!attr_done) {                                                                                     \
      (void)hipFuncSetAttribute((const void*)k_lqer_gemm<DT, LR, BO>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                          GEMM_LDS);                                                                      \
      attr_done = true;                                                                                   \
    }                                                                                                     \
    k_lqer_gemm<DT, LR, BO><<<grid, 512, GEMM_LDS, st>>>(g);                                              \
  } while (0)
  if (lowrank && bout16)
    LQER_GEMM_LAUNCH(true, true);
  else if (lowrank)
    LQER_GEMM_LAUNCH(true, false);
  else
    LQER_GEMM_LAUNCH(false, false);
#undef LQER_GEMM_LAUNCH
  return check_launch("lqer_gemm");
}

int gemm_dispatch(GemmArgs g, int dtype, bool lowrank, hipStream_t st) {
  if (g.M == 0 || g.N == 0) return LQER_OK;
  bool bout16 = false;
  if (lowrank) {
    if (g.bout.kind == LQER_Q_MXINT && g.bout.block == 16)
      bout16 = true;
    else if (g.bout.kind != LQER_Q_PASSTHROUGH) {
      set_error("B_out_quantizer block %d: the fused kernel re-quantizes blocks of 16 output columns only", g.bout.block);
      return LQER_E_UNSUPPORTED;
    }
  }
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = g.Np / BN;
  switch (dtype) {
    case LQER_F32: return launch_gemm<LQER_F32>(g, lowrank, bout16, st);
    case LQER_F16: return launch_gemm<LQER_F16>(g, lowrank, bout16, st);
    case LQER_BF16: return launch_gemm<LQER_BF16>(g, lowrank, bout16, st);
  }
  set_error("unknown dtype %d", dtype);
  return LQER_E_INVALID;
}

}  // namespace lqer
