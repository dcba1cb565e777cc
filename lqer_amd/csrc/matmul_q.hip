// Quantized attention products (reference quantized_functions/matmul.py:12-37; call sites models/llama_decoder.py:263,294,
// opt_decoder.py:125,190):      out[b] = x_quantizer(x[b]) @ w_quantizer(y[b])
// with block_fp quantizers whose blocks of 16 run along the LAST dim of each operand: for x [.., S1, K] that is the
// contraction dim, for y [.., K, S2] it is NOT (llama-7b.toml:110-126) - in Q K^T the 16 elements of a block of y = K^T
// are 16 consecutive tokens of one head feature.
//
//   k_qmm_bimage_*   y -> bf16 image [b][j][k] of w_quantizer(y) (every 8-bit MXINT value is exact in bf16), k contiguous -
//                    the layout an MFMA operand fragment reads - whatever y's own layout: j-contiguous (P V: y = V
//                    [tokens, head_dim], transposed through LDS) or k-contiguous (Q K^T: y is the transposed VIEW of K; one
//                    thread holds 16 consecutive tokens x 8 features in registers, so the block maxima need no cross-lane work)
//   k_qmatmul        batched GEMM on v_mfma_f32_32x32x16_bf16 whose first operand is quantized IN THE LOAD PATH: a thread
//                    loads one block of 16 (32 B, four threads cover 128 B of a row), quantizes it in registers and writes
//                    the bf16 image to an LDS slab - x (the attention probabilities in P V: the big operand) is read from HBM
//                    exactly once and no quantized copy of it ever exists in HBM.  fp32 accumulation of exact products.
#include <type_traits>

#include "common.h"

namespace lqer {

namespace qmm {

constexpr int BM = 128, BN = 128, BK = 64;

__device__ __forceinline__ int swz(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

// 16 consecutive elements (or fewer at the end of a row), 128-bit loads when the piece is aligned and complete
template <int DT>
__device__ __forceinline__ void load16(const void* base, int64_t off, int64_t valid, bool vec, float (&v)[16]) {
  if (vec && valid >= 16) {
    if constexpr (DT == LQER_F32) {
      const float4* p = (const float4*)((const float*)base + off);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 t = p[i];
        v[4 * i] = t.x, v[4 * i + 1] = t.y, v[4 * i + 2] = t.z, v[4 * i + 3] = t.w;
      }
    } else {
      const uint4* p = (const uint4*)((const bf16_t*)base + off);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const uint4 t = p[i];
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (DT == LQER_F16) {
            typedef __attribute__((ext_vector_type(2))) _Float16 h2;
            const h2 h = __builtin_bit_cast(h2, w[j]);
            v[8 * i + 2 * j] = (float)h[0], v[8 * i + 2 * j + 1] = (float)h[1];
          } else {
            v[8 * i + 2 * j] = __uint_as_float(w[j] << 16), v[8 * i + 2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
          }
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = i < valid ? load_elem<DT>(base, off + i) : 0.0f;
  }
}

// 16 consecutive elements as raw 16-byte loads (the aligned fast path: requested chunks ahead, converted at use)
template <int DT>
struct Raw16 {
  static constexpr int N = DT == LQER_F32 ? 4 : 2;
  uint4 r[N];
};
template <int DT>
__device__ __forceinline__ void raw_to_f32(const Raw16<DT>& raw, float (&v)[16]) {
  if constexpr (DT == LQER_F32) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[4 * i] = __uint_as_float(raw.r[i].x), v[4 * i + 1] = __uint_as_float(raw.r[i].y);
      v[4 * i + 2] = __uint_as_float(raw.r[i].z), v[4 * i + 3] = __uint_as_float(raw.r[i].w);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint32_t w[4] = {raw.r[i].x, raw.r[i].y, raw.r[i].z, raw.r[i].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (DT == LQER_F16) {
          typedef __attribute__((ext_vector_type(2))) _Float16 h2;
          const h2 h = __builtin_bit_cast(h2, w[j]);
          v[8 * i + 2 * j] = (float)h[0], v[8 * i + 2 * j + 1] = (float)h[1];
        } else {
          v[8 * i + 2 * j] = __uint_as_float(w[j] << 16), v[8 * i + 2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
        }
      }
    }
  }
}

// one block of 16 -> 16 exact bf16 values (block_fp.py:7-82; zero block -> zeros; |x| <= 1e-8 flushed to 0 - FLUSH_TINY is
// false for fp16 inputs, which cannot hold a non-zero |x| <= 1e-8: four instructions less per pair in the hot loop)
template <bool FLUSH_TINY = true>
__device__ __forceinline__ void quant16_bf16(const float (&v)[16], const QP& q, uint32_t (&w)[8]) {
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) amax = fmaxf(amax, fabsf(v[i]));
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = 0;
  if (amax > 0.f) {
    const int e = block_exponent(amax, q);
    if (mxint16_fast_ok(e, q)) {
      mxint16_bf16_fast<FLUSH_TINY>(v, e, q, w);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t lo = exact_bf16_bits(ldexpf(mxint_mantissa(v[2 * i], e, q), e - q.mbits));
        const uint32_t hi = exact_bf16_bits(ldexpf(mxint_mantissa(v[2 * i + 1], e, q), e - q.mbits));
        w[i] = lo | (hi << 16);
      }
    }
  }
}

// PRE: the values are ALREADY the quantizer's outputs (the standalone quantizer ran over the operand: block lengths other than
// 16 - 16 n or whole rows - of formats whose values are bf16 numbers, width <= 9): the bf16 image is their high halves
template <bool PRE, bool FLUSH_TINY>
__device__ __forceinline__ void image16(const float (&v)[16], const QP& q, uint32_t (&w)[8]) {
  if constexpr (PRE) {
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = exact_bf16_bits(v[2 * i]) | (exact_bf16_bits(v[2 * i + 1]) << 16);
  } else {
    quant16_bf16<FLUSH_TINY>(v, q, w);
  }
}

// ---- y [b][k][j], j contiguous -> img [b][S2p][Kp] (blocks of 16 along j), transposed through LDS -------------------------
// One workgroup = 64 k x 64 j.  Thread t quantizes the block (k = t / 4, j = 16 (t % 4) ..): four threads read 128 B of a
// k row; the bf16 values go to an LDS tile [j][k] and leave as 32-byte pieces of the image's j rows.
template <int DT, bool PRE = false>  // PRE: y is the bf16 image of the standalone quantizer (DT = LQER_BF16): transposed only
__global__ __launch_bounds__(256) void k_qmm_bimage_j(const void* __restrict__ y, int64_t K, int64_t S2, int64_t y_bs, int64_t y_ks, QP q,
                                                      bf16_t* __restrict__ img, int64_t S2p, int64_t Kp, bool vec) {
  __shared__ bf16_t tile[64][64 + 2];  // (+2: the 16 two-byte stores of a thread walk 16 rows - spread them over banks)
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.z, k0 = (int64_t)blockIdx.y * 64, j0 = (int64_t)blockIdx.x * 64;
  {
    const int kl = tid >> 2, jb = (tid & 3) * 16;
    float v[16];
    const int64_t k = k0 + kl, j = j0 + jb;
    uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (k < K && j < S2) {
      load16<DT>(y, b * y_bs + k * y_ks + j, S2 - j, vec, v);
      image16<PRE, DT != LQER_F16>(v, q, w);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      tile[jb + 2 * i][kl] = (bf16_t)(w[i] & 0xffff);
      tile[jb + 2 * i + 1][kl] = (bf16_t)(w[i] >> 16);
    }
  }
  __syncthreads();
  {
    const int jl = tid >> 2, kc = (tid & 3) * 16;
    if (j0 + jl < S2p && k0 + kc < Kp) {
      bf16_t* dst = img + (b * S2p + j0 + jl) * Kp + k0 + kc;
      uint32_t w[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) w[i] = (uint32_t)tile[jl][kc + 2 * i] | ((uint32_t)tile[jl][kc + 2 * i + 1] << 16);
      ((uint4*)dst)[0] = make_uint4(w[0], w[1], w[2], w[3]);
      ((uint4*)dst)[1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
  }
}

// ---- y [b][k][j], k contiguous (the transposed view of a [j][k] tensor) -> img [b][S2p][Kp] -------------------------------
// A 64 (j) x 64 (k) tile per workgroup: four threads read 128 B of a j row, the fp32 values cross an LDS tile, thread (j block of
// 16, k) quantizes its block with the packed routine, the bf16 values cross a second tile and leave as 32-byte pieces of the
// image's rows.  (Round 2's kernel kept 16 x 8 values per thread - one wave per SIMD, 1,400 instructions per thread, a branch
// around a slow path per element: 18-20 us for a [32, 2048, 128] operand against 10 us.)
template <int DT>
__global__ __launch_bounds__(256) void k_qmm_bimage_k(const void* __restrict__ y, int64_t K, int64_t S2, int64_t y_bs, int64_t y_js, QP q,
                                                       bf16_t* __restrict__ img, int64_t S2p, int64_t Kp, bool vec) {
  __shared__ float tf[64][64 + 1];
  __shared__ bf16_t tb[64][64 + 2];
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.z, j0 = (int64_t)blockIdx.y * 64, k0 = (int64_t)blockIdx.x * 64;
  {
    const int jl = tid >> 2, kc = (tid & 3) * 16;
    const int64_t j = j0 + jl, k = k0 + kc;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = 0.f;
    if (j < S2 && k < K) load16<DT>(y, b * y_bs + j * y_js + k, K - k, vec, v);
#pragma unroll
    for (int i = 0; i < 16; ++i) tf[jl][kc + i] = v[i];
  }
  __syncthreads();
  {
    const int jb = tid >> 6, kl = tid & 63;  // block of 16 consecutive j at one k: consecutive lanes = consecutive k (no bank conflict)
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = tf[16 * jb + r][kl];
    uint32_t w[8];
    quant16_bf16<DT != LQER_F16>(v, q, w);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      tb[16 * jb + 2 * i][kl] = (bf16_t)(w[i] & 0xffff);
      tb[16 * jb + 2 * i + 1][kl] = (bf16_t)(w[i] >> 16);
    }
  }
  __syncthreads();
  {
    const int jl = tid >> 2, kc = (tid & 3) * 16;
    if (j0 + jl < S2p && k0 + kc < Kp) {
      bf16_t* dst = img + (b * S2p + j0 + jl) * Kp + k0 + kc;
      uint32_t w[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) w[i] = (uint32_t)tb[jl][kc + 2 * i] | ((uint32_t)tb[jl][kc + 2 * i + 1] << 16);
      ((uint4*)dst)[0] = make_uint4(w[0], w[1], w[2], w[3]);
      ((uint4*)dst)[1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
  }
}

// ---- the product ----------------------------------------------------------------------------------------------------------
// One workgroup = one 128 (i) x 128 (j) tile of out[b]; 4 waves as 2 x 2, each 64 x 64 = 2 x 2 tiles of 32 x 32.
// The MFMA is issued with the image rows (j) as the A operand and the token rows (i) as the B operand, so a lane owns one
// output row i and 4 consecutive columns j per accumulator quad (the store pattern of gemm_w4a8.hip).
// PF (aligned rows, K a multiple of 16): the raw loads of x and of the image run THREE 64-k chunks ahead in a register ring
// indexed statically (the loop is unrolled by its depth) - with one chunk ahead every chunk exposed a memory latency, and in
// P V the attention probabilities are a 268 MB stream.  All loads are unconditional from clamped in-range addresses (a load
// under a branch makes the compiler wait for everything in flight); what lies outside x is zeroed at the quantizer.
// NW = 8 (with PF): the same tile on 512 threads, waves as 2 x 4, each 64 x 32 - one activation block and two image pieces
// per thread and chunk, half the accumulators and prefetch registers per thread: twice the waves per CU hide the latencies
// of a loop that alternates vector work (the quantizer), barriers and MFMAs (P V: 131 -> ? us).
// XPRE (without PF): x is the bf16 image of the standalone quantizer (block lengths other than 16): copied into the slab as it is.
template <int DT, bool PF, int NW = 4, bool XPRE = false>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 1) void k_qmatmul(const void* __restrict__ x, const bf16_t* __restrict__ img, void* __restrict__ out,
                                                 int64_t S1, int64_t K, int64_t S2, int64_t x_bs, int64_t x_rs, int64_t S2p, int64_t Kp,
                                                 QP q, bool vec) {
  static_assert(!(XPRE && PF), "pre-quantized x: the plain loop only");
  constexpr int XDT = XPRE ? LQER_BF16 : DT;  // element type of x as it is read
  __shared__ __attribute__((aligned(16))) unsigned char smem[BM * BK * 2 + BN * BK * 2];
  unsigned char* const sa = smem;                // x tile, quantized: 128 rows x 128 B
  unsigned char* const sb = smem + BM * BK * 2;  // image tile (after the loop the 32 KiB hold the 16-bit output tile)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  static_assert(NW == 4 || (NW == 8 && PF), "8 waves: the prefetching form only");
  constexpr int T = 64 * NW, JT = 8 / NW;      // threads; 32-column tiles per wave (waves: 2 x NW/2, each 64 x 32 JT)
  constexpr int XB = 512 / T, IB = 1024 / T;   // activation blocks / 16-byte image pieces per thread and chunk
  const int wm = wave / (NW / 2), wn = wave % (NW / 2), l31 = lane & 31, lh = lane >> 5;
  const int64_t b = blockIdx.z, i0 = (int64_t)blockIdx.y * BM, j0 = (int64_t)blockIdx.x * BN;
  const bf16_t* const ib = img + (b * S2p + j0) * Kp;
  f32x16 acc[JT][2];  // [j tile][i tile]
#pragma unroll
  for (int a = 0; a < JT; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
  const int nkc = (int)(Kp / BK);
  auto mma_chunk = [&]() {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 fj[JT], fi[2];
#pragma unroll
      for (int t = 0; t < JT; ++t) fj[t] = *(const bf16x8*)(sb + swz(wn * (32 * JT) + t * 32 + l31, 2 * ks + lh));
#pragma unroll
      for (int t = 0; t < 2; ++t) fi[t] = *(const bf16x8*)(sa + swz(wm * 64 + t * 32 + l31, 2 * ks + lh));
#pragma unroll
      for (int a = 0; a < JT; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fj[a], fi[c], acc[a][c], 0, 0, 0);
    }
  };
  if constexpr (PF) {
    constexpr int DEPTH = NW == 8 ? 2 : 3;  // (8 waves: half the registers per thread for the ring, twice the waves to hide latency)
    constexpr int ESZ = DT == LQER_F32 ? 4 : 2;
    constexpr int NR = Raw16<DT>::N;
    struct Stage {  // one chunk's raw loads of this thread (three named stages: an indexed ring ended up in scratch memory)
      uint4 x[XB][NR];
      uint4 b[IB];
    };
    Stage s0, s1, s2;
    const int xrow0 = tid >> 2;
    auto fetch_raw = [&](int kc, Stage& st) {
      const int kcc = kc < nkc ? kc : nkc - 1;
#pragma unroll
      for (int u = 0; u < XB; ++u) {
        const int64_t i = i0 + xrow0 + (T / 4) * u, ic = i < S1 ? i : S1 - 1;
        int64_t k = (int64_t)kcc * BK + (tid & 3) * 16;
        k = k + 16 <= K ? k : K - 16;
        const uint4* ptr = (const uint4*)((const char*)x + (b * x_bs + ic * x_rs + k) * ESZ);
#pragma unroll
        for (int n = 0; n < NR; ++n) st.x[u][n] = ptr[n];
      }
#pragma unroll
      for (int u = 0; u < IB; ++u) {
        const int p = tid + T * u;
        st.b[u] = *(const uint4*)(ib + (int64_t)(p >> 3) * Kp + (int64_t)kcc * BK + (p & 7) * 8);
      }
    };
    auto chunk = [&](int kc, Stage& st) {
      uint32_t w[XB][8];
#pragma unroll
      for (int u = 0; u < XB; ++u) {
        const bool live = i0 + xrow0 + (T / 4) * u < S1 && (int64_t)kc * BK + (tid & 3) * 16 < K;
        float v[16];
        Raw16<DT> raw;
#pragma unroll
        for (int n = 0; n < NR; ++n) raw.r[n] = st.x[u][n];
        raw_to_f32<DT>(raw, v);
        quant16_bf16<DT != LQER_F16>(v, q, w[u]);
#pragma unroll
        for (int e = 0; e < 8; ++e) w[u][e] = live ? w[u][e] : 0u;
      }
      __syncthreads();  // the previous chunk's fragment reads are done
#pragma unroll
      for (int u = 0; u < XB; ++u) {
        const int row = xrow0 + (T / 4) * u, c = 2 * (tid & 3);
        *(uint4*)(sa + swz(row, c)) = make_uint4(w[u][0], w[u][1], w[u][2], w[u][3]);
        *(uint4*)(sa + swz(row, c + 1)) = make_uint4(w[u][4], w[u][5], w[u][6], w[u][7]);
      }
#pragma unroll
      for (int u = 0; u < IB; ++u) {
        const int p = tid + T * u;
        *(uint4*)(sb + swz(p >> 3, p & 7)) = st.b[u];
      }
      __syncthreads();
      fetch_raw(kc + DEPTH, st);  // (past the end: a clamped re-read, never used)
      mma_chunk();
    };
    fetch_raw(0, s0);
    fetch_raw(1, s1);
    if constexpr (DEPTH == 3) {
      fetch_raw(2, s2);
      for (int kc = 0;; kc += DEPTH) {
        chunk(kc, s0);
        if (kc + 1 >= nkc) break;
        chunk(kc + 1, s1);
        if (kc + 2 >= nkc) break;
        chunk(kc + 2, s2);
        if (kc + 3 >= nkc) break;
      }
    } else {
      for (int kc = 0;; kc += DEPTH) {
        chunk(kc, s0);
        if (kc + 1 >= nkc) break;
        chunk(kc + 1, s1);
        if (kc + 2 >= nkc) break;
      }
    }
  } else {
  // this thread's two blocks of the x tile (rows tid / 4 and 64 + tid / 4, block tid % 4) and four 16-byte pieces of the image tile
  float xv[2][16];
  uint4 bv0, bv1, bv2, bv3;  // (named, not an array: indexed under the conditional prefetch it ended up in scratch memory - 80 B per lane)
  auto fetch = [&](int kc) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int row = (tid >> 2) + 64 * u;
      const int64_t i = i0 + row, k = (int64_t)kc * BK + (tid & 3) * 16;
      if (i < S1 && k < K) {
        load16<XDT>(x, b * x_bs + i * x_rs + k, K - k, vec, xv[u]);
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) xv[u][e] = 0.f;
      }
    }
    auto piece = [&](int u) {
      const int p = tid + 256 * u, row = p >> 3, ch = p & 7;
      return *(const uint4*)(ib + (int64_t)row * Kp + (int64_t)kc * BK + ch * 8);  // (rows up to S2p exist, zero beyond S2)
    };
    bv0 = piece(0), bv1 = piece(1), bv2 = piece(2), bv3 = piece(3);
  };
  fetch(0);
  for (int kc = 0; kc < nkc; ++kc) {
    uint32_t w[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u) image16<XPRE, DT != LQER_F16>(xv[u], q, w[u]);
    __syncthreads();  // the previous chunk's fragment reads are done
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int row = (tid >> 2) + 64 * u, c = 2 * (tid & 3);
      *(uint4*)(sa + swz(row, c)) = make_uint4(w[u][0], w[u][1], w[u][2], w[u][3]);
      *(uint4*)(sa + swz(row, c + 1)) = make_uint4(w[u][4], w[u][5], w[u][6], w[u][7]);
    }
    {
      auto put = [&](int u, const uint4& v) {
        const int p = tid + 256 * u;
        *(uint4*)(sb + swz(p >> 3, p & 7)) = v;
      };
      put(0, bv0), put(1, bv1), put(2, bv2), put(3, bv3);
    }
    __syncthreads();
    if (kc + 1 < nkc) fetch(kc + 1);  // the next chunk's loads travel under this chunk's MFMAs
    mma_chunk();
  }
  }
  // ---- store: lane = output row, register r of tile (a, c): column (r & 3) + 8 (r >> 2) + 4 lh
  const int esz = DT == LQER_F32 ? 4 : 2;
  const bool aligned = (((uintptr_t)out) & 15) == 0 && (S2 * esz) % 16 == 0;
  const bool staged = DT != LQER_F32 && aligned && j0 + BN <= S2;  // workgroup-uniform
  if (staged) __syncthreads();  // the last chunk's fragment reads are done: the tiles' LDS is free
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int64_t i = i0 + wm * 64 + c * 32 + l31;
#pragma unroll
    for (int a = 0; a < JT; ++a) {
      const int64_t jb = j0 + wn * (32 * JT) + a * 32;
      if constexpr (DT == LQER_F32) {
        if (i >= S1) continue;
        float* dst = (float*)out + (b * S1 + i) * S2;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int64_t j = jb + 8 * qd + 4 * lh;
          if (aligned && j + 3 < S2) {
            *(float4*)(dst + j) = make_float4(acc[a][c][4 * qd], acc[a][c][4 * qd + 1], acc[a][c][4 * qd + 2], acc[a][c][4 * qd + 3]);
          } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
              if (j + t < S2) dst[j + t] = acc[a][c][4 * qd + t];
          }
        }
      } else {
        // 16-bit outputs: pack 4 columns into 8 B, then merge quads (2p, 2p+1) of lanes l / l^32 into one 16-byte store
        // (gemm_w4a8.hip's store): lanes 0-31 write columns 16p .. 16p+7, lanes 32-63 columns 16p+8 .. 16p+15 - the output
        // is the large stream of Q K^T (S1 x S2 per head)
        uint32_t pk[4][2];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const float v0 = acc[a][c][4 * qd + 2 * h], v1 = acc[a][c][4 * qd + 2 * h + 1];
            if constexpr (DT == LQER_F16) {
              typedef __attribute__((ext_vector_type(2))) _Float16 h2;
              const h2 hv = {(_Float16)v0, (_Float16)v1};
              pk[qd][h] = __builtin_bit_cast(uint32_t, hv);
            } else {
              pk[qd][h] = (uint32_t)f32_to_bf16_rne(v0) | ((uint32_t)f32_to_bf16_rne(v1) << 16);
            }
          }
        bf16_t* dst = (bf16_t*)out + (b * S1 + i) * S2;
        const bool wide = aligned && jb + 32 <= S2;  // wave-uniform
        if (staged) {
          // whole tile inside the output: the 16-byte pieces go to LDS (row-major 256 B per row, chunks XOR-ed with the row
          // so that the 32 rows of a wave's store spread over the banks) and leave as full 256-byte row segments below
          const int row_l = wm * 64 + c * 32 + l31;
#pragma unroll
          for (int p2 = 0; p2 < 2; ++p2) {
            auto r0 = __builtin_amdgcn_permlane32_swap(pk[2 * p2][0], pk[2 * p2 + 1][0], false, false);
            auto r1 = __builtin_amdgcn_permlane32_swap(pk[2 * p2][1], pk[2 * p2 + 1][1], false, false);
            const int chunk = (wn * (32 * JT) + a * 32 + 16 * p2 + 8 * lh) >> 3;
            *(uint4*)(smem + row_l * 256 + ((chunk ^ (row_l & 15)) << 4)) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
          }
        } else if (wide) {
#pragma unroll
          for (int p2 = 0; p2 < 2; ++p2) {
            auto r0 = __builtin_amdgcn_permlane32_swap(pk[2 * p2][0], pk[2 * p2 + 1][0], false, false);
            auto r1 = __builtin_amdgcn_permlane32_swap(pk[2 * p2][1], pk[2 * p2 + 1][1], false, false);
            if (i < S1) *(uint4*)(dst + jb + 16 * p2 + 8 * lh) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
          }
        } else if (i < S1) {
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            const int64_t j = jb + 8 * qd + 4 * lh;
            if (j < S2) dst[j] = (bf16_t)(pk[qd][0] & 0xffff);
            if (j + 1 < S2) dst[j + 1] = (bf16_t)(pk[qd][0] >> 16);
            if (j + 2 < S2) dst[j + 2] = (bf16_t)(pk[qd][1] & 0xffff);
            if (j + 3 < S2) dst[j + 3] = (bf16_t)(pk[qd][1] >> 16);
          }
        }
      }
    }
  }
  if (staged) {
    __syncthreads();
    bf16_t* const ob = (bf16_t*)out + (b * S1) * S2 + j0;
#pragma unroll
    for (int u = 0; u < 2048 / T; ++u) {  // 16 consecutive lanes = one row of the tile: 256 contiguous bytes
      const int idx = tid + T * u, row_l = idx >> 4, chunk = idx & 15;
      const uint4 v = *(const uint4*)(smem + row_l * 256 + ((chunk ^ (row_l & 15)) << 4));
      if (i0 + row_l < S1) *(uint4*)(ob + (i0 + row_l) * S2 + chunk * 8) = v;
    }
  }
}


// ---- the product for a short contraction dim (K <= 128: Q K^T, K = head_dim) ---------------------------------------------
// There the OUTPUT is the stream (S1 x S2 per head) and the first operand is small: one workgroup quantizes its 128-row tile
// of x for ALL of K once into LDS and walks XR_JT column tiles with it, so x is read and quantized once per 512 output
// columns instead of once per 128.  Same tile arithmetic, fragment maps and LDS-staged 16-bit stores as k_qmatmul; the image
// tile of the next column tile is requested before the current tile's MFMAs.
constexpr int XR_JT = 4;      // column tiles per workgroup
constexpr int XR_MAXK = 128;  // two 64-k chunks

template <int DT, int NW = 8>  // NW waves as 2 x NW/2, each 64 x (256 / NW) of the tile (k_qmatmul's two geometries)
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 1) void k_qmatmul_xr(const void* __restrict__ x, const bf16_t* __restrict__ img, void* __restrict__ out,
                                                    int64_t S1, int64_t K, int64_t S2, int64_t x_bs, int64_t x_rs, int64_t S2p, int64_t Kp,
                                                    QP q, bool vec) {
  // [x tile: 2 chunks x 128 rows x 128 B][image tile: 2 chunks x 128 rows x 128 B, reused for the 16-bit output tile]
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BM * BK * 2 + 2 * BN * BK * 2];
  unsigned char* const sa = smem;
  unsigned char* const sb = smem + 2 * BM * BK * 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int T = 64 * NW, JT = 8 / NW, XB = 512 / T, IB = 1024 / T;
  const int wm = wave / (NW / 2), wn = wave % (NW / 2), l31 = lane & 31, lh = lane >> 5;
  const int64_t b = blockIdx.z, i0 = (int64_t)blockIdx.y * BM;
  const int nkc = (int)(Kp / BK);  // 1 or 2
  // ---- the x tile, quantized once: thread -> rows tid / 4 and 64 + tid / 4, block tid % 4 of every chunk
#pragma unroll
  for (int kc = 0; kc < 2; ++kc)
#pragma unroll
    for (int u = 0; u < XB; ++u) {
      const int row = (tid >> 2) + (T / 4) * u;
      const int64_t i = i0 + row, k = (int64_t)kc * BK + (tid & 3) * 16;
      uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (kc < nkc && i < S1 && k < K) {
        float v[16];
        load16<DT>(x, b * x_bs + i * x_rs + k, K - k, vec, v);
        quant16_bf16<DT != LQER_F16>(v, q, w);
      }
      const int c = 2 * (tid & 3);
      *(uint4*)(sa + kc * (BM * BK * 2) + swz(row, c)) = make_uint4(w[0], w[1], w[2], w[3]);
      *(uint4*)(sa + kc * (BM * BK * 2) + swz(row, c + 1)) = make_uint4(w[4], w[5], w[6], w[7]);
    }
  const int esz = DT == LQER_F32 ? 4 : 2;
  const bool aligned = (((uintptr_t)out) & 15) == 0 && (S2 * esz) % 16 == 0;
  const int64_t jt0 = (int64_t)blockIdx.x * XR_JT, njt = S2p / BN;
  uint4 bv[2][IB];  // the image tile of one column tile: [chunk][piece]
  auto fetch = [&](int64_t jt) {
    const bf16_t* const ib = img + (b * S2p + jt * BN) * Kp;
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
      for (int u = 0; u < IB; ++u) {
        const int p = tid + T * u, row = p >> 3, ch = p & 7;
        bv[kc][u] = kc < nkc ? *(const uint4*)(ib + (int64_t)row * Kp + (int64_t)kc * BK + ch * 8) : make_uint4(0, 0, 0, 0);
      }
  };
  if (jt0 < njt) fetch(jt0);
  for (int t = 0; t < XR_JT; ++t) {
    const int64_t jt = jt0 + t, j0 = jt * BN;
    if (jt >= njt) break;  // workgroup-uniform
    __syncthreads();  // the x tile is written (t = 0); the previous tile's output has left LDS (t > 0)
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
      for (int u = 0; u < IB; ++u) {
        const int p = tid + T * u;
        *(uint4*)(sb + kc * (BN * BK * 2) + swz(p >> 3, p & 7)) = bv[kc][u];
      }
    __syncthreads();
    if (jt + 1 < njt && t + 1 < XR_JT) fetch(jt + 1);  // the next tile's loads travel under this tile's MFMAs and stores
    f32x16 acc[JT][2];  // [j tile][i tile]
#pragma unroll
    for (int a = 0; a < JT; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      if (kc >= nkc) break;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 fj[JT], fi[2];
#pragma unroll
        for (int u = 0; u < JT; ++u) fj[u] = *(const bf16x8*)(sb + kc * (BN * BK * 2) + swz(wn * (32 * JT) + u * 32 + l31, 2 * ks + lh));
#pragma unroll
        for (int u = 0; u < 2; ++u) fi[u] = *(const bf16x8*)(sa + kc * (BM * BK * 2) + swz(wm * 64 + u * 32 + l31, 2 * ks + lh));
#pragma unroll
        for (int a = 0; a < JT; ++a)
#pragma unroll
          for (int c = 0; c < 2; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fj[a], fi[c], acc[a][c], 0, 0, 0);
      }
    }
    // ---- store (k_qmatmul's): fp32 directly; 16-bit through LDS as full 256-byte row segments when the tile is inside
    const bool staged = DT != LQER_F32 && aligned && j0 + BN <= S2;  // workgroup-uniform
    if (staged) __syncthreads();  // every wave has read its image fragments: sb is free
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int64_t i = i0 + wm * 64 + c * 32 + l31;
#pragma unroll
      for (int a = 0; a < JT; ++a) {
        const int64_t jb = j0 + wn * (32 * JT) + a * 32;
        if constexpr (DT == LQER_F32) {
          if (i >= S1) continue;
          float* dst = (float*)out + (b * S1 + i) * S2;
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            const int64_t j = jb + 8 * qd + 4 * lh;
            if (aligned && j + 3 < S2) {
              *(float4*)(dst + j) = make_float4(acc[a][c][4 * qd], acc[a][c][4 * qd + 1], acc[a][c][4 * qd + 2], acc[a][c][4 * qd + 3]);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (j + e < S2) dst[j + e] = acc[a][c][4 * qd + e];
            }
          }
        } else {
          uint32_t pk[4][2];
#pragma unroll
          for (int qd = 0; qd < 4; ++qd)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const float v0 = acc[a][c][4 * qd + 2 * h], v1 = acc[a][c][4 * qd + 2 * h + 1];
              if constexpr (DT == LQER_F16) {
                typedef __attribute__((ext_vector_type(2))) _Float16 h2;
                const h2 hv = {(_Float16)v0, (_Float16)v1};
                pk[qd][h] = __builtin_bit_cast(uint32_t, hv);
              } else {
                pk[qd][h] = (uint32_t)f32_to_bf16_rne(v0) | ((uint32_t)f32_to_bf16_rne(v1) << 16);
              }
            }
          bf16_t* dst = (bf16_t*)out + (b * S1 + i) * S2;
          const bool wide = aligned && jb + 32 <= S2;  // wave-uniform
          if (staged) {
            const int row_l = wm * 64 + c * 32 + l31;
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
              auto r0 = __builtin_amdgcn_permlane32_swap(pk[2 * p2][0], pk[2 * p2 + 1][0], false, false);
              auto r1 = __builtin_amdgcn_permlane32_swap(pk[2 * p2][1], pk[2 * p2 + 1][1], false, false);
              const int chunk = (wn * (32 * JT) + a * 32 + 16 * p2 + 8 * lh) >> 3;
              *(uint4*)(sb + row_l * 256 + ((chunk ^ (row_l & 15)) << 4)) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
            }
          } else if (wide) {
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
              auto r0 = __builtin_amdgcn_permlane32_swap(pk[2 * p2][0], pk[2 * p2 + 1][0], false, false);
              auto r1 = __builtin_amdgcn_permlane32_swap(pk[2 * p2][1], pk[2 * p2 + 1][1], false, false);
              if (i < S1) *(uint4*)(dst + jb + 16 * p2 + 8 * lh) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
            }
          } else if (i < S1) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
              const int64_t j = jb + 8 * qd + 4 * lh;
              if (j < S2) dst[j] = (bf16_t)(pk[qd][0] & 0xffff);
              if (j + 1 < S2) dst[j + 1] = (bf16_t)(pk[qd][0] >> 16);
              if (j + 2 < S2) dst[j + 2] = (bf16_t)(pk[qd][1] & 0xffff);
              if (j + 3 < S2) dst[j + 3] = (bf16_t)(pk[qd][1] >> 16);
            }
          }
        }
      }
    }
    if (staged) {
      __syncthreads();
      bf16_t* const ob = (bf16_t*)out + (b * S1) * S2 + j0;
#pragma unroll
      for (int u = 0; u < 2048 / T; ++u) {  // 16 consecutive lanes = one row of the tile: 256 contiguous bytes
        const int idx = tid + T * u, row_l = idx >> 4, chunk = idx & 15;
        const uint4 v = *(const uint4*)(sb + row_l * 256 + ((chunk ^ (row_l & 15)) << 4));
        if (i0 + row_l < S1) *(uint4*)(ob + (i0 + row_l) * S2 + chunk * 8) = v;
      }
    }
  }
}

}  // namespace qmm

size_t qmatmul_workspace_bytes(int64_t batch, int64_t K, int64_t S2) {
  const int64_t S2p = (S2 + qmm::BN - 1) / qmm::BN * qmm::BN, Kp = (K + qmm::BK - 1) / qmm::BK * qmm::BK;
  return (size_t)batch * S2p * Kp * sizeof(bf16_t);
}
// the same plus the bf16 images of the standalone quantizer for operands whose blocks are not 16 (x: [batch S1][Kp], y: [batch K][S2 padded to 64])
static size_t qmm_align(size_t v) { return (v + 255) / 256 * 256; }
size_t qmatmul_workspace_bytes_ex(int64_t batch, int64_t S1, int64_t K, int64_t S2, bool x_pre, bool y_pre) {
  const int64_t Kp = (K + qmm::BK - 1) / qmm::BK * qmm::BK, S2q = (S2 + 63) / 64 * 64;
  size_t n = qmm_align(qmatmul_workspace_bytes(batch, K, S2));
  if (x_pre) n += qmm_align((size_t)lqer_padded_m(batch * S1) * Kp * sizeof(bf16_t));
  if (y_pre) n += qmm_align((size_t)lqer_padded_m(batch * K) * S2q * sizeof(bf16_t));
  return n;
}

template <int DT>
static int launch_qmm(const void* x, const void* y, void* out, int64_t batch, int64_t S1, int64_t K, int64_t S2, int64_t x_bs, int64_t x_rs,
                      int64_t y_bs, int64_t y_ks, int64_t y_js, const QP& qx, const QP& qy, bf16_t* img, hipStream_t st) {
  const int64_t S2p = (S2 + qmm::BN - 1) / qmm::BN * qmm::BN, Kp = (K + qmm::BK - 1) / qmm::BK * qmm::BK;
  const int esz = DT == LQER_F32 ? 4 : 2;
  auto al16 = [&](const void* p, int64_t a, int64_t c) { return ((uintptr_t)p % 16 == 0) && (a * esz) % 16 == 0 && (c * esz) % 16 == 0; };
  // ---- operands whose blocks are not 16 elements: the standalone quantizer writes their bf16 images first (any 16 n, whole
  // rows), the image / product kernels then take those as they are (PRE / XPRE) - the same bits as the fused form would give
  const bool x_pre = qx.block != 16, y_pre = qy.block != 16;
  if (x_pre || y_pre) {
    unsigned char* wsp = (unsigned char*)img + qmm_align(qmatmul_workspace_bytes(batch, K, S2));
    const int64_t S2q = (S2 + 63) / 64 * 64;
    const bf16_t* xi = nullptr;
    const bf16_t* yi = nullptr;
    if (x_pre) {  // rows (b, i) must be evenly spaced: x_bs == S1 x_rs (the caller guarantees it)
      QuantOut o{nullptr, nullptr, nullptr, (bf16_t*)wsp, Kp, 0};
      const int rc = quantize_dispatch(x, DT, batch * S1, K, x_rs, qx, o, st);
      if (rc) return rc;
      xi = (const bf16_t*)wsp;
      wsp += qmm_align((size_t)lqer_padded_m(batch * S1) * Kp * sizeof(bf16_t));
    }
    if (y_pre) {  // y [b][k][j] dense along j, rows (b, k) evenly spaced: y_bs == K y_ks
      QuantOut o{nullptr, nullptr, nullptr, (bf16_t*)wsp, S2q, 0};
      const int rc = quantize_dispatch(y, DT, batch * K, S2, y_ks, qy, o, st);
      if (rc) return rc;
      yi = (const bf16_t*)wsp;
    }
    const dim3 gi((unsigned)(S2p / 64), (unsigned)(Kp / 64), (unsigned)batch);
    if (y_pre)
      qmm::k_qmm_bimage_j<LQER_BF16, true><<<gi, 256, 0, st>>>(yi, K, S2, K * S2q, S2q, qy, img, S2p, Kp, true);
    else if (y_js == 1)
      qmm::k_qmm_bimage_j<DT><<<gi, 256, 0, st>>>(y, K, S2, y_bs, y_ks, qy, img, S2p, Kp, al16(y, y_bs, y_ks));
    else
      qmm::k_qmm_bimage_k<DT><<<dim3((unsigned)(Kp / 64), (unsigned)(S2p / 64), (unsigned)batch), 256, 0, st>>>(y, K, S2, y_bs, y_js, qy, img, S2p, Kp,
                                                                                                          al16(y, y_bs, y_js));
    const dim3 grid((unsigned)(S2p / qmm::BN), (unsigned)((S1 + qmm::BM - 1) / qmm::BM), (unsigned)batch);
    if (x_pre)
      qmm::k_qmatmul<DT, false, 4, true><<<grid, 256, 0, st>>>(xi, img, out, S1, K, S2, S1 * Kp, Kp, S2p, Kp, qx, true);
    else
      qmm::k_qmatmul<DT, false><<<grid, 256, 0, st>>>(x, img, out, S1, K, S2, x_bs, x_rs, S2p, Kp, qx, al16(x, x_bs, x_rs));
    return check_launch("lqer_matmul_q");
  }
  if (y_js == 1) {
    const dim3 grid((unsigned)(S2p / 64), (unsigned)(Kp / 64), (unsigned)batch);
    qmm::k_qmm_bimage_j<DT><<<grid, 256, 0, st>>>(y, K, S2, y_bs, y_ks, qy, img, S2p, Kp, al16(y, y_bs, y_ks));
  } else {
    const dim3 grid((unsigned)(Kp / 64), (unsigned)(S2p / 64), (unsigned)batch);
    qmm::k_qmm_bimage_k<DT><<<grid, 256, 0, st>>>(y, K, S2, y_bs, y_js, qy, img, S2p, Kp, al16(y, y_bs, y_js));
  }
  if (Kp <= qmm::XR_MAXK && S2p / qmm::BN >= 2 * qmm::XR_JT) {  // short contraction, many column tiles: x tile resident in LDS
    const dim3 grid((unsigned)((S2p / qmm::BN + qmm::XR_JT - 1) / qmm::XR_JT), (unsigned)((S1 + qmm::BM - 1) / qmm::BM), (unsigned)batch);
    qmm::k_qmatmul_xr<DT><<<grid, 512, 0, st>>>(x, img, out, S1, K, S2, x_bs, x_rs, S2p, Kp, qx, al16(x, x_bs, x_rs));
    return check_launch("lqer_matmul_q");
  }
  const dim3 grid((unsigned)(S2p / qmm::BN), (unsigned)((S1 + qmm::BM - 1) / qmm::BM), (unsigned)batch);
  const bool vec = al16(x, x_bs, x_rs);
#ifndef LQER_QMM_NW
#define LQER_QMM_NW 8
#endif
  if (vec && K % 16 == 0 && K >= 16)
    qmm::k_qmatmul<DT, true, LQER_QMM_NW><<<grid, 64 * LQER_QMM_NW, 0, st>>>(x, img, out, S1, K, S2, x_bs, x_rs, S2p, Kp, qx, vec);
  else
    qmm::k_qmatmul<DT, false><<<grid, 256, 0, st>>>(x, img, out, S1, K, S2, x_bs, x_rs, S2p, Kp, qx, vec);
  return check_launch("lqer_matmul_q");
}

int qmatmul_dispatch(const void* x, const void* y, void* out, int dtype, int64_t batch, int64_t S1, int64_t K, int64_t S2, int64_t x_bs,
                     int64_t x_rs, int64_t y_bs, int64_t y_ks, int64_t y_js, const QP& qx, const QP& qy, void* workspace, hipStream_t st) {
  if (batch == 0 || S1 == 0 || S2 == 0) return LQER_OK;
  switch (dtype) {
    case LQER_F32: return launch_qmm<LQER_F32>(x, y, out, batch, S1, K, S2, x_bs, x_rs, y_bs, y_ks, y_js, qx, qy, (bf16_t*)workspace, st);
    case LQER_F16: return launch_qmm<LQER_F16>(x, y, out, batch, S1, K, S2, x_bs, x_rs, y_bs, y_ks, y_js, qx, qy, (bf16_t*)workspace, st);
    case LQER_BF16: return launch_qmm<LQER_BF16>(x, y, out, batch, S1, K, S2, x_bs, x_rs, y_bs, y_ks, y_js, qx, qy, (bf16_t*)workspace, st);
  }
  set_error("unknown dtype %d", dtype);
  return LQER_E_INVALID;
}

}  // namespace lqer
