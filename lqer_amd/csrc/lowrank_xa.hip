// Rank-r side path, first half:  xAq = A_out_quantizer( Q_x(x) @ A )   (reference
// quantized_layers/linear.py:154).  [M,K] x [K,r] with r << K: a skinny GEMM that streams the quantized
// activation once, i.e. HBM/L2 bound - so it is cut into ~2048 independent (32-row, K-chunk) pieces, one
// wave each, to put every CU on the stream:
//   k_xa_partial : v_mfma_f32_32x32x16_bf16 over the wave's K chunk, fp32 partial tile -> scratch
//   k_xa_reduce  : sums the chunks in a fixed order (bit-reproducible, no atomics), applies A_out, writes
//                  the exact bf16 image the fused GEMM's prologue consumes.
// A comes as exact bf16 limbs (pack.hip), x as the exact bf16 image of the activation quantizer, so every
// product is exact in fp32; only the fp32 accumulation order differs from the reference's torch.matmul.
#include <type_traits>

#include "common.h"

namespace lqer {

#ifndef LQER_XA_PF2
#define LQER_XA_PF2 1  // int8 activation image: loads two windows ahead (k_xa_partial)
#endif
#ifndef LQER_XAL_WGS
#define LQER_XAL_WGS 256  // workgroups the LDS-staged side GEMM aims for (128-row tiles x K chunks): one per CU - 512 measured 10-25 % slower
#endif
constexpr int XA_ROWS = 32;       // token rows per wave
constexpr int XA_MAX_TILES = 8;   // rp <= 256
#ifndef LQER_XA_TARGET
#define LQER_XA_TARGET 2048
#endif
constexpr int XA_TARGET_WAVES = LQER_XA_TARGET;
#ifndef LQER_QX_K
#define LQER_QX_K 256
#endif
constexpr int QX_K = LQER_QX_K;          // k per workgroup of the fused quantize + side-GEMM kernel
constexpr int QX_WAVES = QX_K / 64;      // one wave per 64 k; QX_K threads, two 16-element blocks each

struct XaPlan {
  int row_groups, nchunk, kc;  // kc = k per chunk (multiple of 64)
};

__host__ __device__ inline XaPlan xa_plan(int64_t M, int64_t Kp) {
  XaPlan p;
  p.row_groups = (int)((M + XA_ROWS - 1) / XA_ROWS);
  const int windows = (int)(Kp / 64);
  int want = p.row_groups > 0 ? (XA_TARGET_WAVES + p.row_groups - 1) / p.row_groups : 1;
  if (want < 1) want = 1;
  if (want > windows) want = windows;
  const int wpc = (windows + want - 1) / want;  // 64-k windows per chunk
  p.kc = wpc * 64;
  p.nchunk = (windows + wpc - 1) / wpc;
  return p;
}

size_t xa_scratch_bytes(int64_t m_max, int64_t K, int64_t rp) {
  // split-K plan: row_groups * nchunk <= XA_TARGET_WAVES + row_groups for every M <= m_max;
  // fused plan (k_quant_xa16): row_groups * ceil(Kp / 256) partial tiles
  const int64_t rg = (m_max + XA_ROWS - 1) / XA_ROWS;
  const int64_t split = XA_TARGET_WAVES + rg;
  const int64_t fused = rg * ((lqer_padded_k(K) + 255) / 256) * (QX_K < 256 ? 256 / QX_K : 1);
  return (size_t)(split > fused ? split : fused) * XA_ROWS * rp * sizeof(float);
}

// One wave = RG x 32 token rows x one K chunk.  Within a 64-k window lane (r = lane & 31, h = lane >> 5) owns the
// 32 consecutive k at 32h of its rows - one 64-byte load per row - and feeds them to 4 MFMAs in the order it holds
// them (the k order inside a window is a free permutation as long as both operands use the same one).  Every A^T
// fragment (fetched from L2) feeds RG MFMAs: with one row group per wave the A^T stream, rp/32 times the activation
// stream, is the bound.
// XF16: fp16 activation image and fp16 A^T (LQER_Q_PASSTHROUGH_F16).  AFPF: the A^T fragments of the next window are
// requested one window ahead as well (a_limbs * NT <= 4 fragment sets = 64 registers; used for rank <= 64): without it
// every 64-k window waits one L2 latency for them, which - not the activation stream - bounds the kernel at large M
// (16384 x 13824, rank 64, two limbs: 358 -> 326 us for quantizer + side GEMM; no gain at rank 128, where it is off).
// XI8: the activation image holds int8 mantissas ([Mp][Kx] bytes, Kx = the image's row stride; one exponent per row, applied
// to the row's sum by the reduce pass): the lane's 32 bytes of a window are converted to bf16 (integers up to 127: exact).
__device__ __forceinline__ void i8x32_to_bf16(const u32x4& lo, const u32x4& hi, bf16x8 (&f)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // fragment i = bytes 8i .. 8i+7
    const uint32_t w0 = i < 2 ? lo[2 * i] : hi[2 * i - 4], w1 = i < 2 ? lo[2 * i + 1] : hi[2 * i - 3];
    u32x4 r;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t w = j ? w1 : w0;
      const float f0 = (float)(int)(int8_t)(w & 0xff), f1 = (float)(int)(int8_t)((w >> 8) & 0xff);
      const float f2 = (float)(int)(int8_t)((w >> 16) & 0xff), f3 = (float)(int)(int8_t)(w >> 24);
      r[2 * j] = (__float_as_uint(f0) >> 16) | (__float_as_uint(f1) & 0xffff0000u);
      r[2 * j + 1] = (__float_as_uint(f2) >> 16) | (__float_as_uint(f3) & 0xffff0000u);
    }
    f[i] = __builtin_bit_cast(bf16x8, r);
  }
}

// the same 32 mantissas as fp16 (XI8 with an fp16 A^T image: the mantissas times an fp16-exact A on v_mfma_*_f16, ONE limb of
// A^T instead of two): byte b -> the half 0x6400 | (b ^ 0x80) = 1024 + (b + 128), minus 1152 - two bytes per v_perm_b32 and
// v_pk_add_f16 instead of three conversions per byte
__device__ __forceinline__ void i8x32_to_f16(const u32x4& lo, const u32x4& hi, bf16x8 (&f)[4]) {
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  const h2 bias = {(_Float16)-1152.0f, (_Float16)-1152.0f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // fragment i = bytes 8i .. 8i+7
    const uint32_t w0 = i < 2 ? lo[2 * i] : hi[2 * i - 4], w1 = i < 2 ? lo[2 * i + 1] : hi[2 * i - 3];
    u32x4 r;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t t = (j ? w1 : w0) ^ 0x80808080u;
      const h2 a = __builtin_bit_cast(h2, __builtin_amdgcn_perm(0x64646464u, t, 0x04010400u)) + bias;  // bytes 0, 1
      const h2 b = __builtin_bit_cast(h2, __builtin_amdgcn_perm(0x64646464u, t, 0x04030402u)) + bias;  // bytes 2, 3
      r[2 * j] = __builtin_bit_cast(uint32_t, a);
      r[2 * j + 1] = __builtin_bit_cast(uint32_t, b);
    }
    f[i] = __builtin_bit_cast(bf16x8, r);
  }
}

// PF2 (XI8 with AFPF): the activation loads run TWO windows ahead (two named register stages, the loop unrolled by two) - at
// M = 16384 a wave holds 64 rows x 64 bytes per window, 4 waves per CU: one window ahead leaves 16 KB per CU in flight and the
// HBM latency, not the bandwidth, sets the pace (84 MB in 49 us = 1.7 TB/s).
template <int NT, int RG, bool XF16 = false, bool AFPF = false, bool XI8 = false, bool PF2 = false>
__global__ __launch_bounds__(256) void k_xa_partial(const bf16_t* __restrict__ xq, int64_t M, int64_t Kp, int64_t Kx,
                                                    const bf16_t* __restrict__ a_t, int a_limbs, int rp, XaPlan plan,
                                                    float* __restrict__ part) {
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int wave_rows = (plan.row_groups + RG - 1) / RG;
  if (wid >= wave_rows * plan.nchunk) return;
  const int wr = wid / plan.nchunk, c = wid - wr * plan.nchunk;
  const int r = lane & 31, h = lane >> 5;
  const int64_t k_begin = (int64_t)c * plan.kc;
  const int64_t k_end = k_begin + plan.kc < Kp ? k_begin + plan.kc : Kp;
  f32x16 acc[RG][NT];
#pragma unroll
  for (int u = 0; u < RG; ++u)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[u][t][j] = 0.f;
  const bf16_t* xrow[RG];  // (XI8: a byte pointer - the lane's 32 consecutive k of a window are 32 bytes)
#pragma unroll
  for (int u = 0; u < RG; ++u) {
    const int rg = wr * RG + u < plan.row_groups ? wr * RG + u : plan.row_groups - 1;  // a clamped duplicate is not stored
    const int64_t row = (int64_t)rg * XA_ROWS + r;  // rows past M repeat the last one (never read beyond the tensor: the
    if constexpr (XI8)                              // image may be the caller's own fp16 tensor); their results are unused
      xrow[u] = (const bf16_t*)((const uint8_t*)xq + (row < M ? row : M - 1) * Kx + 32 * h);
    else
      xrow[u] = xq + (row < M ? row : M - 1) * Kx + 32 * h;
  }
  // the next window's activation loads are issued before this window's MFMAs (the stream from HBM is the bound)
  constexpr int NST = PF2 ? 2 : 1;  // register stages of the activation prefetch
  bf16x8 xn[NST][RG][XI8 ? 2 : 4];  // XI8: the raw 32 bytes
  auto load_x = [&](int64_t k0, auto st_c) {
    constexpr int ST = decltype(st_c)::value;
#pragma unroll
    for (int u = 0; u < RG; ++u) {
      if constexpr (XI8) {
#pragma unroll
        for (int i = 0; i < 2; ++i) xn[ST][u][i] = *(const bf16x8*)((const uint8_t*)xrow[u] + k0 + 16 * i);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) xn[ST][u][i] = *(const bf16x8*)(xrow[u] + k0 + 8 * i);
      }
    }
  };
  using std::integral_constant;
  load_x(k_begin, integral_constant<int, 0>{});
  if constexpr (PF2) {
    if (k_begin + 64 < k_end) load_x(k_begin + 64, integral_constant<int, 1>{});
  }
  // (limb, tile) fragment sets held one window ahead; the fp16 image of the int8 route is ONE limb: half the registers
  constexpr int LMAX = AFPF ? ((XF16 && XI8) ? 1 : 4 / NT) : 1;
  constexpr int PAIRS = AFPF ? LMAX * NT : 1;
  bf16x8 afn[PAIRS][4];
  auto load_af = [&](int64_t k0) {
    if constexpr (AFPF) {
#pragma unroll
      for (int l = 0; l < LMAX; ++l)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int n = t * 32 + r;
          if (l < a_limbs && n < rp) {
            const bf16_t* arow = a_t + ((int64_t)l * rp + n) * Kp + k0 + 32 * h;
#pragma unroll
            for (int i = 0; i < 4; ++i) afn[l * NT + t][i] = *(const bf16x8*)(arow + 8 * i);
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) afn[l * NT + t][i] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
          }
        }
    }
  };
  load_af(k_begin);
  // one 64-k window out of register stage ST; its stage is refilled NST windows ahead
  auto window = [&](int64_t k0, auto st_c) {
    constexpr int ST = decltype(st_c)::value;
    bf16x8 xf[RG][4];
#pragma unroll
    for (int u = 0; u < RG; ++u) {
      if constexpr (XI8 && XF16) {
        i8x32_to_f16(__builtin_bit_cast(u32x4, xn[ST][u][0]), __builtin_bit_cast(u32x4, xn[ST][u][1]), xf[u]);
      } else if constexpr (XI8) {
        i8x32_to_bf16(__builtin_bit_cast(u32x4, xn[ST][u][0]), __builtin_bit_cast(u32x4, xn[ST][u][1]), xf[u]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[u][i] = xn[ST][u][i];
      }
    }
    if (k0 + 64 * NST < k_end) load_x(k0 + 64 * NST, st_c);
    if constexpr (AFPF) {
      bf16x8 afc[PAIRS][4];
#pragma unroll
      for (int p = 0; p < PAIRS; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i) afc[p][i] = afn[p][i];
      if (k0 + 64 < k_end) load_af(k0 + 64);
#pragma unroll
      for (int l = 0; l < LMAX; ++l)
#pragma unroll
        for (int t = 0; t < NT; ++t)
          if (l < a_limbs) {
#pragma unroll
            for (int u = 0; u < RG; ++u)
#pragma unroll
              for (int i = 0; i < 4; ++i) acc[u][t] = mfma_32x32x16<XF16>(xf[u][i], afc[l * NT + t][i], acc[u][t]);
          }
      return;
    }
    for (int l = 0; l < a_limbs; ++l) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int n = t * 32 + r;
        bf16x8 af[4];
        if (n < rp) {
          const bf16_t* arow = a_t + ((int64_t)l * rp + n) * Kp + k0 + 32 * h;
#pragma unroll
          for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(arow + 8 * i);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) af[i] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < RG; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[u][t] = mfma_32x32x16<XF16>(xf[u][i], af[i], acc[u][t]);
      }
    }
  };
  for (int64_t k0 = k_begin; k0 < k_end; k0 += 64 * NST) {
    window(k0, integral_constant<int, 0>{});
    if constexpr (PF2) {
      if (k0 + 64 < k_end) window(k0 + 64, integral_constant<int, 1>{});
    }
  }
  // D layout: col n = lane & 31, row m = (reg & 3) + 8 (reg >> 2) + 4 h.  part[c][rg*32 + m][n]
#pragma unroll
  for (int u = 0; u < RG; ++u) {
    const int rg = wr * RG + u;
    if (rg < plan.row_groups) {
      float* dst = part + ((int64_t)c * plan.row_groups + rg) * XA_ROWS * rp;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int n = t * 32 + r;
        if (n < rp) {
#pragma unroll
          for (int j = 0; j < 16; ++j) dst[((j & 3) + 8 * (j >> 2) + 4 * h) * rp + n] = acc[u][t][j];
        }
      }
    }
  }
}

// ---- side GEMM with both operands staged through LDS (prefill sizes) ---------------------------------------------------------
// k_xa_partial reads both operands with one 16-byte load per lane and ROW: 32-64 different cache lines per wave instruction.
// Counters (tools/r03_sidepmc.sh, C4): the texture addresser is busy for the kernel's whole duration, stalled on the L1 tag
// lookups (TA_BUSY ~ 84 k of ~90 k cycles, TA_ADDR_STALLED_BY_TC 18.5 M summed) - 84 MB of activations move at 1.7-2.2 TB/s.
// Here a workgroup (4 waves) owns 128 token rows x one K chunk and walks it in steps of 128 BYTES of activation row: the
// step's activation tile (128 rows x 128 B) and A^T slab arrive by LDS-DMA in whole 128-byte lines, three slots, two steps
// ahead (source-side XOR swizzles: fragment reads are conflict-free); wave w multiplies rows 32 w .. 32 w + 31.  Two operand
// formats:
//   I8  (the W4A8 INT configurations): int8 mantissas (128 k per step) x ONE fp16 image of A^T (256 B per rank row and step),
//       v_mfma_f32_32x32x16_f16 after the byte -> half conversion of k_xa_partial;
//   B16 (block_fp activations, rank > 64 - the OPT configurations): the bf16 activation image (64 k per step) x one bf16
//       limb of A^T (128 B per rank row and step), v_mfma_f32_32x32x16_bf16.
// Same partial-tile layout as k_xa_partial (the reduce kernels do not change); the k order inside a step is permuted the same
// way for both operands.
namespace xal {
constexpr int ROWS = 128, BKB = 128;  // token rows per workgroup, activation bytes per row and step
constexpr int X_SLOT = ROWS * BKB;    // 16 KiB
typedef __attribute__((address_space(3))) void lds_void;
constexpr int a_row_bytes(bool i8) { return i8 ? 256 : 128; }
#ifndef LQER_XAL_SLOTS
#define LQER_XAL_SLOTS 3
#endif
constexpr int NS = LQER_XAL_SLOTS, AHEAD = NS - 1;  // ring slots; steps in flight ahead of the one being multiplied (a fourth slot -
                                                 // 96 KB per CU in flight - measured +-0 at C4 and C5: tools/ab_xa.py)
constexpr int lds_bytes(int nt, bool i8) { return NS * (X_SLOT + 32 * nt * a_row_bytes(i8)); }

template <int NT, bool I8>  // 32-column rank tiles: rp = 32 NT
__global__ __launch_bounds__(256) void k_xa_partial_lds(const uint8_t* __restrict__ xq, int64_t x_ld, const bf16_t* __restrict__ a_img,
                                                        int64_t Kp, int row_groups, int nchunk, int steps_per_chunk, int steps_total,
                                                        float* __restrict__ part) {
  constexpr int RP = 32 * NT, AROW = a_row_bytes(I8), A_SLOT = RP * AROW, SLOT = X_SLOT + A_SLOT;
  constexpr int APR = 1024 / AROW;        // A^T rows per 1-KiB piece (4 or 8)
  constexpr int NA = RP / APR / 4;        // A^T pieces per wave and step
  constexpr int ACH = AROW / 16;          // 16-byte chunks per A^T row and step (16 or 8)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r31 = lane & 31, h = lane >> 5;
  const int tile = blockIdx.x / nchunk, c = blockIdx.x - tile * nchunk;
  const int s_begin = c * steps_per_chunk;
  const int s_end = s_begin + steps_per_chunk < steps_total ? s_begin + steps_per_chunk : steps_total;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem;
  // staging: wave w brings rows 32 w .. 32 w + 31 of the activation tile (4 pieces of 8 rows x 128 B) and NA pieces of A^T
  // (x_ld: bytes per activation row)
  const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(xq + (int64_t)tile * ROWS * x_ld), 0, (int)(ROWS * x_ld), 0x00020000);
  const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a_img, 0, (int)(RP * Kp * 2), 0x00020000);
  int x_voff[4], a_voff[4];  // (NA entries used: an array sized by a template-dependent constant, captured by a lambda, makes hipcc's
                             // HOST pass drop the kernel's stub without a diagnostic - gemm_w4a8.hip has the same note)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 32 + i * 8 + (lane >> 3);
    x_voff[i] = row * (int)x_ld + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
    a_voff[i] = 0;
  }
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    if constexpr (I8) {  // 4 rows x 256 B per piece, 16 chunks per row
      const int row = 4 * (wave * NA + i) + (lane >> 4);
      a_voff[i] = row * (int)Kp * 2 + (((lane & 15) ^ (row & 15)) << 4);
    } else {             // 8 rows x 128 B per piece: the activation tile's swizzle
      const int row = 8 * (wave * NA + i) + (lane >> 3);
      a_voff[i] = row * (int)Kp * 2 + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
    }
  }
  auto issue = [&](int st) {
    const int slot = st % NS;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void*)(smem + slot * SLOT + (wave * 4 + i) * 1024), 16, x_voff[i], st * BKB, 0, 0);
#pragma unroll
    for (int i = 0; i < NA; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_void*)(smem + slot * SLOT + X_SLOT + (wave * NA + i) * 1024), 16, a_voff[i],
                                               st * AROW, 0, 0);
  };
  // fragment addresses (slot 0): activation row 32 w + r31, 16-byte chunk 2 j + h; A^T row n = 32 t + r31: I8 chunks 4 j + 2 h (+ 1),
  // B16 chunk 2 j + h
  const int xrow = wave * 32 + r31;
  uint32_t xa[4], aa[4][4][2];  // (NT tiles, I8: two reads per (tile, j))
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    xa[j] = lds0 + xrow * 128 + (((2 * j + h) ^ ((xrow >> 1) & 7)) << 4);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = 32 * t + r31;
      if constexpr (I8) {
#pragma unroll
        for (int u = 0; u < 2; ++u) aa[t][j][u] = lds0 + X_SLOT + n * 256 + (((4 * j + 2 * h + u) ^ (n & 15)) << 4);
      } else {
        aa[t][j][0] = lds0 + X_SLOT + n * 128 + (((2 * j + h) ^ ((n >> 1) & 7)) << 4);
        aa[t][j][1] = 0;
      }
    }
  }
  static_assert(ACH == (I8 ? 16 : 8), "A^T chunk geometry");
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
#pragma unroll
  for (int d = 0; d < AHEAD; ++d)
    if (s_begin + d < s_end) issue(s_begin + d);
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  const h2 bias = {(_Float16)-1152.0f, (_Float16)-1152.0f};
  for (int st = s_begin; st < s_end; ++st) {
    // own loads of step st landed (the batches of the steps behind it - up to AHEAD - 1 - may stay in flight), then everybody's
    {
      const int younger = s_end - 1 - st < AHEAD - 1 ? s_end - 1 - st : AHEAD - 1;  // (workgroup-uniform)
      static_assert(AHEAD >= 2 && AHEAD <= 4, "counted waits below");
      if (younger >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (4 + NA)) : "memory");
      else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (4 + NA)) : "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + NA) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_barrier" ::: "memory");  // (also: every wave has finished its reads of step st - 1, whose slot is filled next)
    if (st + AHEAD < s_end) issue(st + AHEAD);
    const uint32_t so = (uint32_t)((st % NS) * SLOT);
    // (asm: hipcc would put vmcnt(0) in front of LDS reads it can see while an LDS-DMA is in flight; every read's result is
    // pinned behind the lgkmcnt(0) below by a "+v" operand)
    u32x4 xr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(xr[j]) : "v"(xa[j] + so));
    bf16x8 ar[4][4][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < (I8 ? 2 : 1); ++u) asm volatile("ds_read_b128 %0, %1" : "=v"(ar[t][j][u]) : "v"(aa[t][j][u] + so));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]));
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < (I8 ? 2 : 1); ++u) asm volatile("" : "+v"(ar[t][j][u]));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (I8) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {  // bytes 8 u .. 8 u + 7 of the chunk -> 8 halves (i8x32_to_f16's arithmetic)
          u32x4 f;
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            const uint32_t tb = xr[j][2 * u + d] ^ 0x80808080u;
            f[2 * d] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, __builtin_amdgcn_perm(0x64646464u, tb, 0x04010400u)) + bias);
            f[2 * d + 1] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, __builtin_amdgcn_perm(0x64646464u, tb, 0x04030402u)) + bias);
          }
          const bf16x8 xf = __builtin_bit_cast(bf16x8, f);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = mfma_32x32x16<true>(xf, ar[t][j][u], acc[t]);
        }
      } else {
        const bf16x8 xf = __builtin_bit_cast(bf16x8, xr[j]);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = mfma_32x32x16<false>(xf, ar[t][j][0], acc[t]);
      }
    }
  }
  // D layout: col n = lane & 31, row m = (reg & 3) + 8 (reg >> 2) + 4 h.  part[c][rg * 32 + m][n]
  const int rg = tile * 4 + wave;
  if (rg < row_groups) {
    float* dst = part + ((int64_t)c * row_groups + rg) * XA_ROWS * RP;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) dst[((e & 3) + 8 * (e >> 2) + 4 * h) * RP + 32 * t + r31] = acc[t][e];
  }
}

// The B16 form of the kernel above with the ACTIVATION QUANTIZER in front of it (block_fp activations in blocks of 16, rank
// 65..128: the OPT configurations and q/k/v groups, which the 32-row fused kernel k_quant_xa16 does not cover - its waves fetch
// their own A^T fragments, 64 KB per workgroup at rank 128).  Instead of an LDS-DMA of the bf16 image, every wave loads the
// 16-bit source of its 32 rows x 64 k (four 16-byte requests per lane and step, 8 lanes per 128-byte line, two steps ahead in
// two register sets), quantizes it - a block of 16 = two neighbouring lanes, the maximum crosses with one DPP - and writes the
// bf16 words both to the step's LDS slot (where the DMA would have put them: same source-side swizzle) and to the image the
// GEMM reads.  One pass over x instead of two launches and a re-read of the image.  vmcnt counts loads, stores and LDS-DMA
// together in issue order (MI355X_MICROARCH.md): behind a step's loads lie the previous step's 4 image stores and the next
// step's 4 + NA requests.
template <int DT, int NT, int WAVES, int TROWS = ROWS>  // WAVES waves share TROWS / 32 row groups x NT rank tiles; TROWS = 128 or 64 token rows
__global__ __launch_bounds__(64 * WAVES) void k_quant_xa128(const uint8_t* __restrict__ x, int64_t M, int64_t K, int64_t ldx_b, QP q,
                                                            uint8_t* __restrict__ xq, int64_t Kp, const bf16_t* __restrict__ a_img,
                                                            int row_groups, int nchunk, int steps_per_chunk, int steps_total,
                                                            float* __restrict__ part) {
  static_assert(DT == LQER_F16 || DT == LQER_BF16, "16-bit sources");
  static_assert(WAVES == 4 || WAVES == 8, "4 or 8 waves");
  static_assert(TROWS == 128 || TROWS == 64, "128- or 64-row tiles");
  constexpr int XS = TROWS * BKB;                  // the tile's activation slot
  constexpr int RP = 32 * NT, AROW = 128, A_SLOT = RP * AROW, SLOT = XS + A_SLOT;
  constexpr int XP = (TROWS / 8) / WAVES;          // activation pieces (8 rows x 128 B of image) a wave quantizes per step
  static_assert(XP >= 1, "at most one wave per activation piece");
  constexpr int AP_ALL = RP / 8;                   // A^T pieces (8 rows x 128 B) per step
  constexpr int AP = (AP_ALL + WAVES - 1) / WAVES; // per wave (a wave without a piece of its own repeats another one: same bytes)
  constexpr int NRG = TROWS / 32;                  // 32-row groups of the tile
  constexpr int TG = WAVES / NRG;                  // waves per 32-row group
  constexpr int TW = (NT + TG - 1) / TG;           // rank tiles per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r31 = lane & 31, h = lane >> 5;
  const int tile = blockIdx.x / nchunk, c = blockIdx.x - tile * nchunk;
  const int s_begin = c * steps_per_chunk;
  const int s_end = s_begin + steps_per_chunk < steps_total ? s_begin + steps_per_chunk : steps_total;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem;
  const int64_t row0 = (int64_t)tile * TROWS;
  const int rows_here = (int)(M - row0 < TROWS ? M - row0 : TROWS);  // (>= 1: the grid has ceil(M / TROWS) tiles)
  auto make_rs = [](const uint8_t* base, uint32_t range) {
    const unsigned long long b64 = (unsigned long long)base;
    return (u32x4){(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b64),
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b64 >> 32)) & 0xffffu,
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)range), 0x00020000u};
  };
  // rows of the tile beyond M read as zeros through the range check; the image has its padded rows
  const u32x4 x_rs = make_rs(x + row0 * ldx_b, (uint32_t)(rows_here * ldx_b));
  const u32x4 q_rs = make_rs(xq + row0 * Kp * 2, (uint32_t)(TROWS * Kp * 2));
  const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a_img, 0, (int)(RP * Kp * 2), 0x00020000);
  int gx_voff[4], qs_voff[4], kc[4], a_voff[4], a_piece[4];
  uint32_t ldw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = wave * XP + (i < XP ? i : 0);
    const int row = piece * 8 + (lane >> 3);
    const int cs = (lane & 7) ^ ((row >> 1) & 7);  // the source chunk this lane's LDS position holds (k_xa_partial_lds's swizzle)
    gx_voff[i] = row * (int)ldx_b + cs * 16;
    qs_voff[i] = row * (int)Kp * 2 + cs * 16;
    kc[i] = cs * 8;
    ldw[i] = lds0 + piece * 1024 + lane * 16;
    a_voff[i] = 0, a_piece[i] = 0;
  }
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    a_piece[i] = (wave + i * WAVES) % AP_ALL;
    const int row = 8 * a_piece[i] + (lane >> 3);
    a_voff[i] = row * (int)Kp * 2 + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
  }
  // Every step issues the requests of step st + 2 - beyond the chunk with out-of-range offsets (zeros, no traffic; the LDS-DMA
  // zero-fills a slot nobody reads) - so that ONE counted wait without a branch around it serves every step: a branch made
  // hipcc copy the destination registers of the loads in front of the wait that retires them.
  auto issue = [&](int st, u32x4& x0, u32x4& x1, u32x4& x2, u32x4& x3) {
    const int slot = st % 3;
    const int oob = st < s_end ? 0 : 0x40000000;  // (the voffset takes part in the range check)
#pragma unroll
    for (int i = 0; i < AP; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_void*)(smem + slot * SLOT + XS + a_piece[i] * 1024), 16, a_voff[i] | oob,
                                               st * AROW, 0, 0);
    const int so = __builtin_amdgcn_readfirstlane(st * BKB);
    if constexpr (XP == 4)
      asm volatile("buffer_load_dwordx4 %0, %4, %8, %9 offen\n\tbuffer_load_dwordx4 %1, %5, %8, %9 offen\n\t"
                   "buffer_load_dwordx4 %2, %6, %8, %9 offen\n\tbuffer_load_dwordx4 %3, %7, %8, %9 offen"
                   : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3)
                   : "v"(gx_voff[0] | oob), "v"(gx_voff[1] | oob), "v"(gx_voff[2] | oob), "v"(gx_voff[3] | oob), "s"(x_rs), "s"(so)
                   : "memory");
    else if constexpr (XP == 2)
      asm volatile("buffer_load_dwordx4 %0, %2, %4, %5 offen\n\tbuffer_load_dwordx4 %1, %3, %4, %5 offen"
                   : "=&v"(x0), "=&v"(x1)
                   : "v"(gx_voff[0] | oob), "v"(gx_voff[1] | oob), "s"(x_rs), "s"(so)
                   : "memory");
    else
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&v"(x0) : "v"(gx_voff[0] | oob), "s"(x_rs), "s"(so) : "memory");
  };
  // fragment reads: activation rows of this wave's 32-row group, rank tiles th * TW .. of the A^T slab
  const int rgi = wave % NRG, th = wave / NRG;
  const int xrow = rgi * 32 + r31;
  uint32_t xa[4], aa[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    xa[j] = lds0 + xrow * 128 + (((2 * j + h) ^ ((xrow >> 1) & 7)) << 4);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int tt = th * TW + t < NT ? th * TW + t : NT - 1;  // (a tile slot beyond the rank: re-reads the last one, never stored)
      const int n = 32 * tt + r31;
      aa[t][j] = lds0 + XS + n * 128 + (((2 * j + h) ^ ((n >> 1) & 7)) << 4);
    }
  }
  f32x16 acc[4];  // (TW used; fixed sizes: an array sized by a template constant and captured by a lambda loses the kernel's host stub)
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  typedef __attribute__((ext_vector_type(2))) float f2;
  // quantize the lane's 8 values of piece i (step st), write LDS + image
  auto quant_piece = [&](int st, int i, u32x4 wd) {
    const bool valid = st * 64 + kc[i] < K;  // (k beyond K inside the padded step: zeros)
    float v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t u = valid ? wd[j] : 0u;
      if constexpr (DT == LQER_F16) {
        typedef __attribute__((ext_vector_type(2))) _Float16 h2;
        const h2 hv = __builtin_bit_cast(h2, u);
        v[2 * j] = (float)hv[0], v[2 * j + 1] = (float)hv[1];
      } else {
        v[2 * j] = __uint_as_float(u << 16), v[2 * j + 1] = __uint_as_float(u & 0xffff0000u);
      }
    }
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[j]));
    // the block's other half lies in lane ^ 1 (chunks 2b, 2b + 1: the swizzle keeps the pair together)
    amax = fmaxf(amax, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(amax), 0xB1, 0xf, 0xf, true)));
    const bool any = amax > 0.f;
    const int e = block_exponent_u(any ? amax : 1.0f, q);
    uint32_t w[4];
    if (__builtin_expect(mxint16_fast_ok(e, q), 1)) {  // (per block, like the standalone quantizer: the two routes differ in the sign of a zero)
      const float sc = __uint_as_float((uint32_t)(127 + q.mbits - e) << 23);
      const float inv = __uint_as_float((uint32_t)(127 + e - q.mbits) << 23);
      const float es = 1e-9f * sc, lo = -q.mneg, hi = q.mmax;
      const f2 magic = {12582912.0f, 12582912.0f};
#pragma unroll
      for (int j = 0; j < 4; ++j) {  // (mxint16_bf16_fast's arithmetic, common.h)
        const f2 xv = {v[2 * j], v[2 * j + 1]};
        const f2 cc = {copysignf(es, xv[0]), copysignf(es, xv[1])};
        f2 r = (__builtin_elementwise_fma(xv, (f2){sc, sc}, cc) + magic) - magic;
        r[0] = __builtin_amdgcn_fmed3f(r[0], lo, hi);
        r[1] = __builtin_amdgcn_fmed3f(r[1], lo, hi);
        const f2 val = r * (f2){inv, inv};
        uint32_t b0 = __float_as_uint(val[0]), b1 = __float_as_uint(val[1]);
        if constexpr (DT != LQER_F16) {  // (fp16 cannot hold a non-zero |x| <= 1e-8)
          b0 = fabsf(xv[0]) <= 1e-8f ? 0u : b0;
          b1 = fabsf(xv[1]) <= 1e-8f ? 0u : b1;
        }
        w[j] = any ? __builtin_amdgcn_perm(b1, b0, 0x07060302u) : 0u;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t b0 = exact_bf16_bits(ldexpf(mxint_mantissa(v[2 * j], e, q), e - q.mbits));
        const uint32_t b1 = exact_bf16_bits(ldexpf(mxint_mantissa(v[2 * j + 1], e, q), e - q.mbits));
        w[j] = any ? (b0 | (b1 << 16)) : 0u;
      }
    }
    const u32x4 wv = {w[0], w[1], w[2], w[3]};
    const int so = __builtin_amdgcn_readfirstlane(st * BKB);
    // (s_nop: a VALU write of the data registers of a > 64-bit store needs a wait state the hazard recogniser cannot add here)
    asm volatile("ds_write_b128 %0, %1\n\tbuffer_store_dwordx4 %1, %2, %3, %4 offen\n\ts_nop 1"
                 :: "v"(ldw[i] + (uint32_t)((st % 3) * SLOT)), "v"(wv), "v"(qs_voff[i]), "s"(q_rs), "s"(so) : "memory");
  };
  auto body = [&](int st, u32x4& x0, u32x4& x1, u32x4& x2, u32x4& x3) {
    // the step's own requests have landed: behind them lie at most the previous step's XP image stores (waited for as well)
    // and the next step's XP + AP requests
    if constexpr (XP == 4) {
      asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : [n] "n"(XP + AP) : "memory");
      quant_piece(st, 0, x0), quant_piece(st, 1, x1), quant_piece(st, 2, x2), quant_piece(st, 3, x3);
    } else if constexpr (XP == 2) {
      asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(x0), "+v"(x1) : [n] "n"(XP + AP) : "memory");
      quant_piece(st, 0, x0), quant_piece(st, 1, x1);
    } else {
      asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(x0) : [n] "n"(XP + AP) : "memory");
      quant_piece(st, 0, x0);
    }
    // everybody's words of step st are in the slot (and every wave is past its reads of step st - 1, whose slot is filled next)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    issue(st + 2, x0, x1, x2, x3);
    const uint32_t so = (uint32_t)((st % 3) * SLOT);
    u32x4 xr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(xr[j]) : "v"(xa[j] + so));
    bf16x8 ar[4][4];
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(ar[t][j]) : "v"(aa[t][j] + so));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]));
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(ar[t][j]));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bf16x8 xf = __builtin_bit_cast(bf16x8, xr[j]);
#pragma unroll
      for (int t = 0; t < TW; ++t) acc[t] = mfma_32x32x16<false>(xf, ar[t][j], acc[t]);
    }
  };
  u32x4 p0, p1, p2, p3, q0, q1, q2, q3;  // the two register sets of source chunks in flight (XP of each used)
  issue(s_begin, p0, p1, p2, p3);
  issue(s_begin + 1, q0, q1, q2, q3);
  for (int st = s_begin; st < s_end; st += 2) {
    body(st, p0, p1, p2, p3);
    if (st + 1 < s_end) body(st + 1, q0, q1, q2, q3);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the last image stores and the requests past the chunk)
  const int rg = tile * NRG + rgi;
  if (rg < row_groups) {
    float* dst = part + ((int64_t)c * row_groups + rg) * XA_ROWS * RP;
#pragma unroll
    for (int t = 0; t < TW; ++t) {
      const int tt = th * TW + t;
      if (tt < NT) {
#pragma unroll
        for (int e = 0; e < 16; ++e) dst[((e & 3) + 8 * (e >> 2) + 4 * h) * RP + 32 * tt + r31] = acc[t][e];
      }
    }
  }
}

#ifndef LQER_QXA128_WAVES
#define LQER_QXA128_WAVES 8
#endif
#ifndef LQER_QXA128_ROWS
#define LQER_QXA128_ROWS 64  // 64-row tiles x twice the K per chunk: half the partial tiles to write and to reduce (-1.3 .. -1.7 us per
#endif                      // rank-96 / 128 forward at K = 4096 against 128-row tiles; +-0 at K = 16384)
constexpr int qx_rows = LQER_QXA128_ROWS;  // token rows per workgroup of the fused quantizer (its K chunks double at 64)
template <int DT, int NT>
static void launch_q(const void* x, int64_t M, int64_t K, int64_t ldx_b, const QP& q, bf16_t* xq, int64_t Kp, const bf16_t* a_img,
                     int row_groups, int tiles, int nch, int spc, int steps_total, float* part, hipStream_t st) {
  constexpr int W = LQER_QXA128_WAVES, R = LQER_QXA128_ROWS;
  constexpr int lds = 3 * (R * BKB + 32 * NT * 128);
  static LdsLimitOnce once;
  once.set((const void*)k_quant_xa128<DT, NT, W, R>, lds);
  k_quant_xa128<DT, NT, W, R><<<(unsigned)(tiles * nch), 64 * W, lds, st>>>((const uint8_t*)x, M, K, ldx_b, q, (uint8_t*)xq, Kp, a_img,
                                                                           row_groups, nch, spc, steps_total, part);
}

template <int NT, bool I8>
static void launch(const void* xq, int64_t x_ld, const bf16_t* a_img, int64_t Kp, int row_groups, int tiles, int nch, int spc,
                   int steps_total, float* part, hipStream_t st) {
  static LdsLimitOnce once;
  once.set((const void*)k_xa_partial_lds<NT, I8>, lds_bytes(NT, I8));
  k_xa_partial_lds<NT, I8><<<(unsigned)(tiles * nch), 256, lds_bytes(NT, I8), st>>>((const uint8_t*)xq, x_ld, a_img, Kp, row_groups, nch,
                                                                                    spc, steps_total, part);
}
}  // namespace xal

// Fixed-order sum over the chunks + A_out + bf16 store.  One lane per 4 consecutive rank entries; the G = L/4
// lanes of a block (G a power of two <= 64, lanes of one wave) share their max through xor-shuffles.
template <int G>
__global__ __launch_bounds__(256) void k_xa_reduce4(const float* __restrict__ part, XaPlan plan, int rp, QP q,
                                                    bf16_t* __restrict__ xaq, const float* __restrict__ rowscale = nullptr) {
  const int64_t total = (int64_t)plan.row_groups * XA_ROWS * rp / 4;  // float4 items (rp/4 per row, a multiple of G)
  const int64_t chunk_stride = (int64_t)plan.row_groups * XA_ROWS * rp;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (64-thread workgroups when the grid would not fill the chip)
  const bool live = idx < total;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
    const float* src = part + idx * 4;
    // chunks summed in ascending order, 16 (then 8) loads in flight at a time: a launch-latency kernel, every round trip counts
    int c = 0;
    for (; c + 16 <= plan.nchunk; c += 16) {
      float4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = *(const float4*)(src + (c + u) * chunk_stride);
#pragma unroll
      for (int u = 0; u < 16; ++u) s.x += v[u].x, s.y += v[u].y, s.z += v[u].z, s.w += v[u].w;
    }
    for (; c + 8 <= plan.nchunk; c += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *(const float4*)(src + (c + u) * chunk_stride);
#pragma unroll
      for (int u = 0; u < 8; ++u) s.x += v[u].x, s.y += v[u].y, s.z += v[u].z, s.w += v[u].w;
    }
    for (; c < plan.nchunk; ++c) {
      const float4 v = *(const float4*)(src + c * chunk_stride);
      s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
  }
  if (rowscale && live) {  // int8 activation image: the sums are of mantissas, the row's power-of-two scale comes last (exact)
    const float sc = rowscale[idx / (rp / 4)];
    s.x *= sc, s.y *= sc, s.z *= sc, s.w *= sc;
  }
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
  *(uint2*)(xaq + idx * 4) = make_uint2(w[0], w[1]);
}

// A_out_quantizer = passthrough (reference quantizers/passthrough.py:1; the *-int.toml templates leave A_out at the
// x quantizer's pass-through, linear.py:120-124): the fp32 sum is handed on as LA bf16 limbs laid side by side,
// xaq [rows][LA * rp], limb l of column n at l * rp + n (16 significand bits with two limbs - the reference keeps
// 11 (fp16 tensors) or 8 (bf16) here -, all 24 with three); B^T is repeated LA times along r to match.
template <int LA>
__global__ __launch_bounds__(256) void k_xa_reduce_limbs(const float* __restrict__ part, XaPlan plan, int rp,
                                                         bf16_t* __restrict__ xaq, const float* __restrict__ rowscale = nullptr) {
  const int64_t total = (int64_t)plan.row_groups * XA_ROWS * rp / 4;
  const int64_t chunk_stride = (int64_t)plan.row_groups * XA_ROWS * rp;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const float* src = part + idx * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  int c = 0;
  for (; c + 8 <= plan.nchunk; c += 8) {  // ascending, 8 loads in flight, as in k_xa_reduce4
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *(const float4*)(src + (c + u) * chunk_stride);
#pragma unroll
    for (int u = 0; u < 8; ++u) s.x += v[u].x, s.y += v[u].y, s.z += v[u].z, s.w += v[u].w;
  }
  for (; c < plan.nchunk; ++c) {
    const float4 v = *(const float4*)(src + c * chunk_stride);
    s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
  }
  const int rq = rp / 4;
  const int64_t row = idx / rq;
  const int n0 = (int)(idx - row * rq) * 4;
  const float sc = rowscale ? rowscale[row] : 1.0f;
  float v[4] = {s.x * sc, s.y * sc, s.z * sc, s.w * sc};
#pragma unroll
  for (int l = 0; l < LA; ++l) {
    uint32_t w[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bf16_t b0 = f32_to_bf16_rne(v[2 * i]), b1 = f32_to_bf16_rne(v[2 * i + 1]);
      v[2 * i] -= __uint_as_float((uint32_t)b0 << 16);
      v[2 * i + 1] -= __uint_as_float((uint32_t)b1 << 16);
      w[i] = (uint32_t)b0 | ((uint32_t)b1 << 16);
    }
    *(uint2*)(xaq + row * (LA * rp) + l * rp + n0) = make_uint2(w[0], w[1]);
  }
}

// Generic block length (not 4 * 2^g): one lane per block, serial.
__global__ __launch_bounds__(256) void k_xa_reduce_blk(const float* __restrict__ part, XaPlan plan, int rp, QP q,
                                                       bf16_t* __restrict__ xaq, const float* __restrict__ rowscale = nullptr) {
  const int L = (q.block <= 0 || q.block >= rp) ? rp : q.block;
  const int nb = rp / L;
  const int64_t total = (int64_t)plan.row_groups * XA_ROWS * nb;
  const int64_t chunk_stride = (int64_t)plan.row_groups * XA_ROWS * rp;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t row = idx / nb;
    const int b0 = (int)(idx - row * nb) * L;
    const float* src = part + row * rp + b0;
    bf16_t* dst = xaq + row * rp + b0;
    const float sc = rowscale ? rowscale[row] : 1.0f;
    float amax = 0.f;
    for (int k = 0; k < L; ++k) {
      float sum = src[k];
      for (int c = 1; c < plan.nchunk; ++c) sum += src[c * chunk_stride + k];
      amax = fmaxf(amax, fabsf(sum * sc));
    }
    const bool any = amax > 0.f;
    const int e = any ? block_exponent(amax, q) : 0;
    for (int k = 0; k < L; ++k) {
      float sum = src[k];
      for (int c = 1; c < plan.nchunk; ++c) sum += src[c * chunk_stride + k];
      sum *= sc;
      const float mv = any ? mxint_mantissa(sum, e, q) : 0.f;
      dst[k] = (bf16_t)exact_bf16_bits(ldexpf(mv, e - q.mbits));
    }
  }
}

// ---- fused: activation quantize (blocks of 16) + side-path partial GEMM --------------------------------
// One workgroup = 32 token rows x 256 k.  Phase 1: every lane quantizes two 16-element blocks (consecutive
// lanes = consecutive 32-byte pieces of a row: fully coalesced), writes the bf16 image to HBM and to an LDS
// slab (XOR-swizzled 16-byte chunks).  Phase 2: the 4 waves run v_mfma_f32_32x32x16_bf16 over 64 k each,
// A^T fragments straight from L2; their partial tiles are summed in a fixed order through LDS and written as
// ONE partial per workgroup.  The activation image is not read back from HBM for the side path.

template <int DT>
__device__ __forceinline__ void qx_load16(const void* x, int64_t base, int64_t k0, int64_t cols, bool vec, float (&v)[16]) {
  if (vec && k0 + 16 <= cols) {
    if constexpr (DT == LQER_F32) {
      const float4* p = (const float4*)((const float*)x + base + k0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 t = p[i];
        v[4 * i] = t.x, v[4 * i + 1] = t.y, v[4 * i + 2] = t.z, v[4 * i + 3] = t.w;
      }
    } else {
      const uint4* p = (const uint4*)((const bf16_t*)x + base + k0);
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
    for (int i = 0; i < 16; ++i) v[i] = (k0 + i < cols) ? load_elem<DT>(x, base + k0 + i) : 0.0f;
  }
}

__device__ __forceinline__ int qx_swz(int row, int chunk) { return row * (QX_K * 2) + ((chunk ^ (row & 15)) << 4); }

template <int DT, int NT>
__global__ __launch_bounds__(QX_K) void k_quant_xa16(const void* __restrict__ x, int64_t M, int64_t K, int64_t ldx, bool vec,
                                                    QP q, bf16_t* __restrict__ xq, int64_t Kp,
                                                    const bf16_t* __restrict__ a_t, int a_limbs, int rp, int row_groups,
                                                    float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) unsigned char slab[32 * QX_K * 2];   // 16 KiB
  __shared__ __attribute__((aligned(16))) float red[(QX_WAVES > 1 ? QX_WAVES - 1 : 1) * 32 * 32 * NT];  // waves 1.. park their tiles here
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nchunk = (int)((Kp + QX_K - 1) / QX_K);
  const int rg = blockIdx.x / nchunk, c = blockIdx.x - rg * nchunk;
  const int64_t kbase = (int64_t)c * QX_K;
  // the wave's A^T fragments (limb 0) are fetched first, so their L2 latency overlaps the activation loads
  const int r = lane & 31, h = lane >> 5;
  const int64_t kw = kbase + 64 * wave;
  bf16x8 af0[4][NT];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = t * 32 + r;
      af0[ks][t] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (n < rp && kw < Kp && a_limbs > 0) af0[ks][t] = *(const bf16x8*)(a_t + (int64_t)n * Kp + kw + 16 * ks + 8 * h);
    }
  // ---- phase 1: quantize
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    const int s = tid + s2 * QX_K;          // 2 QX_K blocks of 16: row = s / (QX_K / 16), segment = s % (QX_K / 16)
    const int row = s / (QX_K / 16), seg = s % (QX_K / 16);
    const int64_t m = (int64_t)rg * 32 + row, k0 = kbase + seg * 16;
    uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (k0 < Kp) {
      if (m < M && k0 < K) {
        float v[16];
        qx_load16<DT>(x, m * ldx, k0, K, vec, v);
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) amax = fmaxf(amax, fabsf(v[i]));
        if (amax > 0.f) {
          const int e = block_exponent_u(amax, q);
          if (mxint16_fast_ok(e, q)) {
            mxint16_bf16_fast<DT != LQER_F16>(v, e, q, w);
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
      uint4* dst = (uint4*)(xq + m * Kp + k0);   // rows up to the padded M are allocated
#ifdef LQER_XQ_STORE_MODS  // cache-policy experiment (tools/ab_step.py): e.g. -DLQER_XQ_STORE_MODS='"nt"'
      {
        const u32x4 lo = {w[0], w[1], w[2], w[3]}, hi = {w[4], w[5], w[6], w[7]};
        asm volatile("global_store_dwordx4 %0, %1, off " LQER_XQ_STORE_MODS "\n\tglobal_store_dwordx4 %0, %2, off offset:16 " LQER_XQ_STORE_MODS
                     ::"v"(dst), "v"(lo), "v"(hi) : "memory");
      }
#else
      dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
      dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
#endif
    }
    *(uint4*)(slab + qx_swz(row, 2 * seg)) = make_uint4(w[0], w[1], w[2], w[3]);
    *(uint4*)(slab + qx_swz(row, 2 * seg + 1)) = make_uint4(w[4], w[5], w[6], w[7]);
  }
  __syncthreads();
  // ---- phase 2: partial side GEMM, wave w covers k [64w, 64w + 64) of the slab
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
  if (kw < Kp) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 xf = *(const bf16x8*)(slab + qx_swz(r, (64 * wave + 16 * ks) / 8 + h));
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, af0[ks][t], acc[t], 0, 0, 0);
      for (int l = 1; l < a_limbs; ++l) {  // fp16 / fp32 A: further exact bf16 limbs
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int n = t * 32 + r;
          bf16x8 af = {0, 0, 0, 0, 0, 0, 0, 0};
          if (n < rp) af = *(const bf16x8*)(a_t + ((int64_t)l * rp + n) * Kp + kw + 16 * ks + 8 * h);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, af, acc[t], 0, 0, 0);
        }
      }
    }
  }
  // fixed-order combine: ((w0 + w1) + w2) + w3
  if (wave > 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) red[(((wave - 1) * NT + t) * 16 + j) * 64 + lane] = acc[t][j];
  }
  __syncthreads();
  if (wave == 0) {
    float* dst = part + ((int64_t)c * row_groups + rg) * XA_ROWS * rp;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = t * 32 + r;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float sum = acc[t][j];
#pragma unroll
        for (int w2 = 0; w2 < QX_WAVES - 1; ++w2) sum += red[((w2 * NT + t) * 16 + j) * 64 + lane];
        if (n < rp) dst[((j & 3) + 8 * (j >> 2) + 4 * h) * rp + n] = sum;
      }
    }
  }
}

// Decode sizes: the small-M GEMM can sum the fused kernel's partial tiles and apply A_out itself (gemm_smallm.hip) when
// A_out blocks are 16 wide - the reduce launch is then skipped.
bool xa_fused_partials_ok(const QP& qx, const QP& qa, int64_t r) {
  const int rp = (int)lqer_padded_r(r);
  return qx.kind == LQER_Q_MXINT && qx.block == 16 && qx.mbits <= 8 && rp <= 64 && qa.kind == LQER_Q_MXINT && qa.mbits <= 8 &&
         qa.block == 16;
}
void xa_fused_plan(int64_t M, int64_t K, int64_t r, int* nchunk, int64_t* cstride) {
  const int64_t Kp = lqer_padded_k(K);
  *nchunk = (int)((Kp + QX_K - 1) / QX_K);
  *cstride = ((M + XA_ROWS - 1) / XA_ROWS) * XA_ROWS * lqer_padded_r(r);
}

// Fused activation quantize + side path.  Returns LQER_E_UNSUPPORTED (without launching) when the shape or
// formats are outside what the fused kernel covers; the caller then runs the two separate steps.
int quant_xa_fused_dispatch(const void* x, int dtype, int64_t M, int64_t K, int64_t ldx, const QP& qx, bf16_t* xq,
                            const bf16_t* a_t, int a_limbs, int64_t r, const QP& qa, bf16_t* xaq, float* scratch,
                            size_t scratch_bytes, hipStream_t st) {
  const int64_t Kp = lqer_padded_k(K);
  const int rp = (int)lqer_padded_r(r);
  if (qx.kind != LQER_Q_MXINT || qx.block != 16 || qx.mbits > 8 || rp > 128 || qa.kind != LQER_Q_MXINT || qa.mbits > 8)
    return LQER_E_UNSUPPORTED;
  const int La = (qa.block <= 0 || qa.block >= rp) ? rp : qa.block;
  const int G = La / 4;
  if (rp % La != 0 || La % 4 != 0 || (G & (G - 1)) != 0 || G > 64) return LQER_E_UNSUPPORTED;
  if (M == 0) return LQER_OK;
#ifndef LQER_NO_QXA128
#ifdef LQER_QXA128_ALL  // experiment: every rank through the tile kernel (rank 32: one rank tile, three of four waves per row group idle in the MFMA part)
  if (rp % 32 == 0 && M >= 512 && a_limbs == 1 && dtype != LQER_F32 && K % 8 == 0 && (uintptr_t)x % 16 == 0 && (ldx * 2) % 16 == 0) {
#else
  if (rp > 64) {
#endif
    // rank 65..128: 128-row tiles, both MFMA operands through LDS (xal::k_quant_xa128), the plan of the LDS-staged side GEMM
    const int64_t ldx_b = ldx * 2;
    if (rp % 32 != 0 || a_limbs != 1 || M < 512 || dtype == LQER_F32 || K % 8 != 0 || (uintptr_t)x % 16 != 0 || ldx_b % 16 != 0 ||
        xal::ROWS * ldx_b >= 0x40000000)
      return LQER_E_UNSUPPORTED;
    const XaPlan base = xa_plan(M, Kp);
    const int steps_total = (int)(Kp * 2 / xal::BKB), tiles = (int)((M + xal::qx_rows - 1) / xal::qx_rows);
    int nch = LQER_XAL_WGS / tiles;
    nch = nch < 1 ? 1 : (nch > base.nchunk ? base.nchunk : nch);
    nch = nch < steps_total ? nch : steps_total;
    const int spc = (steps_total + nch - 1) / nch;
    nch = (steps_total + spc - 1) / spc;
    XaPlan plan = base;
    plan.nchunk = nch;
    const size_t need = (size_t)nch * plan.row_groups * XA_ROWS * rp * sizeof(float);
    if (!scratch || scratch_bytes < need) return LQER_E_UNSUPPORTED;
    const int nt = rp / 32;
#ifdef LQER_QXA128_ALL
#define QX128(DT) (nt == 1 ? xal::launch_q<DT, 1>(x, M, K, ldx_b, qx, xq, Kp, a_t, plan.row_groups, tiles, nch, spc, steps_total, scratch, st) \
                 : nt == 2 ? xal::launch_q<DT, 2>(x, M, K, ldx_b, qx, xq, Kp, a_t, plan.row_groups, tiles, nch, spc, steps_total, scratch, st) \
                 : nt == 3 ? xal::launch_q<DT, 3>(x, M, K, ldx_b, qx, xq, Kp, a_t, plan.row_groups, tiles, nch, spc, steps_total, scratch, st) \
                           : xal::launch_q<DT, 4>(x, M, K, ldx_b, qx, xq, Kp, a_t, plan.row_groups, tiles, nch, spc, steps_total, scratch, st))
#else
#define QX128(DT) (nt == 3 ? xal::launch_q<DT, 3>(x, M, K, ldx_b, qx, xq, Kp, a_t, plan.row_groups, tiles, nch, spc, steps_total, scratch, st) \
                           : xal::launch_q<DT, 4>(x, M, K, ldx_b, qx, xq, Kp, a_t, plan.row_groups, tiles, nch, spc, steps_total, scratch, st))
#endif
    if (dtype == LQER_F16) QX128(LQER_F16); else QX128(LQER_BF16);
#undef QX128
    if (!xaq) return check_launch("quantize_act_xa");
    const int64_t items = (int64_t)plan.row_groups * XA_ROWS * rp / 4;
    const unsigned bs = items <= 128 * 256 ? 64 : 256;
    const unsigned grid2 = (unsigned)((items + bs - 1) / bs);
    switch (G) {
      case 1: k_xa_reduce4<1><<<grid2, bs, 0, st>>>(scratch, plan, rp, qa, xaq); break;
      case 2: k_xa_reduce4<2><<<grid2, bs, 0, st>>>(scratch, plan, rp, qa, xaq); break;
      case 4: k_xa_reduce4<4><<<grid2, bs, 0, st>>>(scratch, plan, rp, qa, xaq); break;
      case 8: k_xa_reduce4<8><<<grid2, bs, 0, st>>>(scratch, plan, rp, qa, xaq); break;
      case 16: k_xa_reduce4<16><<<grid2, bs, 0, st>>>(scratch, plan, rp, qa, xaq); break;
      default: k_xa_reduce4<32><<<grid2, bs, 0, st>>>(scratch, plan, rp, qa, xaq); break;
    }
    return check_launch("quantize_act_xa");
  }
#else
  if (rp > 64) return LQER_E_UNSUPPORTED;
#endif
  XaPlan plan;
  plan.row_groups = (int)((M + XA_ROWS - 1) / XA_ROWS);
  plan.nchunk = (int)((Kp + QX_K - 1) / QX_K);
  plan.kc = QX_K;
  const size_t need = (size_t)plan.nchunk * plan.row_groups * XA_ROWS * rp * sizeof(float);
  if (!scratch || scratch_bytes < need) return LQER_E_UNSUPPORTED;  // sized for the unfused plan: fall back
  const int esz = dtype == LQER_F32 ? 4 : 2;
  const bool vec = ((uintptr_t)x % 16 == 0) && ((ldx * esz) % 16 == 0);
  const unsigned grid = (unsigned)(plan.row_groups * plan.nchunk);
#define QX_LAUNCH(DT, NT) k_quant_xa16<DT, NT><<<grid, QX_K, 0, st>>>(x, M, K, ldx, vec, qx, xq, Kp, a_t, a_limbs, rp, plan.row_groups, scratch)
  const int nt = (rp + 31) / 32;
  switch (dtype) {
    case LQER_F32: if (nt == 1) QX_LAUNCH(LQER_F32, 1); else QX_LAUNCH(LQER_F32, 2); break;
    case LQER_F16: if (nt == 1) QX_LAUNCH(LQER_F16, 1); else QX_LAUNCH(LQER_F16, 2); break;
    case LQER_BF16: if (nt == 1) QX_LAUNCH(LQER_BF16, 1); else QX_LAUNCH(LQER_BF16, 2); break;
    default: set_error("unknown dtype %d", dtype); return LQER_E_INVALID;
  }
#undef QX_LAUNCH
  if (!xaq) return check_launch("quantize_act_xa");  // the consumer reduces the partial tiles itself
  const int64_t items = (int64_t)plan.row_groups * XA_ROWS * rp / 4;
  const unsigned bs = items <= 128 * 256 ? 64 : 256;  // up to 32 Ki items (C2: 16 Ki): one wave per workgroup spreads them over all CUs
  const unsigned grid2 = (unsigned)((items + bs - 1) / bs);
  switch (G) {
    case 1: k_xa_reduce4<1><<<grid2, bs, 0, st>>>(scratch, plan, rp, qa, xaq); break;
    case 2: k_xa_reduce4<2><<<grid2, bs, 0, st>>>(scratch, plan, rp, qa, xaq); break;
    case 4: k_xa_reduce4<4><<<grid2, bs, 0, st>>>(scratch, plan, rp, qa, xaq); break;
    case 8: k_xa_reduce4<8><<<grid2, bs, 0, st>>>(scratch, plan, rp, qa, xaq); break;
    default: k_xa_reduce4<16><<<grid2, bs, 0, st>>>(scratch, plan, rp, qa, xaq); break;
  }
  return check_launch("quantize_act_xa");
}

// x_limbs: the activation image holds that many bf16 limbs side by side ([Mp][x_limbs * Kp], act_limbs.hip) and a_t
// is repeated as often along k; xa_limbs: limbs of the result when A_out is a pass-through (0 otherwise).
// x_limbs = 0: xq holds fp16 bits and a_t is ONE fp16 image [rp][Kp] (pack.hip::a_f16_dispatch) - v_mfma_*_f16.
// x_limbs = -1: xq holds int8 mantissas [Mp][lqer_padded_k8(K)] with the row scales behind them (the int8 route).
int lowrank_xa_dispatch(const bf16_t* xq, int64_t M, int64_t K, int x_limbs, const bf16_t* a_t, int a_limbs, int64_t r,
                        const QP& q, int xa_limbs, bf16_t* xaq, float* scratch, size_t scratch_bytes, hipStream_t st) {
  const bool x_f16 = x_limbs == 0;
  const bool x_i8 = x_limbs == -1;
  const bool a_f16 = x_i8 && a_limbs == -1;  // int8 mantissas x ONE fp16 image of A^T (lqer_f16_prepare) on the fp16 MFMA
  if (x_f16 || a_f16) a_limbs = 1;
  if (x_f16) x_limbs = 1;
  if (x_i8) x_limbs = 1;
  const int64_t Kp = lqer_padded_k(K) * x_limbs;
  const int64_t Kx = x_i8 ? padded_k8(K) : Kp;  // row stride of the activation image (elements)
  const float* rowscale = x_i8 ? i8_row_scales(xq, M, K) : nullptr;
  const int rp = (int)lqer_padded_r(r);
  const bool pass = q.kind == LQER_Q_PASSTHROUGH;
  const bool fixed = q.kind == LQER_Q_INT;  // (integer: the reduce pass quantizes with the pinned exponent, common.h)
  if (pass ? (xa_limbs != 2 && xa_limbs != 3) : !((q.kind == LQER_Q_MXINT && q.mbits <= 8) || (fixed && q.mmax <= 256.f && q.mneg <= 256.f))) {
    set_error("A_out_quantizer must be block_fp or integer with codes up to 256 (a bf16 image), or passthrough with 2 or 3 limbs, on "
              "the HIP path (got kind %d width %d)", q.kind, q.width);
    return LQER_E_UNSUPPORTED;
  }
  const int L = pass ? 4 : ((q.block <= 0 || q.block >= rp) ? rp : q.block);
  if (rp % L != 0 || L % 2 != 0) {
    set_error("A_out_quantizer block %d does not tile the padded rank %d", q.block, rp);
    return LQER_E_UNSUPPORTED;
  }
  if (rp > 32 * XA_MAX_TILES) {
    set_error("rank %d > %d not supported", (int)r, 32 * XA_MAX_TILES);
    return LQER_E_UNSUPPORTED;
  }
  if (M == 0) return LQER_OK;
  const XaPlan plan = xa_plan(M, Kp);
  const size_t need = (size_t)plan.nchunk * plan.row_groups * XA_ROWS * rp * sizeof(float);
  if (!scratch || scratch_bytes < need) {
    set_error("lowrank_xa: scratch %zu B < %zu B", scratch_bytes, need);
    return LQER_E_WORKSPACE;
  }
  const int nt = (rp + 31) / 32;
#ifndef LQER_XA_NO_LDS
  bool staged = false;
  XaPlan plan_l = plan;
  // the LDS-staged kernel (xal): the int8 route's images at any rank tile count up to 2; the bf16 image x one bf16 limb of A^T
  // for ranks beyond the fused quantizer's (rank 65..128: the OPT configurations).  128-row tiles x K chunks of whole steps,
  // about one workgroup per CU (fewer, longer chunks than the wave-per-chunk plan: less for the reduce pass to read)
  const bool lds_i8 = a_f16 && (rp == 32 || rp == 64) && Kx % xal::BKB == 0;
  const bool lds_b16 = !x_f16 && !x_i8 && x_limbs == 1 && a_limbs == 1 && rp % 32 == 0 && rp > 64 && rp <= 128;
  if ((lds_i8 || lds_b16) && M >= 512) {
    const int64_t x_ld = lds_i8 ? Kx : Kp * 2;  // bytes per activation row (both zero-padded to whole 128-byte steps)
    const int steps_total = (int)(x_ld / xal::BKB), tiles = (int)((M + xal::ROWS - 1) / xal::ROWS);
    int nch = LQER_XAL_WGS / tiles;
    nch = nch < 1 ? 1 : (nch > plan.nchunk ? plan.nchunk : nch);
    nch = nch < steps_total ? nch : steps_total;
    const int spc = (steps_total + nch - 1) / nch;
    nch = (steps_total + spc - 1) / spc;
    plan_l.nchunk = nch;
    if (lds_i8) {
      if (nt == 1) xal::launch<1, true>(xq, x_ld, a_t, Kp, plan.row_groups, tiles, nch, spc, steps_total, scratch, st);
      else xal::launch<2, true>(xq, x_ld, a_t, Kp, plan.row_groups, tiles, nch, spc, steps_total, scratch, st);
    } else {
      if (nt == 3) xal::launch<3, false>(xq, x_ld, a_t, Kp, plan.row_groups, tiles, nch, spc, steps_total, scratch, st);
      else xal::launch<4, false>(xq, x_ld, a_t, Kp, plan.row_groups, tiles, nch, spc, steps_total, scratch, st);
    }
    staged = true;
  }
#else
  const bool staged = false;
  const XaPlan plan_l = plan;
#endif
  // row groups per wave: two while RG x NT accumulator tiles + two activation windows fit the register file
  const int rgw = nt <= 4 ? 2 : 1;
  const int wave_rows = (plan.row_groups + rgw - 1) / rgw;
  const unsigned grid = (unsigned)((wave_rows * plan.nchunk + 3) / 4);
#define XA_CASE(NT, RG)                                                                                      \
  case NT:                                                                                                   \
    if (NT <= 2 && a_limbs * NT <= 4 && plan.kc > 64) { /* A^T fragments one window ahead */                 \
      if (x_f16)                                                                                             \
        k_xa_partial<NT, RG, true, (NT <= 2)><<<grid, 256, 0, st>>>(xq, M, Kp, Kx, a_t, a_limbs, rp, plan, scratch);  \
      else if (a_f16)                                                                                        \
        k_xa_partial<NT, RG, true, (NT <= 2), true, LQER_XA_PF2 != 0><<<grid, 256, 0, st>>>(xq, M, Kp, Kx, a_t, a_limbs, rp, plan, scratch); \
      else if (x_i8)                                                                                         \
        k_xa_partial<NT, RG, false, (NT <= 2), true, LQER_XA_PF2 != 0><<<grid, 256, 0, st>>>(xq, M, Kp, Kx, a_t, a_limbs, rp, plan, scratch); \
      else                                                                                                   \
        k_xa_partial<NT, RG, false, (NT <= 2)><<<grid, 256, 0, st>>>(xq, M, Kp, Kx, a_t, a_limbs, rp, plan, scratch); \
    } else if (x_f16)                                                                                        \
      k_xa_partial<NT, RG, true><<<grid, 256, 0, st>>>(xq, M, Kp, Kx, a_t, a_limbs, rp, plan, scratch);      \
    else if (a_f16)                                                                                          \
      k_xa_partial<NT, RG, true, false, true><<<grid, 256, 0, st>>>(xq, M, Kp, Kx, a_t, a_limbs, rp, plan, scratch); \
    else if (x_i8)                                                                                           \
      k_xa_partial<NT, RG, false, false, true><<<grid, 256, 0, st>>>(xq, M, Kp, Kx, a_t, a_limbs, rp, plan, scratch); \
    else                                                                                                     \
      k_xa_partial<NT, RG><<<grid, 256, 0, st>>>(xq, M, Kp, Kx, a_t, a_limbs, rp, plan, scratch);            \
    break;
  if (!staged) switch (nt) {
    XA_CASE(1, 2) XA_CASE(2, 2) XA_CASE(3, 2) XA_CASE(4, 2) XA_CASE(5, 1) XA_CASE(6, 1) XA_CASE(7, 1) XA_CASE(8, 1)
  }
#undef XA_CASE
  const int G = L / 4;
  if (pass) {
    const int64_t items = (int64_t)plan.row_groups * XA_ROWS * rp / 4;
    const unsigned grid2 = (unsigned)((items + 255) / 256);
    if (xa_limbs == 2)
      k_xa_reduce_limbs<2><<<grid2, 256, 0, st>>>(scratch, plan_l, rp, xaq, rowscale);
    else
      k_xa_reduce_limbs<3><<<grid2, 256, 0, st>>>(scratch, plan_l, rp, xaq, rowscale);
  } else if (L % 4 == 0 && (G & (G - 1)) == 0 && G <= 64) {
    const int64_t items = (int64_t)plan.row_groups * XA_ROWS * rp / 4;
    const unsigned bs = items <= 128 * 256 ? 64 : 256;  // (as in the fused quantizer's reduce above)
    const unsigned grid2 = (unsigned)((items + bs - 1) / bs);
    switch (G) {
      case 1: k_xa_reduce4<1><<<grid2, bs, 0, st>>>(scratch, plan_l, rp, q, xaq, rowscale); break;
      case 2: k_xa_reduce4<2><<<grid2, bs, 0, st>>>(scratch, plan_l, rp, q, xaq, rowscale); break;
      case 4: k_xa_reduce4<4><<<grid2, bs, 0, st>>>(scratch, plan_l, rp, q, xaq, rowscale); break;
      case 8: k_xa_reduce4<8><<<grid2, bs, 0, st>>>(scratch, plan_l, rp, q, xaq, rowscale); break;
      case 16: k_xa_reduce4<16><<<grid2, bs, 0, st>>>(scratch, plan_l, rp, q, xaq, rowscale); break;
      case 32: k_xa_reduce4<32><<<grid2, bs, 0, st>>>(scratch, plan_l, rp, q, xaq, rowscale); break;
      default: k_xa_reduce4<64><<<grid2, bs, 0, st>>>(scratch, plan_l, rp, q, xaq, rowscale); break;
    }
  } else {
    const int64_t total = (int64_t)plan.row_groups * XA_ROWS * (rp / L);
    const unsigned grid2 = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    k_xa_reduce_blk<<<grid2, 256, 0, st>>>(scratch, plan_l, rp, q, xaq, rowscale);
  }
  return check_launch("lowrank_xa");
}

}  // namespace lqer
