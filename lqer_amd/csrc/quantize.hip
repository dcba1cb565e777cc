// MXINT (block floating point) quantizers over matrix rows - the HBM-bound kernels of the path.
//
//  k_quant_seg16 : one lane per 16-element block (the x_quantizer / A_out / B_out default,
//                  reference llama-7b.toml:82-88).  128-bit loads and stores, no cross-lane work.
//  k_quant_row   : one workgroup per row, one shared exponent per row (block_size [1,-1]).
//  k_quant_blk   : one lane per block of any length, serial (cold paths: L = 32, 128, ragged).
//
// Replaces reference quantizers/block_fp.py:7-82 together with the pad/unfold/fold blocking of
// quantizers/utils.py:127-158 and :211-258 (an activation's blocks are runs of L consecutive
// elements of one row, for 2-D and 3-D inputs alike - SURVEY.md §4).
#include <type_traits>

#include "common.h"

namespace lqer {

template <int DT, bool VEC>
__device__ __forceinline__ void load16(const void* x, int64_t base, int64_t k0, int64_t cols, float (&v)[16]) {
  if constexpr (VEC) {
    if constexpr (DT == LQER_F32) {
      const float4* p = (const float4*)((const float*)x + base + k0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float4 t = p[i];
        v[4 * i] = t.x, v[4 * i + 1] = t.y, v[4 * i + 2] = t.z, v[4 * i + 3] = t.w;
      }
    } else {
      const uint4* p = (const uint4*)((const bf16_t*)x + base + k0);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        uint4 t = p[i];
        uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (DT == LQER_F16) {
            typedef __attribute__((ext_vector_type(2))) _Float16 h2;
            h2 h = __builtin_bit_cast(h2, w[j]);
            v[8 * i + 2 * j] = (float)h[0];
            v[8 * i + 2 * j + 1] = (float)h[1];
          } else {
            v[8 * i + 2 * j] = __uint_as_float(w[j] << 16);
            v[8 * i + 2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
          }
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = (k0 + i < cols) ? load_elem<DT>(x, base + k0 + i) : 0.0f;
  }
}

// Quantize 16 values that share exponent e (or are all zero when !any) and emit every requested image.
__device__ __forceinline__ void emit16(const float (&v)[16], bool any, int e, const QP& q, const QuantOut& o,
                                       int64_t row, int64_t k0, int64_t cols) {
  if (o.xq && !o.deq && !o.codes && any && mxint16_fast_ok(e, q)) {  // the activation image alone: packed-fp32 route
    uint32_t w[8];
    mxint16_bf16_fast<true>(v, e, q, w);
    uint4* dst = (uint4*)(o.xq + row * o.cols_p + k0);
    dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
    dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
    return;
  }
  if (o.xq8 && !o.xq && !o.deq && !o.codes && any && mxint16_fast_ok(e, q)) {  // the int8 image alone: packed-fp32 route
    uint32_t w[4];
    mxint16_i8_fast<true>(v, e, q, w);
    *(uint4*)(o.xq8 + row * o.cols_p8 + k0) = make_uint4(w[0], w[1], w[2], w[3]);
    return;
  }
  float m[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) m[i] = any ? mxint_mantissa(v[i], e, q) : 0.0f;
  if (o.xq8) {  // the int8 image: 16 two's-complement mantissas = one 16-byte store
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      w[i] = ((uint32_t)(int)m[4 * i] & 0xffu) | (((uint32_t)(int)m[4 * i + 1] & 0xffu) << 8) |
             (((uint32_t)(int)m[4 * i + 2] & 0xffu) << 16) | (((uint32_t)(int)m[4 * i + 3] & 0xffu) << 24);
    *(uint4*)(o.xq8 + row * o.cols_p8 + k0) = make_uint4(w[0], w[1], w[2], w[3]);
  }
  if (o.xq) {
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t lo = exact_bf16_bits(ldexpf(m[2 * i], e - q.mbits));
      const uint32_t hi = exact_bf16_bits(ldexpf(m[2 * i + 1], e - q.mbits));
      w[i] = lo | (hi << 16);
    }
    uint4* dst = (uint4*)(o.xq + row * o.cols_p + k0);
    dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
    dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
  }
  if (o.deq || o.codes) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (k0 + i < cols) {
        if (o.deq) o.deq[row * cols + k0 + i] = fabsf(v[i]) <= 1e-8f ? v[i] : ldexpf(m[i], e - q.mbits);
        if (o.codes) o.codes[row * cols + k0 + i] = (int8_t)(int)m[i];
      }
    }
  }
}

template <int DT, bool VEC>
__global__ __launch_bounds__(256) void k_quant_seg16(const void* __restrict__ x, int64_t rows, int64_t cols,
                                                     int64_t ld, QP q, QuantOut o) {
  const int64_t segs = o.xq ? o.cols_p / 16 : (cols + 15) / 16;
  const int64_t total = rows * segs;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t row = idx / segs;
    const int64_t k0 = (idx - row * segs) * 16;
    float v[16];
    if (k0 + 16 <= cols) {
      load16<DT, VEC>(x, row * ld, k0, cols, v);
    } else {
      load16<DT, false>(x, row * ld, k0, cols, v);
    }
    float amax = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) amax = fmaxf(amax, fabsf(v[i]));
    const bool any = amax > 0.0f;
    const int e = any ? block_exponent_u(amax, q) : 0;
    emit16(v, any, e, q, o, row, k0, cols);
    if (o.exps && k0 < cols) o.exps[row * o.nblk + k0 / 16] = (int8_t)(e > 127 ? 127 : e);
  }
}

// One workgroup per row, one exponent per row.  Rows of up to 256 x QR_KEEP x 16 elements are read ONCE: every
// lane keeps its 16-element segments in registers between the row-maximum reduction and the quantization (longer
// rows fall back to a second read).
constexpr int QR_KEEP = 4;  // segments per lane held in registers: rows up to 16384 elements

template <int DT>
__global__ __launch_bounds__(256) void k_quant_row(const void* __restrict__ x, int64_t rows, int64_t cols, int64_t ld,
                                                   QP q, QuantOut o, bool vec) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const int64_t segs = o.xq8 ? o.cols_p8 / 16 : (o.xq ? o.cols_p / 16 : (cols + 15) / 16);
  const bool keep = segs <= 256 * QR_KEEP;
  float v[QR_KEEP][16];
  float amax = 0.0f;
  if (keep) {
#pragma unroll
    for (int j = 0; j < QR_KEEP; ++j) {
      const int64_t sg = threadIdx.x + 256 * j;
#pragma unroll
      for (int i = 0; i < 16; ++i) v[j][i] = 0.f;
      if (sg < segs && sg * 16 < cols) {
        if (vec && sg * 16 + 16 <= cols)
          load16<DT, true>(x, row * ld, sg * 16, cols, v[j]);
        else
          load16<DT, false>(x, row * ld, sg * 16, cols, v[j]);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) amax = fmaxf(amax, fabsf(v[j][i]));
    }
  } else {
    for (int64_t k = threadIdx.x; k < cols; k += 256) amax = fmaxf(amax, fabsf(load_elem<DT>(x, row * ld + k)));
  }
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) amax = fmaxf(amax, __shfl_xor(amax, s, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const bool any = amax > 0.0f;
  const int e = any ? block_exponent_u(amax, q) : 0;
  if (keep) {
#pragma unroll
    for (int j = 0; j < QR_KEEP; ++j) {
      const int64_t sg = threadIdx.x + 256 * j;
      if (sg < segs) emit16(v[j], any, e, q, o, row, sg * 16, cols);
    }
  } else {
    for (int64_t s = threadIdx.x; s < segs; s += 256) {
      float t[16];
      load16<DT, false>(x, row * ld, s * 16, cols, t);
      emit16(t, any, e, q, o, row, s * 16, cols);
    }
  }
  if (o.exps && threadIdx.x == 0) o.exps[row * o.nblk] = (int8_t)(e > 127 ? 127 : e);
  if (o.xscale && threadIdx.x == 0) o.xscale[row] = any ? ldexpf(1.0f, e - q.mbits) : 1.0f;
}

// The int8 image of per-token activations (one exponent per row, the int8 route) from 16-bit tensors: ONE WAVE per row, the
// whole row in registers as raw 16-byte chunks - lane l holds chunks l, l + 64, ... (a wave's request is a contiguous KiB),
// MAXCH requests in flight -, the row maximum by packed 16-bit maxima and one wave reduction (no LDS, no barrier), then every
// chunk's 8 elements to 8 int8 (one 8-byte store per lane, 512 contiguous bytes per wave) with the packed signed arithmetic of
// mxint16_i8_fast.  With one exponent per row nothing needs the 16-element block structure.  Same codes and scales as
// k_quant_row (which stays for fp32 inputs, unaligned rows and rows longer than 64 x 8 x MAXCH elements).
template <int DT, int MAXCH>
__global__ __launch_bounds__(256) void k_quant_row8(const void* __restrict__ x, int64_t rows, int64_t cols, int64_t ld, QP q,
                                                    int8_t* __restrict__ xq8, int64_t cols_p8, float* __restrict__ xscale) {
  static_assert(DT != LQER_F32, "16-bit inputs");
  typedef __attribute__((ext_vector_type(2))) float f2;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = (int)(cols / 8), nch_p = (int)(cols_p8 / 8);
  const u32x4* p = (const u32x4*)((const bf16_t*)x + row * ld);
  u32x4 raw[MAXCH];
#pragma unroll
  for (int u = 0; u < MAXCH; ++u) raw[u] = lane + 64 * u < nch ? p[lane + 64 * u] : (u32x4){0, 0, 0, 0};
  float amax = row8_amax<DT, MAXCH>(raw);
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) amax = fmaxf(amax, __shfl_xor(amax, s, 64));
  const bool any = amax > 0.f;
  const int e = any ? block_exponent_u(amax, q) : 0;
  if (lane == 0 && xscale) xscale[row] = any ? ldexpf(1.0f, e - q.mbits) : 1.0f;
  int8_t* const dst = xq8 + row * cols_p8;
  const bool fast = mxint16_fast_ok(e, q);  // (wave-uniform)
  const float s = __uint_as_float((uint32_t)(127 + (fast ? q.mbits - e : 0)) << 23);
  const float es = 1e-9f * s;
  auto emit = [&](auto fast_c) {  // (two copies under ONE wave-uniform branch: as a select the slow arithmetic ran for every element)
    constexpr bool FAST = decltype(fast_c)::value;
#pragma unroll
    for (int u = 0; u < MAXCH; ++u) {
      const int c = lane + 64 * u;
      if (c >= nch_p) continue;
      *(u32x2*)(dst + (int64_t)c * 8) = row8_chunk<DT, FAST>(raw[u], c < nch && any, e, q, s, es);
    }
  };
  if (DT == LQER_F16 && any && row8_h16_ok(e, q)) {  // (wave-uniform; round 6) packed half arithmetic: same bytes, a third of the instructions
#pragma unroll
    for (int u = 0; u < MAXCH; ++u) {
      const int c = lane + 64 * u;
      if (c >= nch_p) continue;
      *(u32x2*)(dst + (int64_t)c * 8) = row8_chunk_h16(raw[u], c < nch, e, q);
    }
  } else if (fast)
    emit(std::true_type{});
  else
    emit(std::false_type{});
}

// One lane per block of L elements (L a multiple of 16), serial.
template <int DT>
__global__ __launch_bounds__(256) void k_quant_blk(const void* __restrict__ x, int64_t rows, int64_t cols, int64_t ld,
                                                   QP q, QuantOut o) {
  const int64_t L = q.block;
  const int64_t width = o.xq ? o.cols_p : cols;
  const int64_t nb = (width + L - 1) / L;
  const int64_t total = rows * nb;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t row = idx / nb;
    const int64_t b0 = (idx - row * nb) * L;
    const int64_t b1 = b0 + L < width ? b0 + L : width;
    float amax = 0.0f;
    for (int64_t k = b0; k < b1 && k < cols; ++k) amax = fmaxf(amax, fabsf(load_elem<DT>(x, row * ld + k)));
    const bool any = amax > 0.0f;
    const int e = any ? block_exponent_u(amax, q) : 0;
    for (int64_t k0 = b0; k0 < b1; k0 += 16) {
      float v[16];
      load16<DT, false>(x, row * ld, k0, cols, v);
      emit16(v, any, e, q, o, row, k0, cols);
    }
    if (o.exps && b0 < cols) o.exps[row * o.nblk + b0 / L] = (int8_t)(e > 127 ? 127 : e);
  }
}

// "integer" (fixed point, reference quantizers/integer.py:10-43): elementwise, one lane per 16 elements - no shared
// exponent, so no reduction.  Every image of the block_fp kernels: fp32 values, int8 codes (width <= 8), the bf16 image
// (|code| <= 256) - and the "exponent" output holds -frac_width, one entry per row (value = code 2^exponent).
template <int DT>
__global__ __launch_bounds__(256) void k_quant_int(const void* __restrict__ x, int64_t rows, int64_t cols, int64_t ld, QP q,
                                                   QuantOut o) {
  const int64_t segs = o.xq ? o.cols_p / 16 : (cols + 15) / 16;
  const int64_t total = rows * segs;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t row = idx / segs;
    const int64_t k0 = (idx - row * segs) * 16;
    float v[16], m[16];
    load16<DT, false>(x, row * ld, k0, cols, v);  // (elements past `cols` read as 0 -> code 0)
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = mxint_mantissa(v[i], 0, q);
    if (o.xq) {
      uint32_t w[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        w[i] = exact_bf16_bits(ldexpf(m[2 * i], -q.mbits)) | (exact_bf16_bits(ldexpf(m[2 * i + 1], -q.mbits)) << 16);
      uint4* dst = (uint4*)(o.xq + row * o.cols_p + k0);
      dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
      dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (k0 + i < cols) {
        if (o.deq) o.deq[row * cols + k0 + i] = ldexpf(m[i], -q.mbits);
        if (o.codes) o.codes[row * cols + k0 + i] = (int8_t)(int)m[i];
      }
    if (o.exps && k0 == 0) o.exps[row * o.nblk] = (int8_t)(-q.mbits < -128 ? -128 : -q.mbits);  // (one "block" per row)
  }
}

template <int DT>
static int launch_quant(const void* x, int64_t rows, int64_t cols, int64_t ld, const QP& q, const QuantOut& o,
                        hipStream_t st) {
  if (rows == 0 || cols == 0) return LQER_OK;
  const int64_t width = o.xq ? o.cols_p : cols;
  if (q.kind == LQER_Q_INT) {
    const int64_t total = rows * ((width + 15) / 16);
    const unsigned grid = (unsigned)((total + 255) / 256 < 1 << 20 ? (total + 255) / 256 : 1 << 20);
    k_quant_int<DT><<<grid, 256, 0, st>>>(x, rows, cols, ld, q, o);
    return check_launch("quantize (integer)");
  }
  const bool whole = q.block <= 0 || q.block >= cols;
  if (whole) {
    const int esz0 = DT == LQER_F32 ? 4 : 2;
    const bool vec0 = ((uintptr_t)x % 16 == 0) && ((ld * esz0) % 16 == 0);
    // the int8 image alone, from an aligned 16-bit tensor: one wave per row, the row in registers (k_quant_row8)
    bool done = false;
    if constexpr (DT != LQER_F32) {
#ifndef LQER_NO_QUANT_ROW8
      if (o.xq8 && !o.xq && !o.deq && !o.codes && !o.exps && vec0 && cols % 8 == 0 && q.mbits <= 7) {
        const unsigned grid = (unsigned)((rows + 3) / 4);
        const int64_t nch_p = o.cols_p8 / 8;
        if (nch_p <= 64 * 12) {
          k_quant_row8<DT, 12><<<grid, 256, 0, st>>>(x, rows, cols, ld, q, o.xq8, o.cols_p8, o.xscale);
          done = true;
        } else if (nch_p <= 64 * 28) {
          k_quant_row8<DT, 28><<<grid, 256, 0, st>>>(x, rows, cols, ld, q, o.xq8, o.cols_p8, o.xscale);
          done = true;
        }
      }
#endif
    }
    if (!done) k_quant_row<DT><<<dim3((unsigned)rows), 256, 0, st>>>(x, rows, cols, ld, q, o, vec0);
  } else if (q.block == 16) {
    const int64_t total = rows * ((width + 15) / 16);
    const unsigned grid = (unsigned)((total + 255) / 256 < 1 << 20 ? (total + 255) / 256 : 1 << 20);
    const int esz = DT == LQER_F32 ? 4 : 2;
    const bool vec = ((uintptr_t)x % 16 == 0) && ((ld * esz) % 16 == 0);
    if (vec)
      k_quant_seg16<DT, true><<<grid, 256, 0, st>>>(x, rows, cols, ld, q, o);
    else
      k_quant_seg16<DT, false><<<grid, 256, 0, st>>>(x, rows, cols, ld, q, o);
  } else if (q.block % 16 == 0) {
    const int64_t total = rows * ((width + q.block - 1) / q.block);
    const unsigned grid = (unsigned)((total + 255) / 256 < 1 << 20 ? (total + 255) / 256 : 1 << 20);
    k_quant_blk<DT><<<grid, 256, 0, st>>>(x, rows, cols, ld, q, o);
  } else {
    set_error("MXINT block %d: block must be 16*n or cover the whole row", q.block);
    return LQER_E_UNSUPPORTED;
  }
  return check_launch("quantize_mxint");
}

int quantize_dispatch(const void* x, int dtype, int64_t rows, int64_t cols, int64_t ld, const QP& q,
                      const QuantOut& o, hipStream_t st) {
  switch (dtype) {
    case LQER_F32: return launch_quant<LQER_F32>(x, rows, cols, ld, q, o, st);
    case LQER_F16: return launch_quant<LQER_F16>(x, rows, cols, ld, q, o, st);
    case LQER_BF16: return launch_quant<LQER_BF16>(x, rows, cols, ld, q, o, st);
  }
  set_error("unknown dtype %d", dtype);
  return LQER_E_INVALID;
}

// ---- 2-D tiles of an activation (blocks that span token rows) --------------------------------------------------------------------
// Reference quantizers/utils.py:211-237 (`_block_3d_activation`: [batch, tokens, features] with skip_first_dim = true, tiles of
// R x L over (tokens, features) of every batch element) and :161-183 (`_block_2d_weight`, which utils.py:261-270 also applies to a
// 2-D activation when skip_first_dim = false).  Zero padding of ragged edges cannot raise a maximum, so the tiles are simply
// clipped.  A cold path (no template configuration has such blocks): two passes, no staging -
//   k_tile_amax : one wave per (batch, row, column tile) segment, lanes stride the segment (coalesced), one atomicMax of the
//                 fp32 bit pattern per wave into amax[batch][row / R][column tile] (max is order-independent: reproducible);
//   k_tile_quant: one thread per element, the element routine of every other quantizer (block_exponent / mxint_mantissa).
template <int DT>
__global__ __launch_bounds__(256) void k_tile_amax(const void* __restrict__ x, int64_t batches, int64_t rows, int64_t cols, int64_t R,
                                                   int64_t L, int64_t tr, int64_t tc, unsigned int* __restrict__ amax) {
  const int lane = threadIdx.x & 63;
  const int64_t segs = batches * rows * tc;
  for (int64_t sgi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); sgi < segs; sgi += (int64_t)gridDim.x * 4) {
    const int64_t c = sgi % tc, br = sgi / tc, r = br % rows, b = br / rows;
    const int64_t k0 = c * L, k1 = k0 + L < cols ? k0 + L : cols;
    float m = 0.0f;
    for (int64_t k = k0 + lane; k < k1; k += 64) m = fmaxf(m, fabsf(load_elem<DT>(x, (b * rows + r) * cols + k)));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0 && m > 0.0f) atomicMax(amax + (b * tr + r / R) * tc + c, __float_as_uint(m));
  }
}

template <int DT>
__global__ __launch_bounds__(256) void k_tile_quant(const void* __restrict__ x, int64_t batches, int64_t rows, int64_t cols, int64_t R,
                                                    int64_t L, int64_t tr, int64_t tc, const float* __restrict__ amax, QP q,
                                                    float* __restrict__ out) {
  const int64_t total = batches * rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t k = i % cols, br = i / cols, r = br % rows, b = br / rows;
    const float a = amax[(b * tr + r / R) * tc + k / L];
    const float v = load_elem<DT>(x, i);
    const bool any = a > 0.0f;
    const int e = any ? block_exponent(a, q) : 0;
    const float m = any ? mxint_mantissa(v, e, q) : 0.0f;
    out[i] = fabsf(v) <= 1e-8f ? v : ldexpf(m, e - q.mbits);
  }
}

__global__ void k_zero_u32(uint32_t* p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0u;
}

template <int DT>
static int launch_tiles(const void* x, int64_t batches, int64_t rows, int64_t cols, int64_t R, int64_t L, const QP& q, float* out,
                        float* amax, hipStream_t st) {
  const int64_t tr = (rows + R - 1) / R, tc = (cols + L - 1) / L;
  // (a kernel, not hipMemsetAsync: as a memset node of a captured graph the fill leaves cells untouched on replay - gemm_w4a8.hip)
  {
    const int64_t nz = batches * tr * tc;
    k_zero_u32<<<(unsigned)((nz + 255) / 256), 256, 0, st>>>((uint32_t*)amax, nz);
    const int rc = check_launch("quantize_mxint_tiles (zero fill of the tile maxima)");
    if (rc) return rc;
  }
  const int64_t segs = batches * rows * tc, total = batches * rows * cols;
  const unsigned g1 = (unsigned)((segs + 3) / 4 < 65536 ? (segs + 3) / 4 : 65536);
  const unsigned g2 = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  k_tile_amax<DT><<<g1, 256, 0, st>>>(x, batches, rows, cols, R, L, tr, tc, (unsigned int*)amax);
  k_tile_quant<DT><<<g2, 256, 0, st>>>(x, batches, rows, cols, R, L, tr, tc, amax, q, out);
  return check_launch("quantize_mxint_tiles");
}

int quantize_tiles_dispatch(const void* x, int dtype, int64_t batches, int64_t rows, int64_t cols, int64_t R, int64_t L, const QP& q,
                            float* out, float* amax, hipStream_t st) {
  switch (dtype) {
    case LQER_F32: return launch_tiles<LQER_F32>(x, batches, rows, cols, R, L, q, out, amax, st);
    case LQER_F16: return launch_tiles<LQER_F16>(x, batches, rows, cols, R, L, q, out, amax, st);
    case LQER_BF16: return launch_tiles<LQER_BF16>(x, batches, rows, cols, R, L, q, out, amax, st);
  }
  set_error("unknown dtype %d", dtype);
  return LQER_E_INVALID;
}

}  // namespace lqer
