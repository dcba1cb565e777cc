// One-time operand packing: replaces the reference's lazy in-place weight/bias quantization on the
// first forward (quantized_layers/linear.py:149-153 -> quantizers/block_fp.py:111, blocking by
// quantizers/utils.py:161-208) with a packed 4-bit image the GEMM kernel streams.
//
// Packed W layout (DESIGN.md "Data layout"): panels of 16 rows x 64 k, panel (pn, pk) at byte
// ((pn * Kp/64) + pk) * 576:   [16 rows][32 B]  4-bit sign-magnitude mantissas (bit 3 = sign)
//                              [16 rows][4]     int8 exponent of each 16-k segment
// Within each 32-bit word (8 consecutive k) nibble p holds k = p/2 for even p and 4 + p/2 for odd p,
// so that the GEMM kernel's byte-wise expand (even nibbles, then odd nibbles) emits k in order.
// A row's 8 words (64 k) are stored as words {0,2,4,6} then {1,3,5,7}: the MFMA lane that needs the
// words of chunks 2 ks + h (ks = 0..3) reads them with one 16-byte LDS access.  Exponents are stored
// biased, ready to be shifted into an fp32 exponent field: byte = clamp(e - mbits + 127, 1, 254).
// A coarser weight block (32, 128, whole row) repeats its exponent per segment, so the GEMM kernel
// handles every block length with one code path.
#include "common.h"

namespace lqer {

// pass 1: exponent of every (row, block) -> scratch[N][nblk]
template <int DT>
__global__ __launch_bounds__(256) void k_w_exps(const void* __restrict__ W, int64_t N, int64_t K, int64_t ld, QP q,
                                                int64_t L, int64_t nblk, int8_t* __restrict__ scratch) {
  const int64_t total = N * nblk;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t row = idx / nblk, b = idx - row * nblk;
    const int64_t k1 = (b + 1) * L < K ? (b + 1) * L : K;
    float amax = 0.0f;
    for (int64_t k = b * L; k < k1; ++k) amax = fmaxf(amax, fabsf(load_elem<DT>(W, row * ld + k)));
    int e = amax > 0.0f ? block_exponent(amax, q) : -128;  // -128 marks an all-zero block
    scratch[idx] = (int8_t)(e > 127 ? 127 : e);
  }
}

// pass 1b, 2-D tiles (block_size [R, L], reference quantizers/utils.py:161-183): the exponent of a tile is that of its largest
// element = the largest of its rows' exponents (ceil(log2 .) is monotone; -128 marks all-zero row blocks and loses every max)
__global__ __launch_bounds__(256) void k_w_exps_rows(int8_t* __restrict__ scratch, int64_t N, int64_t nblk, int64_t R) {
  const int64_t groups = (N + R - 1) / R, total = groups * nblk;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t gr = idx / nblk, b = idx - gr * nblk;
    const int64_t r1 = (gr + 1) * R < N ? (gr + 1) * R : N;
    int e = -128;
    for (int64_t row = gr * R; row < r1; ++row) e = scratch[row * nblk + b] > e ? scratch[row * nblk + b] : e;
    for (int64_t row = gr * R; row < r1; ++row) scratch[row * nblk + b] = (int8_t)e;
  }
}

// Weights of 5..8 bits (the reference's W8A8 baseline: sweep_baseline_no_lqer.sh:73-76, block_fp width 8): the mantissa m,
// |m| <= 127, is written in signed base-8 digits, m = 64 a + 8 b + c with a in [-2, 2], b, c in [-4, 3] - three 4-bit sign-magnitude
// LIMBS with block exponents e - mbits + 6, + 3, + 0.  The image holds them side by side along k (panel (pn, l Kp/64 + pk): limb l),
// the activation image is repeated three times to match (include/lqer_hip.h "weights of 5..8 bits"): every 4-bit kernel multiplies
// such a weight exactly, at three times the work - the universal route; the int8 MFMA kernel has an image of its own.
__device__ __forceinline__ void w8_digits(int m, int (&d)[3]) {
  const int c = ((m + 4) & 7) - 4;
  const int m1 = (m - c) >> 3;  // (exact: m - c is a multiple of 8)
  const int b = ((m1 + 4) & 7) - 4;
  d[0] = (m1 - b) >> 3, d[1] = b, d[2] = c;
}

// pass 2 for those: one lane per (padded row, 16-k segment), three panels
template <int DT>
__global__ __launch_bounds__(256) void k_w_pack8(const void* __restrict__ W, int64_t N, int64_t K, int64_t ld, QP q,
                                                 int64_t L, int64_t nblk, const int8_t* __restrict__ scratch,
                                                 int64_t Np, int64_t Kp, uint8_t* __restrict__ out) {
  const int64_t segs = Kp / 16, npk = Kp / 64;
  const int64_t total = Np * segs;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t row = idx / segs, seg = idx - row * segs, k0 = seg * 16;
    uint32_t lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    int e = 0;
    if (row < N && k0 < K) {
      const int es = scratch[row * nblk + k0 / L];
      // (a block whose lowest limb's exponent byte e - mbits + 127 would leave [1, 254] keeps all-zero digits: the three limbs' clamped
      // bytes would no longer stand 2^3 apart.  Nothing is lost: such a block has e <= mbits - 127, every |w| <= 2^e < 1e-8 - the
      // range packed images flush to 0 anyway (mxint_mantissa's `tiny`); the upper end cannot clamp, e <= 128)
      if (es != -128 && es - q.mbits + 127 >= 1) {
        e = es;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float w = (k0 + i < K) ? load_elem<DT>(W, row * ld + k0 + i) : 0.0f;
          int d[3];
          w8_digits((int)mxint_mantissa(w, e, q), d);
          const int j = i & 7, pos = j < 4 ? 2 * j : 2 * (j - 4) + 1;
#pragma unroll
          for (int l = 0; l < 3; ++l) {
            const uint32_t c = (uint32_t)(d[l] < 0 ? (8 - d[l]) : d[l]);  // sign-magnitude, +0 canonical
            if (i < 8) lo[l] |= c << (4 * pos);
            else hi[l] |= c << (4 * pos);
          }
        }
      }
    }
#pragma unroll
    for (int l = 0; l < 3; ++l) {
      uint8_t* panel = out + ((row / 16) * (3 * npk) + l * npk + seg / 4) * LQER_PANEL_BYTES;
      *(uint32_t*)(panel + (row % 16) * 32 + (seg % 4) * 4) = lo[l];
      *(uint32_t*)(panel + (row % 16) * 32 + 16 + (seg % 4) * 4) = hi[l];
      int eb = e - q.mbits + 3 * (2 - l) + 127;
      eb = eb < 1 ? 1 : (eb > 254 ? 254 : eb);
      panel[512 + (row % 16) * 4 + (seg % 4)] = (uint8_t)eb;
    }
  }
}

// pass 2: one lane per (padded row, 16-k segment)
template <int DT>
__global__ __launch_bounds__(256) void k_w_pack(const void* __restrict__ W, int64_t N, int64_t K, int64_t ld, QP q,
                                                int64_t L, int64_t nblk, const int8_t* __restrict__ scratch,
                                                int64_t Np, int64_t Kp, uint8_t* __restrict__ out) {
  const int64_t segs = Kp / 16;
  const int64_t total = Np * segs;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t row = idx / segs, seg = idx - row * segs, k0 = seg * 16;
    uint32_t lo = 0, hi = 0;
    int e = 0;
    if (row < N && k0 < K) {
      const int es = scratch[row * nblk + k0 / L];
      if (es != -128) {
        e = es;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float w = (k0 + i < K) ? load_elem<DT>(W, row * ld + k0 + i) : 0.0f;
          const int mi = (int)mxint_mantissa(w, e, q);
          // sign-magnitude, +0 canonical; the `integer` quantizer's codes -2^(w-1) .. 2^(w-1)-1 (quantizers/integer.py:37-40) do
          // not fit that - they travel as two's-complement nibbles (GemmArgs.w_twos: the tile kernel's second expand)
          const uint32_t c = q.kind == LQER_Q_INT ? ((uint32_t)mi & 0xfu) : (uint32_t)(mi < 0 ? (8 - mi) : mi);
          const int j = i & 7, pos = j < 4 ? 2 * j : 2 * (j - 4) + 1;
          if (i < 8)
            lo |= c << (4 * pos);
          else
            hi |= c << (4 * pos);
        }
      }
    }
    uint8_t* panel = out + ((row / 16) * (Kp / 64) + seg / 4) * LQER_PANEL_BYTES;
    *(uint32_t*)(panel + (row % 16) * 32 + (seg % 4) * 4) = lo;        // chunk 2 * (seg % 4)
    *(uint32_t*)(panel + (row % 16) * 32 + 16 + (seg % 4) * 4) = hi;   // chunk 2 * (seg % 4) + 1
    int eb = e - q.mbits + 127;
    eb = eb < 1 ? 1 : (eb > 254 ? 254 : eb);
    panel[512 + (row % 16) * 4 + (seg % 4)] = (uint8_t)eb;
  }
}

// (limbs = 3: the image of a 5..8-bit weight - the value is the sum of its three limbs, each digit x 2^(exponent byte - 127))
__global__ __launch_bounds__(256) void k_w_unpack(const uint8_t* __restrict__ in, int64_t N, int64_t K, int64_t Kp,
                                                  int mbits, bool twos, int limbs, float* __restrict__ out) {
  const int64_t segs = Kp / 16, npk = Kp / 64;
  const int64_t total = N * segs;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t row = idx / segs, seg = idx - row * segs, k0 = seg * 16;
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int l = 0; l < limbs; ++l) {
      const uint8_t* panel = in + ((row / 16) * (limbs * npk) + l * npk + seg / 4) * LQER_PANEL_BYTES;
      uint2 c;
      c.x = *(const uint32_t*)(panel + (row % 16) * 32 + (seg % 4) * 4);
      c.y = *(const uint32_t*)(panel + (row % 16) * 32 + 16 + (seg % 4) * 4);
      const int e = (int)panel[512 + (row % 16) * 4 + (seg % 4)] - 127 + mbits;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const uint32_t word = i < 8 ? c.x : c.y;
        const int j = i & 7, pos = j < 4 ? 2 * j : 2 * (j - 4) + 1;
        const uint32_t nib = (word >> (4 * pos)) & 0xfu;
        const int v = twos ? ((int)nib >= 8 ? (int)nib - 16 : (int)nib) : ((nib & 8u) ? -(int)(nib & 7u) : (int)nib);
        acc[i] += ldexpf((float)v, e - mbits);  // (limbs of one value: digits x powers of two - the partial sums are exact)
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (k0 + i < K) out[row * K + k0 + i] = acc[i];
  }
}

// A [K,r] -> a_t [3][rp][Kp] ; B [r,N] -> b_t [3][Np][rp]  (bf16 limbs, transposed, zero padded).
template <int DT>
__global__ __launch_bounds__(256) void k_pack_lowrank(const void* __restrict__ src, int64_t rows, int64_t cols,
                                                      int64_t rows_p, int64_t cols_p, bf16_t* __restrict__ dst,
                                                      int32_t* __restrict__ nlimbs) {
  // dst[l][c][r] = limb l of src[r][c]   (dst is [3][cols_p][rows_p])
  const int64_t total = rows_p * cols_p;
  int used = 0;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t c = idx / rows_p, r = idx - c * rows_p;
    float v = (r < rows && c < cols) ? load_elem<DT>(src, r * cols + c) : 0.0f;
#pragma unroll
    for (int l = 0; l < 3; ++l) {
      const bf16_t b = f32_to_bf16_rne(v);
      dst[(l * cols_p + c) * rows_p + r] = b;
      if (b & 0x7fff) used = l + 1;
      v -= __uint_as_float((uint32_t)b << 16);  // exact: the residual has fewer significant bits
    }
  }
  if (used) atomicMax(nlimbs, used);
}

template <int DT>
__global__ void k_bias_passthrough(const void* __restrict__ b, int64_t N, int64_t Np, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < Np) out[i] = i < N ? load_elem<DT>(b, i) : 0.0f;
}

template <int DT>
static int pack_w(const void* W, int64_t N, int64_t K, int64_t ld, const QP& q, int64_t block_rows, uint8_t* out, int8_t* scratch,
                  hipStream_t st) {
  const int64_t Np = lqer_padded_n(N), Kp = lqer_padded_k(K);
  const int64_t L = (q.block <= 0 || q.block >= K) ? Kp : q.block;
  const int64_t nblk = (K + L - 1) / L;
  {
    const int64_t total = N * nblk;
    const unsigned grid = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    k_w_exps<DT><<<grid, 256, 0, st>>>(W, N, K, ld, q, L, nblk, scratch);
  }
  if (block_rows != 1) {  // tiles of R rows (R <= 0 or >= N: all rows)
    const int64_t R = (block_rows <= 0 || block_rows >= N) ? N : block_rows;
    const int64_t total = ((N + R - 1) / R) * nblk;
    const unsigned grid = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    k_w_exps_rows<<<grid, 256, 0, st>>>(scratch, N, nblk, R);
  }
  {
    const int64_t total = Np * (Kp / 16);
    const unsigned grid = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    if (q.width > 4) k_w_pack8<DT><<<grid, 256, 0, st>>>(W, N, K, ld, q, L, nblk, scratch, Np, Kp, out);
    else k_w_pack<DT><<<grid, 256, 0, st>>>(W, N, K, ld, q, L, nblk, scratch, Np, Kp, out);
  }
  return check_launch("pack_weight");
}

int pack_weight_dispatch(const void* W, int dtype, int64_t N, int64_t K, int64_t ld, const QP& q, int64_t block_rows, void* out,
                         void* scratch, hipStream_t st) {
  if (q.width < 2 || q.width > 8 || (q.kind == LQER_Q_INT && q.width > 4)) {
    set_error("packed weights hold 4-bit codes (block_fp widths 5..8: three 4-bit limbs): w_quantizer width must be 2..8 (integer: 2..4), got %d", q.width);
    return LQER_E_UNSUPPORTED;
  }
  if (!(q.block <= 0 || q.block >= K || q.block % 16 == 0)) {
    set_error("w_quantizer block %d: must be a multiple of 16 or cover the row", q.block);
    return LQER_E_UNSUPPORTED;
  }
  switch (dtype) {
    case LQER_F32: return pack_w<LQER_F32>(W, N, K, ld, q, block_rows, (uint8_t*)out, (int8_t*)scratch, st);
    case LQER_F16: return pack_w<LQER_F16>(W, N, K, ld, q, block_rows, (uint8_t*)out, (int8_t*)scratch, st);
    case LQER_BF16: return pack_w<LQER_BF16>(W, N, K, ld, q, block_rows, (uint8_t*)out, (int8_t*)scratch, st);
  }
  set_error("unknown dtype %d", dtype);
  return LQER_E_INVALID;
}

int unpack_weight_dispatch(const void* in, int64_t N, int64_t K, int mbits, bool twos, float* out, hipStream_t st) {
  const int64_t Kp = lqer_padded_k(K);
  const int64_t total = N * (Kp / 16);
  if (total == 0) return LQER_OK;
  const unsigned grid = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  k_w_unpack<<<grid, 256, 0, st>>>((const uint8_t*)in, N, K, Kp, mbits, twos, mbits > 3 ? 3 : 1, out);
  return check_launch("unpack_weight");
}

template <int DT>
static int pack_lr(const void* A, const void* B, int64_t K, int64_t N, int64_t r, bf16_t* a_t, bf16_t* b_t,
                   int32_t* flags, hipStream_t st) {
  const int64_t Kp = lqer_padded_k(K), Np = lqer_padded_n(N), rp = lqer_padded_r(r);
  (void)hipMemsetAsync(flags, 0, 2 * sizeof(int32_t), st);
  {  // A [K, r] -> a_t [3][rp][Kp]
    const int64_t total = Kp * rp;
    const unsigned grid = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    k_pack_lowrank<DT><<<grid, 256, 0, st>>>(A, K, r, Kp, rp, a_t, flags);
  }
  {  // B [r, N] -> b_t [3][Np][rp]
    const int64_t total = rp * Np;
    const unsigned grid = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    k_pack_lowrank<DT><<<grid, 256, 0, st>>>(B, r, N, rp, Np, b_t, flags + 1);
  }
  return check_launch("pack_lowrank");
}

int pack_lowrank_dispatch(const void* A, const void* B, int dtype, int64_t K, int64_t N, int64_t r, void* a_t,
                          void* b_t, int32_t* flags, hipStream_t st) {
  switch (dtype) {
    case LQER_F32: return pack_lr<LQER_F32>(A, B, K, N, r, (bf16_t*)a_t, (bf16_t*)b_t, flags, st);
    case LQER_F16: return pack_lr<LQER_F16>(A, B, K, N, r, (bf16_t*)a_t, (bf16_t*)b_t, flags, st);
    case LQER_BF16: return pack_lr<LQER_BF16>(A, B, K, N, r, (bf16_t*)a_t, (bf16_t*)b_t, flags, st);
  }
  set_error("unknown dtype %d", dtype);
  return LQER_E_INVALID;
}

// ---- fp16 fast path of pass-through fp16 activations (LQER_Q_PASSTHROUGH_F16) -----------------------------------
// The main loops then expand the weights to fp16 (code * 2^(eb - 127), eb the stored exponent byte) and the side GEMM
// reads A as ONE fp16 image.  Both must be exact: flags[0] is raised when a weight block's scale leaves the fp16
// range (subnormals included: 2^-24 .. 2^13 keeps 7 * scale finite), flags[1] when an element of A is not an fp16
// number.  The caller falls back to bf16 limbs (act_limbs.hip) if either is set.
__global__ __launch_bounds__(256) void k_w_f16_range(const uint8_t* __restrict__ wp, int64_t panels, int32_t* __restrict__ flags) {
  bool bad = false;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < panels * 16; idx += (int64_t)gridDim.x * 256) {
    const uint32_t e4 = *(const uint32_t*)(wp + (idx >> 4) * LQER_PANEL_BYTES + 512 + (idx & 15) * 4);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int eb = (int)((e4 >> (8 * b)) & 0xffu);
      bad |= eb < 127 - 24 || eb > 127 + 13;
    }
  }
  if (bad) atomicOr(flags, 1);
}

__global__ __launch_bounds__(256) void k_a_f16(const bf16_t* __restrict__ limbs, int a_limbs, int64_t total, _Float16* __restrict__ out,
                                               int32_t* __restrict__ flags) {
  bool bad = false;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    float v = 0.f;
    for (int l = a_limbs - 1; l >= 0; --l) v += __uint_as_float((uint32_t)limbs[l * total + idx] << 16);  // exact: limbs of one fp32
    const _Float16 h = (_Float16)v;
    bad |= (float)h != v;
    out[idx] = h;
  }
  if (bad) atomicOr(flags + 1, 1);
}

int f16_prepare_dispatch(const void* w_packed, int64_t N, int64_t K, const void* a_limbs_img, int a_limbs, int64_t r, void* a_f16,
                         int32_t* flags, hipStream_t st) {
  const int64_t Kp = lqer_padded_k(K), Np = lqer_padded_n(N), rp = lqer_padded_r(r);
  (void)hipMemsetAsync(flags, 0, 2 * sizeof(int32_t), st);
  const int64_t panels = (Np / 16) * (Kp / 64);
  k_w_f16_range<<<(unsigned)((panels * 16 + 255) / 256 < 4096 ? (panels * 16 + 255) / 256 : 4096), 256, 0, st>>>((const uint8_t*)w_packed, panels, flags);
  if (r > 0) {
    const int64_t total = rp * Kp;
    k_a_f16<<<(unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096), 256, 0, st>>>((const bf16_t*)a_limbs_img, a_limbs, total,
                                                                                                   (_Float16*)a_f16, flags);
  }
  return check_launch("f16_prepare");
}

int bias_passthrough_dispatch(const void* b, int dtype, int64_t N, float* out, hipStream_t st) {
  const int64_t Np = lqer_padded_n(N);
  const unsigned grid = (unsigned)((Np + 255) / 256);
  switch (dtype) {
    case LQER_F32: k_bias_passthrough<LQER_F32><<<grid, 256, 0, st>>>(b, N, Np, out); break;
    case LQER_F16: k_bias_passthrough<LQER_F16><<<grid, 256, 0, st>>>(b, N, Np, out); break;
    case LQER_BF16: k_bias_passthrough<LQER_BF16><<<grid, 256, 0, st>>>(b, N, Np, out); break;
    default: set_error("unknown dtype %d", dtype); return LQER_E_INVALID;
  }
  return check_launch("pack_bias");
}

}  // namespace lqer
