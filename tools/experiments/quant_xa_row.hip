// EXPERIMENT (not compiled into liblqer_hip.so): per-token activations (block_size [1, -1], the int8 route, C4) - a read-only
// row-maximum pass (k_row_scale) followed by ONE pass that quantizes with the rows' exponents, writes the int8 image and
// multiplies the mantissas with A on the MFMA (k_quant_xa_row: 32 rows x several 256-k slabs per workgroup, next slab's
// activation blocks prefetched as raw bytes, both bf16 limbs of A^T preloaded, few K chunks so that the partial tiles stay
// small), instead of k_quant_row + k_xa_partial.  Parity-green on tests/test_gpu_int8.py and the in-run parity check of
// bench.py --workload c4.  Result on MI355X (16384 x 5120 / 13824 activations, rank 64, two limbs of A, averages of the C4
// sweep under rocprofv3): k_row_scale 48.6 us + k_quant_xa_row 190 us (177 without the prefetch, 187 with one partial tile per
// 256 k = 84-226 MB of partial tiles) against k_quant_row 74-81 us + k_xa_partial 65 us: C4 1720 -> 1580 TFLOP/s-equiv.
// Why: the side GEMM at this size is bound by the A^T stream from L2, not by the activation bytes - every 32-row group reads
// all of A^T (2 limbs x 64 x K x 2 B = 1.3-3.5 MB; 512 groups = 0.67-1.8 GB per call), which is what k_xa_partial's 65 us
// are as well; fusing the quantizer in front of it serialises the HBM stream behind that L2 stream inside every workgroup.
// What would pay: more token rows per A^T fragment (64-128 rows per workgroup) or one fp16 limb on v_mfma_*_f16.
// The pieces below were wired in through lqer_quantize_act_xa (x_is_i8 && rank > 0 -> quant_xa_fused_row_dispatch).

// ---- common.h helper ----
// The int8 image AND a bf16 copy of the same 16 signed mantissas (exact: |m| <= 127 has 7 significant bits) - the fused
// per-token quantizer feeds the copy to the side GEMM's MFMA and stores the int8 words.  Needs mxint16_fast_ok(e, q).
template <bool FLUSH_TINY>
__device__ __forceinline__ void mxint16_i8_bf16_fast(const float (&v)[16], int e, const QP& q, uint32_t (&w)[4], uint32_t (&wb)[8]) {
  typedef __attribute__((ext_vector_type(2))) float f2;
  const float s = __uint_as_float((uint32_t)(127 + q.mbits - e) << 23);
  const f2 magic = {12582912.0f, 12582912.0f}, eps = {1e-9f, 1e-9f};
  uint32_t h[8];  // pairs of int16
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const f2 x = {v[2 * i], v[2 * i + 1]};
    const f2 a = {fabsf(x[0]), fabsf(x[1])};
    const f2 t = a + eps;
    f2 r = __builtin_elementwise_fma(t, (f2){s, s}, magic) - magic;
    r[0] = copysignf(fminf(r[0], q.mmax), x[0]);
    r[1] = copysignf(fminf(r[1], q.mmax), x[1]);
    if constexpr (FLUSH_TINY) {
      r[0] = a[0] <= 1e-8f ? 0.0f : r[0];
      r[1] = a[1] <= 1e-8f ? 0.0f : r[1];
    }
    h[i] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16((int)r[0], (int)r[1]));
    // (-0.0f, from a negative x that rounds to 0, would be the bf16 pattern 0x8000: harmless in a product, but keep +0)
    wb[i] = ((__float_as_uint(r[0] + 0.0f)) >> 16) | (__float_as_uint(r[1] + 0.0f) & 0xffff0000u);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = __builtin_amdgcn_perm(h[2 * i + 1], h[2 * i], 0x06040200u);
}


// ---- lowrank_xa.hip ----
// ---- per-token activations (block_size [1, -1], the int8 route): row scales first, then quantize + side GEMM ----------
// A row's exponent needs the whole row, so the fused pass is preceded by a read-only pass: k_row_scale, one wave per token
// row, 16-byte loads, writes 2^(e - mbits) per row (what k_quant_row writes).  k_quant_xa_row is k_quant_xa16 with the
// exponent taken from that table: 32 rows x QX_K k per workgroup, the int8 mantissas go to HBM (the image the int8 GEMM
// reads), a bf16 copy of them to the LDS slab, the split-K partial tiles of (mantissas) x A to the scratch; the reduce pass
// applies the row scale last (exact).  Against k_quant_row + k_xa_partial this reads x twice (once only for the maximum)
// but never reads the int8 image back and converts nothing: 146 -> ~85 us per 16384 x 5120 activation.
template <int DT>
__global__ __launch_bounds__(256) void k_row_scale(const void* __restrict__ x, int64_t M, int64_t K, int64_t ldx, bool vec, QP q,
                                                   float* __restrict__ xscale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float amax = 0.f;
  constexpr int EPC = DT == LQER_F32 ? 4 : 8;  // elements per 16-byte chunk
  if (vec && K % EPC == 0) {
    // the maximum needs no block structure: lane l takes the 16-byte chunks l, l + 64, ... of the row (a wave's request is
    // one contiguous KiB), eight requests in flight
    const u32x4* p = (const u32x4*)((const char*)x + row * ldx * (DT == LQER_F32 ? 4 : 2));
    const int64_t nch = K / EPC;
    for (int64_t cb = lane; cb < nch; cb += 8 * 64) {
      u32x4 c[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) c[u] = cb + u * 64 < nch ? p[cb + u * 64] : (u32x4){0, 0, 0, 0};
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t w = c[u][j];
          if constexpr (DT == LQER_F32) {
            amax = fmaxf(amax, fabsf(__uint_as_float(w)));
          } else if constexpr (DT == LQER_F16) {
            typedef __attribute__((ext_vector_type(2))) _Float16 h2;
            const h2 hv = __builtin_bit_cast(h2, w & 0x7fff7fffu);
            amax = fmaxf(amax, fmaxf((float)hv[0], (float)hv[1]));
          } else {
            amax = fmaxf(amax, fmaxf(__uint_as_float((w << 16) & 0x7fffffffu), __uint_as_float(w & 0x7fff0000u)));
          }
        }
    }
  } else {
    for (int64_t k0 = (int64_t)lane * 16; k0 < K; k0 += 64 * 16) {
      float v[16];
      qx_load16<DT>(x, row * ldx, k0, K, vec, v);
#pragma unroll
      for (int i = 0; i < 16; ++i) amax = fmaxf(amax, fabsf(v[i]));
    }
  }
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) amax = fmaxf(amax, __shfl_xor(amax, s, 64));
  if (lane == 0) xscale[row] = amax > 0.f ? ldexpf(1.0f, block_exponent(amax, q) - q.mbits) : 1.0f;
}

// PF (aligned input, K a multiple of 16): the next slab's activation blocks are requested, as raw bytes, before this slab's
// barrier and MFMAs - the HBM stream does not stop while a workgroup computes.
template <int DT, int NT, bool PF>
__global__ __launch_bounds__(QX_K) void k_quant_xa_row(const void* __restrict__ x, int64_t M, int64_t K, int64_t ldx, bool vec,
                                                      QP q, int8_t* __restrict__ xq8, int64_t Kp8, int64_t Kp,
                                                      const float* __restrict__ xscale, const bf16_t* __restrict__ a_t,
                                                      int a_limbs, int rp, XaPlan plan, float* __restrict__ part) {
  // One workgroup = 32 token rows x plan.kc k (a multiple of QX_K), walked in slabs of QX_K: the side product accumulates in
  // the MFMA over the slabs, ONE partial tile per workgroup at the end (at M = 16384 two chunks fill the chip: the partial
  // tiles are 2 x M x rp floats, not K / 256 of them).
  __shared__ __attribute__((aligned(16))) unsigned char slab[32 * QX_K * 2];
  __shared__ __attribute__((aligned(16))) float red[(QX_WAVES > 1 ? QX_WAVES - 1 : 1) * 32 * 32 * NT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rg = blockIdx.x / plan.nchunk, c = blockIdx.x - rg * plan.nchunk;
  const int r = lane & 31, h = lane >> 5;
  const int64_t k_begin = (int64_t)c * plan.kc;
  const int64_t k_end = k_begin + plan.kc < Kp8 ? k_begin + plan.kc : (Kp8 > Kp ? Kp8 : Kp);  // (the last chunk also zero-fills up to Kp8)
  // this thread's two blocks of a slab: rows s / (QX_K / 16), segments s % (QX_K / 16)
  int brow[2], bseg[2];
  int eb[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    const int s = tid + s2 * QX_K;
    brow[s2] = s / (QX_K / 16), bseg[s2] = s % (QX_K / 16);
    const int64_t m = (int64_t)rg * 32 + brow[s2];
    eb[s2] = m < M ? ilogbf(xscale[m]) + q.mbits : 0;  // the row's exponent (the table holds 2^(e - mbits), exact also when subnormal)
  }
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
  constexpr int RW = DT == LQER_F32 ? 4 : 2;  // 16-byte words of a raw block
  u32x4 cur[2][RW], nxt[2][RW];
  auto request = [&](int64_t kb, u32x4 (&rw)[2][RW]) {  // (PF) blocks past K or past the chunk: zeros, no memory access
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int64_t m = (int64_t)rg * 32 + brow[s2], k0 = kb + bseg[s2] * 16;
      const bool live = m < M && k0 < K && kb < k_end;
      const u32x4* p = (const u32x4*)((const char*)x + (live ? m * ldx + k0 : 0) * (DT == LQER_F32 ? 4 : 2));
#pragma unroll
      for (int i = 0; i < RW; ++i) {
        const u32x4 t = p[i];  // (always in range: offset 0 for the dead ones)
        rw[s2][i] = live ? t : (u32x4){0, 0, 0, 0};
      }
    }
  };
  if constexpr (PF) request(k_begin, cur);
  for (int64_t kbase = k_begin; kbase < k_end; kbase += QX_K) {
    const int64_t kw = kbase + 64 * wave;
    // this wave's A^T fragments (limbs 0 and 1) first: their L2 latency overlaps the activation loads
    bf16x8 af0[4][NT], af1[4][NT];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int n = t * 32 + r;
        af0[ks][t] = af1[ks][t] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (n < rp && kw < Kp && a_limbs > 0) af0[ks][t] = *(const bf16x8*)(a_t + (int64_t)n * Kp + kw + 16 * ks + 8 * h);
        if (n < rp && kw < Kp && a_limbs > 1) af1[ks][t] = *(const bf16x8*)(a_t + ((int64_t)rp + n) * Kp + kw + 16 * ks + 8 * h);
      }
    if constexpr (PF) request(kbase + QX_K, nxt);
    // ---- quantize the slab with the rows' exponents
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int row = brow[s2], seg = bseg[s2];
      const int64_t m = (int64_t)rg * 32 + row, k0 = kbase + seg * 16;
      uint32_t w[4] = {0, 0, 0, 0}, wb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (m < M && k0 < K) {
        float v[16];
        if constexpr (PF) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            if constexpr (DT == LQER_F32) {
              v[i] = __uint_as_float(cur[s2][i >> 2][i & 3]);
            } else {
              const uint32_t wd = cur[s2][i >> 3][(i >> 1) & 3];
              if constexpr (DT == LQER_F16) {
                typedef __attribute__((ext_vector_type(2))) _Float16 h2;
                v[i] = (float)__builtin_bit_cast(h2, wd)[i & 1];
              } else {
                v[i] = (i & 1) ? __uint_as_float(wd & 0xffff0000u) : __uint_as_float(wd << 16);
              }
            }
          }
        } else {
          qx_load16<DT>(x, m * ldx, k0, K, vec, v);
        }
        const int e = eb[s2];
        if (mxint16_fast_ok(e, q)) {
          mxint16_i8_bf16_fast<DT != LQER_F16>(v, e, q, w, wb);
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float mv = mxint_mantissa(v[i], e, q) + 0.0f;
            w[i >> 2] |= ((uint32_t)(int)mv & 0xffu) << (8 * (i & 3));
            wb[i >> 1] |= (__float_as_uint(mv) >> 16) << (16 * (i & 1));
          }
        }
      }
      if (k0 < Kp8) *(uint4*)(xq8 + m * Kp8 + k0) = make_uint4(w[0], w[1], w[2], w[3]);  // rows up to the padded M are allocated
      *(uint4*)(slab + qx_swz(row, 2 * seg)) = make_uint4(wb[0], wb[1], wb[2], wb[3]);
      *(uint4*)(slab + qx_swz(row, 2 * seg + 1)) = make_uint4(wb[4], wb[5], wb[6], wb[7]);
    }
    __syncthreads();
    // ---- side GEMM on the mantissas: wave w covers k [64w, 64w + 64) of the slab (as k_quant_xa16)
    if (kw < Kp) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 xf = *(const bf16x8*)(slab + qx_swz(r, (64 * wave + 16 * ks) / 8 + h));
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, af0[ks][t], acc[t], 0, 0, 0);
        if (a_limbs > 1) {
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, af1[ks][t], acc[t], 0, 0, 0);
        }
        for (int l = 2; l < a_limbs; ++l) {  // (fp32 A: a third limb, fetched where it is used)
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
    __syncthreads();  // the slab is rewritten by the next trip
    if constexpr (PF) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int i = 0; i < RW; ++i) cur[s2][i] = nxt[s2][i];
    }
  }
  // fixed-order combine of the four waves' tiles: ((w0 + w1) + w2) + w3
  if (wave > 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) red[(((wave - 1) * NT + t) * 16 + j) * 64 + lane] = acc[t][j];
  }
  __syncthreads();
  if (wave == 0) {
    float* dst = part + ((int64_t)c * plan.row_groups + rg) * XA_ROWS * rp;
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

int quant_xa_fused_row_dispatch(const void* x, int dtype, int64_t M, int64_t K, int64_t ldx, const QP& qx, void* xq8,
                                const bf16_t* a_t, int a_limbs, int64_t r, const QP& qa, bf16_t* xaq, float* scratch,
                                size_t scratch_bytes, hipStream_t st) {
  const int64_t Kp = lqer_padded_k(K), Kp8 = padded_k8(K);
  const int rp = (int)lqer_padded_r(r);
  if (qx.kind != LQER_Q_MXINT || !(qx.block <= 0 || qx.block >= K) || qx.mbits > 7 || rp > 64 || a_limbs < 1 || a_limbs > 3 ||
      qa.kind != LQER_Q_MXINT || qa.mbits > 8 || !xaq)
    return LQER_E_UNSUPPORTED;
  const int La = (qa.block <= 0 || qa.block >= rp) ? rp : qa.block;
  const int G = La / 4;
  if (rp % La != 0 || La % 4 != 0 || (G & (G - 1)) != 0 || G > 64) return LQER_E_UNSUPPORTED;
  if (M == 0) return LQER_OK;
  XaPlan plan;
  plan.row_groups = (int)((M + XA_ROWS - 1) / XA_ROWS);
  // chunks of K: as few as fill the chip (~4 workgroups per CU) - every chunk costs a partial tile of M x rp floats
  const int slabs = (int)((Kp8 + QX_K - 1) / QX_K);
  int nch = (int)((1024 + plan.row_groups - 1) / plan.row_groups);
  nch = nch < 1 ? 1 : (nch > slabs ? slabs : nch);
  const int per = (slabs + nch - 1) / nch;
  plan.kc = per * QX_K;
  plan.nchunk = (slabs + per - 1) / per;
  const size_t need = (size_t)plan.nchunk * plan.row_groups * XA_ROWS * rp * sizeof(float);
  if (!scratch || scratch_bytes < need) return LQER_E_UNSUPPORTED;
  const int esz = dtype == LQER_F32 ? 4 : 2;
  const bool vec = ((uintptr_t)x % 16 == 0) && ((ldx * esz) % 16 == 0);
  float* xscale = const_cast<float*>(i8_row_scales(xq8, M, K));
  const bool pf = vec && K % 16 == 0;
  const unsigned grid1 = (unsigned)((M + 3) / 4), grid = (unsigned)(plan.row_groups * plan.nchunk);
  const int nt = (rp + 31) / 32;
#define QXR_ONE(DT, NT)                                                                                                        \
  do {                                                                                                                         \
    if (pf)                                                                                                                    \
      k_quant_xa_row<DT, NT, true><<<grid, QX_K, 0, st>>>(x, M, K, ldx, vec, qx, (int8_t*)xq8, Kp8, Kp, xscale, a_t, a_limbs,   \
                                                          rp, plan, scratch);                                                  \
    else                                                                                                                       \
      k_quant_xa_row<DT, NT, false><<<grid, QX_K, 0, st>>>(x, M, K, ldx, vec, qx, (int8_t*)xq8, Kp8, Kp, xscale, a_t, a_limbs,  \
                                                           rp, plan, scratch);                                                 \
  } while (0)
#define QXR_LAUNCH(DT)                                                                                                         \
  do {                                                                                                                         \
    k_row_scale<DT><<<grid1, 256, 0, st>>>(x, M, K, ldx, vec, qx, xscale);                                                     \
    if (nt == 1)                                                                                                               \
      QXR_ONE(DT, 1);                                                                                                          \
    else                                                                                                                       \
      QXR_ONE(DT, 2);                                                                                                          \
  } while (0)
  switch (dtype) {
    case LQER_F32: QXR_LAUNCH(LQER_F32); break;
    case LQER_F16: QXR_LAUNCH(LQER_F16); break;
    case LQER_BF16: QXR_LAUNCH(LQER_BF16); break;
    default: set_error("unknown dtype %d", dtype); return LQER_E_INVALID;
  }
#undef QXR_LAUNCH
#undef QXR_ONE
  const int64_t items = (int64_t)plan.row_groups * XA_ROWS * rp / 4;
  const unsigned grid2 = (unsigned)((items + 255) / 256);
  switch (G) {
    case 1: k_xa_reduce4<1><<<grid2, 256, 0, st>>>(scratch, plan, rp, qa, xaq, xscale); break;
    case 2: k_xa_reduce4<2><<<grid2, 256, 0, st>>>(scratch, plan, rp, qa, xaq, xscale); break;
    case 4: k_xa_reduce4<4><<<grid2, 256, 0, st>>>(scratch, plan, rp, qa, xaq, xscale); break;
    case 8: k_xa_reduce4<8><<<grid2, 256, 0, st>>>(scratch, plan, rp, qa, xaq, xscale); break;
    default: k_xa_reduce4<16><<<grid2, 256, 0, st>>>(scratch, plan, rp, qa, xaq, xscale); break;
  }
  return check_launch("quantize_act_xa (per-token, fused)");
}

