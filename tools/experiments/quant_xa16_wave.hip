// EXPERIMENT (not compiled into liblqer_hip.so): the fused activation quantizer + split-K side GEMM of the block-16 formats
// (k_quant_xa16, C2's 9.7-10.7 us) with ONE WAVE per (32 rows, 256 k) unit instead of a 4-wave workgroup with an LDS slab, two
// barriers and a cross-wave combine - after k_quant_row8 (one wave per row) had taken the per-token quantizer from 3.8 to
// 5 TB/s.  Two versions, both parity-green through tests/ and slower (tools/ab_step.py, tools/r02_steptrace.sh, C2):
//   * lane = (row, 8-k half) = its MFMA operand fragment, no LDS at all: 15.4 us (every load / store instruction touches
//     32-byte runs of 32 rows);
//   * whole-line loads and stores (lane i takes pieces i, i + 64, ...), wave-private 4 KiB of LDS for the fragment
//     transposition, next sub-step's pieces one step ahead (below): 13.8 us.
// A wave that walks four dependent sub-steps (wait for pieces, quantize, LDS, MFMA) has a longer chain than four waves
// that each do a quarter and meet at a barrier; with one exponent per 16 elements the block structure keeps the LDS step.
// It was wired in behind -DLQER_QX_WAVE in quant_xa_fused_dispatch (rank <= 32, 16-bit aligned input, K % 8 == 0, M > 64).

// ---- the same fused pass with ONE WAVE per (32 rows, 256 k) unit: no workgroup barrier, no cross-wave combine --------------
// The unit is walked in four sub-steps of 64 k.  A sub-step's 32 rows x 128 B are fetched as whole cache lines - lane i
// takes the 16-byte pieces i, i + 64, i + 128, i + 192: 8 rows x 128 B per instruction -, a block of 16 is two neighbouring
// lanes (one DPP exchange for its maximum), each lane quantizes its 8 values and stores them to the activation image in the
// same whole-line pattern; the bf16 pieces also go to this wave's own 4 KiB of LDS (XOR-swizzled) and come back as MFMA
// operand fragments (row = lane & 31), the 16 MFMAs of the unit accumulate its partial tile of x A in the wave's registers.
// The next sub-step's pieces and A^T fragments are requested before the current one is processed.  LDS operations of one
// wave execute in order: no barrier anywhere.  Padded rank <= 32, 16-bit aligned input, K a multiple of 8 (else k_quant_xa16).
template <int DT>
__global__ __launch_bounds__(256, 4) void k_quant_xa16w(const void* __restrict__ x, int64_t M, int64_t K, int64_t ldx, QP q,
                                                        bf16_t* __restrict__ xq, int64_t Kp, const bf16_t* __restrict__ a_t,
                                                        int a_limbs, int rp, int row_groups, float* __restrict__ part) {
  static_assert(DT != LQER_F32, "16-bit inputs");
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * 4096];
  unsigned char* const wl = lds + (threadIdx.x >> 6) * 4096;
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int nchunk = (int)((Kp + QX_K - 1) / QX_K);
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= (int64_t)row_groups * nchunk) return;
  const int rg = (int)(wid / nchunk), c = (int)(wid - (int64_t)rg * nchunk);
  const int64_t k_unit = (int64_t)c * QX_K;
  const bool a_ok = r < rp;
  const bf16_t* const arow = a_t + (int64_t)(a_ok ? r : 0) * Kp + 8 * h;
  // this lane's four pieces of a sub-step: rows prow[j], 8 elements at 8 pq within the sub-step's 64 k
  const int pq = lane & 7;
  auto load_x = [&](int sub, u32x4 (&xr)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t m = (int64_t)rg * 32 + 8 * j + (lane >> 3), k0 = k_unit + 64 * sub + 8 * pq;
      xr[j] = (m < M && k0 < K) ? *(const u32x4*)((const bf16_t*)x + m * ldx + k0) : (u32x4){0, 0, 0, 0};
    }
  };
  auto load_a = [&](int sub, bf16x8 (&ar)[4]) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int64_t k0 = k_unit + 64 * sub + 16 * ks;
      ar[ks] = (a_ok && k0 < Kp) ? *(const bf16x8*)(arow + k0) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  };
  f32x16 acc;
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  auto step = [&](int sub, const u32x4 (&xr)[4], const bf16x8 (&ar)[4]) {
    const int64_t kbase = k_unit + 64 * sub;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = 8 * j + (lane >> 3);
      const int64_t m = (int64_t)rg * 32 + row, k0 = kbase + 8 * pq;
      float v[8];
      const uint32_t wd[4] = {xr[j][0], xr[j][1], xr[j][2], xr[j][3]};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if constexpr (DT == LQER_F16) {
          typedef __attribute__((ext_vector_type(2))) _Float16 h2;
          const h2 hv = __builtin_bit_cast(h2, wd[t]);
          v[2 * t] = (float)hv[0], v[2 * t + 1] = (float)hv[1];
        } else {
          v[2 * t] = __uint_as_float(wd[t] << 16), v[2 * t + 1] = __uint_as_float(wd[t] & 0xffff0000u);
        }
      }
      float amax = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t) amax = fmaxf(amax, fabsf(v[t]));
      // the block's other 8 elements: the neighbouring lane (quad_perm [1,0,3,2])
      amax = fmaxf(amax, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(amax), 0xB1, 0xf, 0xf, true)));
      uint32_t w[4] = {0, 0, 0, 0};
      if (amax > 0.f) {
        const int e = block_exponent(amax, q);
        if (mxint16_fast_ok(e, q)) {
          typedef __attribute__((ext_vector_type(2))) float f2;
          const float s = __uint_as_float((uint32_t)(127 + q.mbits - e) << 23), inv = __uint_as_float((uint32_t)(127 + e - q.mbits) << 23);
          const float es = 1e-9f * s, lo = -q.mneg, hi = q.mmax;
          const f2 magic = {12582912.0f, 12582912.0f};
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const f2 xv = {v[2 * t], v[2 * t + 1]};
            const f2 cc = {copysignf(es, xv[0]), copysignf(es, xv[1])};
            f2 rr = (__builtin_elementwise_fma(xv, (f2){s, s}, cc) + magic) - magic;
            rr[0] = __builtin_amdgcn_fmed3f(rr[0], lo, hi);
            rr[1] = __builtin_amdgcn_fmed3f(rr[1], lo, hi);
            const f2 val = rr * (f2){inv, inv};
            uint32_t b0 = __float_as_uint(val[0]), b1 = __float_as_uint(val[1]);
            if constexpr (DT != LQER_F16) {
              b0 = fabsf(xv[0]) <= 1e-8f ? 0u : b0;
              b1 = fabsf(xv[1]) <= 1e-8f ? 0u : b1;
            }
            w[t] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
          }
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const uint32_t lo16 = exact_bf16_bits(ldexpf(mxint_mantissa(v[2 * t], e, q), e - q.mbits));
            const uint32_t hi16 = exact_bf16_bits(ldexpf(mxint_mantissa(v[2 * t + 1], e, q), e - q.mbits));
            w[t] = lo16 | (hi16 << 16);
          }
        }
      }
      const u32x4 wq = {w[0], w[1], w[2], w[3]};
      if (k0 < Kp) *(u32x4*)(xq + m * Kp + k0) = wq;  // rows up to the padded M are allocated
      *(u32x4*)(wl + row * 128 + ((pq ^ (row & 7)) << 4)) = wq;
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 xf = *(const bf16x8*)(wl + r * 128 + (((2 * ks + h) ^ (r & 7)) << 4));
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, ar[ks], acc, 0, 0, 0);
      for (int l = 1; l < a_limbs; ++l) {  // fp16 / fp32 A: further exact bf16 limbs, fetched where they are used
        const int64_t k0 = kbase + 16 * ks;
        bf16x8 af = {0, 0, 0, 0, 0, 0, 0, 0};
        if (a_ok && k0 < Kp) af = *(const bf16x8*)(a_t + ((int64_t)l * rp + r) * Kp + k0 + 8 * h);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, af, acc, 0, 0, 0);
      }
    }
  };
  // (the A^T fragments of a sub-step are requested at its start - their L2 latency passes under its quantizer -, the
  // activation pieces one sub-step ahead; a second A^T buffer spilled registers at 128 per thread)
  u32x4 xa_[4], xb_[4];
  bf16x8 aa_[4];
  load_x(0, xa_);
  load_a(0, aa_);
  load_x(1, xb_);
  step(0, xa_, aa_);
  load_a(1, aa_);
  load_x(2, xa_);
  step(1, xb_, aa_);
  load_a(2, aa_);
  load_x(3, xb_);
  step(2, xa_, aa_);
  load_a(3, aa_);
  step(3, xb_, aa_);
  // D layout: col n = lane & 31, row (j & 3) + 8 (j >> 2) + 4 h.  part[c][rg * 32 + row][n]
  if (a_ok) {
    float* dst = part + ((int64_t)c * row_groups + rg) * XA_ROWS * rp;
#pragma unroll
    for (int j = 0; j < 16; ++j) dst[((j & 3) + 8 * (j >> 2) + 4 * h) * rp + r] = acc[j];
  }
}

