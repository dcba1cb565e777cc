// Activation quantizer + the whole first half of the side path in ONE launch, for prefill sizes:
//
//   xq  = Q_x(x)                      (reference quantized_layers/linear.py:148, blocks of 16 along k)
//   xAq = A_out_quantizer(xq @ A)     (linear.py:154)
//
// One workgroup owns ROWS token rows (8 or 16) over ALL of K, so x @ A is complete inside the workgroup: no split-K
// partial tiles in HBM, no reduce launch (k_quant_xa16 + k_xa_reduce4 are two launches and 4 MiB of partials at
// 2048 x 4096).  The price is that every workgroup streams the whole A^T image from L2 (rp x Kp bf16: 256 KiB at
// rank 32, K 4096) - L2 bandwidth that is idle in an HBM-bound kernel.
//
// The workgroup walks K in stages of 1024 with two wave roles (loads complete in order per wave, so the L2-hot A^T
// fragment loads must not queue behind the HBM loads of the activations):
//   producers (waves 0-3): every thread quantizes two (ROWS 16: four) 16-element blocks per stage - lane = consecutive
//     32-byte pieces of a row, a wave = 2 KiB of one row -, writes the bf16 image to HBM and to an LDS slab (XOR-swizzled
//     16-byte chunks, double-buffered); raw x loads run QR_PD pieces ahead in registers;
//   consumers (waves 4-7): v_mfma_f32_16x16x32_bf16 (tokens x rank tiles of 16) over 256 k each of the slab the
//     producers finished one barrier ago, A^T fragments straight from L2, requested one stage ahead.
// One barrier per stage: producers arrive with slab s written, consumers with slab s - 1 read.  At the end the 4
// consumer waves' tiles are summed in a fixed order through LDS (bit-reproducible), A_out is applied (a block of 16 rank
// entries of a token = the 16 lanes of a DPP row) and the bf16 image of xAq is written.
#include <type_traits>

#include "common.h"

namespace lqer {

constexpr int QR_THREADS = 768;  // 8 producer waves + 4 consumer waves
constexpr int QR_PWAVES = 8;
constexpr int QR_KS = 1024;  // k per stage
#ifndef LQER_QR_PD
#define LQER_QR_PD 8
#endif
#ifndef LQER_QR_PERM
#define LQER_QR_PERM 0
#endif
constexpr int QR_PD = LQER_QR_PD;  // 32-byte pieces of raw activation loads in flight per thread

template <int DT>
struct QrRaw {
  static constexpr int N = DT == LQER_F32 ? 4 : 2;  // 16-byte loads per 16-element block
  uint4 r[N];
};

// 16-byte loads through a buffer descriptor: a lane whose offset lies outside the descriptor's range reads zeros, so row / k
// tails need no branch around the load (branches make hipcc fall back to s_waitcnt vmcnt(0) and drain the prefetch)
typedef __attribute__((ext_vector_type(4))) uint32_t qr_u4;
constexpr uint32_t QR_OOB = 0x7ffffff0u;
template <int DT, typename RS>
__device__ __forceinline__ void qr_load(const RS& rsrc, uint32_t byte_off, QrRaw<DT>& raw) {
  constexpr int N = QrRaw<DT>::N;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const qr_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off + 16 * i, 0, 0);
    raw.r[i] = make_uint4(v[0], v[1], v[2], v[3]);
  }
}

template <int DT>
__device__ __forceinline__ void qr_convert(const QrRaw<DT>& raw, float (&v)[16]) {
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

// bf16 image (8 words) of one block of 16 - the same arithmetic as k_quant_xa16 / k_quant_seg16 (bit-identical images)
template <int DT>
__device__ __forceinline__ void qr_quantize(const QrRaw<DT>& raw, const QP& q, uint32_t (&w)[8]) {
  float v[16];
  qr_convert<DT>(raw, v);
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) amax = fmaxf(amax, fabsf(v[i]));
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = 0;
  if (amax > 0.f) {
    const int e = block_exponent(amax, q);
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

// slab: [ROWS][QR_KS] bf16, 16-byte chunk c of row r at r * 2048 + ((c ^ sw(r)) << 4); sw spreads the rows that one
// ds_read_b128 lane group touches over the 16 chunk slots of a 256-byte bank row (8 rows: 2 r; 16 rows: r)
template <int ROWS>
__device__ __forceinline__ int qr_slab(int row, int chunk) {
  return row * (QR_KS * 2) + ((chunk ^ (ROWS == 8 ? 2 * row : row)) << 4);
}

template <int DT, int NT, int ROWS>
__global__ __launch_bounds__(QR_THREADS) void k_quant_xa_rows(const void* __restrict__ x, int64_t M, int64_t K, int64_t ldx, QP q,
                                                              bf16_t* __restrict__ xq, int64_t Kp,
                                                              const bf16_t* __restrict__ a_t, int rp, QP qa,
                                                              bf16_t* __restrict__ xaq) {
  constexpr int NB = ROWS / 8;                       // blocks of 16 per producer thread and stage
  constexpr int PD0 = QR_PD / (NB * (DT == LQER_F32 ? 2 : 1));
  constexpr int PD = PD0 < 1 ? 1 : (PD0 > 4 ? 4 : PD0);  // stages of raw activation loads in flight
  constexpr bool AF2 = NT <= 2;                      // A^T fragments one stage ahead (else requested at the head of their stage)
  constexpr int SLAB = ROWS * QR_KS * 2;             // bytes of one slab
  constexpr int RED = 3 * NT * 4 * 64 * 4;           // the consumer waves' tiles at the end (reuses the slabs)
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * SLAB > RED ? 2 * SLAB : RED];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  const int nstage = (int)((Kp + QR_KS - 1) / QR_KS);
  constexpr int UNR = PD < 2 ? 2 : PD;  // both roles run whole groups of UNR stages (= barriers); a stage past the end of K is empty
  const int nst = (nstage + UNR - 1) / UNR * UNR;
  const int l15 = lane & 15, lq = lane >> 4;

  if (wave < QR_PWAVES) {
    // ---- producers: quantize.  Thread's blocks: row = wave + 8 b, k = stage * 1024 + 16 lane.  Descriptor = this
    // workgroup's rows that exist (rows past M read zeros); k >= K is sent out of range explicitly (ldx may equal K)
    constexpr int ESZ = DT == LQER_F32 ? 4 : 2;
    const int64_t rows_here = M - row0 < ROWS ? M - row0 : ROWS;
    const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)x + row0 * ldx * ESZ), 0,
                                                          (int)(((rows_here - 1) * ldx + K) * ESZ), 0x00020000);
    auto x_off = [&](int s, int b) -> uint32_t {
      const int64_t k0 = (int64_t)s * QR_KS + 16 * lane;
      return k0 < K ? (uint32_t)(((wave + 8 * b) * ldx + k0) * ESZ) : QR_OOB;
    };
    QrRaw<DT> raw[PD][NB];
#pragma unroll
    for (int d = 0; d < PD; ++d)
#pragma unroll
      for (int b = 0; b < NB; ++b) qr_load<DT>(x_rsrc, x_off(d, b), raw[d][b]);
    // the ring of raw loads is indexed statically (the loop is unrolled by its depth): a register copy of a load still in
    // flight would wait for it - i.e. for the newest request - every stage
    auto pstage = [&](int s, auto slot_c) {
      constexpr int SLOT = decltype(slot_c)::value % PD;
      unsigned char* slab = smem + (s & 1) * SLAB;
      const int64_t k0 = (int64_t)s * QR_KS + 16 * lane;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int row = wave + 8 * b;
        uint32_t w[8];
#ifdef LQER_QR_ABL_NOQUANT
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = i < 4 ? raw[SLOT][b].r[0].x + i : raw[SLOT][b].r[1].y + i;
#else
        qr_quantize<DT>(raw[SLOT][b], q, w);
#endif
#ifdef LQER_QR_ABL_NOSTORE
        if (k0 < 0) {
#else
        if (k0 < Kp) {
#endif
          uint4* dst = (uint4*)(xq + (row0 + row) * Kp + k0);  // rows up to the padded M are allocated
          dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
          dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
        }
        *(uint4*)(slab + qr_slab<ROWS>(row, 2 * lane)) = make_uint4(w[0], w[1], w[2], w[3]);
        *(uint4*)(slab + qr_slab<ROWS>(row, 2 * lane + 1)) = make_uint4(w[4], w[5], w[6], w[7]);
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) qr_load<DT>(x_rsrc, x_off(s + PD, b), raw[SLOT][b]);
      // (not __syncthreads(): its fence waits for vmcnt(0), i.e. for the prefetches just issued - only the LDS writes matter)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // barrier s: slab s is complete
    };
    for (int s = 0; s < nst; s += UNR) {
      pstage(s, std::integral_constant<int, 0>{});
      pstage(s + 1, std::integral_constant<int, 1>{});
      if constexpr (UNR >= 4) {
        pstage(s + 2, std::integral_constant<int, 2>{});
        pstage(s + 3, std::integral_constant<int, 3>{});
      }
      if constexpr (UNR >= 8) {
        pstage(s + 4, std::integral_constant<int, 4>{});
        pstage(s + 5, std::integral_constant<int, 5>{});
        pstage(s + 6, std::integral_constant<int, 6>{});
        pstage(s + 7, std::integral_constant<int, 7>{});
      }
    }
    asm volatile("s_barrier" ::: "memory");  // the two barriers of the combine below
    asm volatile("s_barrier" ::: "memory");
    return;
  }

  // ---- consumers: wave cw multiplies k [256 cw, 256 cw + 256) of every stage.  The order of k inside the wave's 256 is a
  // free permutation as long as both operands use it: slot (lq, j) of MFMA i stands for k = 64 lq + 8 i + j, so that a lane
  // reads 128 CONTIGUOUS bytes of its A^T row over the 8 MFMAs (whole cache lines, instead of 16-byte pieces 64 bytes apart).
  // A^T fragments of one stage: MFMA i, rank tile t: lane (n = 16 t + l15, k = 64 lq + 8 i).  Descriptor = limb 0 of the image: rank tiles past rp are out
  // of range by themselves, k >= Kp explicitly.  Their own waves, because loads complete in order per wave: behind the
  // HBM loads of the activations an L2-hot fragment load would wait one HBM latency per stage
  const int cw = wave - QR_PWAVES;
  const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a_t, 0, (int)(rp * Kp * 2), 0x00020000);
  bf16x8 af[AF2 ? 2 : 1][AF2 ? 8 : 4][NT];
  auto load_af = [&](int s, auto buf_c, auto i0_c) {  // MFMAs i0 .. i0 + (AF2 ? 8 : 4)
    constexpr int BUF = decltype(buf_c)::value, I0 = decltype(i0_c)::value;
    const int64_t kl = (int64_t)s * QR_KS + 256 * cw + 64 * lq;  // Kp is a multiple of 64: the lane's 64 k are inside or outside
#pragma unroll
    for (int i = 0; i < (AF2 ? 8 : 4); ++i) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#if LQER_QR_PERM
        const uint32_t off = kl < Kp ? (uint32_t)(((16 * t + l15) * Kp + kl + 8 * (I0 + i)) * 2) : QR_OOB;
#else
        const uint32_t off = kl < Kp ? (uint32_t)(((16 * t + l15) * Kp + kl - 64 * lq + 32 * (I0 + i) + 8 * lq) * 2) : QR_OOB;
#endif
        af[BUF][i][t] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, off, 0, 0));
      }
    }
  };
  using std::integral_constant;
  if constexpr (AF2) load_af(0, integral_constant<int, 0>{}, integral_constant<int, 0>{});
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = ROWS == 8 ? (lane & 7) : l15;  // token row of this lane's x fragment (8 rows: rows 8..15 duplicate 0..7)
  auto stage = [&](int s, auto buf_c) {
    constexpr int BUF = AF2 ? decltype(buf_c)::value : 0;
    const unsigned char* slab = smem + (s & 1) * SLAB;
    // (unconditional: past the end of K the offsets are out of range - a load under a branch would spoil the compiler's
    // counted vmcnt)
#ifndef LQER_QR_ABL_NOAF
    if constexpr (AF2) load_af(s + 1, integral_constant<int, BUF ^ 1>{}, integral_constant<int, 0>{});
    else load_af(s, integral_constant<int, 0>{}, integral_constant<int, 0>{});
#endif
    asm volatile("s_barrier" ::: "memory");  // barrier s: slab s is complete; the producers go on with slab s + 1
    // (no k < Kp test: past the end of K the A^T fragments are zeros and the slab holds zeros as well)
    auto mm = [&](auto i0_c) {
      constexpr int I0 = decltype(i0_c)::value;
#pragma unroll
      for (int i = 0; i < (AF2 ? 8 : 4); ++i) {
#if LQER_QR_PERM
        const bf16x8 xf = *(const bf16x8*)(slab + qr_slab<ROWS>(frow, 32 * cw + 8 * lq + I0 + i));
#else
        const bf16x8 xf = *(const bf16x8*)(slab + qr_slab<ROWS>(frow, 32 * cw + 4 * (I0 + i) + lq));
#endif
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, af[BUF][i][t], acc[t], 0, 0, 0);
      }
    };
    mm(integral_constant<int, 0>{});
    if constexpr (!AF2) {  // rank tiles 3, 4: the second half of the stage's fragments through the same registers
#ifndef LQER_QR_ABL_NOAF
      load_af(s, integral_constant<int, 0>{}, integral_constant<int, 4>{});
#endif
      mm(integral_constant<int, 4>{});
    }
  };
  for (int s = 0; s < nst; s += 2) {  // stages in pairs, no test between them: one straight-line loop body
    stage(s, integral_constant<int, 0>{});
    stage(s + 1, integral_constant<int, 1>{});
  }

  // ---- fixed-order combine ((c0 + c1) + c2) + c3, A_out, store
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave is done with the slabs
  float* red = (float*)smem;
  if (cw > 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[(((cw - 1) * NT + t) * 4 + j) * 64 + lane] = acc[t][j];
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (cw != 0) return;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float sum = acc[t][j];
#pragma unroll
      for (int w2 = 0; w2 < 3; ++w2) sum += red[((w2 * NT + t) * 4 + j) * 64 + lane];
      acc[t][j] = sum;
    }
  // lane holds token row 4 lq + j, rank entries 16 t + l15.  A_out block = TB consecutive rank tiles of one token
  const int La = (qa.block <= 0 || qa.block >= rp) ? rp : qa.block;
  const int TB = La / 16;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 4 * lq + j;
    float tmax[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) tmax[t] = row16_max(fabsf(acc[t][j]));
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (16 * t >= rp) continue;
      float amax = 0.f;
      const int t0 = t / TB * TB;
#pragma unroll
      for (int u = 0; u < NT; ++u)
        if (u >= t0 && u < t0 + TB) amax = fmaxf(amax, tmax[u]);
      float out = 0.f;
      if (amax > 0.f) {
        const int e = block_exponent(amax, qa);
        out = ldexpf(mxint_mantissa(acc[t][j], e, qa), e - qa.mbits);
      }
      if (row < ROWS) xaq[(row0 + row) * rp + 16 * t + l15] = (bf16_t)exact_bf16_bits(out);
    }
  }
}

// Prefill sizes only: returns LQER_E_UNSUPPORTED (without launching) for anything the kernel does not cover - the caller then
// takes the split-K route (k_quant_xa16 + k_xa_reduce4, or the separate quantizer and side GEMM).
#ifndef LQER_QR_MIN_M
#define LQER_QR_MIN_M 65
#endif
int quant_xa_rows_dispatch(const void* x, int dtype, int64_t M, int64_t K, int64_t ldx, const QP& qx, bf16_t* xq, const bf16_t* a_t,
                           int a_limbs, int64_t r, const QP& qa, bf16_t* xaq, hipStream_t st) {
  const int64_t Kp = lqer_padded_k(K);
  const int rp = (int)lqer_padded_r(r);
  if (M < LQER_QR_MIN_M || !xaq || !a_t || a_limbs != 1) return LQER_E_UNSUPPORTED;  // (fp16 / fp32 A: further limbs - the split-K route)
  if (qx.kind != LQER_Q_MXINT || qx.block != 16 || qx.mbits > 8 || qa.kind != LQER_Q_MXINT || qa.mbits > 8 || rp > 64)
    return LQER_E_UNSUPPORTED;
  const int La = (qa.block <= 0 || qa.block >= rp) ? rp : qa.block;
  if (La % 16 != 0 || rp % La != 0) return LQER_E_UNSUPPORTED;
  const int esz = dtype == LQER_F32 ? 4 : 2;
  if (K % 16 != 0 || (uintptr_t)x % 16 != 0 || (ldx * esz) % 16 != 0) return LQER_E_UNSUPPORTED;
  if (16 * ldx * esz >= (int64_t)QR_OOB || (int64_t)rp * Kp * 2 >= (int64_t)QR_OOB) return LQER_E_UNSUPPORTED;  // 32-bit buffer offsets
  // 16-row workgroups halve the A^T stream; taken once they still fill the chip
  const bool r16 = (M + 15) / 16 >= 512;
  const unsigned grid = (unsigned)((M + (r16 ? 15 : 7)) / (r16 ? 16 : 8));
#define QR_LAUNCH(DT, NT)                                                                                              \
  do {                                                                                                                 \
    if (r16)                                                                                                           \
      k_quant_xa_rows<DT, NT, 16><<<grid, QR_THREADS, 0, st>>>(x, M, K, ldx, qx, xq, Kp, a_t, rp, qa, xaq);   \
    else                                                                                                               \
      k_quant_xa_rows<DT, NT, 8><<<grid, QR_THREADS, 0, st>>>(x, M, K, ldx, qx, xq, Kp, a_t, rp, qa, xaq);    \
  } while (0)
#define QR_CASE(DT)                                                                                                    \
  case DT:                                                                                                             \
    if (rp <= 16) QR_LAUNCH(DT, 1); else if (rp <= 32) QR_LAUNCH(DT, 2); else if (rp <= 48) QR_LAUNCH(DT, 3); else QR_LAUNCH(DT, 4); \
    break;
  switch (dtype) {
    QR_CASE(LQER_F32) QR_CASE(LQER_F16) QR_CASE(LQER_BF16)
    default: set_error("unknown dtype %d", dtype); return LQER_E_INVALID;
  }
#undef QR_CASE
#undef QR_LAUNCH
  return check_launch("quantize_act_xa (rows)");
}

}  // namespace lqer
