// The int8 route's activation side as ONE launch (round 6): per-token quantizer (reference x_quantizer with block_size [1, -1],
// quantizers/block_fp.py:55-82 through linear.py:154) + x_q A (linear.py:155) + A_out_quantizer (linear.py:156) - what
// k_quant_row8 + k_xa_partial_lds + k_xa_reduce4 did in three launches of 5-9 us beside a 37-us GEMM at M = 2048.
//
// A workgroup owns ROWS = 8 token rows over ALL of K: no split-K partial tiles in HBM, no reduce launch, no cross-workgroup protocol.
//   phase 1 (wave w = row w): k_quant_row8's code - the row in registers as raw 16-byte chunks (lane l: chunks l, l + 64, ...: every
//           request a contiguous KiB), packed row maximum, one wave reduction, row8_chunk's arithmetic -> 8 int8 per chunk, stored to
//           the image (512 contiguous bytes per wave) AND to an LDS slab [8 rows][Kp8 + 32] (the pitch puts the 8 rows' 8-byte
//           fragments of one k into 16 distinct bank pairs);
//   phase 2 (wave w = one eighth of the 32-k steps): v_mfma_f32_16x16x32_f16 with the 8 rows as rows 0-7 of the 16-row operand
//           (rows 8-15 repeat them; their results are dropped) - token fragment = one ds_read_b64 + the byte -> half conversion of
//           k_xa_partial_lds; A^T fragment = ONE fully coalesced 16-byte load per lane from the FRAGMENT-MAJOR copy of the fp16 image
//           that lqer_f16_prepare writes behind [rp][Kp] (block (step, rank tile): [64 lanes][8 halves] = 1 KiB), batches of 16 / RT
//           steps in two register sets, the first batch requested at the head of the kernel beside the row;
//   phase 3: the 8 waves' partial tiles through LDS, summed in wave order (fixed: run-to-run bit-stable) by rows x rp / 4 threads,
//           row scale, A_out exactly as k_xa_reduce4, bf16 store.
// Earlier one-launch attempts (tools/experiments/quant_xa_rows.hip, quant_rows_xa.hip) held the row in the MFMA operand layout:
// 16-row gathers of 64-byte pieces for x and A^T (1.35 TB/s) on half of the CUs.  Here every global access is a contiguous KiB.
#include <type_traits>

#include "common.h"

namespace lqer {
namespace a8f {

constexpr int ROWS = 8, WAVES = 8;

#ifdef LQER_CLOCKPROBE
// diagnostic build (tools/clock_probe_a8.py): shader cycles at the phase boundaries of every wave, written to a buffer nothing else reads
__device__ unsigned long long* g_a8_stamp_buf = nullptr;
#define A8_STAMP(c) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c)::"memory")
#endif

__host__ __device__ inline int pitch_of(int64_t cols_p8) { return (int)cols_p8 + 32; }
__host__ inline size_t lds_bytes(int64_t cols_p8, int rp) {
  return (size_t)ROWS * pitch_of(cols_p8) + (size_t)WAVES * ROWS * rp * sizeof(float) + ROWS * sizeof(float);
}

// fragment-major copy of the fp16 A^T image: block (s = 32-k step, t = 16-rank tile) at ((s * RT + t) * 64 + lane) * 8 halves;
// lane (n = lane & 15, g = lane >> 4) holds A^T[16 t + n][32 s + 8 g .. + 8) - the B operand of v_mfma_f32_16x16x32_f16
__global__ __launch_bounds__(256) void k_a_frag(const _Float16* __restrict__ a16, int64_t Kp, int rp, int steps, _Float16* __restrict__ frag) {
  const int RT = rp / 16;
  const int64_t total = (int64_t)steps * RT * 64;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int lane = (int)(idx & 63);
    const int64_t blk = idx >> 6;
    const int t = (int)(blk % RT);
    const int64_t s = blk / RT;
    const int n = 16 * t + (lane & 15);
    const int64_t k0 = 32 * s + 8 * (lane >> 4);
    u32x4 v = {0, 0, 0, 0};
    if (k0 + 8 <= Kp) v = *(const u32x4*)(a16 + (int64_t)n * Kp + k0);  // (Kp is a multiple of 64: whole chunks)
    *(u32x4*)(frag + idx * 8) = v;
  }
}

template <int DT, int MAXCH, int RT>
__global__ __launch_bounds__(512) void k_act8_fused(const void* __restrict__ x, int64_t M, int64_t K, int64_t ld, QP qx, int8_t* __restrict__ xq8,
                                                     int64_t cols_p8, float* __restrict__ xscale, const _Float16* __restrict__ a_frag, QP qa,
                                                     int L_aout, bf16_t* __restrict__ xaq, float* __restrict__ zero_p, int zero_n) {
  constexpr int RP = 16 * RT;
  constexpr int SB = 16 / RT;  // steps per batch of A^T fragments (16 fragments = 64 registers per set)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef LQER_CLOCKPROBE
  unsigned long long cp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  A8_STAMP(cp[0]);
#endif
  const int pitch = pitch_of(cols_p8);
  unsigned char* const xb = smem;                                    // [ROWS][pitch] int8 mantissas
  float* const red = (float*)(smem + (size_t)ROWS * pitch);          // [WAVES][ROWS][RP] partial tiles
  float* const rsc = red + WAVES * ROWS * RP;                        // [ROWS] row scales 2^(e - mbits)
  const int64_t m0 = (int64_t)blockIdx.x * ROWS;
  const int64_t row = m0 + wave;
  const bool live_row = row < M;
  const int nch = (int)(K / 8), nch_p = (int)(cols_p8 / 8);
  // ---- requests: the row, then the first batch of this wave's A^T fragments (independent of the row: their latency passes under the
  // quantizer's arithmetic)
  const u32x4* p = (const u32x4*)((const bf16_t*)x + row * ld);
  // (the GEMM's pre-pass behind this launch wants its atomicMax cells zero - gemm_amax_zero_bytes: one store per thread here instead of a
  // memset launch there)
  if ((int)(blockIdx.x * 512 + threadIdx.x) < zero_n) zero_p[blockIdx.x * 512 + threadIdx.x] = 0.f;
  // register u of lane l holds chunk 2 l + (u & 1) + 128 (u >> 1): a lane's registers 2 v, 2 v + 1 are NEIGHBOURING chunks, so their 16
  // mantissas leave as ONE 16-byte store (a KiB per wave instruction, to the image and to the LDS slab) - half the store instructions
  // of k_quant_row8's lane + 64 u order, which cost the texture path a full slot for 512 bytes each
  static_assert(MAXCH % 2 == 0, "chunk pairs");
  auto chunk_of = [&](int u) { return 2 * lane + (u & 1) + 128 * (u >> 1); };
  u32x4 raw[MAXCH];
#pragma unroll
  for (int u = 0; u < MAXCH; ++u) raw[u] = (live_row && chunk_of(u) < nch) ? p[chunk_of(u)] : (u32x4){0, 0, 0, 0};
  const int steps = (int)(cols_p8 / 32);
  const int spw = (steps + WAVES - 1) / WAVES;  // steps per wave
  const int s_begin = wave * spw, s_end = s_begin + spw < steps ? s_begin + spw : steps;
  const u32x4* const fr = (const u32x4*)a_frag + lane;  // block (s, t): fr[(s * RT + t) * 64]
  u32x4 fa[SB][RT], fb[SB][RT];
  auto load_batch = [&](u32x4 (&f)[SB][RT], int s0) {
#pragma unroll
    for (int i = 0; i < SB; ++i)
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const int s = s0 + i < s_end ? s0 + i : s_end - 1;  // (past the end: a valid block again - multiplied by nothing)
        f[i][t] = s_begin < s_end ? fr[((int64_t)s * RT + t) * 64] : (u32x4){0, 0, 0, 0};
      }
  };
  // The CU's texture path moves 64 B per cycle and this kernel is bound by it (352 KiB per workgroup at K = 4096: 64 of rows, 256 of A^T
  // fragments, 32 of image stores); its queue is shallow, so a wave that issues 32 fragment requests in a row stands still until the
  // path has taken them - requested beside the row (in front of it or, behind an issue barrier, after it) they held the quantizer back by
  // ~3,500 cycles (tools/clock_probe_a8.py).  Order kept: the row; once its maximum is known the first batch of fragments, which streams
  // under the quantizer's arithmetic; the second batch behind the image stores, under the barrier and the first batch's MFMAs.
  // (rows of up to 6144 elements: the registers also hold the SECOND batch - at K = 4096 that is all of a wave's A^T)
  constexpr bool TWO = MAXCH <= 8 || (MAXCH <= 12 && DT == LQER_F16);  // (bf16 rows of 12 chunks: the wider conversion would spill)

  // ---- phase 1: the row -> int8 image + LDS slab (k_quant_row8's arithmetic)
#ifdef LQER_CLOCKPROBE
  A8_STAMP(cp[1]);  // requests out
#endif
  float amax = row8_amax<DT, MAXCH>(raw);
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) amax = fmaxf(amax, __shfl_xor(amax, s, 64));
#ifdef LQER_CLOCKPROBE
  asm volatile("" ::"v"(amax));
  A8_STAMP(cp[2]);  // the row has landed, its maximum is known
#endif
  load_batch(fa, s_begin);
  const bool any = amax > 0.f;
  const int e = any ? block_exponent(amax, qx) : 0;
  const float rs = any ? ldexpf(1.0f, e - qx.mbits) : 1.0f;
  if (lane == 0) {
    rsc[wave] = rs;
    if (live_row) xscale[row] = rs;
  }
  int8_t* const dst = xq8 + row * cols_p8;
  unsigned char* const xrow = xb + wave * pitch;
  const bool fast = mxint16_fast_ok(e, qx);  // (wave-uniform)
  const float sc = __uint_as_float((uint32_t)(127 + (fast ? qx.mbits - e : 0)) << 23);
  const float es = 1e-9f * sc;
  // (nch_p is even - the image is padded to 128 k -, so a pair of chunks is inside it or outside it as a whole)
  auto put = [&](int c0, const u32x2 w0, const u32x2 w1) {
    const u32x4 w = {w0[0], w0[1], w1[0], w1[1]};
    if (live_row) *(u32x4*)(dst + (int64_t)c0 * 8) = w;
    *(u32x4*)(xrow + c0 * 8) = w;
  };
  auto emit = [&](auto fast_c) {
    constexpr bool FAST = decltype(fast_c)::value;
#pragma unroll
    for (int u = 0; u < MAXCH; u += 2) {
      const int c = chunk_of(u);
      if (c >= nch_p) continue;
      put(c, row8_chunk<DT, FAST>(raw[u], c < nch && any, e, qx, sc, es), row8_chunk<DT, FAST>(raw[u + 1], c + 1 < nch && any, e, qx, sc, es));
    }
  };
  if (DT == LQER_F16 && any && row8_h16_ok(e, qx)) {  // (wave-uniform) packed half arithmetic: same bytes, a third of the instructions
#pragma unroll
    for (int u = 0; u < MAXCH; u += 2) {
      const int c = chunk_of(u);
      if (c >= nch_p) continue;
      put(c, row8_chunk_h16(raw[u], c < nch, e, qx), row8_chunk_h16(raw[u + 1], c + 1 < nch, e, qx));
    }
  } else if (fast)
    emit(std::true_type{});
  else
    emit(std::false_type{});
  if constexpr (TWO) load_batch(fb, s_begin + SB);
#ifdef LQER_CLOCKPROBE
  A8_STAMP(cp[3]);  // quantized, stores issued
#endif
  __syncthreads();
#ifdef LQER_CLOCKPROBE
  A8_STAMP(cp[4]);  // every row's slab is in LDS
#endif

  // ---- phase 2: this wave's steps of x_q A on the fp16 MFMA (int8 mantissas are exact halves; the row scale comes last)
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  typedef __attribute__((ext_vector_type(8))) _Float16 h8;
  const h2 bias = {(_Float16)-1152.0f, (_Float16)-1152.0f};
  const int g = lane >> 4;
  const unsigned char* const tok = xb + (lane & 7) * pitch + 8 * g;  // + 32 s
  f32x4 acc[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto compute_batch = [&](const u32x4 (&f)[SB][RT], int s0) {
    u32x2 tb8[SB];  // the batch's token fragments first (one LDS round trip for all of them, not one per step)
#pragma unroll
    for (int i = 0; i < SB; ++i) tb8[i] = *(const u32x2*)(tok + 32 * (s0 + i < s_end ? s0 + i : s_end - 1));
#pragma unroll
    for (int i = 0; i < SB; ++i) {
      if (s0 + i >= s_end) break;  // (wave-uniform)
      const u32x2 b8 = tb8[i];
      u32x4 hf;  // bytes -> 8 halves (i8x32_to_f16's arithmetic: 0x6400 | (b ^ 0x80) = 1024 + 128 + b, then - 1152)
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const uint32_t tb = b8[d] ^ 0x80808080u;
        hf[2 * d] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, __builtin_amdgcn_perm(0x64646464u, tb, 0x04010400u)) + bias);
        hf[2 * d + 1] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, __builtin_amdgcn_perm(0x64646464u, tb, 0x04030402u)) + bias);
      }
#pragma unroll
      for (int t = 0; t < RT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, hf), __builtin_bit_cast(h8, f[i][t]), acc[t], 0, 0, 0);
    }
  };
  for (int s0 = s_begin; s0 < s_end; s0 += 2 * SB) {
    if (!(TWO && s0 == s_begin) && s0 + SB < s_end) load_batch(fb, s0 + SB);
    compute_batch(fa, s0);
    if (s0 + 2 * SB < s_end) load_batch(fa, s0 + 2 * SB);
    if (s0 + SB < s_end) compute_batch(fb, s0 + SB);
  }
#ifdef LQER_CLOCKPROBE
  asm volatile("" ::"v"(acc[0][0]));
  A8_STAMP(cp[5]);  // this wave's steps multiplied
#endif
  // D layout: column n = lane & 15, rows 4 g + j: token rows 0-7 live in g = 0, 1
  if (g < 2) {
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[(wave * ROWS + 4 * g + j) * RP + 16 * t + (lane & 15)] = acc[t][j];
  }
  __syncthreads();
#ifdef LQER_CLOCKPROBE
  A8_STAMP(cp[6]);
  if (g_a8_stamp_buf && lane == 0) {
    unsigned long long* o = g_a8_stamp_buf + ((size_t)blockIdx.x * WAVES + wave) * 8;
#pragma unroll
    for (int i = 1; i < 7; ++i) o[i] = cp[i] - cp[0];
    o[0] = cp[0];
  }
#endif

  // ---- phase 3: fixed-order sum of the 8 partial tiles, row scale, A_out (k_xa_reduce4's arithmetic), bf16 store
  const int tid = threadIdx.x;
  const bool live = tid < ROWS * RP / 4;
  const int r = tid / (RP / 4), c4 = tid - r * (RP / 4);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      const float4 v = *(const float4*)(red + (w * ROWS + r) * RP + 4 * c4);
      s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
    const float scr = rsc[r];
    s.x *= scr, s.y *= scr, s.z *= scr, s.w *= scr;
  }
  if (tid >= 64 * ((ROWS * RP / 4 + 63) / 64)) return;  // (whole waves only: the shuffles below need their partners)
  float bmax = fmaxf(fmaxf(fabsf(s.x), fabsf(s.y)), fmaxf(fabsf(s.z), fabsf(s.w)));
  const int G = L_aout / 4;
  for (int d = 1; d < G; d <<= 1) bmax = fmaxf(bmax, __shfl_xor(bmax, d, 64));
  if (!live || m0 + r >= M) return;
  const bool anyb = bmax > 0.f;
  const int eb = anyb ? block_exponent(bmax, qa) : 0;
  const float v[4] = {s.x, s.y, s.z, s.w};
  uint32_t w2[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float m0v = anyb ? mxint_mantissa(v[2 * i], eb, qa) : 0.f;
    const float m1v = anyb ? mxint_mantissa(v[2 * i + 1], eb, qa) : 0.f;
    w2[i] = exact_bf16_bits(ldexpf(m0v, eb - qa.mbits)) | (exact_bf16_bits(ldexpf(m1v, eb - qa.mbits)) << 16);
  }
  *(uint2*)(xaq + ((m0 + r) * RP + 4 * c4)) = make_uint2(w2[0], w2[1]);
}

template <int DT, int MAXCH>
static int launch(const void* x, int64_t M, int64_t K, int64_t ld, const QP& qx, int8_t* xq8, int64_t cols_p8, float* xscale, const _Float16* a_frag,
                  const QP& qa, int L, int rp, bf16_t* xaq, float* zero_p, int zero_n, hipStream_t st) {
  const unsigned grid = (unsigned)((M + ROWS - 1) / ROWS);
  const int lds = (int)lds_bytes(cols_p8, rp);
#define A8F_LAUNCH(RTv)                                                                                                            \
  do {                                                                                                                             \
    static LdsLimitOnce once;                                                                                                      \
    once.set((const void*)k_act8_fused<DT, MAXCH, RTv>, 160 * 1024);                                                                \
    k_act8_fused<DT, MAXCH, RTv><<<grid, 512, lds, st>>>(x, M, K, ld, qx, xq8, cols_p8, xscale, a_frag, qa, L, xaq, zero_p, zero_n); \
  } while (0)
  switch (rp / 16) {
    case 1: A8F_LAUNCH(1); break;
    case 2: A8F_LAUNCH(2); break;
    default: A8F_LAUNCH(4); break;
  }
#undef A8F_LAUNCH
  return check_launch("quantize_act_xa (fused int8 route)");
}

}  // namespace a8f

#ifdef LQER_CLOCKPROBE
extern "C" int lqer_debug_set_a8_stamp_buffer(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(a8f::g_a8_stamp_buf), &p, sizeof(p)); }
#endif

size_t a_f16_image_bytes(int64_t K, int64_t r) {  // [rp][Kp] fp16, then its fragment-major copy over padded_k8(K) / 32 steps
  const int64_t rp = lqer_padded_r(r);
  return (size_t)rp * (lqer_padded_k(K) + padded_k8(K)) * sizeof(_Float16);
}

int a_frag_dispatch(void* a_f16, int64_t K, int64_t r, hipStream_t st) {
  const int64_t rp = lqer_padded_r(r), Kp = lqer_padded_k(K);
  if (rp % 16 != 0) return LQER_OK;  // (no fragment image for such a rank: the fused kernel is not taken)
  const int steps = (int)(padded_k8(K) / 32);
  const int64_t total = (int64_t)steps * (rp / 16) * 64;
  a8f::k_a_frag<<<(unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096), 256, 0, st>>>((const _Float16*)a_f16, Kp, (int)rp, steps,
                                                                                                      (_Float16*)a_f16 + rp * Kp);
  return check_launch("f16_prepare (fragment-major A^T)");
}

// LQER_E_UNSUPPORTED: not this kernel's case (the caller takes the three-launch route)
int act8_fused_dispatch(const void* x, int dtype, int64_t M, int64_t K, int64_t ldx, const QP& qx, void* xq_i8, const void* a_f16, int64_t r,
                        const QP& qa, bf16_t* xaq, int tuning, hipStream_t st, float* zero_p, size_t zero_bytes, bool* zeroed) {
  if (zeroed) *zeroed = false;
#ifdef LQER_NO_ACT8_FUSED
  return LQER_E_UNSUPPORTED;
#endif
  if (tuning & LQER_TUNE_ACT8_SPLIT) return LQER_E_UNSUPPORTED;
  const int64_t rp = lqer_padded_r(r);
  if (dtype == LQER_F32 || !a_f16 || !xaq || r <= 0 || M <= 0) return LQER_E_UNSUPPORTED;
  if (!(rp == 16 || rp == 32 || rp == 64) || qx.mbits > 7) return LQER_E_UNSUPPORTED;
  if (qx.kind != LQER_Q_MXINT || !(qx.block <= 0 || qx.block >= K)) return LQER_E_UNSUPPORTED;
  if (((uintptr_t)x % 16) != 0 || ((ldx * 2) % 16) != 0 || K % 8 != 0) return LQER_E_UNSUPPORTED;
  if (!(qa.kind == LQER_Q_MXINT || qa.kind == LQER_Q_INT)) return LQER_E_UNSUPPORTED;
  if (qa.kind == LQER_Q_MXINT ? qa.mbits > 8 : !(qa.mmax <= 256.f && qa.mneg <= 256.f)) return LQER_E_UNSUPPORTED;
  const int L = (qa.block <= 0 || qa.block >= rp) ? (int)rp : qa.block;
  const int G = L / 4;
  if (rp % L != 0 || L % 4 != 0 || (G & (G - 1)) != 0 || G > 64) return LQER_E_UNSUPPORTED;
  // every workgroup streams the whole A^T image: worth it while the token count is small (M = 2048: 64 MB through L2 against three
  // launches; at M = 16384 the split-K kernels read A^T once per 128 rows)
  // (... and large enough for its grid of M / 8 workgroups to cover the chip's better half: below, the split-K kernels spread the same
  // rows over more CUs)
  if (!(tuning & LQER_TUNE_ACT8_FUSED) && (M > LQER_ACT8_FUSED_MAX_M || M < LQER_ACT8_FUSED_MIN_M)) return LQER_E_UNSUPPORTED;
  const int64_t cols_p8 = padded_k8(K);
  const int64_t nch_p = cols_p8 / 8;
  if (a8f::lds_bytes(cols_p8, (int)rp) > 160 * 1024) return LQER_E_UNSUPPORTED;
  int8_t* const xq8 = (int8_t*)xq_i8;
  float* const xscale = const_cast<float*>(i8_row_scales(xq_i8, M, K));
  const _Float16* const a_frag = (const _Float16*)a_f16 + rp * lqer_padded_k(K);
  // zero fill on the GEMM's behalf: one store per thread of the grid, or not at all (the caller keeps its memset)
  const bool zfit = zero_p && zero_bytes > 0 && zero_bytes / 4 <= (size_t)((M + a8f::ROWS - 1) / a8f::ROWS) * 512;
  float* const zp = zfit ? zero_p : nullptr;
  const int zn = zfit ? (int)(zero_bytes / 4) : 0;
  if (zeroed) *zeroed = zfit;  // (every return below this line that is not LQER_E_UNSUPPORTED has launched the kernel)
#define A8F_DT(DTv)                                                                                                                       \
  do {                                                                                                                                    \
    if (nch_p <= 64 * 8) return a8f::launch<DTv, 8>(x, M, K, ldx, qx, xq8, cols_p8, xscale, a_frag, qa, L, (int)rp, xaq, zp, zn, st);       \
    if (nch_p <= 64 * 12) return a8f::launch<DTv, 12>(x, M, K, ldx, qx, xq8, cols_p8, xscale, a_frag, qa, L, (int)rp, xaq, zp, zn, st);     \
    if constexpr (DTv == LQER_F16) /* (bf16 rows beyond 6144 elements: 112 raw registers + the wider conversion spill - three launches) */ \
      if (nch_p <= 64 * 28) return a8f::launch<DTv, 28>(x, M, K, ldx, qx, xq8, cols_p8, xscale, a_frag, qa, L, (int)rp, xaq, zp, zn, st);   \
  } while (0)
  if (dtype == LQER_F16) A8F_DT(LQER_F16);
  else if (dtype == LQER_BF16) A8F_DT(LQER_BF16);
#undef A8F_DT
  return LQER_E_UNSUPPORTED;
}

}  // namespace lqer
