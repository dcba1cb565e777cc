// The block-16 MXINT activation side as ONE launch (round 6): x_quantizer in blocks of [1, 16] (reference quantizers/block_fp.py:55-82
// through quantized_layers/linear.py:154, llama-7b.toml:78-105) + x_q A (linear.py:155) + A_out_quantizer (linear.py:156) - what
// k_quant_xa16 (quantizer + split-K partial tiles of x A) + k_xa_reduce4 (fixed-order sum + A_out) did in two launches: the reduce is a
// 4.9-us launch-floor kernel beside a 56-us GEMM at BASELINE configs[1] and resisted three fusions into its neighbours (NOTEBOOK 7-9).
//
// A workgroup owns ROWS = 8 token rows over ALL of K (no partial tiles in HBM, no reduce launch, no cross-workgroup protocol), and inside
// it every WAVE is a pipeline of its own over slabs of 512 k (slab s belongs to wave s % 8) - a block's exponent needs nothing outside
// its 16 elements, so nothing waits for a whole row:
//   load     the slab of each of the 8 rows: one 16-byte request per lane and row, a contiguous KiB per wave instruction;
//   quantize a block of 16 = two neighbouring lanes (the maximum crosses with one DPP); k_quant_xa16's arithmetic (mxint16_bf16_fast or,
//            at extreme exponents, the element routine) on the lane's 8 values -> 8 bf16 = 16 bytes, stored to the activation image
//            (a KiB per wave instruction) and to the wave's PRIVATE LDS slab [8 rows][1040 B] - no workgroup barrier;
//   multiply 16 steps of v_mfma_f32_16x16x32_bf16: the 8 rows as rows 0-7 of the 16-row operand (one ds_read_b128 per step: row
//            lane & 7, chunk 4 t + lane / 16; the pitch of 1040 B keeps a 16-lane group on 16 distinct bank quads), A^T fragments as
//            ONE coalesced 16-byte load per lane from the fragment-major copy behind the bf16 image (lqer_a_b16_prepare), both halves of
//            the slab's fragments requested in front of the quantizer's arithmetic;
//   finally  the 8 waves' partial tiles through LDS, summed in wave order (fixed: run-to-run bit-stable), A_out exactly as k_xa_reduce4.
// The image is bit for bit k_quant_xa16's; x A is summed in another order (xAq inside the summation-order envelope, tests/_envelope.py).
#include "common.h"

namespace lqer {
namespace a16f {

constexpr int ROWS = 8, WAVES = 8, SLAB = 512, PITCH = 1040;

__host__ inline size_t lds_bytes(int rp) { return (size_t)WAVES * ROWS * PITCH + (size_t)WAVES * ROWS * rp * sizeof(float); }

typedef __attribute__((ext_vector_type(8))) __bf16 bf16v8;

#ifdef LQER_CLOCKPROBE
__device__ unsigned long long* g_a16_stamp_buf = nullptr;  // diagnostic build: shader cycles at the phase boundaries of a wave's FIRST slab
#define A16_STAMP(i) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(cp[i])::"memory")
#else
#define A16_STAMP(i)
#endif

template <int DT, int RT>
__global__ __launch_bounds__(512) void k_act16_fused(const void* __restrict__ x, int64_t M, int64_t K, int64_t ld, QP qx, bf16_t* __restrict__ xq,
                                                      int64_t Kp, const bf16_t* __restrict__ a_frag, QP qa, int L_aout, bf16_t* __restrict__ xaq) {
  constexpr int RP = 16 * RT;
  constexpr int HB = RT == 8 ? 2 : (RT == 4 ? 4 : 8);  // steps per part of a slab (16 / HB parts, two register sets of HB x RT fragments: <= 128 registers)
  constexpr int NPART = 16 / HB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char* const wb = smem + (size_t)wave * ROWS * PITCH;                 // this wave's slab: [ROWS][PITCH]
  float* const red = (float*)(smem + (size_t)WAVES * ROWS * PITCH);             // [WAVES][ROWS][RP] partial tiles
  const int64_t m0 = (int64_t)blockIdx.x * ROWS;
  const int nslab = (int)((Kp + SLAB - 1) / SLAB);
  const int g = lane >> 4;
  const unsigned char* const tok = wb + (lane & 7) * PITCH + 16 * g;  // + 64 t: row lane & 7, chunk 4 t + g
  const u32x4* const fr = (const u32x4*)a_frag + lane;                // block (step, rank tile): fr[(step * RT + tile) * 64]
  f32x4 acc[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifdef LQER_CLOCKPROBE
  unsigned long long cp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  A16_STAMP(0);
#endif

  for (int s = wave; s < nslab; s += WAVES) {
    const int64_t k0 = (int64_t)s * SLAB + 8 * lane;  // this lane's 8 elements of every row
    const bool in_img = k0 < Kp, in_x = k0 < K;       // (K % 8 == 0: a chunk is inside x or outside it as a whole)
    u32x4 raw[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
      raw[r] = (in_x && m0 + r < M) ? *(const u32x4*)((const bf16_t*)x + (m0 + r) * ld + k0) : (u32x4){0, 0, 0, 0};
    // the slab's A^T fragments: 16 steps x RT tiles, requested behind the rows (they land under the quantizer's arithmetic).  Steps past
    // the padded K read block 0 again and multiply zeros (the slab's tail is zero-filled below).
    const int st0 = s * (SLAB / 32), nst = (int)(Kp / 32);
    u32x4 fa[HB][RT], fb[HB][RT];
    auto load_part = [&](u32x4 (&f)[HB][RT], int part) {
#pragma unroll
      for (int i = 0; i < HB; ++i)
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          const int st = st0 + part * HB + i;
          f[i][t] = fr[((int64_t)(st < nst ? st : 0) * RT + t) * 64];
        }
    };
    // (requested in the MIDDLE of the quantizer's loop, not beside the rows: the texture path's queue is shallow - a wave that issues 32
    // fragment requests in a row stands still until the path has taken them, and with it its rows' arithmetic: act8_fused.hip found the
    // same, tools/clock_probe_a8.py)
    // ---- quantize: the lane's 8 values of each row; the block's other half sits in lane ^ 1
    if (s == wave) A16_STAMP(1);  // the rows' requests are out
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      if (r == 1 && s == wave) A16_STAMP(2);  // row 0 landed and quantized
      if (r == 2) {
        asm volatile("" ::: "memory");
        load_part(fa, 0);
        asm volatile("" ::: "memory");
      }
      if (r == 5) {
        asm volatile("" ::: "memory");
        load_part(fb, 1);
        asm volatile("" ::: "memory");
      }
      float v[8];
      const uint32_t wd[4] = {raw[r][0], raw[r][1], raw[r][2], raw[r][3]};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (DT == LQER_F16) {
          typedef __attribute__((ext_vector_type(2))) _Float16 h2;
          const h2 h = __builtin_bit_cast(h2, wd[j]);
          v[2 * j] = (float)h[0], v[2 * j + 1] = (float)h[1];
        } else {
          v[2 * j] = __uint_as_float(wd[j] << 16), v[2 * j + 1] = __uint_as_float(wd[j] & 0xffff0000u);
        }
      }
      float amax = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(v[i]));
      amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
      uint32_t w[4] = {0, 0, 0, 0};
      if (amax > 0.f) {
        const int e = block_exponent_u(amax, qx);
        if (mxint16_fast_ok(e, qx)) {
          mxint16_bf16_fast<DT != LQER_F16, 8>(v, e, qx, w);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const uint32_t lo = exact_bf16_bits(ldexpf(mxint_mantissa(v[2 * i], e, qx), e - qx.mbits));
            const uint32_t hi = exact_bf16_bits(ldexpf(mxint_mantissa(v[2 * i + 1], e, qx), e - qx.mbits));
            w[i] = lo | (hi << 16);
          }
        }
      }
      const u32x4 wv = {w[0], w[1], w[2], w[3]};
      if (in_img) *(u32x4*)(xq + (m0 + r) * Kp + k0) = wv;  // (rows up to the padded M are allocated; rows past M and k past K: zeros)
      *(u32x4*)(wb + r * PITCH + 16 * lane) = wv;
    }
    if (s == wave) A16_STAMP(3);  // all rows quantized, image stores issued
    // (the wave reads back what it wrote itself: LDS executes a wave's accesses in order - no workgroup barrier; the fence keeps the
    // compiler from moving the reads up)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- multiply: 16 steps of 32 k
    auto half = [&](const u32x4 (&f)[HB][RT], int t0) {
      u32x4 tk[HB];
#pragma unroll
      for (int i = 0; i < HB; ++i) tk[i] = *(const u32x4*)(tok + 64 * (t0 + i));
#pragma unroll
      for (int i = 0; i < HB; ++i)
#pragma unroll
        for (int t = 0; t < RT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16v8, tk[i]), __builtin_bit_cast(bf16v8, f[i][t]), acc[t], 0, 0, 0);
    };
    // parts alternate between the two register sets; a set is refilled (part + 2) as soon as its steps have been multiplied
#pragma unroll
    for (int pp = 0; pp < NPART; pp += 2) {
      half(fa, pp * HB);
      if (pp + 2 < NPART) load_part(fa, pp + 2);
      half(fb, (pp + 1) * HB);
      if (pp + 3 < NPART) load_part(fb, pp + 3);
    }
    // (the next slab overwrites the wave's LDS slab: its reads above must have been issued - they have, in program order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#ifdef LQER_CLOCKPROBE
    if (s == wave) {
      asm volatile("" ::"v"(acc[0]));
      A16_STAMP(4);  // the slab's fragments landed, 16 steps multiplied
    }
#endif
  }
  // D layout: column n = lane & 15, rows 4 g + j: token rows 0-7 live in g = 0, 1
  if (g < 2) {
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[(wave * ROWS + 4 * g + j) * RP + 16 * t + (lane & 15)] = acc[t][j];
  }
  __syncthreads();
#ifdef LQER_CLOCKPROBE
  A16_STAMP(5);  // every wave's partial tile is in LDS
  if (g_a16_stamp_buf && lane == 0) {
    unsigned long long* o = g_a16_stamp_buf + ((size_t)blockIdx.x * WAVES + wave) * 8;
    for (int i = 1; i < 6; ++i) o[i] = cp[i] - cp[0];
    unsigned long long rt;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt)::"memory");
    o[0] = cp[0], o[6] = rt;
  }
#endif
  // ---- fixed-order sum of the 8 partial tiles, A_out (k_xa_reduce4's arithmetic), bf16 store
  const int tid = threadIdx.x;
  const bool live = tid < ROWS * RP / 4;
  const int r = tid / (RP / 4), c4 = tid - r * (RP / 4);
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      const float4 v = *(const float4*)(red + (w * ROWS + r) * RP + 4 * c4);
      sum.x += v.x, sum.y += v.y, sum.z += v.z, sum.w += v.w;
    }
  }
  if (tid >= 64 * ((ROWS * RP / 4 + 63) / 64)) return;  // (whole waves only: the shuffles below need their partners)
  float bmax = fmaxf(fmaxf(fabsf(sum.x), fabsf(sum.y)), fmaxf(fabsf(sum.z), fabsf(sum.w)));
  const int G = L_aout / 4;
  for (int d = 1; d < G; d <<= 1) bmax = fmaxf(bmax, __shfl_xor(bmax, d, 64));
  if (!live || m0 + r >= M) return;
  const bool anyb = bmax > 0.f;
  const int eb = anyb ? block_exponent(bmax, qa) : 0;
  const float v[4] = {sum.x, sum.y, sum.z, sum.w};
  uint32_t w2[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float m0v = anyb ? mxint_mantissa(v[2 * i], eb, qa) : 0.f;
    const float m1v = anyb ? mxint_mantissa(v[2 * i + 1], eb, qa) : 0.f;
    w2[i] = exact_bf16_bits(ldexpf(m0v, eb - qa.mbits)) | (exact_bf16_bits(ldexpf(m1v, eb - qa.mbits)) << 16);
  }
  *(uint2*)(xaq + ((m0 + r) * RP + 4 * c4)) = make_uint2(w2[0], w2[1]);
}

// the bf16 image [rp][Kp] (limb 0 of lqer_pack_lowrank's a_t) and, behind it, its fragment-major copy: block (s = 32-k step, t = 16-rank
// tile) at ((s * RT + t) * 64 + lane) * 8 elements; lane (n = lane & 15, g = lane >> 4) holds A^T[16 t + n][32 s + 8 g .. + 8)
__global__ __launch_bounds__(256) void k_a_b16(const bf16_t* __restrict__ limb0, int64_t Kp, int rp, bf16_t* __restrict__ out) {
  const int RT = rp / 16;
  const int64_t n_img = (int64_t)rp * Kp / 8, total = 2 * n_img;  // 16-byte pieces: the image, then the fragments (same count)
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    if (idx < n_img) {
      ((u32x4*)out)[idx] = ((const u32x4*)limb0)[idx];
    } else {
      const int64_t f = idx - n_img;
      const int lane = (int)(f & 63);
      const int64_t blk = f >> 6;
      const int t = (int)(blk % RT);
      const int64_t s = blk / RT;
      ((u32x4*)out)[idx] = *(const u32x4*)(limb0 + (int64_t)(16 * t + (lane & 15)) * Kp + 32 * s + 8 * (lane >> 4));
    }
  }
}

}  // namespace a16f

#ifdef LQER_CLOCKPROBE
extern "C" int lqer_debug_set_a16_stamp_buffer(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(a16f::g_a16_stamp_buf), &p, sizeof(p)); }
#endif

size_t a_b16_image_bytes(int64_t K, int64_t r) { return (size_t)2 * lqer_padded_r(r) * lqer_padded_k(K) * sizeof(bf16_t); }

int a_b16_prepare_dispatch(const void* a_t_limbs, int64_t K, int64_t r, void* out, hipStream_t st) {
  const int64_t rp = lqer_padded_r(r), Kp = lqer_padded_k(K);
  if (rp % 16 != 0 || Kp % 32 != 0) {
    set_error("a_b16_prepare: padded rank %d / padded K %d", (int)rp, (int)Kp);
    return LQER_E_UNSUPPORTED;
  }
  const int64_t total = 2 * rp * Kp / 8;
  a16f::k_a_b16<<<(unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096), 256, 0, st>>>((const bf16_t*)a_t_limbs, Kp, (int)rp, (bf16_t*)out);
  return check_launch("a_b16_prepare");
}

// LQER_E_UNSUPPORTED: not this kernel's case (the caller takes k_quant_xa16 + k_xa_reduce4 on the image's first part)
int act16_fused_dispatch(const void* x, int dtype, int64_t M, int64_t K, int64_t ldx, const QP& qx, bf16_t* xq, const void* a_b16, int64_t r,
                         const QP& qa, bf16_t* xaq, int tuning, hipStream_t st) {
#ifdef LQER_NO_ACT16_FUSED
  return LQER_E_UNSUPPORTED;
#endif
  if (tuning & LQER_TUNE_ACT16_SPLIT) return LQER_E_UNSUPPORTED;
  const int64_t rp = lqer_padded_r(r), Kp = lqer_padded_k(K);
  if (dtype == LQER_F32 || !a_b16 || !xaq || !xq || r <= 0 || M <= 0) return LQER_E_UNSUPPORTED;
  // (rank 128 - RT = 8, eight parts of two steps - was built and measured in round 6: c5 1049 against 1129 with k_quant_xa128 + k_xa_reduce4:
  // every 8-row workgroup streams 1-4 MB of A^T fragments; not instantiated)
  if (!(rp == 16 || rp == 32 || rp == 64)) return LQER_E_UNSUPPORTED;
  if (qx.kind != LQER_Q_MXINT || qx.block != 16 || qx.mbits > 8) return LQER_E_UNSUPPORTED;
  if (((uintptr_t)x % 16) != 0 || ((ldx * 2) % 16) != 0 || K % 16 != 0) return LQER_E_UNSUPPORTED;
  if (qa.kind != LQER_Q_MXINT || qa.mbits > 8) return LQER_E_UNSUPPORTED;
  const int L = (qa.block <= 0 || qa.block >= rp) ? (int)rp : qa.block;
  const int G = L / 4;
  if (rp % L != 0 || L % 4 != 0 || (G & (G - 1)) != 0 || G > 64) return LQER_E_UNSUPPORTED;
  // every workgroup of 8 rows streams the whole A^T image: worth it while the token count is small (see act8_fused.hip)
  if (!(tuning & LQER_TUNE_ACT16_FUSED) && (M > LQER_ACT8_FUSED_MAX_M || M < LQER_ACT8_FUSED_MIN_M)) return LQER_E_UNSUPPORTED;
  const bf16_t* const a_frag = (const bf16_t*)a_b16 + rp * Kp;
  const unsigned grid = (unsigned)((M + a16f::ROWS - 1) / a16f::ROWS);
  const int lds = (int)a16f::lds_bytes((int)rp);
#define A16F_LAUNCH(DTv, RTv)                                                                                              \
  do {                                                                                                                     \
    static LdsLimitOnce once;                                                                                              \
    once.set((const void*)a16f::k_act16_fused<DTv, RTv>, 160 * 1024);                                                       \
    a16f::k_act16_fused<DTv, RTv><<<grid, 512, lds, st>>>(x, M, K, ldx, qx, xq, Kp, a_frag, qa, L, xaq);                     \
  } while (0)
#define A16F_DT(DTv)                     \
  switch (rp / 16) {                     \
    case 1: A16F_LAUNCH(DTv, 1); break;  \
    case 2: A16F_LAUNCH(DTv, 2); break;  \
    default: A16F_LAUNCH(DTv, 4); break; \
  }
  if (dtype == LQER_F16) {
    A16F_DT(LQER_F16)
  } else {
    A16F_DT(LQER_BF16)
  }
#undef A16F_DT
#undef A16F_LAUNCH
  return check_launch("quantize_act_xa (fused block-16 route)");
}

}  // namespace lqer
