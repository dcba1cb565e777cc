// The fused W4 x A8 Linear kernel:
//
//   y[m,n] = sum_k xq[m,k] * Wq[n,k]  +  bq[n]  +  Q_Bout( sum_j xAq[m,j] * B[j,n] )
//
// replaces reference quantized_layers/linear.py:155-156 (torch.matmul(xA, B), B_out_quantizer,
// F.linear, add).  One workgroup = one 128(m) x 256(n) output tile (wide in n: a weight costs 0.56 B
// of L2->LDS traffic per element, an activation 2 B), 8 waves as 2(m) x 4(n), each
// wave 64 x 64 = 2 x 2 tiles of v_mfma_f32_32x32x16_bf16 (bf16 holds every MXINT value
// m * 2^e, |m| < 256, exactly; fp32 accumulation - SURVEY.md §7 H1 strategy S1).
//
// The MFMA is issued "transposed" (A operand = weight rows, B operand = token rows), so a lane owns
// one token row m and, per accumulator quad, 4 consecutive output columns n:
//   acc reg k of a 32x32 tile:  m = lane & 31,  n = (k & 3) + 8 (k >> 2) + 4 (lane >> 5).
// A B_out block (16 consecutive n of one token) is then 8 registers of lane l and 8 of lane l^32:
// 7 in-lane max + one v_permlane32_swap; and the output leaves as 16-byte stores.
//
//  * prologue: the rank-r product xAq @ B runs on the MFMA straight from global memory, is
//    re-quantized in registers, the bias is added, and the result is the INITIAL accumulator of the
//    main loop - there is no epilogue pass over the tile.
//  * main loop, BK = 64, one barrier per k-step, every global access a global_load_lds (16 B/lane)
//    so that loads stay in flight across barriers behind a COUNTED s_waitcnt vmcnt:
//      - activation tile: 3-slot LDS ring, loaded two k-steps ahead (XOR swizzle on the source address)
//      - packed weight panel (4-bit codes + block exponents, 576 B per wave): 3-slot ring, three steps
//        ahead; one step ahead of its use the owning wave expands it to bf16 (VALU) into a 2-slot tile.
//  * tiles are numbered so that each of the 8 XCDs works on a contiguous run of tiles (same token
//    rows -> the activation slab stays in that XCD's L2).
#include "common.h"

namespace lqer {

constexpr int BM = 128, BN = 256, BK = 64;
constexpr int A_SLOT = BM * BK * 2;                   // 16 KiB
constexpr int W_SLOT = BN * BK * 2;                   // 32 KiB (expanded bf16)
constexpr int R_SLOT = (BN / 16) * LQER_PANEL_BYTES;  // 9216 B (raw panels)
constexpr int OFF_A = 0;
constexpr int OFF_W = 3 * A_SLOT;
constexpr int OFF_R = OFF_W + 2 * W_SLOT;
constexpr int GEMM_LDS = OFF_R + 3 * R_SLOT;  // 142336 B

// byte offset of 16-byte chunk `c` (8 bf16 along k) of tile row `r`; rows are 128 B.
// chunk ^ ((row >> 1) & 7): the 16 lanes of a ds_read_b128 group then hit 16 distinct 16-B slots.
__device__ __forceinline__ int swz(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

// Expand 16 sign-magnitude 4-bit codes (lo = k 0..7, hi = k 8..15; nibble order see pack.hip) times
// 2^(e - mbits) to bf16: magnitude -> fp8 (e4m3) byte through a v_perm_b32 table, sign bit OR-ed in,
// then v_cvt_scalef32_pk_bf16_fp8 converts two elements per instruction and applies the block scale.
// ~1.5 VALU ops per weight; every step is exact (integers 0..7 and powers of two).
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ void expand16(uint32_t lo, uint32_t hi, int e, int mbits, uint32_t (&w)[8]) {
  int ef = e - mbits + 127;
  ef = ef < 1 ? 1 : ef;  // codes of such a block are all zero (|w| <= 1e-8 is flushed)
  const float scale = __uint_as_float((uint32_t)ef << 23);
  constexpr uint32_t LUT_LO = 0x44403800u, LUT_HI = 0x4E4C4A48u;  // e4m3 bytes of 0,1,2,3 | 4,5,6,7
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint32_t word = h ? hi : lo;
    const uint32_t me = word & 0x07070707u, mo = (word >> 4) & 0x07070707u;   // k 0..3 | k 4..7
    uint32_t fe = __builtin_amdgcn_perm(LUT_HI, LUT_LO, me);
    uint32_t fo = __builtin_amdgcn_perm(LUT_HI, LUT_LO, mo);
    fe |= (word << 4) & 0x80808080u;
    fo |= word & 0x80808080u;
    w[4 * h + 0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(fe, scale, false));
    w[4 * h + 1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(fe, scale, true));
    w[4 * h + 2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(fo, scale, false));
    w[4 * h + 3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(fo, scale, true));
  }
}

// max over lanes l and l^32
__device__ __forceinline__ float pair32_max(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}


// ---- LDS access in inline asm -------------------------------------------------------------------
// hipcc's waitcnt pass treats every global_load_lds in flight as a pending LDS write and puts
// s_waitcnt vmcnt(0) in front of any LDS access it can see, which would drain the prefetch ring every
// k-step.  Inside the main loop all LDS reads/writes are therefore asm statements the pass cannot
// see; completion is waited for explicitly (cdna_hip_programming.md §5.7 form (ii): the wait
// statement names every destination register "+v", so no consumer can be scheduled above it).
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

template <int OFF>
__device__ __forceinline__ bf16x8 lds_read128(uint32_t addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
  return v;
}
__device__ __forceinline__ u32x2 lds_read64(uint32_t addr) {
  u32x2 v;
  asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
__device__ __forceinline__ int lds_read_i8_512(uint32_t addr) {
  int v;
  asm volatile("ds_read_i8 %0, %1 offset:512" : "=v"(v) : "v"(addr));
  return v;
}
__device__ __forceinline__ void lds_write128(uint32_t addr, u32x4 v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_wait(bf16x8& a, bf16x8& b, bf16x8& c, bf16x8& d) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void lds_wait(u32x2& a, int& b) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b));
}

#ifdef LQER_STAMPS
// Diagnostic build only: per-section cycle sums (s_memtime) of the main loop, written to a buffer that
// nothing else reads.  Never quote this build's run time (the stamps serialise the sections).
__device__ unsigned long long* g_stamp_buf = nullptr;
#define STAMP(i)                                                                        \
  do {                                                                                  \
    unsigned long long t_;                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");         \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    st_sum[i] += t_ - st_prev;                                                          \
    st_prev = t_;                                                                       \
  } while (0)
#else
#define STAMP(i)
#endif

template <int DT, bool LOWRANK, bool BOUT16>
__global__ __launch_bounds__(512) void k_lqer_gemm(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
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
  const int nk = g.Kp / BK;

  // ---- staging ------------------------------------------------------------------------------
  // activation: wave w stages tile rows [16w, 16w+16): 2 x LDS-DMA of 8 rows x 128 B.  Buffer
  // addressing: wave-uniform descriptor + per-lane byte offset fixed for the whole kernel + the
  // k-step as scalar offset, so a prefetch costs no vector ALU work.
  const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(g.xq + (int64_t)m0 * g.Kp), 0, 0x7fffffff, 0x00020000);
  int a_voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wave * 16 + i * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    a_voff[i] = (row * g.Kp + chunk * 8) * 2;
  }
  auto issue_a = [&](int kt, int slot) {
#ifdef LQER_ABL_NO_A_LOAD
    return;
#endif
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      unsigned char* dst = smem + OFF_A + slot * A_SLOT + (wave * 16 + i * 8) * 128;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_void*)dst, 16, a_voff[i], kt * (BK * 2), 0, 0);
    }
  };
  // weights: wave w owns panels 2w, 2w+1 of the tile (rows [32w, 32w+32); a panel = 16 rows x 64 k =
  // 576 B = lanes 0..35 x 16 B): it fetches them, expands them to bf16 and publishes them for all waves.
  const uint8_t* w_base = g.wp + ((int64_t)(n0 / 16 + 2 * wave) * nk) * LQER_PANEL_BYTES;
  const auto w_rsrc0 = __builtin_amdgcn_make_buffer_rsrc((void*)w_base, 0, 0x7fffffff, 0x00020000);
  const auto w_rsrc1 = __builtin_amdgcn_make_buffer_rsrc((void*)(w_base + (int64_t)nk * LQER_PANEL_BYTES), 0, 0x7fffffff, 0x00020000);
  const int w_voff = (lane < 36 ? lane : 35) * 16;
  auto issue_w = [&](int kt, int slot) {
#ifdef LQER_ABL_NO_W_LOAD
    return;
#endif
    unsigned char* dst = smem + OFF_R + slot * R_SLOT + 2 * wave * LQER_PANEL_BYTES;
    if (lane < 36) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc0, (lds_void*)dst, 16, w_voff, kt * LQER_PANEL_BYTES, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc1, (lds_void*)(dst + LQER_PANEL_BYTES), 16, w_voff, kt * LQER_PANEL_BYTES, 0, 0);
    }
  };
  // raw panel j of this wave in ring slot rs: lane -> 16 codes (8 B) + the exponent of their 16-k block
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem;
  const uint32_t raw_addr = lds0 + OFF_R + 2 * wave * LQER_PANEL_BYTES + lane * 8;  // + j * 576 + rs * R_SLOT
  const uint32_t rawe_addr = lds0 + OFF_R + 2 * wave * LQER_PANEL_BYTES + lane;     // + 512 + ...
  // expanded tile rows [32w + 16j, +16): lane -> (row = lane / 4, 16-k segment = lane % 4) = chunks 2seg, 2seg+1
  uint32_t wexp_addr[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    wexp_addr[j][0] = lds0 + OFF_W + swz(wave * 32 + j * 16 + (lane >> 2), 2 * (lane & 3));  // + ws * W_SLOT
    wexp_addr[j][1] = lds0 + OFF_W + swz(wave * 32 + j * 16 + (lane >> 2), 2 * (lane & 3) + 1);
  }
  auto expand_regs = [&](u32x2 codes, int e, u32x4& lo, u32x4& hi) {
    uint32_t w[8];
#ifdef LQER_ABL_NO_EXPAND
    for (int i = 0; i < 8; ++i) w[i] = codes[i & 1] + i + e;
#else
    expand16(codes[0], codes[1], e, g.w_mbits, w);
#endif
    lo = (u32x4){w[0], w[1], w[2], w[3]};
    hi = (u32x4){w[4], w[5], w[6], w[7]};
  };
  auto store_expanded = [&](const u32x4& lo, const u32x4& hi, int j, int ws) {
    lds_write128(wexp_addr[j][0] + ws * W_SLOT, lo);
    lds_write128(wexp_addr[j][1] + ws * W_SLOT, hi);
  };
  // fragment read addresses (slot 0): row = wave tile row + lane & 31; chunk 2 ks + (lane >> 5), swizzled
  uint32_t fa_addr[4], fw_addr[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    fa_addr[ks] = lds0 + OFF_A + swz(wm * 64 + l31, 2 * ks + lh);  // second m tile: +32 rows = +4096 B
    fw_addr[ks] = lds0 + OFF_W + swz(wn * 64 + l31, 2 * ks + lh);  // (row + 32 keeps (row >> 1) & 7)
  }

  // prologue loads: A(0), A(1) and raw W(0..2), every wave its own rows / panel
  const bool late = wave >= 4;  // waves 4-7 run one barrier behind waves 0-3 (see main loop)
  issue_a(0, 0);
  if (nk > 1) issue_a(1, 1);
  issue_w(0, 0);
  if (nk > 1) issue_w(1, 1);
  if (nk > 2) issue_w(2, 2);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;

  // ---- low-rank prologue: acc = Q_Bout(xAq @ B) + bias ----------------------------------------
  if constexpr (LOWRANK) {
    for (int l = 0; l < g.b_limbs; ++l) {
      for (int ks = 0; ks < g.rp / 16; ++ks) {
        bf16x8 xa[2], bb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          xa[i] = *(const bf16x8*)(g.xaq + (int64_t)(m0 + wm * 64 + i * 32 + l31) * g.rp + ks * 16 + 8 * lh);
          bb[i] = *(const bf16x8*)(g.bt + ((int64_t)l * g.Np + n0 + wn * 64 + i * 32 + l31) * g.rp + ks * 16 + 8 * lh);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bb[j], xa[i], acc[i][j], 0, 0, 0);
      }
    }
    if constexpr (BOUT16) {
      const int mb = g.bout.mbits;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            float amax = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) amax = fmaxf(amax, fabsf(acc[i][j][8 * b + k]));
            amax = pair32_max(amax);
            const int e = block_exponent(amax, g.bout);  // amax = 0: every element takes the pass-through
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const float t = acc[i][j][8 * b + k];
              const float m = fminf(rintf(ldexpf(fabsf(t) + 1e-9f, mb - e)), g.bout.mmax);
              const float q = copysignf(ldexpf(m, e - mb), t);
              acc[i][j][8 * b + k] = fabsf(t) <= 1e-8f ? t : q;
            }
          }
    }
  }
  if (g.bias) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const float bv = g.bias[n0 + wn * 64 + j * 32 + (k & 3) + 8 * (k >> 2) + 4 * lh];
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][j][k] += bv;
      }
  }

  // W(0) -> expanded slot 0 (own panels: only this wave's vmcnt matters), publish; W(1) expanded into
  // registers for the first LOAD section
  u32x4 wx[2][2] = {{{0, 0, 0, 0}, {0, 0, 0, 0}}, {{0, 0, 0, 0}, {0, 0, 0, 0}}};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    u32x2 c0 = lds_read64(raw_addr + j * LQER_PANEL_BYTES), c1 = lds_read64(raw_addr + j * LQER_PANEL_BYTES + R_SLOT);
    int e0 = lds_read_i8_512(rawe_addr + j * LQER_PANEL_BYTES), e1 = lds_read_i8_512(rawe_addr + j * LQER_PANEL_BYTES + R_SLOT);
    lds_wait(c0, e0);
    lds_wait(c1, e1);
    expand_regs(c0, e0, wx[j][0], wx[j][1]);
    store_expanded(wx[j][0], wx[j][1], j, 0);
    if (nk > 1) expand_regs(c1, e1, wx[j][0], wx[j][1]);
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  // ---- main loop ------------------------------------------------------------------------------
  // Ping-pong: the two waves that share a SIMD (w and w+4) run one barrier apart.  A k-step is a LOAD
  // section and a COMPUTE section, each closed by a barrier; while waves 0-3 compute, waves 4-7 load,
  // and vice versa, so the matrix pipe always has a wave feeding it.
  //   barrier index      2kt-1        2kt          2kt+1         2kt+2
  //   waves 0-3:    ... | LOAD(kt)  | COMPUTE(kt) | LOAD(kt+1)  | ...
  //   waves 4-7:    ... | COMP(kt-1)| LOAD(kt)    | COMPUTE(kt) | ...
  // LOAD(kt):    write the expanded panel W(kt+1) held in registers (4 ds_write_b128, first, so that their
  //              latency hides under what follows), read the 16 operand fragments of step kt, issue the
  //              prefetch A(kt+2) x2 + raw W(kt+3) x2, wait: LDS done, all but these 4 loads done; barrier.
  // COMPUTE(kt): read the raw panel W(kt+2), 16 MFMAs, the expand of W(kt+2) (12 VALU + 8 converts) in
  //              their shadow, result kept in 8 registers for LOAD(kt+1); barrier.
  // Hazards: the slot of expanded W(kt+1) was last read in LOAD(kt-1) (both groups are past it: two
  // barriers earlier for the other group) and is first read in LOAD(kt+1), after barrier 2kt+1 which
  // every writer reaches with lgkmcnt(0).  A(kt+2) / raw W(kt+3) go to the ring slots of A(kt-1) /
  // W(kt), last read by this wave's own group in LOAD(kt-1) / by this wave in COMPUTE(kt-2).  The counted
  // vmcnt at the end of LOAD(kt) retires what LOAD(kt-1) issued: A(kt+1) for LOAD(kt+1) (activation rows
  // never cross the two groups, and the group passes a barrier first) and raw W(kt+2) for COMPUTE(kt).
  if (late) asm volatile("s_barrier" ::: "memory");
  int sa = 0, sr = 2, sw = 0;  // A(kt) in ring slot sa, raw W(kt+2) in slot sr, expanded W(kt) in slot sw
#ifdef LQER_STAMPS
  unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif
  for (int kt = 0; kt < nk; ++kt) {
    const int sa2 = sa == 0 ? 2 : sa - 1;  // (sa + 2) % 3: slot of A(kt+2)
    const int sr1 = sr == 2 ? 0 : sr + 1;  // slot of raw W(kt+3)
    const bool full = kt + 3 < nk;
    const uint32_t oa = sa * A_SLOT, ow = sw * W_SLOT;
    STAMP(7);
    // ---- LOAD(kt)
#ifndef LQER_ABL_DMA_ONLY
    if (kt + 1 < nk) {
      store_expanded(wx[0][0], wx[0][1], 0, sw ^ 1);
      store_expanded(wx[1][0], wx[1][1], 1, sw ^ 1);
    }
#endif
    bf16x8 xa[4][2], wb[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#ifdef LQER_ABL_DMA_ONLY
      xa[ks][0] = xa[ks][1] = wb[ks][0] = wb[ks][1] = (bf16x8){1, 2, 3, 4, 5, 6, 7, 8};
#else
      xa[ks][0] = lds_read128<0>(fa_addr[ks] + oa), xa[ks][1] = lds_read128<4096>(fa_addr[ks] + oa);
      wb[ks][0] = lds_read128<0>(fw_addr[ks] + ow), wb[ks][1] = lds_read128<4096>(fw_addr[ks] + ow);
#endif
    }
    if (kt + 2 < nk) issue_a(kt + 2, sa2);
    if (full) issue_w(kt + 3, sr1);
#define LQER_FRAGS                                                                                          \
  "+v"(xa[0][0]), "+v"(xa[0][1]), "+v"(xa[1][0]), "+v"(xa[1][1]), "+v"(xa[2][0]), "+v"(xa[2][1]), "+v"(xa[3][0]), \
      "+v"(xa[3][1]), "+v"(wb[0][0]), "+v"(wb[0][1]), "+v"(wb[1][0]), "+v"(wb[1][1]), "+v"(wb[2][0]), "+v"(wb[2][1]), \
      "+v"(wb[3][0]), "+v"(wb[3][1])
    STAMP(0);  // LOAD section issue (stamped builds also wait for the LDS traffic here)
    if (full)
      asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" : LQER_FRAGS::"memory");
    else
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" : LQER_FRAGS::"memory");
#undef LQER_FRAGS
    __builtin_amdgcn_sched_barrier(0);
    STAMP(4);  // waits + barrier after LOAD
    // ---- COMPUTE(kt)
    u32x2 wc0, wc1;
    int we0, we1;
#ifdef LQER_ABL_DMA_ONLY
    const bool do_expand = false;
#else
    const bool do_expand = kt + 2 < nk;
#endif
    if (do_expand) {
      wc0 = lds_read64(raw_addr + sr * R_SLOT), wc1 = lds_read64(raw_addr + LQER_PANEL_BYTES + sr * R_SLOT);
      we0 = lds_read_i8_512(rawe_addr + sr * R_SLOT), we1 = lds_read_i8_512(rawe_addr + LQER_PANEL_BYTES + sr * R_SLOT);
    }
#if defined(LQER_ABL_NO_MFMA) || defined(LQER_ABL_DMA_ONLY)
#define LQER_MFMA4(ks) asm volatile("" ::"v"(wb[ks][0]), "v"(wb[ks][1]), "v"(xa[ks][0]), "v"(xa[ks][1]))
#else
#define LQER_MFMA4(ks)                                                                                   \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[ks][0], xa[ks][0], acc[0][0], 0, 0, 0);        \
  acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[ks][1], xa[ks][0], acc[0][1], 0, 0, 0);        \
  acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[ks][0], xa[ks][1], acc[1][0], 0, 0, 0);        \
  acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[ks][1], xa[ks][1], acc[1][1], 0, 0, 0)
#endif
    LQER_MFMA4(0);
    if (do_expand) {
      lds_wait(wc0, we0);
      lds_wait(wc1, we1);
      expand_regs(wc0, we0, wx[0][0], wx[0][1]);
    }
    LQER_MFMA4(1);
    LQER_MFMA4(2);
    if (do_expand) expand_regs(wc1, we1, wx[1][0], wx[1][1]);
    LQER_MFMA4(3);
#undef LQER_MFMA4
    __builtin_amdgcn_sched_barrier(0);
    STAMP(5);  // COMPUTE section issue
    asm volatile("s_barrier" : "+v"(wx[0][0]), "+v"(wx[0][1]), "+v"(wx[1][0]), "+v"(wx[1][1])::"memory");
    __builtin_amdgcn_sched_barrier(0);
    STAMP(6);  // barrier after COMPUTE
    sa = sa == 2 ? 0 : sa + 1;
    sr = sr == 2 ? 0 : sr + 1;
    sw ^= 1;
  }
  if (!late) asm volatile("s_barrier" ::: "memory");
#ifdef LQER_STAMPS
  if (g_stamp_buf && lane == 0)
    for (int i = 0; i < 8; ++i) g_stamp_buf[(blockIdx.x * 8 + wave) * 8 + i] = st_sum[i];
#endif

  // ---- store ----------------------------------------------------------------------------------
  // per 32x32 tile and quad q: regs 4q..4q+3 = columns n = nb + 8q + 4 lh + (0..3) of token row m
  const bool aligned16 = (((uintptr_t)g.y) & 15) == 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 64 + i * 32 + l31;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int nb = n0 + wn * 64 + j * 32;
      if constexpr (DT == LQER_F32) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = nb + 8 * q + 4 * lh;
          if (m < g.M) {
            float* dst = (float*)g.y + (int64_t)m * g.ldy + n;
            if (n + 3 < g.N && (g.ldy & 3) == 0 && aligned16) {
              *(float4*)dst = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
            } else {
#pragma unroll
              for (int t = 0; t < 4; ++t)
                if (n + t < g.N) dst[t] = acc[i][j][4 * q + t];
            }
          }
        }
      } else {
        // 16-bit outputs: pack 4 columns into 8 B, then merge quads (q, q+1) of lanes l / l^32 into one
        // 16-B store: lanes 0-31 get columns 16p .. 16p+7, lanes 32-63 columns 16p+8 .. 16p+15
        uint32_t pk[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const float v0 = acc[i][j][4 * q + 2 * h], v1 = acc[i][j][4 * q + 2 * h + 1];
            if constexpr (DT == LQER_F16) {
              typedef __attribute__((ext_vector_type(2))) _Float16 h2;
              h2 hv = {(_Float16)v0, (_Float16)v1};
              pk[q][h] = __builtin_bit_cast(uint32_t, hv);
            } else {
              pk[q][h] = (uint32_t)f32_to_bf16_rne(v0) | ((uint32_t)f32_to_bf16_rne(v1) << 16);
            }
          }
        const bool wide = (g.ldy & 7) == 0 && nb + 32 <= g.N && aligned16;  // wave-uniform
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const uint32_t a0 = pk[2 * p][0], a1 = pk[2 * p][1], b0 = pk[2 * p + 1][0], b1 = pk[2 * p + 1][1];
          if (wide) {
            auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
            auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
            // lanes 0-31: {own quad 2p, upper lane's quad 2p}; lanes 32-63: {lower lane's quad 2p+1, own quad 2p+1}
            if (m < g.M) {
              bf16_t* dst = (bf16_t*)g.y + (int64_t)m * g.ldy + nb + 16 * p + 8 * lh;
              *(uint4*)dst = make_uint4(r0[0], r1[0], r0[1], r1[1]);
            }
          } else if (m < g.M) {
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
              const int n = nb + 8 * (2 * p + qq) + 4 * lh;
              const uint32_t lo = qq ? b0 : a0, hi = qq ? b1 : a1;
              bf16_t* dst = (bf16_t*)g.y + (int64_t)m * g.ldy + n;
              if (n < g.N) dst[0] = (bf16_t)(lo & 0xffff);
              if (n + 1 < g.N) dst[1] = (bf16_t)(lo >> 16);
              if (n + 2 < g.N) dst[2] = (bf16_t)(hi & 0xffff);
              if (n + 3 < g.N) dst[3] = (bf16_t)(hi >> 16);
            }
          }
        }
      }
    }
  }
}

template <int DT>
static int launch_gemm(const GemmArgs& g, bool lowrank, bool bout16, hipStream_t st) {
  const unsigned grid = (unsigned)(g.tiles_m * g.tiles_n);
  // raising the dynamic-LDS limit is idempotent; the flag only saves the call on later launches
#define LQER_GEMM_LAUNCH(LR, BO)                                                                                \
  do {                                                                                                          \
    static bool attr_done = false;                                                                              \
    if (!attr_done) {                                                                                           \
      (void)hipFuncSetAttribute((const void*)k_lqer_gemm<DT, LR, BO>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                GEMM_LDS);                                                                      \
      attr_done = true;                                                                                         \
    }                                                                                                           \
    k_lqer_gemm<DT, LR, BO><<<grid, 512, GEMM_LDS, st>>>(g);                                                    \
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

#ifdef LQER_STAMPS
extern "C" int lqer_debug_set_stamp_buffer(void* p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &p, sizeof(p));
}
#endif

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
