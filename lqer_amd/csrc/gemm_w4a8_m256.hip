// Large-M variant of the fused W4 x A8 Linear kernel (gemm_w4a8.hip): one workgroup = one 256(m) x 256(n) tile.
//
//   y[m,n] = sum_k xq[m,k] * Wq[n,k]  +  bq[n]  +  Q_Bout( sum_j xAq[m,j] * B[j,n] )      (linear.py:155-156)
//
// Same arithmetic, operand formats, LDS-DMA ring discipline and transposed MFMA issue as the 128 x 256 kernel; what
// changes is the shape of a wave's work: 8 waves side by side along n, each 256 token rows x 32 columns = 8 x 1
// tiles of v_mfma_f32_32x32x16_bf16 (128 accumulator registers).  One expanded weight fragment now feeds 8 MFMAs
// instead of 4, so per MFMA there is half the expand VALU, half the LDS-DMA issue and half the prologue / store /
// pipeline-fill cost - in the 128-row kernel the LOAD and COMPUTE sections of the ping-pong are both ~600 cycles
// against 512 cycles of MFMA work; here the MFMA work of a section is the long pole.
//
// A 64-deep k-step is processed as two half-steps (k slices 0,1 and 2,3), because the fragments of a whole step
// would not fit beside the accumulators: LOAD(h) reads the 16 activation fragments and the two code words of
// half-step h, issues half of the step's prefetch (k-step + 2 into a 3-slot ring), waits, barrier; COMPUTE(h)
// expands 2 weight fragments and issues 16 MFMAs, barrier.  The two waves of a SIMD run one barrier apart.
// Used for shapes with at least two rounds of 256 x 256 tiles (gemm_dispatch).
#include <type_traits>

#include "common.h"

namespace lqer {

namespace m256 {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int DEPTH = 2;                              // k-steps of prefetch in flight
constexpr int NSLOT = DEPTH + 1;                      // LDS ring slots
constexpr int A_SLOT = BM * BK * 2;                   // 32 KiB  activation tile, bf16
constexpr int R_SLOT = (BN / 16) * LQER_PANEL_BYTES;  // 9216 B  packed weight panels
constexpr int OFF_A = 0;
constexpr int OFF_R = NSLOT * A_SLOT;
constexpr int GEMM_LDS = OFF_R + NSLOT * R_SLOT;      // 125952 B

__device__ __forceinline__ int swz(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ float pair32_max(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

template <int DT, bool LOWRANK, int BOUT>
__global__ __launch_bounds__(512) void k_lqer_gemm_m256(GemmArgs g) {
  constexpr bool XF16 = DT == LQER_F16X;  // (gemm_w4a8.hip)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave;
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

  // ---- staging: per k-step and wave 4 activation pieces (8 rows x 128 B each) + 1 weight piece (wave 0: 2) --------
  const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(g.xq + (int64_t)m0 * g.Kp), 0, 0x7fffffff, 0x00020000);
  int a_voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 32 + i * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    a_voff[i] = (row * g.Kp + chunk * 8) * 2;
  }
  const uint8_t* w_base = g.wp + ((int64_t)(n0 / 16) * nk) * LQER_PANEL_BYTES;
  const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)w_base, 0, 0x7fffffff, 0x00020000);
  auto w_piece_voff = [&](int piece) {
    const int byte = piece * 1024 + lane * 16;
    const int pnl = byte / LQER_PANEL_BYTES;
    return pnl * nk * LQER_PANEL_BYTES + (byte - pnl * LQER_PANEL_BYTES);
  };
  const int w_voff = w_piece_voff(wave), w_voff8 = w_piece_voff(8);
  unsigned char* const a_dst0 = smem + OFF_A + wave * 32 * 128;  // + slot * A_SLOT + piece * 1024
  unsigned char* const w_dst0 = smem + OFF_R + wave * 1024;      // + slot * R_SLOT
  // half 0: activation pieces 0,1 + the wave's weight piece (3 loads); half 1: pieces 2,3 (+ wave 0: piece 8)
  auto issue_half = [&](int kt, int slot, int half) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_void*)(a_dst0 + slot * A_SLOT + (2 * half + i) * 1024), 16,
                                               a_voff[2 * half + i], kt * (BK * 2), 0, 0);
    if (half == 0)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(w_dst0 + slot * R_SLOT), 16, w_voff, kt * LQER_PANEL_BYTES, 0, 0);
    else if (wave == 0)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(smem + OFF_R + 8192 + slot * R_SLOT), 16, w_voff8,
                                               kt * LQER_PANEL_BYTES, 0, 0);
  };
  // fragment read addresses (slot 0): activation row = lane & 31 (+ 32 i: +4096 B, the swizzle term is unchanged),
  // chunk 2 ks + (lane >> 5); weights as in the 128-row kernel, the two words of a half at + 8 half
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem;
  // (one address register per slot and slice: a DS offset field holds 16 bits, slot 2 + row 224 lies beyond it)
  uint32_t fa_addr[NSLOT][4];
#pragma unroll
  for (int sl = 0; sl < NSLOT; ++sl)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fa_addr[sl][ks] = lds0 + OFF_A + sl * A_SLOT + swz(l31, 2 * ks + lh);
  const int nw = wn * 32 + l31;
  const uint32_t fw_addr = lds0 + OFF_R + (nw >> 4) * LQER_PANEL_BYTES + (nw & 15) * 32 + lh * 16;
  const uint32_t fe_addr = lds0 + OFF_R + (nw >> 4) * LQER_PANEL_BYTES + (nw & 15) * 4;  // + 512

  // side path, first half (gemm_w4a8.hip): the loads of pass 0 - the tile's rows of xAq, to be staged through the free
  // ring slot, and this wave's B^T fragments - are issued before the ring prefetch
  constexpr int STG = BM * 8 / 512;  // staged 16-byte chunks per thread and pass
  u32x4 stg[STG];
  bf16x8 sb[3][4];
  const uint32_t stage = lds0 + OFF_A + (NSLOT - 1) * A_SLOT;
  const bf16_t* const bt_row = LOWRANK ? g.bt + (int64_t)(n0 + wn * 32 + l31) * g.rp + 8 * lh : nullptr;
  auto side_fetch = [&](int p0) {
    const int cols = g.rp - p0 < 64 ? g.rp - p0 : 64;  // a multiple of 16
    const int cpr = cols >> 3;                          // 16-byte chunks per row
#pragma unroll
    for (int j = 0; j < STG; ++j) {
      const int c = tid + 512 * j;
      if (c < BM * cpr) {
        const int row = c / cpr, ch = c - row * cpr;
        stg[j] = *(const u32x4*)(g.xaq + (int64_t)(m0 + row) * g.xaq_ld + p0 + 8 * ch);
      }
    }
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        if (l < g.b_limbs && ks * 16 < cols) sb[l][ks] = *(const bf16x8*)(bt_row + (int64_t)l * g.Np * g.rp + p0 + ks * 16);
  };
  if constexpr (LOWRANK) side_fetch(0);

  // prologue loads: steps 0 .. DEPTH-1 (both halves)
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (d < nk) {
      issue_half(d, d, 0);
      issue_half(d, d, 1);
    }

  // ---- low-rank prologue: acc = Q_Bout(xAq @ B) + bias, one 32 x 32 accumulator at a time -------------------------
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
  if constexpr (LOWRANK) {
    // xAq @ B for the whole tile first (8 independent accumulators per fragment), 64 rank entries per pass
    for (int p0 = 0; p0 < g.rp; p0 += 64) {
      const int cols = g.rp - p0 < 64 ? g.rp - p0 : 64;
      const int cpr = cols >> 3;
      if (p0) {
        asm volatile("s_barrier" ::: "memory");  // the previous pass's fragment reads are done
        side_fetch(p0);
      }
#pragma unroll
      for (int j = 0; j < STG; ++j) {
        const int c = tid + 512 * j;
        if (c < BM * cpr) {
          const int row = c / cpr, ch = c - row * cpr;
          asm volatile("ds_write_b128 %0, %1" ::"v"(stage + swz(row, ch)), "v"(stg[j]) : "memory");
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
      for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          if (l < g.b_limbs && ks * 16 < cols) {
            const uint32_t fa = stage + swz(l31, 2 * ks + lh);  // m tile i: + i * 4096 (row + 32 keeps the swizzle)
            bf16x8 xv[8];
            asm volatile(
                "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:4096\n\tds_read_b128 %2, %8 offset:8192\n\t"
                "ds_read_b128 %3, %8 offset:12288\n\tds_read_b128 %4, %8 offset:16384\n\tds_read_b128 %5, %8 offset:20480\n\t"
                "ds_read_b128 %6, %8 offset:24576\n\tds_read_b128 %7, %8 offset:28672\n\ts_waitcnt lgkmcnt(0)"
                : "=&v"(xv[0]), "=&v"(xv[1]), "=&v"(xv[2]), "=&v"(xv[3]), "=&v"(xv[4]), "=&v"(xv[5]), "=&v"(xv[6]), "=&v"(xv[7])
                : "v"(fa)
                : "memory");
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sb[l][ks], xv[i], acc[i], 0, 0, 0);
          }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    f32x16 t = acc[i];
    if constexpr (LOWRANK) {
      if constexpr (BOUT != 0) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          float amax;
          if constexpr (BOUT == 1) {
            amax = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) amax = fmaxf(amax, fabsf(t[8 * b + k]));
            amax = pair32_max(amax);
          } else {
            amax = g.bout_amax ? g.bout_amax[(int64_t)(m0 + i * 32 + l31) * g.bout_nblk + (n0 + wn * 32 + 16 * b) / g.bout_L] : 1.0f;  // (null: integer B_out)
          }
          const int e = block_exponent(amax, g.bout);  // amax = 0: every element takes the pass-through
          float blk[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) blk[k] = t[8 * b + k];
          mxint_requant_fast(blk, e, g.bout);  // two elements per packed fp32 instruction (common.h)
#pragma unroll
          for (int k = 0; k < 8; ++k) t[8 * b + k] = blk[k];
        }
      }
    }
    if (g.bias) {
#pragma unroll
      for (int k = 0; k < 16; ++k) t[k] += g.bias[n0 + wn * 32 + (k & 3) + 8 * (k >> 2) + 4 * lh];
    }
    acc[i] = t;
  }

  // ---- main loop: half-steps h = 2 kt + p ------------------------------------------------------------------------
  //   barrier index      2h-1        2h          2h+1
  //   waves 0-3:    ... | LOAD(h)  | COMPUTE(h) | LOAD(h+1) | ...
  //   waves 4-7:    ... | COMP(h-1)| LOAD(h)    | COMPUTE(h)| ...
  // RAW: a wave ends LOAD(2kt+1) with a counted vmcnt that retires its own loads of step kt+1 (the batch of step kt+2,
  // issued during step kt, stays in flight) and passes a barrier before anyone reads step kt+1.  WAR: slot (kt+2) % 3
  // held step kt-1, last read in LOAD(2kt-1) of waves 4-7, which ends (lgkmcnt(0)) before the barrier that precedes
  // LOAD(2kt) of waves 0-3 and LOAD(2kt) of waves 4-7 alike; the overwriting loads are issued in LOAD(2kt) or later.
  const bool late = wave >= 4;
  {
    // loads(0) landed; loads(1) (5 per wave, wave 0: 6) may stay in flight
    if (nk >= 2)
      asm volatile("s_waitcnt vmcnt(5)\n\ts_barrier" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  }
  if (late) asm volatile("s_barrier" ::: "memory");
  // hand-built buffer descriptors for the in-asm LDS-DMA (wave-uniform words).  Exact ranges: every half-step issues
  // its prefetch, also past the end of K - those lanes read inside the tile's rows or are dropped by the range check
  const unsigned long long a_base = (unsigned long long)(g.xq + (int64_t)m0 * g.Kp);
  const unsigned long long w_base64 = (unsigned long long)w_base;
  const u32x4 a_rs = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a_base),
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a_base >> 32)) & 0xffffu,
                      (uint32_t)(BM * g.Kp * 2), 0x00020000u};
  const u32x4 w_rs = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)w_base64),
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(w_base64 >> 32)) & 0xffffu,
                      (uint32_t)(16 * nk * LQER_PANEL_BYTES), 0x00020000u};
  const uint32_t m0_a = lds0 + OFF_A + wave * 32 * 128;  // + slot * A_SLOT + piece * 1024
  const uint32_t m0_w = lds0 + OFF_R + wave * 1024;      // + slot * R_SLOT
  const uint32_t m0_w8 = lds0 + OFF_R + 8192;            // + slot * R_SLOT (piece 8, wave 0)
  // One half-step.  LOAD is ONE asm statement (a compiler-visible gap between LDS reads and their wait lets hipcc
  // copy registers that have not landed): the 18 LDS reads first, then the LDS-DMA prefetch of step kt+2 - the reads'
  // latency passes while the DMA instructions issue -, then the waits.  The first weight fragment is expanded in the
  // slack that is left before the barrier, so COMPUTE opens with an MFMA.
  auto half_step = [&](int kt, auto slot_c, auto half_c) {
    constexpr int SLOT = decltype(slot_c)::value;
    constexpr int P = decltype(half_c)::value;
    constexpr int slot_new = (SLOT + DEPTH) % NSLOT;
    __builtin_amdgcn_s_setprio(1);
    const int ktn = __builtin_amdgcn_readfirstlane(kt + DEPTH);
    const int a_soff = ktn * (BK * 2), w_soff = ktn * LQER_PANEL_BYTES;
    const uint32_t m0a0 = m0_a + slot_new * A_SLOT + (2 * P) * 1024, m0a1 = m0a0 + 1024;
    const uint32_t m0w = (P == 0 ? m0_w : m0_w8) + slot_new * R_SLOT;
    bf16x8 xa[2][8];  // [slice j of this half][m tile]
    u32x2 wr;
    uint32_t we;
#define LQER_READS                                                                                                  \
      "ds_read_b64 %0, %18 offset:%c23\n\tds_read_b32 %1, %19 offset:%c22+512\n\t"                                  \
      "ds_read_b128 %2, %20\n\tds_read_b128 %3, %20 offset:4096\n\t"                                                \
      "ds_read_b128 %4, %20 offset:8192\n\tds_read_b128 %5, %20 offset:12288\n\t"                                   \
      "ds_read_b128 %6, %20 offset:16384\n\tds_read_b128 %7, %20 offset:20480\n\t"                                  \
      "ds_read_b128 %8, %20 offset:24576\n\tds_read_b128 %9, %20 offset:28672\n\t"                                  \
      "ds_read_b128 %10, %21\n\tds_read_b128 %11, %21 offset:4096\n\t"                                              \
      "ds_read_b128 %12, %21 offset:8192\n\tds_read_b128 %13, %21 offset:12288\n\t"                                 \
      "ds_read_b128 %14, %21 offset:16384\n\tds_read_b128 %15, %21 offset:20480\n\t"                                \
      "ds_read_b128 %16, %21 offset:24576\n\tds_read_b128 %17, %21 offset:28672\n\t"
#define LQER_DMA_A                                                                                                  \
      "s_mov_b32 m0, %29\n\ts_nop 0\n\tbuffer_load_dwordx4 %24, %27, %31 offen lds\n\t"                             \
      "s_mov_b32 m0, %30\n\ts_nop 0\n\tbuffer_load_dwordx4 %25, %27, %31 offen lds\n\t"
#define LQER_OUTS                                                                                                   \
      "=&v"(wr), "=&v"(we), "=&v"(xa[0][0]), "=&v"(xa[0][1]), "=&v"(xa[0][2]), "=&v"(xa[0][3]), "=&v"(xa[0][4]),    \
      "=&v"(xa[0][5]), "=&v"(xa[0][6]), "=&v"(xa[0][7]), "=&v"(xa[1][0]), "=&v"(xa[1][1]), "=&v"(xa[1][2]),         \
      "=&v"(xa[1][3]), "=&v"(xa[1][4]), "=&v"(xa[1][5]), "=&v"(xa[1][6]), "=&v"(xa[1][7])
#define LQER_INS                                                                                                    \
      "v"(fw_addr), "v"(fe_addr), "v"(fa_addr[SLOT][2 * P]), "v"(fa_addr[SLOT][2 * P + 1]), /* 18..21 */            \
      "i"(SLOT * R_SLOT), "i"(SLOT * R_SLOT + 8 * P),                                       /* 22, 23 */            \
      "v"(a_voff[2 * P]), "v"(a_voff[2 * P + 1]), "v"(P == 0 ? w_voff : w_voff8),           /* 24..26 */            \
      "s"(a_rs), "s"(w_rs), "s"(m0a0), "s"(m0a1), "s"(a_soff), "s"(m0w), "s"(w_soff), "s"(wave) /* 27..34 */
    if constexpr (P == 0) {
      // within a step only this wave's LDS reads are waited for (vmcnt(15) never blocks: <= 11 loads are in flight)
      asm volatile(LQER_READS LQER_DMA_A
                   "s_mov_b32 m0, %32\n\ts_nop 0\n\tbuffer_load_dwordx4 %26, %28, %33 offen lds\n\t"
                   "s_waitcnt vmcnt(15) lgkmcnt(0)"
                   : LQER_OUTS
                   : LQER_INS
                   : "memory");
    } else {
      // the next reads are of step kt+1: its loads have landed once only the 5 (wave 0: 6) of step kt+2 are in flight
      asm volatile(LQER_READS LQER_DMA_A
                   "s_cmp_lg_u32 %34, 0\n\ts_cbranch_scc1 1f\n\t"
                   "s_mov_b32 m0, %32\n\ts_nop 0\n\tbuffer_load_dwordx4 %26, %28, %33 offen lds\n\t"
                   "1:\n\ts_waitcnt vmcnt(5) lgkmcnt(0)"
                   : LQER_OUTS
                   : LQER_INS
                   : "memory", "scc");  // (s_cmp inside)
    }
#undef LQER_READS
#undef LQER_DMA_A
#undef LQER_OUTS
#undef LQER_INS
    auto scale_bits = [&](int j) { return ((we >> (8 * (2 * P + j))) & 0xffu) << 23; };  // exponent byte -> 2^(e - mbits)
    bf16x8 wb0 = expand_frag_t<XF16>(wr[0], scale_bits(0));
    asm volatile("s_barrier" : "+v"(wb0)::"memory");
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    // ---- COMPUTE(h)
    {
      const bf16x8 wb1 = expand_frag_t<XF16>(wr[1], scale_bits(1));
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = mfma_32x32x16<XF16>(wb0, xa[0][i], acc[i]);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = mfma_32x32x16<XF16>(wb1, xa[1][i], acc[i]);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  using std::integral_constant;
  for (int kt = 0; kt < nk; kt += NSLOT) {  // unrolled by the ring size: slots are compile-time constants
    half_step(kt, integral_constant<int, 0>{}, integral_constant<int, 0>{});
    half_step(kt, integral_constant<int, 0>{}, integral_constant<int, 1>{});
    if (kt + 1 < nk) {
      half_step(kt + 1, integral_constant<int, 1>{}, integral_constant<int, 0>{});
      half_step(kt + 1, integral_constant<int, 1>{}, integral_constant<int, 1>{});
    }
    if (kt + 2 < nk) {
      half_step(kt + 2, integral_constant<int, 2>{}, integral_constant<int, 0>{});
      half_step(kt + 2, integral_constant<int, 2>{}, integral_constant<int, 1>{});
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the prefetches issued past the end of K have drained
  if (!late) asm volatile("s_barrier" ::: "memory");

  // ---- store ----------------------------------------------------------------------------------
  const bool aligned16 = (((uintptr_t)g.y) & 15) == 0;
  const int nb = n0 + wn * 32;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + i * 32 + l31;
    if constexpr (DT == LQER_F32) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = nb + 8 * q + 4 * lh;
        if (m < g.M) {
          float* dst = (float*)g.y + (int64_t)m * g.ldy + n;
          if (n + 3 < g.N && (g.ldy & 3) == 0 && aligned16) {
            *(float4*)dst = make_float4(acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]);
          } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
              if (n + t < g.N) dst[t] = acc[i][4 * q + t];
          }
        }
      }
    } else {
      uint32_t pk[4][2];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float v0 = acc[i][4 * q + 2 * h], v1 = acc[i][4 * q + 2 * h + 1];
          if constexpr (DT == LQER_F16 || DT == LQER_F16X) {
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

template <int DT>
static int launch(GemmArgs g, bool lowrank, int bout, hipStream_t st) {
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = g.Np / BN;
  const unsigned grid = (unsigned)(g.tiles_m * g.tiles_n);
#define LQER_M256_LAUNCH(LR, BO)                                                                                    \
  do {                                                                                                              \
    static LdsLimitOnce lds_once;                                                                                   \
    lds_once.set((const void*)k_lqer_gemm_m256<DT, LR, BO>, GEMM_LDS);                                              \
    k_lqer_gemm_m256<DT, LR, BO><<<grid, 512, GEMM_LDS, st>>>(g);                                                   \
  } while (0)
  if (!lowrank)
    LQER_M256_LAUNCH(false, 0);
  else if (bout == 1)
    LQER_M256_LAUNCH(true, 1);
  else if (bout == 2)
    LQER_M256_LAUNCH(true, 2);
  else
    LQER_M256_LAUNCH(true, 0);
#undef LQER_M256_LAUNCH
  return check_launch("lqer_gemm_m256");
}

}  // namespace m256

// A 256 x 256 tile costs about 1.9 tiles of 128 x 256 (one weight expand per 8 MFMAs instead of 4, half the per-tile
// fixed work), but both kernels run in whole rounds of one tile per CU: take the large tiles only when they still need
// less time after rounding up - e.g. 16384 x 5120: 5 rounds against 10, 4096 x 4096: 1 against 2, but 2048 x 11008:
// 2 (344 tiles) against 3 (688) keeps the small tiles (measured: 157 vs 178 us).
#ifndef LQER_M256_MIN_M
#define LQER_M256_MIN_M 512
#endif
bool m256_eligible(const GemmArgs& g) {
  constexpr int64_t CUS = 256;
  const int64_t t256 = (int64_t)((g.M + m256::BM - 1) / m256::BM) * (g.Np / m256::BN);
  const int64_t t128 = (int64_t)((g.M + 127) / 128) * (g.Np / m256::BN);
  const int64_t r256 = (t256 + CUS - 1) / CUS, r128 = (t128 + CUS - 1) / CUS;
  return g.M >= LQER_M256_MIN_M && r256 * 19 < r128 * 10;
}

int m256_dispatch(const GemmArgs& g, int dtype, bool lowrank, int bout, hipStream_t st) {
  switch (dtype) {
    case LQER_F32: return m256::launch<LQER_F32>(g, lowrank, bout, st);
    case LQER_F16: return g.x_f16 ? m256::launch<LQER_F16X>(g, lowrank, bout, st) : m256::launch<LQER_F16>(g, lowrank, bout, st);
    case LQER_BF16: return m256::launch<LQER_BF16>(g, lowrank, bout, st);
  }
  set_error("unknown dtype %d", dtype);
  return LQER_E_INVALID;
}

}  // namespace lqer
