// The int8 route on v_mfma_i32_16x16x64_i8 (gemm_w4a8_i8.hip has the arithmetic, the images and the exactness argument;
// this file is the same kernel re-mapped onto the 16 x 16 matrix instruction):
//
//   y[m,n] = 2^(ex[m]-7) * sum_g 2^(ew[n,g]-3) * ( sum_{k in g} cx[m,k] * cw[n,k] )  +  bq[n]  +  Q_Bout( xAq @ B )
//
// Why: the 16x16x64 and 32x32x32 forms have the same per-clock rate, but under sustained load the chip holds 2.0-2.1 GHz on
// the 16x16 form against 1.7 GHz on the 32x32 form (tools/ubench/mfma_shape.hip: 4.07 vs 3.47 POP/s bare, 3.25 vs 2.97 with
// this kernel's expand + shift-fold beside the MFMAs), and this kernel's vector work per MFMA is small enough to keep it
// (3 instructions per 8 weights, one v_lshl_add_u32 per accumulator element and 128-k group) - unlike the bf16 kernels'
// 14-instruction expand (tools/experiments/README.md).
//
// Fragment maps (tokens as rows = A operand, weight rows as columns = B operand, as in gemm_w4a8_i8.hip):
//   A: lane l supplies token row l & 15 of a 16-row tile, k = 16 (l >> 4) .. + 16 of a 64-deep half: one ds_read_b128 of the
//      int8 activation tile (rows of 128 B, the same swizzle);
//   B: lane l supplies weight row n = l & 15 of a 16-column tile, the same 16 k = two words of two's-complement nibbles; the
//      weight image (k_i8_codes) keeps the lane's four words of a 128-k step in ONE 16-byte chunk;
//   D: lane l holds column l & 15, token rows 4 (l >> 4) + j, j = 0..3: the per-column shift and scale stay one register per
//      lane and column tile, per-row constants are 16-byte table reads.
// Tile, ring, LOAD / COMPUTE ping-pong, the epilogue's structure: gemm_w4a8_i8.hip / gemm_w4a8_m256.hip.
#include <type_traits>

#include "common.h"

namespace lqer {
namespace i8t {

constexpr int BM = 256, BN = 256;
constexpr int DEPTH = 2, NSLOT = DEPTH + 1;
constexpr int A_SLOT = BM * I8_BK;   // 32 KiB  int8 activation tile
constexpr int W_SLOT = I8_WBLOCK;    // 16640 B nibbles + shift bytes
constexpr int OFF_A = 0;
constexpr int OFF_W = NSLOT * A_SLOT;
constexpr int GEMM_LDS = OFF_W + NSLOT * W_SLOT;  // 148224 B
constexpr int EP_STAGE = 0;              // xAq tile: up to two 64-column panels of 32 KiB (filled by LDS-DMA)
constexpr int EP_OUT = 65536;            // per wave: 64 rows x 80 B (32 fp16 columns + pad)
constexpr int EP_OUT_WAVE = 64 * 80;
static_assert(EP_OUT + 8 * EP_OUT_WAVE <= GEMM_LDS, "epilogue regions exceed the ring");
constexpr int EP_TAB = GEMM_LDS;         // fp32 [256] x 3: x row scales, B_out 2^(mbits-e[m]), 2^(e[m]-mbits)
constexpr int KERNEL_LDS = GEMM_LDS + 3 * 1024;

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((ext_vector_type(2))) float f2;

__device__ __forceinline__ int swz(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

// BOUT: 0 pass-through, 2 one block per row (exponent from the pre-pass k_bout_amax16 below).  SHIFT: per-group shifts present.
template <int DT, bool LOWRANK, int BOUT, bool SHIFT>
__global__ __launch_bounds__(512) void k_lqer_gemm_i8t(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;

  const int nt = g.tiles_m * g.tiles_n;
  int tile;
  {
    const int b = blockIdx.x, xcd = b & 7, q8 = nt >> 3, r8 = nt & 7;
    tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
  }
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int Kp8 = g.Kp;  // (the int8 image's row stride)
  const int nk = Kp8 / I8_BK;
  const uint8_t* const xq8 = (const uint8_t*)g.xq;

  // ---- staging addresses (identical to gemm_w4a8_i8.hip: the LDS images of the activations are the same)
  int a_voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 32 + i * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    a_voff[i] = row * Kp8 + chunk * 16;
  }
  const int w_voff0 = (2 * wave) * 1024 + lane * 16, w_voff1 = w_voff0 + 1024, s_voff = 256 * 64 + lane * 4;
  const uint8_t* const a_base = xq8 + (int64_t)m0 * Kp8;
  const uint8_t* const w_base = g.w8 + (size_t)tn * nk * I8_WBLOCK;
  const unsigned long long a_base64 = (unsigned long long)a_base, w_base64 = (unsigned long long)w_base;
  const u32x4 a_rs = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a_base64),
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a_base64 >> 32)) & 0xffffu, (uint32_t)(BM * Kp8),
                      0x00020000u};
  const u32x4 w_rs = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)w_base64),
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(w_base64 >> 32)) & 0xffffu,
                      (uint32_t)(nk * I8_WBLOCK), 0x00020000u};
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem;
  const uint32_t m0_a = lds0 + OFF_A + wave * 32 * 128;    // + slot * A_SLOT + piece * 1024
  const uint32_t m0_w = lds0 + OFF_W + (2 * wave) * 1024;  // + slot * W_SLOT (+ 1024: second piece)
  const uint32_t m0_s = lds0 + OFF_W + 256 * 64;           // + slot * W_SLOT

  // fragment read addresses.  Activation: token row l15 (+ 16 per token tile = + 2048 B, swizzle unchanged), chunk 4 kh + lq of
  // the 128-k step; slots 0 and 1 through the DS offset field, slot 2 (beyond 16 bits) through its own base registers.
  // Weights: row 32 wave + 16 ct + l15, the lane's chunk lq (k_i8_codes: chunk XOR-ed with (4 - (row >> 2)) & 3)
  uint32_t fa_lo[2], fa_hi[2], fw[2], fs[2];
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    fa_lo[kh] = lds0 + OFF_A + swz(l15, 4 * kh + lq);
    fa_hi[kh] = fa_lo[kh] + 2 * A_SLOT;
  }
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int rw = wave * 32 + 16 * ct + l15;
    fw[ct] = lds0 + OFF_W + rw * 64 + ((lq ^ ((4 - ((rw >> 2) & 3)) & 3)) << 4);
    fs[ct] = lds0 + OFF_W + 256 * 64 + rw;
  }

  auto issue_step = [&](int kt, int slot) {
    const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, BM * Kp8, 0x00020000);
    const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)w_base, 0, nk * I8_WBLOCK, 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_void*)(smem + OFF_A + slot * A_SLOT + wave * 32 * 128 + i * 1024), 16,
                                               a_voff[i], kt * I8_BK, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(smem + OFF_W + slot * W_SLOT + (2 * wave) * 1024), 16, w_voff0,
                                             kt * I8_WBLOCK, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(smem + OFF_W + slot * W_SLOT + (2 * wave + 1) * 1024), 16, w_voff1,
                                             kt * I8_WBLOCK, 0, 0);
    if (wave == 0)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(smem + OFF_W + slot * W_SLOT + 256 * 64), 4, s_voff, kt * I8_WBLOCK,
                                               0, 0);
  };
  // per-row constants of the epilogue (gemm_w4a8_i8.hip)
  float t_xs = 0.f, t_amax = 0.f;
  if (tid < 256) {
    t_xs = g.xscale[m0 + tid];
    if constexpr (LOWRANK && BOUT == 2) t_amax = g.bout_amax[(int64_t)(m0 + tid) * g.bout_nblk];
  }
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue_step(d, d);  // (past the end of K: dropped by the buffer range check)
  if (tid < 256) {
    const uint32_t ta = lds0 + EP_TAB + 4 * tid;
    asm volatile("ds_write_b32 %0, %1" ::"v"(ta), "v"(t_xs) : "memory");
    if constexpr (LOWRANK && BOUT == 2) {
      int up = g.bout.mbits - block_exponent(t_amax, g.bout);
      up = up > 126 ? 126 : (up < -126 ? -126 : up);
      asm volatile("ds_write_b32 %0, %1 offset:1024\n\tds_write_b32 %0, %2 offset:2048" ::"v"(ta),
                   "v"((uint32_t)(127 + up) << 23), "v"((uint32_t)(127 - up) << 23)
                   : "memory");
    }
  }

  i32x4 R[16][2];  // [token tile][column tile]
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int c = 0; c < 2; ++c) R[i][c] = i32x4{0, 0, 0, 0};

  // ---- main loop: half-steps h = 2 kt + P (barrier / RAW / WAR argument: gemm_w4a8_m256.hip) -------------------------------
  // LOAD(kt, 0): the step's weight words (2 x 16 B) and shift bytes, the activation fragments of token tiles 0-7 (16 x 16 B),
  // half of the prefetch of step kt+2; LOAD(kt, 1): tiles 8-15, the other half.
  const bool late = wave >= 4;
  asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // step 0 landed, the tables written
  if (late) asm volatile("s_barrier" ::: "memory");
  i32x4 wf[2][2];      // the step's expanded weight fragments [column tile][64-deep half]: live across both half-steps
  uint32_t sv[2] = {0, 0};  // this lane's (column's) shifts of the step's 128-k group
  auto half_step = [&](int kt, auto slot_c, auto half_c) {
    constexpr int SLOT = decltype(slot_c)::value;
    constexpr int P = decltype(half_c)::value;
    constexpr int slot_new = (SLOT + DEPTH) % NSLOT;
    constexpr int A_IMM = (SLOT == 2 ? 0 : SLOT * A_SLOT) + 8 * P * 2048;  // token tile t of this half: + 2048 t
    __builtin_amdgcn_s_setprio(1);
    const int ktn = __builtin_amdgcn_readfirstlane(kt + DEPTH);
    const int a_soff = ktn * I8_BK, w_soff = ktn * I8_WBLOCK;
    const uint32_t m0a0 = m0_a + slot_new * A_SLOT + (2 * P) * 1024, m0a1 = m0a0 + 1024;
    const uint32_t m0w = m0_w + slot_new * W_SLOT + P * 1024, m0s = m0_s + slot_new * W_SLOT;
    i32x4 xa[8][2];  // [token tile of this half][64-deep half]
    u32x4 wr[2];
#define I8T_READS_X                                                                                                    \
      "ds_read_b128 %[x00], %[fa0] offset:%c[aimm]\n\tds_read_b128 %[x01], %[fa1] offset:%c[aimm]\n\t"                   \
      "ds_read_b128 %[x10], %[fa0] offset:%c[aimm]+2048\n\tds_read_b128 %[x11], %[fa1] offset:%c[aimm]+2048\n\t"         \
      "ds_read_b128 %[x20], %[fa0] offset:%c[aimm]+4096\n\tds_read_b128 %[x21], %[fa1] offset:%c[aimm]+4096\n\t"         \
      "ds_read_b128 %[x30], %[fa0] offset:%c[aimm]+6144\n\tds_read_b128 %[x31], %[fa1] offset:%c[aimm]+6144\n\t"         \
      "ds_read_b128 %[x40], %[fa0] offset:%c[aimm]+8192\n\tds_read_b128 %[x41], %[fa1] offset:%c[aimm]+8192\n\t"         \
      "ds_read_b128 %[x50], %[fa0] offset:%c[aimm]+10240\n\tds_read_b128 %[x51], %[fa1] offset:%c[aimm]+10240\n\t"       \
      "ds_read_b128 %[x60], %[fa0] offset:%c[aimm]+12288\n\tds_read_b128 %[x61], %[fa1] offset:%c[aimm]+12288\n\t"       \
      "ds_read_b128 %[x70], %[fa0] offset:%c[aimm]+14336\n\tds_read_b128 %[x71], %[fa1] offset:%c[aimm]+14336\n\t"
#define I8T_DMA                                                                                                        \
      "s_mov_b32 m0, %[m0a0]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[av0], %[ars], %[asoff] offen lds\n\t"                   \
      "s_mov_b32 m0, %[m0a1]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[av1], %[ars], %[asoff] offen lds\n\t"                   \
      "s_mov_b32 m0, %[m0w]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[wv], %[wrs], %[wsoff] offen lds\n\t"
#define I8T_OUTS_X                                                                                                     \
      [x00] "=&v"(xa[0][0]), [x01] "=&v"(xa[0][1]), [x10] "=&v"(xa[1][0]), [x11] "=&v"(xa[1][1]), [x20] "=&v"(xa[2][0]),      \
      [x21] "=&v"(xa[2][1]), [x30] "=&v"(xa[3][0]), [x31] "=&v"(xa[3][1]), [x40] "=&v"(xa[4][0]), [x41] "=&v"(xa[4][1]),      \
      [x50] "=&v"(xa[5][0]), [x51] "=&v"(xa[5][1]), [x60] "=&v"(xa[6][0]), [x61] "=&v"(xa[6][1]), [x70] "=&v"(xa[7][0]),      \
      [x71] "=&v"(xa[7][1])
#define I8T_INS                                                                                                        \
      [fa0] "v"(SLOT == 2 ? fa_hi[0] : fa_lo[0]), [fa1] "v"(SLOT == 2 ? fa_hi[1] : fa_lo[1]), [aimm] "i"(A_IMM),              \
      [av0] "v"(a_voff[2 * P]), [av1] "v"(a_voff[2 * P + 1]), [wv] "v"(P == 0 ? w_voff0 : w_voff1), [ars] "s"(a_rs),          \
      [wrs] "s"(w_rs), [m0a0] "s"(m0a0), [m0a1] "s"(m0a1), [asoff] "s"(a_soff), [m0w] "s"(m0w), [wsoff] "s"(w_soff)
    if constexpr (P == 0) {
      asm volatile(I8T_READS_X
                   "ds_read_b128 %[wr0], %[fw0] offset:%c[wimm]\n\tds_read_b128 %[wr1], %[fw1] offset:%c[wimm]\n\t"
                   "ds_read_u8 %[sv0], %[fs0] offset:%c[wimm]\n\tds_read_u8 %[sv1], %[fs1] offset:%c[wimm]\n\t"
                   I8T_DMA "s_waitcnt lgkmcnt(0)"
                   : I8T_OUTS_X, [wr0] "=&v"(wr[0]), [wr1] "=&v"(wr[1]), [sv0] "=&v"(sv[0]), [sv1] "=&v"(sv[1])
                   : I8T_INS, [fw0] "v"(fw[0]), [fw1] "v"(fw[1]), [fs0] "v"(fs[0]), [fs1] "v"(fs[1]), [wimm] "i"(SLOT * W_SLOT)
                   : "memory");
    } else {
      asm volatile(I8T_READS_X I8T_DMA
                   "s_cmp_lg_u32 %[wave], 0\n\ts_cbranch_scc1 1f\n\t"
                   "s_mov_b32 m0, %[m0s]\n\ts_nop 0\n\tbuffer_load_dword %[sv4], %[wrs], %[wsoff] offen lds\n\t"
                   "1:\n\ts_waitcnt vmcnt(6) lgkmcnt(0)"
                   : I8T_OUTS_X
                   : I8T_INS, [wave] "s"(wave), [sv4] "v"(s_voff), [m0s] "s"(m0s)
                   : "memory", "scc");
    }
#undef I8T_READS_X
#undef I8T_DMA
#undef I8T_OUTS_X
#undef I8T_INS
    // two words of two's-complement nibbles -> 16 int8 lanes equal to 16 * code (3 instructions per 8 weights)
    auto expand = [](uint32_t w0, uint32_t w1) {
      return (i32x4){(int)((w0 << 4) & 0xF0F0F0F0u), (int)(w0 & 0xF0F0F0F0u), (int)((w1 << 4) & 0xF0F0F0F0u), (int)(w1 & 0xF0F0F0F0u)};
    };
    if constexpr (P == 0) {
      wf[0][0] = expand(wr[0][0], wr[0][1]);
      asm volatile("s_barrier" : "+v"(wf[0][0])::"memory");
    } else {
      asm volatile("s_barrier" ::: "memory");
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    // ---- COMPUTE: column tile major, so that the second column tile's fragments are expanded under the first one's MFMAs
    if constexpr (P == 0) {
      wf[0][1] = expand(wr[0][2], wr[0][3]);
      wf[1][0] = expand(wr[1][0], wr[1][1]);
      wf[1][1] = expand(wr[1][2], wr[1][3]);
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if constexpr (!SHIFT) {
          R[8 * P + t][ct] = __builtin_amdgcn_mfma_i32_16x16x64_i8(xa[t][0], wf[ct][0], R[8 * P + t][ct], 0, 0, 0);
          R[8 * P + t][ct] = __builtin_amdgcn_mfma_i32_16x16x64_i8(xa[t][1], wf[ct][1], R[8 * P + t][ct], 0, 0, 0);
        } else {
          const i32x4 z = {0, 0, 0, 0};
          i32x4 G = __builtin_amdgcn_mfma_i32_16x16x64_i8(xa[t][0], wf[ct][0], z, 0, 0, 0);
          G = __builtin_amdgcn_mfma_i32_16x16x64_i8(xa[t][1], wf[ct][1], G, 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 4; ++j) R[8 * P + t][ct][j] = (int)(((uint32_t)G[j] << sv[ct]) + (uint32_t)R[8 * P + t][ct][j]);
        }
      }
    // issue order: every MFMA is followed by the vector instructions that fit its shadow (the expands first, then the
    // folds of the tile finished two MFMAs earlier)
    constexpr int V_HEAD = SHIFT ? (P == 0 ? 4 : 2) : (P == 0 ? 2 : 0);  // slots 0..8: the expands of P = 0 on top
    constexpr int V_REST = SHIFT ? 2 : 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if constexpr (V_HEAD > 0) __builtin_amdgcn_sched_group_barrier(0x002, V_HEAD, 0);
    }
#pragma unroll
    for (int i = 9; i < 32; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if constexpr (V_REST > 0) __builtin_amdgcn_sched_group_barrier(0x002, V_REST, 0);
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
  asm volatile("s_barrier" ::: "memory");  // every wave is past its last LDS read of the ring: the epilogue may overwrite it

  // ---- epilogue ---------------------------------------------------------------------------------------------------------
  // lane: output columns n = n0 + 32 wave + 16 ct + l15; register j of tile (t, ct): token row m0 + 16 t + 4 lq + j
  const int nb = n0 + wave * 32;
  const float* const wscale = (const float*)(g.w8 + (size_t)g.tiles_n * nk * I8_WBLOCK);
  const int nsl = LOWRANK ? (g.rp + 31) / 32 : 1;  // 32-deep slices per limb (the last one may be half empty)
  const int nfrag = LOWRANK ? g.b_limbs * nsl : 0;  // (limb, slice) pairs per column tile
  if constexpr (LOWRANK) {
    // the tile's rows of xAq by LDS-DMA into the ring's place (panels of 64 columns, the activation tile's row pitch and swizzle)
    const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(g.xaq + (int64_t)m0 * g.xaq_ld), 0, BM * g.xaq_ld * 2, 0x00020000);
    const int npanel = (g.rp + 63) >> 6;
    for (int pn = 0; pn < npanel; ++pn)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wave * 32 + i * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void*)(smem + EP_STAGE + pn * 32768 + wave * 32 * 128 + i * 1024), 16,
                                                 row * g.xaq_ld * 2 + chunk * 16, pn * 128, 0, 0);
      }
  }
  // this wave's B^T fragments (limb l, 32-deep slice ks, column tile ct): lane (n, rank entries 32 ks + 8 lq ..); in registers
  // when there are at most 4 pairs per column tile (rank 64 with fp16 B, rank 128 with 8-bit B), else re-fetched per token tile
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  bf16x8 sb[LOWRANK ? 4 : 1][2];
  auto load_b = [&](int f, int ct) -> bf16x8 {  // pair f = l * nsl + ks
    const int l = f / nsl, ks = f - l * nsl;
    const int r0 = 32 * ks + 8 * lq;
    if (r0 >= g.rp) return zero8;
    return *(const bf16x8*)(g.bt + ((int64_t)l * g.Np + nb + 16 * ct + l15) * g.rp + r0);
  };
  const bool sb_regs = nfrag <= 4;
  if constexpr (LOWRANK) {
    if (sb_regs) {
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) sb[f][ct] = f < nfrag ? load_b(f, ct) : zero8;
    }
  }
  float ws[2], bv[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    ws[ct] = wscale[nb + 16 * ct + l15];
    bv[ct] = g.bias ? g.bias[nb + 16 * ct + l15] : 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // the xAq tile has landed for every wave
  // xAq fragment addresses: token row l15 (+ 16 t: + 2048 B), chunk 4 (ks & 1) + lq of panel ks >> 1
  uint32_t xaddr[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) xaddr[c] = EP_STAGE + swz(l15, 4 * c + lq);
  unsigned char* const out_w = smem + EP_OUT + wave * EP_OUT_WAVE;
  const bool wide = DT != LQER_F32 && (g.ldy & 7) == 0 && nb + 32 <= g.N && (((uintptr_t)g.y) & 15) == 0;  // wave-uniform
  const int rows_left = g.M - m0 < BM ? g.M - m0 : BM;
  const auto y_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((bf16_t*)g.y + (int64_t)m0 * g.ldy + nb), 0,
                                                        (int)((int64_t)(rows_left - 1) * g.ldy * 2 + 64), 0x00020000);
  const float mmax = g.bout.mmax;
  const f2 magic = {12582912.0f, 12582912.0f};
  const uint32_t ttab = lds0 + EP_TAB + 16 * lq;  // rows 16 t + 4 lq ..: + 64 t bytes
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    // per-row constants of the tile's 4 rows of this lane (asm: a table read hipcc can see would be preceded by a vmcnt(0))
    f32x4 xs4, up4 = {0, 0, 0, 0}, dn4 = {0, 0, 0, 0};
    if constexpr (LOWRANK && BOUT == 2) {
      asm volatile("ds_read_b128 %0, %3 offset:%c4\n\tds_read_b128 %1, %3 offset:%c4+1024\n\tds_read_b128 %2, %3 offset:%c4+2048\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(xs4), "=&v"(up4), "=&v"(dn4)
                   : "v"(ttab), "i"(64 * t)
                   : "memory");
    } else {
      asm volatile("ds_read_b128 %0, %1 offset:%c2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(xs4) : "v"(ttab), "i"(64 * t) : "memory");
    }
    float yv[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      f32x4 sp = {0.f, 0.f, 0.f, 0.f};
      if constexpr (LOWRANK) {
        // the side product in the order of the pre-pass (k_bout_amax16): limb-major, slices ascending
        if (sb_regs) {
#pragma unroll
          for (int f = 0; f < 4; ++f)
            if (f < nfrag) {
              const int ks = f % nsl;
              const bf16x8 xf = *(const bf16x8*)(smem + xaddr[ks & 1] + (ks >> 1) * 32768 + t * 2048);
              sp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, sb[f][ct], sp, 0, 0, 0);
            }
        } else {
          for (int f = 0; f < nfrag; ++f) {
            const int ks = f % nsl;
            const bf16x8 xf = *(const bf16x8*)(smem + xaddr[ks & 1] + (ks >> 1) * 32768 + t * 2048);
            sp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, load_b(f, ct), sp, 0, 0, 0);
          }
        }
      }
      const f2 ws2 = {ws[ct], ws[ct]}, bv2 = {bv[ct], bv[ct]};
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        const f2 rf = {(float)R[t][ct][j], (float)R[t][ct][j + 1]};
        f2 v = __builtin_elementwise_fma(rf * (f2){xs4[j], xs4[j + 1]}, ws2, bv2);
        if constexpr (LOWRANK) {
          f2 sv2 = {sp[j], sp[j + 1]};
          if constexpr (BOUT == 2) {
            // block_fp.py:55-65 on the signed value (every step is odd-symmetric), gemm_w4a8_i8.hip
            const f2 eps = {copysignf(1e-9f, sv2[0]), copysignf(1e-9f, sv2[1])};
            const f2 tt = (sv2 + eps) * (f2){up4[j], up4[j + 1]};
            f2 r = (tt + magic) - magic;
            r[0] = __builtin_amdgcn_fmed3f(r[0], -mmax, mmax);
            r[1] = __builtin_amdgcn_fmed3f(r[1], -mmax, mmax);
            const f2 qv = r * (f2){dn4[j], dn4[j + 1]};
            sv2[0] = fabsf(sv2[0]) <= 1e-8f ? sv2[0] : qv[0];
            sv2[1] = fabsf(sv2[1]) <= 1e-8f ? sv2[1] : qv[1];
          }
          v += sv2;
        }
        yv[ct][j] = v[0], yv[ct][j + 1] = v[1];
      }
    }
    if constexpr (DT == LQER_F32) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int m = m0 + 16 * t + 4 * lq + j, n = nb + 16 * ct + l15;
          if (m < g.M && n < g.N) ((float*)g.y)[(int64_t)m * g.ldy + n] = yv[ct][j];
        }
    } else {
      // 16-bit outputs: four token tiles (64 rows x 32 columns) at a time through this wave's LDS region, then 16-byte stores
      unsigned char* const dst = out_w + ((t & 3) * 16 + 4 * lq) * 80 + l15 * 2;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          uint16_t hv;
          if constexpr (DT == LQER_BF16) hv = f32_to_bf16_rne(yv[ct][j]);
          else hv = __builtin_bit_cast(uint16_t, (_Float16)yv[ct][j]);
          *(uint16_t*)(dst + j * 80 + ct * 32) = hv;
        }
      if ((t & 3) == 3) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int row = u * 16 + (lane >> 2), ch = lane & 3;
          const u32x4 v = *(const u32x4*)(out_w + row * 80 + ch * 16);
          const int mrow = 16 * (t - 3) + row;  // row within the tile
          if (wide) {
            __builtin_amdgcn_raw_buffer_store_b128(v, y_rsrc, (mrow * (int)g.ldy + 8 * ch) * 2, 0, 0);
          } else if (m0 + mrow < g.M) {
            bf16_t* gdst = (bf16_t*)g.y + (int64_t)(m0 + mrow) * g.ldy + nb + 8 * ch;
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (nb + 8 * ch + e < g.N) gdst[e] = (bf16_t)(v[e >> 1] >> (16 * (e & 1)));
          }
        }
      }
    }
  }
}

// Pre-pass for the per-row B_out of this kernel: max |xAq @ B| over every token row, computed with the SAME instruction,
// operand roles and summation order as the epilogue above (limb-major, 32-deep slices ascending), so that the maxima are those
// of the very sums the epilogue quantizes.  One wave = 4 token tiles of 16 rows x a run of 16-column tiles; the running maxima
// stay in registers, one atomicMax (fp32 bit pattern: order-independent) per row at the end.  amax is zeroed before.
__global__ __launch_bounds__(256) void k_bout_amax16(GemmArgs g, int tiles_n16, int seg_tiles) {
  constexpr int RG = 4, MAXF = 8;  // (limb, slice) pairs held in registers per token tile
  const int lane = threadIdx.x & 63, l15 = lane & 15, lq = lane >> 4;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t groups = ((g.M + 15) / 16 + RG - 1) / RG;
  const int nseg = (tiles_n16 + seg_tiles - 1) / seg_tiles;
  if (wid >= groups * nseg) return;
  const int tg = (int)(wid / nseg), sg = (int)(wid - (int64_t)tg * nseg);
  const int nsl = (g.rp + 31) / 32;
  const int Mp = (g.M + LQER_M_ALIGN - 1) / LQER_M_ALIGN * LQER_M_ALIGN;
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  bf16x8 xa[RG][4];  // slices of this lane's token rows (rank <= 128)
#pragma unroll
  for (int u = 0; u < RG; ++u) {
    const int row = (tg * RG + u) * 16 + l15;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      xa[u][ks] = zero8;
      const int r0 = 32 * ks + 8 * lq;
      if (ks < nsl && r0 < g.rp && row < Mp) xa[u][ks] = *(const bf16x8*)(g.xaq + (int64_t)row * g.xaq_ld + r0);
    }
  }
  (void)MAXF;
  float cur[RG][4];
#pragma unroll
  for (int u = 0; u < RG; ++u)
#pragma unroll
    for (int j = 0; j < 4; ++j) cur[u][j] = 0.f;
  const int t_begin = sg * seg_tiles;
  const int t_end = t_begin + seg_tiles < tiles_n16 ? t_begin + seg_tiles : tiles_n16;
  for (int tn = t_begin; tn < t_end; ++tn) {
    f32x4 acc[RG];
#pragma unroll
    for (int u = 0; u < RG; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < g.b_limbs; ++l) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        if (ks < nsl) {
          const int r0 = 32 * ks + 8 * lq;
          bf16x8 bb = zero8;
          if (r0 < g.rp) bb = *(const bf16x8*)(g.bt + ((int64_t)l * g.Np + tn * 16 + l15) * g.rp + r0);
#pragma unroll
          for (int u = 0; u < RG; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[u][ks], bb, acc[u], 0, 0, 0);
        }
    }
#pragma unroll
    for (int u = 0; u < RG; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) cur[u][j] = fmaxf(cur[u][j], fabsf(acc[u][j]));
  }
  // lane holds token rows 4 lq + j, column l15: the row maximum is over the 16 lanes of a DPP row
#pragma unroll
  for (int u = 0; u < RG; ++u)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float mx = row16_max(cur[u][j]);
      const int row = (tg * RG + u) * 16 + 4 * lq + j;
      if (l15 == 0 && row < Mp) atomicMax((unsigned int*)g.bout_amax + (int64_t)row * g.bout_nblk, __float_as_uint(mx));
    }
}

template <int DT>
static int launch(GemmArgs g, bool lowrank, int bout, hipStream_t st) {
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = g.Np / BN;
  const unsigned grid = (unsigned)(g.tiles_m * g.tiles_n);
#define LQER_I8T_LAUNCH(LR, BO)                                                                   \
  do {                                                                                            \
    if (g.i8_shift) {                                                                             \
      static LdsLimitOnce lds_once;                                                               \
      lds_once.set((const void*)k_lqer_gemm_i8t<DT, LR, BO, true>, KERNEL_LDS);                     \
      k_lqer_gemm_i8t<DT, LR, BO, true><<<grid, 512, KERNEL_LDS, st>>>(g);                          \
    } else {                                                                                      \
      static LdsLimitOnce lds_once;                                                               \
      lds_once.set((const void*)k_lqer_gemm_i8t<DT, LR, BO, false>, KERNEL_LDS);                    \
      k_lqer_gemm_i8t<DT, LR, BO, false><<<grid, 512, KERNEL_LDS, st>>>(g);                         \
    }                                                                                             \
  } while (0)
  if (!lowrank)
    LQER_I8T_LAUNCH(false, 0);
  else if (bout == 2)
    LQER_I8T_LAUNCH(true, 2);
  else
    LQER_I8T_LAUNCH(true, 0);
#undef LQER_I8T_LAUNCH
  return check_launch("lqer_gemm_i8t");
}

}  // namespace i8t

int i8t_dispatch(const GemmArgs& g, int dtype, bool lowrank, int bout, hipStream_t st) {
  switch (dtype) {
    case LQER_F32: return i8t::launch<LQER_F32>(g, lowrank, bout, st);
    case LQER_F16: return i8t::launch<LQER_F16>(g, lowrank, bout, st);
    case LQER_BF16: return i8t::launch<LQER_BF16>(g, lowrank, bout, st);
  }
  set_error("unknown dtype %d", dtype);
  return LQER_E_INVALID;
}

// the row maxima of xAq @ B for this kernel's per-row B_out (g.bout_amax zeroed by the caller; g.bout_nblk == 1)
int i8t_bout_amax_dispatch(const GemmArgs& g, hipStream_t st) {
  const int tiles_n16 = g.Np / 16;
  const int64_t groups = ((g.M + 15) / 16 + 3) / 4;
  int nseg = (int)(4096 / groups);  // about 16 waves per CU in total
  nseg = nseg < 1 ? 1 : (nseg > tiles_n16 ? tiles_n16 : nseg);
  const int seg_tiles = (tiles_n16 + nseg - 1) / nseg;
  const int64_t waves = groups * ((tiles_n16 + seg_tiles - 1) / seg_tiles);
  i8t::k_bout_amax16<<<(unsigned)((waves + 3) / 4), 256, 0, st>>>(g, tiles_n16, seg_tiles);
  return check_launch("bout_amax16");
}

}  // namespace lqer
