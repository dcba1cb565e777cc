// The 128-row fused W4 x A8 Linear kernel on v_mfma_f32_16x16x32_bf16:
//
//   y[m,n] = sum_k xq[m,k] * Wq[n,k]  +  bq[n]  +  Q_Bout( sum_j xAq[m,j] * B[j,n] )
//
// Same tile (128 x 256 per workgroup, 8 waves side by side along n, 128 x 32 per wave), same LDS images, same 4-slot
// LDS-DMA ring and LOAD / COMPUTE ping-pong as gemm_w4a8.hip (read that file for the pipeline) - only the matrix
// instruction differs.  Why: the 16x16x32 and 32x32x16 forms have the same per-clock rate, but under sustained MFMA load the
// chip holds a higher clock on the 16x16 form (tools/ubench/mfma_shape.hip on random operands: 2.05 against 1.78 GHz,
// 2.09 against 1.81 PFLOP/s), and this kernel's main loop is MFMA-paced.
//
// Fragment maps (MFMA issued "transposed": A operand = weight rows, B operand = token rows):
//   A: lane l supplies weight row n = l & 15 of a 16-column tile, k = 8 (l >> 4) .. + 8 of a 32-deep slice = exactly one
//      32-bit word of 4-bit codes -> one expand per fragment; a fragment feeds the 8 token tiles of the wave;
//   B: lane l supplies token row l & 15 of a 16-row tile, the same k: one ds_read_b128 of the bf16 activation tile;
//   D: lane l holds token l & 15, output columns 4 (l >> 4) + j, j = 0..3, of the 16 x 16 tile.
// A B_out block (16 consecutive columns of one token) is 4 registers of the 4 lanes l, l^16, l^32, l^48: 3 in-lane max,
// one v_permlane16_swap and one v_permlane32_swap; 16-bit outputs leave as 16-byte stores after one v_permlane16_swap per
// 8 bytes (lanes l, l^16 exchange the halves they do not store).
// Side path: the direct route of gemm_w4a8.hip only (padded rank x limbs <= 32: ONE 32-deep MFMA per tile and limb);
// longer side products stay on the 32x32 kernel (launch_gemm there).
#include <type_traits>

#include "common.h"

namespace lqer {
namespace t16 {

constexpr int BM = 128, BN = 256, BK = 64;
constexpr int DEPTH = 3;                              // k-steps of prefetch in flight
constexpr int NSLOT = DEPTH + 1;                      // LDS ring slots
constexpr int A_SLOT = BM * BK * 2;                   // 16 KiB  activation tile, bf16
constexpr int R_SLOT = (BN / 16) * LQER_PANEL_BYTES;  // 9216 B  packed weight panels
constexpr int OFF_A = 0;
constexpr int OFF_R = NSLOT * A_SLOT;
constexpr int GEMM_LDS = OFF_R + NSLOT * R_SLOT;  // 102400 B

__device__ __forceinline__ int swz(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

typedef __attribute__((address_space(3))) void lds_void;

// max over the lanes l, l^16, l^32, l^48 (vector ALU only: an LDS-based shuffle the compiler can see would make its
// waitcnt pass drain the LDS-DMA ring)
__device__ __forceinline__ float quad16_max(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// BOUT: 0 pass-through, 1 blocks of 16 (max in registers), 2 any block (max from the pre-pass k_bout_amax)
template <int DT, bool LOWRANK, int BOUT>
__global__ __launch_bounds__(512) void k_lqer_gemm_t16(GemmArgs g) {
  constexpr bool XF16 = DT == LQER_F16X;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave;
  const int l15 = lane & 15, lq = lane >> 4;
  const bool hi = (lq >> 1) != 0;

  // XCD-aware tile order (gemm_w4a8.hip): each XCD works on a contiguous run of tiles
  const int nt = g.tiles_m * g.tiles_n;
  int tile;
  {
    const int b = blockIdx.x, xcd = b & 7, q8 = nt >> 3, r8 = nt & 7;
    tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
  }
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk = g.Kp / BK;

  // ---- staging (identical to gemm_w4a8.hip: the LDS images are the same)
  const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(g.xq + (int64_t)m0 * g.Kp), 0, BM * g.Kp * 2, 0x00020000);
  int a_voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wave * 16 + i * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    a_voff[i] = (row * g.Kp + chunk * 8) * 2;
  }
  const uint8_t* w_base = g.wp + ((int64_t)(n0 / 16) * nk) * LQER_PANEL_BYTES;
  const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)w_base, 0, 16 * nk * LQER_PANEL_BYTES, 0x00020000);
  auto w_piece_voff = [&](int piece) {
    const int byte = piece * 1024 + lane * 16;
    const int pnl = byte / LQER_PANEL_BYTES;
    return pnl * nk * LQER_PANEL_BYTES + (byte - pnl * LQER_PANEL_BYTES);
  };
  const int w_voff = w_piece_voff(wave), w_voff8 = w_piece_voff(8);
  unsigned char* const a_dst0 = smem + OFF_A + wave * 16 * 128;
  unsigned char* const w_dst0 = smem + OFF_R + wave * 1024;
  auto issue_loads = [&](int kt, int slot) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_void*)(a_dst0 + slot * A_SLOT + i * 1024), 16, a_voff[i],
                                               kt * (BK * 2), 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(w_dst0 + slot * R_SLOT), 16, w_voff, kt * LQER_PANEL_BYTES, 0, 0);
    if (wave == 0)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(smem + OFF_R + 8192 + slot * R_SLOT), 16, w_voff8,
                                               kt * LQER_PANEL_BYTES, 0, 0);
  };
  // fragment read addresses (slot 0).  Activation: token row l15 (+ 16 per token tile = + 2048 B, same swizzle), chunk
  // 4 kh + lq of the 64-deep step (kh = 32-deep half).  Weights: row 16 ct + l15 of the wave's 32 columns = panel 2 wn + ct,
  // row l15; the row's 32 B of codes hold the words of chunks {0,2,4,6} then {1,3,5,7}: the lane's chunks lq and 4 + lq
  // are words lq >> 1 and 2 + (lq >> 1) of the 16-byte half lq & 1; their block exponents are bytes lq >> 1 and 2 + (lq >> 1).
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem;
  uint32_t fa_addr[2], fw_addr[2], fe_addr[2];
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) fa_addr[kh] = lds0 + OFF_A + swz(l15, 4 * kh + lq);
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    fw_addr[ct] = lds0 + OFF_R + (2 * wn + ct) * LQER_PANEL_BYTES + l15 * 32 + (lq & 1) * 16;
    fe_addr[ct] = lds0 + OFF_R + (2 * wn + ct) * LQER_PANEL_BYTES + 512 + l15 * 4;
  }

  // ---- side path: operands requested ahead of the ring prefetch (loads return in order)
  // one 32-deep slice per limb: lanes whose rank entries 8 lq .. lie beyond the padded rank supply zeros
  bf16x8 db[LOWRANK ? 2 : 1][2], dx[LOWRANK ? 8 : 1];
  if constexpr (LOWRANK) {
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool in_rank = 8 * lq < g.rp;
#pragma unroll
    for (int l = 0; l < 2; ++l)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        db[l][ct] = zero8;
        if (in_rank && l < g.b_limbs) db[l][ct] = *(const bf16x8*)(g.bt + ((int64_t)l * g.Np + n0 + wn * 32 + 16 * ct + l15) * g.rp + 8 * lq);
      }
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      dx[mt] = zero8;
      if (in_rank) dx[mt] = *(const bf16x8*)(g.xaq + (int64_t)(m0 + 16 * mt + l15) * g.xaq_ld + 8 * lq);
    }
  }

#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue_loads(d, d);  // (past the end of K: dropped by the buffer range check)

  f32x4 acc[8][2];  // [token tile][column tile]
#pragma unroll
  for (int mt = 0; mt < 8; ++mt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) acc[mt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- low-rank prologue: acc = Q_Bout(xAq @ B) + bias
  if constexpr (LOWRANK) {
#pragma unroll
    for (int l = 0; l < 2; ++l)
      if (l < g.b_limbs) {
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) acc[mt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(db[l][ct], dx[mt], acc[mt][ct], 0, 0, 0);
      }
    if constexpr (BOUT != 0) {
      const int mb = g.bout.mbits;
#pragma unroll
      for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          float amax;
          if constexpr (BOUT == 1) {
            amax = fmaxf(fmaxf(fabsf(acc[mt][ct][0]), fabsf(acc[mt][ct][1])), fmaxf(fabsf(acc[mt][ct][2]), fabsf(acc[mt][ct][3])));
            amax = quad16_max(amax);
          } else {
            amax = g.bout_amax[(int64_t)(m0 + 16 * mt + l15) * g.bout_nblk + (n0 + wn * 32 + 16 * ct) / g.bout_L];
          }
          const int e = block_exponent(amax, g.bout);  // amax = 0: every element takes the pass-through
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float t = acc[mt][ct][j];
            const float m = fminf(rintf(ldexpf(fabsf(t) + 1e-9f, mb - e)), g.bout.mmax);
            const float q = copysignf(ldexpf(m, e - mb), t);
            acc[mt][ct][j] = fabsf(t) <= 1e-8f ? t : q;
          }
        }
    }
  }
  if (g.bias) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float bv = g.bias[n0 + wn * 32 + 16 * ct + 4 * lq + j];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) acc[mt][ct][j] += bv;
      }
  }

  // ---- main loop (pipeline, barriers, RAW / WAR argument: gemm_w4a8.hip) --------------------------------------------
  const bool late = wave >= 4;
  asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");  // loads(0) landed; two batches of 3 may stay in flight
  if (late) asm volatile("s_barrier" ::: "memory");
  const unsigned long long a_base64 = (unsigned long long)(g.xq + (int64_t)m0 * g.Kp);
  const unsigned long long w_base64 = (unsigned long long)w_base;
  const u32x4 a_rs = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a_base64),
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a_base64 >> 32)) & 0xffffu,
                      (uint32_t)(BM * g.Kp * 2), 0x00020000u};
  const u32x4 w_rs = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)w_base64),
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(w_base64 >> 32)) & 0xffffu,
                      (uint32_t)(16 * nk * LQER_PANEL_BYTES), 0x00020000u};
  const uint32_t m0_a = lds0 + OFF_A + wave * 16 * 128;
  const uint32_t m0_w = lds0 + OFF_R + wave * 1024;
  const uint32_t m0_w8 = lds0 + OFF_R + 8192;
  // this lane's block exponents are bytes hi (first 32-deep half) and 2 + hi of the row's exponent word; alignbit(w, w, s)
  // rotates right by s: byte at bit offset o lands on bit 23 for s = (o + 9) & 31
  const uint32_t rot_lo = hi ? 17u : 9u, rot_hi = hi ? 1u : 25u;
  auto step = [&](int kt, auto slot_c) {
    constexpr int SLOT = decltype(slot_c)::value;
    constexpr int slot_new = SLOT == 0 ? NSLOT - 1 : SLOT - 1;  // (slot + DEPTH) % NSLOT
    __builtin_amdgcn_s_setprio(1);
    const int ktn = __builtin_amdgcn_readfirstlane(kt + DEPTH);
    const int a_soff = ktn * (BK * 2), w_soff = ktn * LQER_PANEL_BYTES;
    const uint32_t m0a0 = m0_a + slot_new * A_SLOT, m0a1 = m0a0 + 1024;
    const uint32_t m0w = m0_w + slot_new * R_SLOT, m0w8 = m0_w8 + slot_new * R_SLOT;
    bf16x8 xa[2][8];  // [kh][token tile]
    u32x4 wr[2];
    uint32_t we[2];
    // LOAD(kt): ONE asm statement - the 20 LDS reads, then the LDS-DMA prefetch of step kt + 3, then the counted waits
    asm volatile(
        "ds_read_b128 %[wr0], %[fw0] offset:%c[rimm]\n\tds_read_b32 %[we0], %[fe0] offset:%c[rimm]\n\t"
        "ds_read_b128 %[x00], %[fa0] offset:%c[aimm]\n\tds_read_b128 %[x01], %[fa0] offset:%c[aimm]+2048\n\t"
        "ds_read_b128 %[x02], %[fa0] offset:%c[aimm]+4096\n\tds_read_b128 %[x03], %[fa0] offset:%c[aimm]+6144\n\t"
        "ds_read_b128 %[x04], %[fa0] offset:%c[aimm]+8192\n\tds_read_b128 %[x05], %[fa0] offset:%c[aimm]+10240\n\t"
        "ds_read_b128 %[x06], %[fa0] offset:%c[aimm]+12288\n\tds_read_b128 %[x07], %[fa0] offset:%c[aimm]+14336\n\t"
        "ds_read_b128 %[wr1], %[fw1] offset:%c[rimm]\n\tds_read_b32 %[we1], %[fe1] offset:%c[rimm]\n\t"
        "ds_read_b128 %[x10], %[fa1] offset:%c[aimm]\n\tds_read_b128 %[x11], %[fa1] offset:%c[aimm]+2048\n\t"
        "ds_read_b128 %[x12], %[fa1] offset:%c[aimm]+4096\n\tds_read_b128 %[x13], %[fa1] offset:%c[aimm]+6144\n\t"
        "ds_read_b128 %[x14], %[fa1] offset:%c[aimm]+8192\n\tds_read_b128 %[x15], %[fa1] offset:%c[aimm]+10240\n\t"
        "ds_read_b128 %[x16], %[fa1] offset:%c[aimm]+12288\n\tds_read_b128 %[x17], %[fa1] offset:%c[aimm]+14336\n\t"
        "s_mov_b32 m0, %[m0a0]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[av0], %[ars], %[asoff] offen lds\n\t"
        "s_mov_b32 m0, %[m0a1]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[av1], %[ars], %[asoff] offen lds\n\t"
        "s_mov_b32 m0, %[m0w]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[wv], %[wrs], %[wsoff] offen lds\n\t"
        "s_cmp_lg_u32 %[wave], 0\n\ts_cbranch_scc1 1f\n\t"
        "s_mov_b32 m0, %[m0w8]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[wv8], %[wrs], %[wsoff] offen lds\n\t"
        "1:\n\ts_waitcnt vmcnt(6) lgkmcnt(0)"
        : [wr0] "=&v"(wr[0]), [we0] "=&v"(we[0]), [wr1] "=&v"(wr[1]), [we1] "=&v"(we[1]), [x00] "=&v"(xa[0][0]),
          [x01] "=&v"(xa[0][1]), [x02] "=&v"(xa[0][2]), [x03] "=&v"(xa[0][3]), [x04] "=&v"(xa[0][4]), [x05] "=&v"(xa[0][5]),
          [x06] "=&v"(xa[0][6]), [x07] "=&v"(xa[0][7]), [x10] "=&v"(xa[1][0]), [x11] "=&v"(xa[1][1]), [x12] "=&v"(xa[1][2]),
          [x13] "=&v"(xa[1][3]), [x14] "=&v"(xa[1][4]), [x15] "=&v"(xa[1][5]), [x16] "=&v"(xa[1][6]), [x17] "=&v"(xa[1][7])
        : [fw0] "v"(fw_addr[0]), [fw1] "v"(fw_addr[1]), [fe0] "v"(fe_addr[0]), [fe1] "v"(fe_addr[1]), [fa0] "v"(fa_addr[0]),
          [fa1] "v"(fa_addr[1]), [aimm] "i"(SLOT * A_SLOT), [rimm] "i"(SLOT * R_SLOT), [av0] "v"(a_voff[0]),
          [av1] "v"(a_voff[1]), [wv] "v"(w_voff), [wv8] "v"(w_voff8), [ars] "s"(a_rs), [wrs] "s"(w_rs), [m0a0] "s"(m0a0),
          [m0a1] "s"(m0a1), [m0w] "s"(m0w), [m0w8] "s"(m0w8), [asoff] "s"(a_soff), [wsoff] "s"(w_soff), [wave] "s"(wave)
        : "memory", "scc");
    // the lane's code words and block scales of the two 32-deep halves, per column tile
    // (written out per half: an index 2 kh + hi into the vector would become a chain of three selects)
    auto word = [&](int ct, int kh) {
      const uint32_t a = kh ? wr[ct][2] : wr[ct][0], b = kh ? wr[ct][3] : wr[ct][1];
      return hi ? b : a;
    };
    // biased exponent byte -> bits 23..30: one rotate + one mask (rot_lo / rot_hi: per-lane rotate amounts)
    auto scale = [&](int ct, int kh) { return __builtin_amdgcn_alignbit(we[ct], we[ct], kh ? rot_hi : rot_lo) & 0x7f800000u; };
    bf16x8 wb_first = expand_frag_t<XF16>(word(0, 0), scale(0, 0));
    asm volatile("s_barrier" : "+v"(wb_first)::"memory");
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    // ---- COMPUTE(kt): 4 weight fragments (column tile x half), each expanded in the shadow of the 8 MFMAs before it
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const bf16x8 wb = (kh == 0 && ct == 0) ? wb_first : expand_frag_t<XF16>(word(ct, kh), scale(ct, kh));
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) acc[mt][ct] = mfma_16x16x32<XF16>(wb, xa[kh][mt], acc[mt][ct]);
      }
    // issue order: every MFMA is followed by the two vector instructions of the NEXT fragment's expand that fit its shadow
    // (8 MFMAs x 2 = the 15 instructions of one fragment); the last fragment's MFMAs run bare
#pragma unroll
    for (int i = 0; i < 24; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  using std::integral_constant;
  for (int kt = 0;; kt += NSLOT) {
    step(kt, integral_constant<int, 0>{});
    if (kt + 1 >= nk) break;
    step(kt + 1, integral_constant<int, 1>{});
    if (kt + 2 >= nk) break;
    step(kt + 2, integral_constant<int, 2>{});
    if (kt + 3 >= nk) break;
    step(kt + 3, integral_constant<int, 3>{});
    if (kt + 4 >= nk) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the prefetches issued past the end of K have drained
  if (!late) asm volatile("s_barrier" ::: "memory");

  // ---- store: lane = token row m0 + 16 mt + l15, columns nb + 16 ct + 4 lq + j
  const bool aligned16 = (((uintptr_t)g.y) & 15) == 0;
  const int nb = n0 + wn * 32;
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    const int m = m0 + 16 * mt + l15;
    if constexpr (DT == LQER_F32) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const int n = nb + 16 * ct + 4 * lq;
        if (m < g.M) {
          float* dst = (float*)g.y + (int64_t)m * g.ldy + n;
          if (n + 3 < g.N && (g.ldy & 3) == 0 && aligned16) {
            *(float4*)dst = make_float4(acc[mt][ct][0], acc[mt][ct][1], acc[mt][ct][2], acc[mt][ct][3]);
          } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
              if (n + t < g.N) dst[t] = acc[mt][ct][t];
          }
        }
      }
    } else {
      uint32_t pk[2][2];  // [column tile][columns 0-1 / 2-3 of the lane's quad]
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float v0 = acc[mt][ct][2 * h], v1 = acc[mt][ct][2 * h + 1];
          if constexpr (DT == LQER_F16 || DT == LQER_F16X) {
            typedef __attribute__((ext_vector_type(2))) _Float16 h2;
            h2 hv = {(_Float16)v0, (_Float16)v1};
            pk[ct][h] = __builtin_bit_cast(uint32_t, hv);
          } else {
            pk[ct][h] = (uint32_t)f32_to_bf16_rne(v0) | ((uint32_t)f32_to_bf16_rne(v1) << 16);
          }
        }
      const bool wide = (g.ldy & 7) == 0 && nb + 32 <= g.N && aligned16;  // wave-uniform
      if (wide) {
        // lanes with even lq keep column tile 0 and receive the quad of lane l + 16; lanes with odd lq keep column tile 1
        // and receive the quad of lane l - 16: 8 consecutive columns = one 16-byte store
        auto r0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
        auto r1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
        if (m < g.M) {
          bf16_t* dst = (bf16_t*)g.y + (int64_t)m * g.ldy + nb + 16 * (lq & 1) + 8 * (lq >> 1);
          *(uint4*)dst = make_uint4(r0[0], r1[0], r0[1], r1[1]);
        }
      } else if (m < g.M) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          const int n = nb + 16 * ct + 4 * lq;
          bf16_t* dst = (bf16_t*)g.y + (int64_t)m * g.ldy + n;
          if (n < g.N) dst[0] = (bf16_t)(pk[ct][0] & 0xffff);
          if (n + 1 < g.N) dst[1] = (bf16_t)(pk[ct][0] >> 16);
          if (n + 2 < g.N) dst[2] = (bf16_t)(pk[ct][1] & 0xffff);
          if (n + 3 < g.N) dst[3] = (bf16_t)(pk[ct][1] >> 16);
        }
      }
    }
  }
}

template <int DT>
static int launch(const GemmArgs& g, bool lowrank, int bout, hipStream_t st) {
  const unsigned grid = (unsigned)(g.tiles_m * g.tiles_n);
#define T16_LAUNCH(LR, BO)                                                          \
  do {                                                                              \
    static LdsLimitOnce lds_once;                                                   \
    lds_once.set((const void*)k_lqer_gemm_t16<DT, LR, BO>, GEMM_LDS);               \
    k_lqer_gemm_t16<DT, LR, BO><<<grid, 512, GEMM_LDS, st>>>(g);                    \
  } while (0)
  if (!lowrank)
    T16_LAUNCH(false, 0);
  else if (bout == 1)
    T16_LAUNCH(true, 1);
  else if (bout == 2)
    T16_LAUNCH(true, 2);
  else
    T16_LAUNCH(true, 0);
#undef T16_LAUNCH
  return check_launch("lqer_gemm_t16");
}

}  // namespace t16

// the 128-row tile route with at most one 32-deep slice of side product per limb (rank 32 with 8-bit A / B: C2, C3)
bool t16_eligible(const GemmArgs& g, bool lowrank) {
#ifdef LQER_NO_T16
  return false;
#else
  return !lowrank || (g.rp <= 32 && g.b_limbs <= 2 && g.rp * g.b_limbs <= 32);
#endif
}

int t16_dispatch(const GemmArgs& g, int dtype, bool lowrank, int bout, hipStream_t st) {
  switch (dtype) {
    case LQER_F32: return t16::launch<LQER_F32>(g, lowrank, bout, st);
    case LQER_F16: return g.x_f16 ? t16::launch<LQER_F16X>(g, lowrank, bout, st) : t16::launch<LQER_F16>(g, lowrank, bout, st);
    case LQER_BF16: return t16::launch<LQER_BF16>(g, lowrank, bout, st);
  }
  set_error("unknown dtype %d", dtype);
  return LQER_E_INVALID;
}

}  // namespace lqer
