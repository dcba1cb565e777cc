// The int8 route of the fused Linear kernel, for configurations whose scales allow exact integer accumulation:
// activations with ONE exponent per token (block_size [1,-1], width <= 8: the "W4A8 INT" configurations, reference
// experiments/pipeline/sweep_lqer_act_int.sh:83, experiments/configs/template/llama-7b-int.toml:87) and 4-bit weights
// whose blocks span 128 k or more.
//
//   y[m,n] = 2^(ex[m]-7) * sum_g 2^(ew[n,g]-3) * ( sum_{k in g} cx[m,k] * cw[n,k] )  +  bq[n]  +  Q_Bout( xAq @ B )
//
// with integer mantissas cx in [-127,127], cw in [-7,7] and g the 128-k groups.  The inner sums run on
// v_mfma_i32_32x32x32_i8 (twice the bf16 rate), exactly:
//  * weights travel as two's-complement nibbles; (w << 4) & 0xF0F0F0F0 and w & 0xF0F0F0F0 ARE the int8 lanes 16 * cw
//    (3 VALU per 8 weights; the 1/16 is folded into the row scale);
//  * activations travel as int8 mantissas (1 B per element: half the LDS-DMA and LDS-read bytes of the bf16 image);
//  * a row's groups share one scale 2^(emin[n]-3) and differ by a left shift s[n,g] = ew[n,g] - emin[n]: the group sum
//    (a 32 x 32 i32 tile per 4 MFMAs) is folded into the running i32 tile with ONE v_lshl_add_u32 per element - the
//    MFMA is issued with tokens as rows and weight rows as columns, so the shift is one register per lane.  A weight
//    whose rows all have s = 0 (one block per row) accumulates straight into the running tile.
//    The i32 range is guaranteed per weight row at pack time (k_i8_rows: sum_g 2^s sum_k |16 cw| * 127 < 2^31), otherwise the
//    Linear stays on the bf16 route (lqer_i8_prepare reports it) - never an inexact result.
//  * epilogue: y = float(R) * 2^(emin[n]-7) * 2^(ex[m]-7) + bias + Q_Bout(xAq @ B); the side product runs on the bf16
//    MFMA from an LDS-staged xAq tile, in the summation order of the k_bout_amax pre-pass; 16-bit outputs are
//    transposed through a per-wave LDS region so that every lane stores 16 bytes.
//
// Tile, ring and wave structure are those of gemm_w4a8_m256.hip: 256(m) x 256(n) per workgroup, 8 waves side by side
// along n, LDS-DMA into a 3-slot ring two steps ahead (one step = 128 k = 128 B per activation row, the same row
// pitch and swizzle), LOAD / COMPUTE ping-pong between the two waves of a SIMD, half a step (4 token tiles) per phase.
#include <type_traits>

#include "common.h"

namespace lqer {

namespace i8 {

constexpr int BM = 256, BN = 256;
constexpr int DEPTH = 2, NSLOT = DEPTH + 1;
constexpr int A_SLOT = BM * I8_BK;   // 32 KiB  int8 activation tile
constexpr int W_SLOT = I8_WBLOCK;    // 16640 B nibbles + shift bytes
constexpr int OFF_A = 0;
constexpr int OFF_W = NSLOT * A_SLOT;
constexpr int GEMM_LDS = OFF_W + NSLOT * W_SLOT;  // 148224 B
// epilogue regions (the ring is free then)
constexpr int EP_STAGE = 0;              // xAq tile: up to two 64-column panels of 32 KiB
constexpr int EP_TAB = 65536;            // x row scales fp32 [256], B_out row exponents int [256]
constexpr int EP_OUT = EP_TAB + 2048;    // per wave: 64 rows x 80 B (32 fp16 columns + pad)
constexpr int EP_OUT_WAVE = 64 * 80;
static_assert(EP_OUT + 8 * EP_OUT_WAVE <= GEMM_LDS, "epilogue regions exceed the ring");

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ int swz(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

// ---- weight image ---------------------------------------------------------------------------------------------------
// Block (n tile tn, step s) at (tn * nk8 + s) * I8_WBLOCK: 256 rows x 64 B of nibbles, then 256 shift bytes.  A row's
// 64 B hold its 16 words of 8 k (word W: k = 8 W .. 8 W + 7; nibble p of a word: k = p/2 for even p, 4 + p/2 for odd p -
// the sign-magnitude image's order) at byte offset h * 32 + ks * 8 + (W & 1) * 4 with ks = W >> 2 (32-k MFMA slice),
// h = (W >> 1) & 1 (lane half): a lane's operand bytes of the four slices are two 16-byte reads.  The 16-byte chunks of a
// row are XOR-ed with (row >> 2) & 3, so that the 16 lanes of a ds_read_b128 group hit 16 distinct 16-byte slots.
__device__ __forceinline__ int w_byte_offset(int rl, int W) {
  const int off = ((W >> 1) & 1) * 32 + (W >> 2) * 8 + (W & 1) * 4;
  return rl * 64 + ((((off >> 4) ^ ((rl >> 2) & 3)) << 4) | (off & 15));
}

// per weight row: group exponents -> base exponent, shifts, row scale; i32 range and format checks (flags[0] != 0: not eligible)
__global__ __launch_bounds__(256) void k_i8_rows(const uint8_t* __restrict__ wp, int64_t N, int64_t Np, int nk, int nk8,
                                                  uint8_t* __restrict__ img, int32_t* __restrict__ flags) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= Np) return;
  float* wscale = (float*)(img + (size_t)(Np / 256) * nk8 * I8_WBLOCK);
  const int64_t tn = n / 256;
  const int rl = (int)(n - tn * 256);
  int bmin = 255;
  bool bad = false;
  // pass 1: the biased exponent byte of every 128-k group (0 = no non-zero code in it), minimum over the row
  for (int s = 0; s < nk8; ++s) {
    int bg = 0;
    for (int half = 0; half < 2; ++half) {
      const int kp = 2 * s + half;
      if (kp >= nk || n >= N) continue;
      const uint8_t* pnl = wp + ((n / 16) * nk + kp) * LQER_PANEL_BYTES;
      const uint32_t* words = (const uint32_t*)(pnl + (n & 15) * 32);
      const uint8_t* eb = pnl + 512 + (n & 15) * 4;
      for (int b = 0; b < 4; ++b) {  // 16-k block b of the panel: words 2b, 2b+1, stored at positions {0,2,4,6,1,3,5,7}^-1
        const uint32_t w0 = words[b], w1 = words[4 + b];
        if (((w0 | w1) & 0x77777777u) == 0) continue;
        if (bg == 0) bg = eb[b];
        else if (bg != eb[b]) bad = true;  // two exponents inside one 128-k group: weight blocks shorter than 128
      }
    }
    if (bg && bg < bmin) bmin = bg;
    img[(tn * nk8 + s) * I8_WBLOCK + 256 * 64 + rl] = (uint8_t)bg;  // (the byte for now; the shift in pass 2)
  }
  // pass 2: shifts and the i32 bound  sum_g 2^s * sum_k |16 c| * 127 < 2^31
  unsigned long long bound = 0;
  bool any_shift = false;
  for (int s = 0; s < nk8; ++s) {
    uint8_t* sp = img + (tn * nk8 + s) * I8_WBLOCK + 256 * 64 + rl;
    const int bg = *sp;
    int sh = 0;
    if (bg) {
      sh = bg - bmin;
      unsigned asum = 0;
      for (int half = 0; half < 2; ++half) {
        const int kp = 2 * s + half;
        if (kp >= nk) continue;
        const uint32_t* words = (const uint32_t*)(wp + ((n / 16) * nk + kp) * LQER_PANEL_BYTES + (n & 15) * 32);
        for (int j = 0; j < 8; ++j) {
          const uint32_t m = words[j] & 0x77777777u;
          for (int p = 0; p < 8; ++p) asum += (m >> (4 * p)) & 7u;
        }
      }
      if (sh > 20) bad = true;
      else bound += ((unsigned long long)asum << sh) * (16ull * 127ull);
      any_shift |= sh != 0;
    }
    *sp = (uint8_t)sh;
  }
  if (bound >= (1ull << 31)) bad = true;
  if (bmin != 255 && bmin < 5) bad = true;  // row scale 2^(e - mbits - 4) would not be a normal float
  wscale[n] = bmin == 255 ? 0.0f : __uint_as_float((uint32_t)(bmin - 4) << 23);
  if (bad) atomicOr(flags, 1);
  if (any_shift) atomicOr(flags + 1, 1);
}

// per (row, step): 16 words of sign-magnitude nibbles -> two's complement, reordered
__global__ __launch_bounds__(256) void k_i8_codes(const uint8_t* __restrict__ wp, int64_t N, int64_t Np, int nk, int nk8,
                                                   uint8_t* __restrict__ img) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= Np * nk8) return;
  const int64_t n = idx / nk8;
  const int s = (int)(idx - n * nk8);
  const int64_t tn = n / 256;
  const int rl = (int)(n - tn * 256);
  uint8_t* blk = img + (tn * nk8 + s) * I8_WBLOCK;
  for (int W = 0; W < 16; ++W) {
    const int kp = 2 * s + (W >> 3), widx = W & 7;
    uint32_t w = 0;
    if (kp < nk && n < N) {
      const int pos = (widx & 1) ? 4 + (widx >> 1) : (widx >> 1);
      w = ((const uint32_t*)(wp + ((n / 16) * nk + kp) * LQER_PANEL_BYTES + (n & 15) * 32))[pos];
    }
    uint32_t out = 0;
    for (int p = 0; p < 8; ++p) {
      const uint32_t nib = (w >> (4 * p)) & 0xfu, mag = nib & 7u;
      const uint32_t tc = (nib & 8u) && mag ? (16u - mag) : mag;
      out |= tc << (4 * p);
    }
    *(uint32_t*)(blk + w_byte_offset(rl, W)) = out;
  }
}

// test hook: the image back to dequantized fp32 [N,K]
__global__ __launch_bounds__(256) void k_i8_unpack(const uint8_t* __restrict__ img, int64_t N, int64_t K, int64_t Np, int nk8,
                                                    float* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * K) return;
  const int64_t n = idx / K, k = idx - n * K;
  const float* wscale = (const float*)(img + (size_t)(Np / 256) * nk8 * I8_WBLOCK);
  const int64_t tn = n / 256;
  const int rl = (int)(n - tn * 256), s = (int)(k / 128), kk = (int)(k % 128);
  const uint8_t* blk = img + (tn * nk8 + s) * I8_WBLOCK;
  const int W = kk >> 3, q = kk & 7;
  const uint32_t w = *(const uint32_t*)(blk + w_byte_offset(rl, W));
  const int p = q < 4 ? 2 * q : 2 * (q - 4) + 1;
  int c = (int)((w >> (4 * p)) & 0xfu);
  c = c >= 8 ? c - 16 : c;
  const int sh = blk[256 * 64 + rl];
  out[idx] = (float)(16 * c * (1 << sh)) * wscale[n];
}

// ---- the GEMM -----------------------------------------------------------------------------------------------------------
// BOUT: 0 pass-through, 2 one block per row (exponent from the k_bout_amax pre-pass).  SHIFT: per-group shifts present.
template <int DT, bool LOWRANK, int BOUT, bool SHIFT>
__global__ __launch_bounds__(512) void k_lqer_gemm_i8(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;

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

  // ---- staging addresses (per lane, fixed for the kernel) ------------------------------------------------------------
  // activations: wave w stages rows [32w, 32w+32) of the tile as 4 pieces of 8 rows x 128 B, chunk-swizzled on the source side
  int a_voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 32 + i * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    a_voff[i] = row * Kp8 + chunk * 16;
  }
  // weights: the block's 16 KiB of nibbles are 16 contiguous 1-KiB pieces: wave w stages pieces 2w and 2w+1; wave 0 also the
  // 256 shift bytes (4 B per lane)
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

  // fragment read addresses: activation row = lane & 31 (+ 32 i: + 4096 B, swizzle unchanged), chunk 2 ks + lh; slots 0 and 1
  // through the DS offset field, slot 2 (beyond 16 bits) through its own base registers
  uint32_t fa_lo[4], fa_hi[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    fa_lo[ks] = lds0 + OFF_A + swz(l31, 2 * ks + lh);
    fa_hi[ks] = fa_lo[ks] + 2 * A_SLOT;
  }
  const int rw = wave * 32 + l31;  // this lane's weight row within the tile
  const uint32_t fw_a = lds0 + OFF_W + rw * 64 + (((2 * lh) ^ ((rw >> 2) & 3)) << 4);      // slices 0, 1
  const uint32_t fw_b = lds0 + OFF_W + rw * 64 + (((2 * lh + 1) ^ ((rw >> 2) & 3)) << 4);  // slices 2, 3
  const uint32_t fs_addr = lds0 + OFF_W + 256 * 64 + rw;

  // one LDS-DMA batch = the operands of one step: 6 loads per wave (wave 0: 7)
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
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue_step(d, d);  // (past the end of K: dropped by the buffer range check)

  i32x16 R[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) R[i][j] = 0;

  // ---- main loop: half-steps h = 2 kt + P (gemm_w4a8_m256.hip has the barrier / RAW / WAR argument) ----------------------
  //   waves 0-3:    ... | LOAD(h)  | COMPUTE(h) | LOAD(h+1) | ...
  //   waves 4-7:    ... | COMP(h-1)| LOAD(h)    | COMPUTE(h)| ...
  // LOAD(kt, 0): the step's weight words (2 x 16 B) and shift byte, the activation fragments of token tiles 0-3 (16 x 16 B),
  // half of the prefetch of step kt+2; LOAD(kt, 1): tiles 4-7, the other half.  A wave ends LOAD(kt, 1) with vmcnt(6): its own
  // batch of step kt+1 has landed (the batch of kt+2 - 6 loads, wave 0: 7 - may stay in flight), then passes a barrier before
  // anyone reads step kt+1.
  const bool late = wave >= 4;
  asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");  // step 0 landed
  if (late) asm volatile("s_barrier" ::: "memory");
  i32x4 wf[4];     // the step's expanded weight fragments (slices 0..3): live across both half-steps
  uint32_t sv = 0;  // this lane's (column's) shift of the step's 128-k group
  auto half_step = [&](int kt, auto slot_c, auto half_c) {
    constexpr int SLOT = decltype(slot_c)::value;
    constexpr int P = decltype(half_c)::value;
    constexpr int slot_new = (SLOT + DEPTH) % NSLOT;
    constexpr int A_IMM = (SLOT == 2 ? 0 : SLOT * A_SLOT) + 4 * P * 4096;  // tile t of this half: + 4096 t
    __builtin_amdgcn_s_setprio(1);
    const int ktn = __builtin_amdgcn_readfirstlane(kt + DEPTH);
    const int a_soff = ktn * I8_BK, w_soff = ktn * I8_WBLOCK;
    const uint32_t m0a0 = m0_a + slot_new * A_SLOT + (2 * P) * 1024, m0a1 = m0a0 + 1024;
    const uint32_t m0w = m0_w + slot_new * W_SLOT + P * 1024, m0s = m0_s + slot_new * W_SLOT;
    i32x4 xa[4][4];  // [tile of this half][slice]
    u32x4 wr0, wr1;
    // (symbolic operand names: x<tile><slice> activation fragments, fa<slice> their address registers)
#define I8_READS_X                                                                                                     \
      "ds_read_b128 %[x00], %[fa0] offset:%c[aimm]\n\tds_read_b128 %[x01], %[fa1] offset:%c[aimm]\n\t"                   \
      "ds_read_b128 %[x02], %[fa2] offset:%c[aimm]\n\tds_read_b128 %[x03], %[fa3] offset:%c[aimm]\n\t"                   \
      "ds_read_b128 %[x10], %[fa0] offset:%c[aimm]+4096\n\tds_read_b128 %[x11], %[fa1] offset:%c[aimm]+4096\n\t"         \
      "ds_read_b128 %[x12], %[fa2] offset:%c[aimm]+4096\n\tds_read_b128 %[x13], %[fa3] offset:%c[aimm]+4096\n\t"         \
      "ds_read_b128 %[x20], %[fa0] offset:%c[aimm]+8192\n\tds_read_b128 %[x21], %[fa1] offset:%c[aimm]+8192\n\t"         \
      "ds_read_b128 %[x22], %[fa2] offset:%c[aimm]+8192\n\tds_read_b128 %[x23], %[fa3] offset:%c[aimm]+8192\n\t"         \
      "ds_read_b128 %[x30], %[fa0] offset:%c[aimm]+12288\n\tds_read_b128 %[x31], %[fa1] offset:%c[aimm]+12288\n\t"       \
      "ds_read_b128 %[x32], %[fa2] offset:%c[aimm]+12288\n\tds_read_b128 %[x33], %[fa3] offset:%c[aimm]+12288\n\t"
#define I8_DMA                                                                                                         \
      "s_mov_b32 m0, %[m0a0]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[av0], %[ars], %[asoff] offen lds\n\t"                   \
      "s_mov_b32 m0, %[m0a1]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[av1], %[ars], %[asoff] offen lds\n\t"                   \
      "s_mov_b32 m0, %[m0w]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[wv], %[wrs], %[wsoff] offen lds\n\t"
#define I8_OUTS_X                                                                                                      \
      [x00] "=&v"(xa[0][0]), [x01] "=&v"(xa[0][1]), [x02] "=&v"(xa[0][2]), [x03] "=&v"(xa[0][3]), [x10] "=&v"(xa[1][0]),      \
      [x11] "=&v"(xa[1][1]), [x12] "=&v"(xa[1][2]), [x13] "=&v"(xa[1][3]), [x20] "=&v"(xa[2][0]), [x21] "=&v"(xa[2][1]),      \
      [x22] "=&v"(xa[2][2]), [x23] "=&v"(xa[2][3]), [x30] "=&v"(xa[3][0]), [x31] "=&v"(xa[3][1]), [x32] "=&v"(xa[3][2]),      \
      [x33] "=&v"(xa[3][3])
#define I8_INS                                                                                                         \
      [fa0] "v"(SLOT == 2 ? fa_hi[0] : fa_lo[0]), [fa1] "v"(SLOT == 2 ? fa_hi[1] : fa_lo[1]),                                 \
      [fa2] "v"(SLOT == 2 ? fa_hi[2] : fa_lo[2]), [fa3] "v"(SLOT == 2 ? fa_hi[3] : fa_lo[3]), [aimm] "i"(A_IMM),              \
      [av0] "v"(a_voff[2 * P]), [av1] "v"(a_voff[2 * P + 1]), [wv] "v"(P == 0 ? w_voff0 : w_voff1), [ars] "s"(a_rs),          \
      [wrs] "s"(w_rs), [m0a0] "s"(m0a0), [m0a1] "s"(m0a1), [asoff] "s"(a_soff), [m0w] "s"(m0w), [wsoff] "s"(w_soff)
    if constexpr (P == 0) {
      asm volatile(I8_READS_X
                   "ds_read_b128 %[wr0], %[fwa] offset:%c[wimm]\n\tds_read_b128 %[wr1], %[fwb] offset:%c[wimm]\n\t"
                   "ds_read_u8 %[sv], %[fs] offset:%c[wimm]\n\t"
                   I8_DMA "s_waitcnt lgkmcnt(0)"
                   : I8_OUTS_X, [wr0] "=&v"(wr0), [wr1] "=&v"(wr1), [sv] "=&v"(sv)
                   : I8_INS, [fwa] "v"(fw_a), [fwb] "v"(fw_b), [fs] "v"(fs_addr), [wimm] "i"(SLOT * W_SLOT)
                   : "memory");
    } else {
      asm volatile(I8_READS_X I8_DMA
                   "s_cmp_lg_u32 %[wave], 0\n\ts_cbranch_scc1 1f\n\t"
                   "s_mov_b32 m0, %[m0s]\n\ts_nop 0\n\tbuffer_load_dword %[sv4], %[wrs], %[wsoff] offen lds\n\t"
                   "1:\n\ts_waitcnt vmcnt(6) lgkmcnt(0)"
                   : I8_OUTS_X
                   : I8_INS, [wave] "s"(wave), [sv4] "v"(s_voff), [m0s] "s"(m0s)
                   : "memory", "scc");
    }
#undef I8_READS_X
#undef I8_DMA
#undef I8_OUTS_X
#undef I8_INS
    auto expand = [](uint32_t w0, uint32_t w1) {
      return (i32x4){(int)((w0 << 4) & 0xF0F0F0F0u), (int)(w0 & 0xF0F0F0F0u), (int)((w1 << 4) & 0xF0F0F0F0u), (int)(w1 & 0xF0F0F0F0u)};
    };
    if constexpr (P == 0) {
      wf[0] = expand(wr0[0], wr0[1]);
      asm volatile("s_barrier" : "+v"(wf[0])::"memory");
    } else {
      asm volatile("s_barrier" ::: "memory");
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    // ---- COMPUTE
    if constexpr (P == 0) {
      wf[1] = expand(wr0[2], wr0[3]);
      wf[2] = expand(wr1[0], wr1[1]);
      wf[3] = expand(wr1[2], wr1[3]);
    }
    if constexpr (!SHIFT) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int t = 0; t < 4; ++t) R[4 * P + t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa[t][ks], wf[ks], R[4 * P + t], 0, 0, 0);
    } else {
      const i32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        i32x16 G = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa[t][0], wf[0], z, 0, 0, 0);
#pragma unroll
        for (int ks = 1; ks < 4; ++ks) G = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa[t][ks], wf[ks], G, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 16; ++j) R[4 * P + t][j] = (int)(((uint32_t)G[j] << sv) + (uint32_t)R[4 * P + t][j]);
      }
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
  // every wave is past its last LDS read of the ring: the epilogue may overwrite it after one more barrier
  asm volatile("s_barrier" ::: "memory");

  // ---- epilogue ---------------------------------------------------------------------------------------------------------
  // lane: output column n = n0 + 32 wave + (lane & 31); register j of tile i: token row m0 + 32 i + (j&3) + 8 (j>>2) + 4 lh.
  const int n = n0 + wave * 32 + l31;
  const float* const xscale = g.xscale;
  const float* const wscale = (const float*)(g.w8 + (size_t)g.tiles_n * nk * I8_WBLOCK);
  float* const tab_xs = (float*)(smem + EP_TAB);
  int* const tab_be = (int*)(smem + EP_TAB + 1024);
  if (tid < 256) {
    tab_xs[tid] = xscale[m0 + tid];
    if constexpr (LOWRANK && BOUT == 2) tab_be[tid] = block_exponent(g.bout_amax[(int64_t)(m0 + tid) * g.bout_nblk], g.bout);
  }
  // the side product's operands: the tile's rows of xAq through LDS (panels of 64 columns, the activation tile's swizzle);
  // this wave's B^T fragments in registers when there are at most 8 (limb, 16-deep slice) pairs - rank 64 with fp16 A / B,
  // rank 128 with 8-bit A / B -, else re-fetched from L2 for every token tile
  bf16x8 sb[LOWRANK ? 8 : 1];
  const int nslices = LOWRANK ? g.rp / 16 : 0;  // 16-deep slices per limb
  const int nfrag = LOWRANK ? g.b_limbs * nslices : 0;
  const bool sbreg = nfrag <= 8;  // (wave-uniform)
  if constexpr (LOWRANK) {
    const int cpr = g.rp >> 3;  // 16-byte chunks per row
    for (int c = tid; c < BM * cpr; c += 512) {
      const int row = c / cpr, ch = c - row * cpr;
      const u32x4 v = *(const u32x4*)(g.xaq + (int64_t)(m0 + row) * g.xaq_ld + 8 * ch);
      *(u32x4*)(smem + EP_STAGE + (ch >> 3) * 32768 + swz(row, ch & 7)) = v;
    }
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      const int l = f / (nslices > 0 ? nslices : 1), ks = f - l * nslices;
      sb[f] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      if (sbreg && l < g.b_limbs) sb[f] = *(const bf16x8*)(g.bt + ((int64_t)l * g.Np + n) * g.rp + ks * 16 + 8 * lh);
    }
  }
  __syncthreads();
  const float ws = wscale[n];
  const float bv = g.bias ? g.bias[n] : 0.f;
  unsigned char* const out_w = smem + EP_OUT + wave * EP_OUT_WAVE;
  const bool aligned16 = (((uintptr_t)g.y) & 15) == 0;
  const int nb = n0 + wave * 32;
  const bool wide = DT != LQER_F32 && (g.ldy & 7) == 0 && nb + 32 <= g.N && aligned16;  // wave-uniform
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float yv[16];
    f32x16 sp;
#pragma unroll
    for (int j = 0; j < 16; ++j) sp[j] = 0.f;
    if constexpr (LOWRANK) {
      if (sbreg) {
#pragma unroll
        for (int f = 0; f < 8; ++f) {
          const int l = f / (nslices > 0 ? nslices : 1), ks = f - l * nslices;
          if (l < g.b_limbs) {
            const bf16x8 xf = *(const bf16x8*)(smem + EP_STAGE + (ks >> 2) * 32768 + swz(l31 + 32 * i, 2 * (ks & 3) + lh));
            sp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, sb[f], sp, 0, 0, 0);
          }
        }
      } else {
        for (int l = 0; l < g.b_limbs; ++l)  // (the same order: limb-major, slices ascending)
          for (int ks = 0; ks < nslices; ++ks) {
            const bf16x8 bf = *(const bf16x8*)(g.bt + ((int64_t)l * g.Np + n) * g.rp + ks * 16 + 8 * lh);
            const bf16x8 xf = *(const bf16x8*)(smem + EP_STAGE + (ks >> 2) * 32768 + swz(l31 + 32 * i, 2 * (ks & 3) + lh));
            sp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, bf, sp, 0, 0, 0);
          }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rloc = 32 * i + 8 * q + 4 * lh;  // + (j & 3)
      const f32x4 xs4 = *(const f32x4*)(tab_xs + rloc);
      i32x4 be4 = {0, 0, 0, 0};
      if constexpr (LOWRANK && BOUT == 2) be4 = *(const i32x4*)(tab_be + rloc);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int j = 4 * q + t;
        float v = (float)R[i][j] * (ws * xs4[t]) + bv;
        if constexpr (LOWRANK) {
          float s = sp[j];
          if constexpr (BOUT == 2) {
            const int e = be4[t], mb = g.bout.mbits;
            const float m = fminf(rintf(ldexpf(fabsf(s) + 1e-9f, mb - e)), g.bout.mmax);
            const float qv = copysignf(ldexpf(m, e - mb), s);
            s = fabsf(s) <= 1e-8f ? s : qv;
          }
          v += s;
        }
        yv[j] = v;
      }
    }
    if constexpr (DT == LQER_F32) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int m = m0 + 32 * i + (j & 3) + 8 * (j >> 2) + 4 * lh;
        if (m < g.M && n < g.N) ((float*)g.y)[(int64_t)m * g.ldy + n] = yv[j];
      }
    } else {
      // 16-bit outputs: two tiles (64 rows x 32 columns) at a time through this wave's LDS region, then 16-byte stores
      unsigned char* const dst = out_w + (i & 1) * 32 * 80 + l31 * 2;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int rl = (j & 3) + 8 * (j >> 2) + 4 * lh;
        uint16_t hv;
        if constexpr (DT == LQER_BF16) hv = f32_to_bf16_rne(yv[j]);
        else hv = __builtin_bit_cast(uint16_t, (_Float16)yv[j]);
        *(uint16_t*)(dst + rl * 80) = hv;
      }
      if (i & 1) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int row = u * 16 + (lane >> 2), ch = lane & 3;
          const uint4 v = *(const uint4*)(out_w + row * 80 + ch * 16);
          const int m = m0 + 32 * (i - 1) + row;
          if (m < g.M) {
            bf16_t* gdst = (bf16_t*)g.y + (int64_t)m * g.ldy + nb + 8 * ch;
            if (wide) {
              *(uint4*)gdst = v;
            } else {
              const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
              for (int e = 0; e < 8; ++e)
                if (nb + 8 * ch + e < g.N) gdst[e] = (bf16_t)(w4[e >> 1] >> (16 * (e & 1)));
            }
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
#define LQER_I8_LAUNCH(LR, BO)                                                                    \
  do {                                                                                            \
    if (g.i8_shift) {                                                                             \
      static LdsLimitOnce lds_once;                                                               \
      lds_once.set((const void*)k_lqer_gemm_i8<DT, LR, BO, true>, GEMM_LDS);                      \
      k_lqer_gemm_i8<DT, LR, BO, true><<<grid, 512, GEMM_LDS, st>>>(g);                           \
    } else {                                                                                      \
      static LdsLimitOnce lds_once;                                                               \
      lds_once.set((const void*)k_lqer_gemm_i8<DT, LR, BO, false>, GEMM_LDS);                     \
      k_lqer_gemm_i8<DT, LR, BO, false><<<grid, 512, GEMM_LDS, st>>>(g);                          \
    }                                                                                             \
  } while (0)
  if (!lowrank)
    LQER_I8_LAUNCH(false, 0);
  else if (bout == 2)
    LQER_I8_LAUNCH(true, 2);
  else
    LQER_I8_LAUNCH(true, 0);
#undef LQER_I8_LAUNCH
  return check_launch("lqer_gemm_i8");
}

}  // namespace i8

// The int8 main loop needs: the int8 images (g.w8 set by the caller for an LQER_Q_MXINT_I8 descriptor), M large enough for
// 256-row tiles to fill the chip in rounds that beat the 128-row bf16 kernel (an int8 256 x 256 tile costs about 1.15
// bf16 128 x 256 tiles), B_out pass-through or one block per row, at most two 64-column panels of xAq.
bool i8_eligible(const GemmArgs& g, int bout) {
  if (!g.w8 || g.M < 512) return false;
  if (!(bout == 0 || (bout == 2 && g.bout_nblk == 1))) return false;
  if (g.rp > 128) return false;
  constexpr int64_t CUS = 256;
  const int64_t t256 = (int64_t)((g.M + i8::BM - 1) / i8::BM) * (g.Np / i8::BN);
  const int64_t t128 = (int64_t)((g.M + 127) / 128) * (g.Np / i8::BN);
  const int64_t r256 = (t256 + CUS - 1) / CUS, r128 = (t128 + CUS - 1) / CUS;
  return r256 * 23 <= r128 * 20;
}

int i8_dispatch(const GemmArgs& g, int dtype, bool lowrank, int bout, hipStream_t st) {
  switch (dtype) {
    case LQER_F32: return i8::launch<LQER_F32>(g, lowrank, bout, st);
    case LQER_F16: return i8::launch<LQER_F16>(g, lowrank, bout, st);
    case LQER_BF16: return i8::launch<LQER_BF16>(g, lowrank, bout, st);
  }
  set_error("unknown dtype %d", dtype);
  return LQER_E_INVALID;
}

int i8_prepare_dispatch(const void* w_packed, int64_t N, int64_t K, int mbits, void* w_i8, int32_t* flags, hipStream_t st) {
  (void)mbits;  // (the exponent bytes of the sign-magnitude image are already biased by the mantissa width)
  const int64_t Np = lqer_padded_n(N);
  const int nk = (int)(lqer_padded_k(K) / 64), nk8 = (int)(padded_k8(K) / I8_BK);
  (void)hipMemsetAsync(flags, 0, 2 * sizeof(int32_t), st);
  i8::k_i8_rows<<<(unsigned)((Np + 255) / 256), 256, 0, st>>>((const uint8_t*)w_packed, N, Np, nk, nk8, (uint8_t*)w_i8, flags);
  const int64_t items = Np * nk8;
  i8::k_i8_codes<<<(unsigned)((items + 255) / 256), 256, 0, st>>>((const uint8_t*)w_packed, N, Np, nk, nk8, (uint8_t*)w_i8);
  return check_launch("lqer_i8_prepare");
}

int i8_unpack_dispatch(const void* w_i8, int64_t N, int64_t K, float* out, hipStream_t st) {
  const int64_t Np = lqer_padded_n(N);
  const int nk8 = (int)(padded_k8(K) / I8_BK);
  i8::k_i8_unpack<<<(unsigned)((N * K + 255) / 256), 256, 0, st>>>((const uint8_t*)w_i8, N, K, Np, nk8, out);
  return check_launch("lqer_unpack_weight_i8");
}

}  // namespace lqer
