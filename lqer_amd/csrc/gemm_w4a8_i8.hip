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
// Persistent tile loop: the grid is at most one workgroup per CU and a workgroup walks its tiles (virtual block id
// blockIdx.x + i * gridDim.x through the XCD-aware tile map).  With one 64-column panel of xAq (padded rank x limbs <= 64)
// the epilogue keeps out of ring slot 0, and the NEXT tile's first step is requested before the epilogue starts - its
// latency passes under the conversion / side product / stores of the current tile.
//
// Tile, ring and wave structure are those of gemm_w4a8_m256.hip: 256(m) x 256(n) per workgroup, 8 waves side by side
// along n, LDS-DMA into a 3-slot ring two steps ahead (one step = 128 k = 128 B per activation row, the same row
// pitch and swizzle), LOAD / COMPUTE ping-pong between the two waves of a SIMD, half a step (4 token tiles) per phase.
#include <atomic>
#include <type_traits>

#include "common.h"

namespace lqer {

namespace i8 {

constexpr int BN = 256;
constexpr int W_SLOT = I8_WBLOCK;    // 16640 B nibbles + shift bytes
constexpr int EP_OUT_WAVE = 64 * 80;  // epilogue, per wave: 64 rows x 80 B (32 fp16 columns + pad)
// Geometry by token tiles per workgroup (NT tiles of 32 rows).  NT = 8: 256 x 256 tiles, 3-slot ring two steps ahead (one
// step = 2048 cycles of MFMA issue per SIMD).  NT = 4: 128 x 256 tiles for token counts whose 256-row grid leaves CUs idle
// (Llama-7B projections at M = 2048: 128 tiles of 256 rows on 256 CUs) - a step is half as long, so the ring is one slot
// deeper (three steps ahead: the same prefetch distance in time).
template <int NT>
struct Geo {
  static constexpr int BM = 32 * NT;
  static constexpr int DEPTH = NT == 8 ? 2 : 3, NSLOT = DEPTH + 1;
  static constexpr int A_SLOT = BM * I8_BK;  // 32 / 16 KiB  int8 activation tile
  static constexpr int NPA = NT / 2;         // 1-KiB activation pieces per wave and step
  static constexpr int BATCH = NPA + 2;      // LDS-DMA loads per wave and step (wave 0: one more, the shift bytes)
  static constexpr int OFF_A = 0;
  static constexpr int OFF_W = NSLOT * A_SLOT;
  static constexpr int GEMM_LDS = OFF_W + NSLOT * W_SLOT;  // 148224 B (NT 8) / 132096 B (NT 4)
  // epilogue regions (the ring is free then)
  static constexpr int PANEL = BM * 128;   // one 64-column panel of the xAq tile (filled by LDS-DMA)
  static constexpr int EP_STAGE = 0;       // up to two panels
  static constexpr int EP_OUT = 2 * PANEL;
  static_assert(EP_OUT + 8 * EP_OUT_WAVE <= GEMM_LDS, "epilogue regions exceed the ring");
  // behind the ring, written at the start of a tile: per-row constants of the epilogue - the x row scales, the B_out
  // scales 2^(mbits - e[m]) and 2^(e[m] - mbits) - fp32 [256] each
  static constexpr int EP_TAB = GEMM_LDS;
  // ... and 1e-9 * 2^(mbits - e[m]), the B_out quantizer's epsilon after scaling; behind the tables 256 B per wave that nobody reads: where
  // the waves other than wave 0 point the ring fill's shift-byte request (issue_step: every wave issues the same number of requests,
  // without a branch)
  static constexpr int EP_DUMP = GEMM_LDS + 4 * 1024;
  static constexpr int KERNEL_LDS = EP_DUMP + 8 * 256;
  // XCH instantiation: the tile's one panel of xAq lives behind the tables from the prologue on (requested right behind the first ring
  // step, read by the epilogue's side product - nothing is staged after the main loop)
  static constexpr int EP_XAQ = KERNEL_LDS;
  // ... and the 4 KiB of gathered granules (32 row quads x 16 column tiles x {bytes, tag}), outside the ring: the gather is requested
  // from INSIDE the main loop, in the request slots of the first step past the end of K (step4)
  static constexpr int EP_GATHER = EP_XAQ + PANEL;
  static constexpr int KERNEL_LDS_XCH = EP_GATHER + 4096;
  static_assert(KERNEL_LDS_XCH <= 160 * 1024 || NT == 8, "LDS of the exchange instantiation");
  // one panel of xAq: the epilogue lives in ring slots 1.. (stage: activation slot 1; output transposes: the activation slots
  // behind it, waves 6 and 7 in weight slot 2), so that slot 0 can take the next tile's first step meanwhile
  static_assert(2 * A_SLOT + 6 * EP_OUT_WAVE <= NSLOT * A_SLOT && 2 * EP_OUT_WAVE <= W_SLOT && PANEL <= A_SLOT,
                "epilogue regions of the persistent loop");
  static_assert((NSLOT - 1) * A_SLOT + (NT - 1) * 4096 < 65536 || NT == 8, "activation fragment offsets fit the DS offset field");
};

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_g;  // a {value, tag} granule of the row-maxima exchange

// llvm.amdgcn.dispatch.id (see decode1.hip): the same value in every workgroup of a launch, a new one for every launch - replayed
// hipGraph nodes included
extern "C" __device__ unsigned long long lqer_dispatch_id() __asm("llvm.amdgcn.dispatch.id");
// The gather's FIRST read of the granules: sc0 alone (past the CU's L1, served by the XCD's L2).  The tile map puts a row band's column
// tiles on one XCD whenever a band has at most 32 of them, and the publisher's sc1 store goes through that same L2: a hit costs a
// fraction of the agent-scope round trip behind the L2.  A stale or foreign line just fails the tag test and is polled at agent scope.
#ifndef LQER_XCH_GATHER_AUX
#define LQER_XCH_GATHER_AUX 1
#endif
constexpr int XCH_GATHER_AUX = LQER_XCH_GATHER_AUX;
// polls of a missing granule (s_sleep 8 + an agent-scope re-read: ~1.5 us each) before the workgroup computes the band's maxima itself.
// Round 6: 6 instead of 64 - every workgroup publishes in its prologue and gathers ~25 us later, so a granule that is still missing
// belongs to a workgroup that is not resident (another stream holds its CU): waiting longer than the fall-back costs (the band's side
// products: tiles_n x 16 MFMAs, a few us) buys nothing, and every late tile paid the full bound
#ifndef LQER_XCH_SWEEPS
#define LQER_XCH_SWEEPS 6
#endif
constexpr int XCH_SWEEPS = LQER_XCH_SWEEPS;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ int swz(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

#ifdef LQER_CLOCKPROBE
// diagnostic build (tools/clock_probe_i8.py): shader cycles (s_memtime) and 100 MHz ticks (s_memrealtime) at the start of the
// kernel, around the main loop and at the end, written to a buffer nothing else reads
__device__ unsigned long long* g_i8_stamp_buf = nullptr;
#define I8_STAMP(c, r) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c), "=s"(r)::"memory")
#endif

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

// per weight row: group exponents -> base exponent, shifts, row scale; i32 range and format checks (flags[0] != 0: not
// eligible).  One workgroup = one n tile of 256 rows: the tile's MODE (common.h I8_MODE_*) is decided here - PRESHIFT when no
// row of the tile spreads its group exponents over more than 4 binades (the int8 lane cw << (4 - q) then carries the group's
// exponent itself and the main loop accumulates without folds), else FOLD; NONE when no group of the tile differs from its row.
__global__ __launch_bounds__(256) void k_i8_rows(const uint8_t* __restrict__ wp, int64_t N, int64_t Np, int nk, int nk8,
                                                  uint8_t* __restrict__ img, int32_t* __restrict__ flags) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (Np is a multiple of 256: every thread owns a row)
  float* wscale = (float*)(img + (size_t)(Np / 256) * nk8 * I8_WBLOCK);
  uint8_t* mode_tab = img + i8_weight_mode_offset(Np, nk8);
  const int64_t tn = n / 256;
  const int rl = (int)(n - tn * 256);
  int bmin = 255, bmax = 0;
  bool bad = false;
  // pass 1: the biased exponent byte of every 128-k group (0 = no non-zero code in it), minimum and maximum over the row
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
    if (bg > bmax) bmax = bg;
    img[(tn * nk8 + s) * I8_WBLOCK + 256 * 64 + rl] = (uint8_t)bg;  // (the byte for now; the shift in pass 2)
  }
  const bool row_any = bmin != 255 && bmax != bmin;
  const int tile_wide = __syncthreads_or(bmin != 255 && bmax - bmin > 4);
  const int tile_two = __syncthreads_or(bmin != 255 && bmax - bmin > 1);
  const int tile_any = __syncthreads_or(row_any);
  const int mode = !tile_any ? I8_MODE_NONE : (tile_wide ? I8_MODE_FOLD : (tile_two ? I8_MODE_PRESHIFT : I8_MODE_PRESHIFT1));
  if (threadIdx.x == 0) mode_tab[tn] = (uint8_t)mode;
  // pass 2: shift bytes and the i32 bound  sum_g sum_k |lane| * 127 < 2^31  (lane = 16 c 2^s, or c 2^(4-q))
  unsigned long long bound = 0;
  for (int s = 0; s < nk8; ++s) {
    uint8_t* sp = img + (tn * nk8 + s) * I8_WBLOCK + 256 * 64 + rl;
    const int bg = *sp;
    int sh = 0;
    if (bg) {
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
      if (mode == I8_MODE_PRESHIFT || mode == I8_MODE_PRESHIFT1) {
        sh = bmax - bg;  // q: the lane is cw << (4 - q)
        bound += ((unsigned long long)asum << (4 - sh)) * 127ull;
      } else {
        sh = bg - bmin;
        if (sh > 20) bad = true;
        else bound += ((unsigned long long)asum << sh) * (16ull * 127ull);
      }
    }
    *sp = (uint8_t)sh;
  }
  if (bound >= (1ull << 31)) bad = true;
  const int bbase = (mode == I8_MODE_PRESHIFT || mode == I8_MODE_PRESHIFT1) ? bmax : bmin;  // value = lane * 2^(bbase - 4 - 127)
  if (bmin != 255 && bbase < 5) bad = true;  // row scale 2^(e - mbits - 4) would not be a normal float
  wscale[n] = bmin == 255 ? 0.0f : __uint_as_float((uint32_t)(bbase - 4) << 23);
  if (bad) atomicOr(flags, 1);
  if (row_any) atomicOr(flags + 1, 1);
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
  const int mode = img[i8_weight_mode_offset(Np, nk8) + tn];
  const int rl = (int)(n - tn * 256), s = (int)(k / 128), kk = (int)(k % 128);
  const uint8_t* blk = img + (tn * nk8 + s) * I8_WBLOCK;
  const int W = kk >> 3, q = kk & 7;
  const uint32_t w = *(const uint32_t*)(blk + w_byte_offset(rl, W));
  const int p = q < 4 ? 2 * q : 2 * (q - 4) + 1;
  int c = (int)((w >> (4 * p)) & 0xfu);
  c = c >= 8 ? c - 16 : c;
  const int sh = blk[256 * 64 + rl];
  // the int8 lane the main loop multiplies: 16 c 2^s (FOLD / NONE: s applied to the group sum), c 2^(4-q) (PRESHIFT)
  out[idx] = (float)((mode == I8_MODE_PRESHIFT || mode == I8_MODE_PRESHIFT1) ? c * (1 << (4 - sh)) : 16 * c * (1 << sh)) * wscale[n];
}


// ---- 8-bit weights (W8): the int8 image holds the CODES ------------------------------------------------------------------------
// Source: the three 4-bit limb images of lqer_pack_weight_mxint (pack.hip: m = 64 a + 8 b + c, limb l at panel column l Kp/64 + pk,
// limb 2's exponent byte = e - mbits + 127).  Image: FRAGMENT-MAJOR - per (n tile of 256 rows, step of 128 k) 32 KiB laid out as
// [wave = row / 32][32-k slice 0..3][lane = row % 32 + 32 (k % 32 / 16)][16 codes]: the 1 KiB a wave's MFMA B-operand of one slice is,
// in lane order, so that one 16-byte load per lane fetches it fully coalesced (NT = 4: straight into registers) and an LDS copy of
// it is read back linearly (NT = 8); then the row scales 2^(e[n] - mbits) fp32 [Np].
__host__ __device__ inline size_t w8_image_offset(int64_t n, int64_t k, int nk8) {
  const int64_t tn = n >> 8;
  const int rl = (int)(n & 255), kk = (int)(k & 127);
  return ((size_t)tn * nk8 + (size_t)(k >> 7)) * 32768 + (size_t)(rl >> 5) * 4096 + (size_t)(kk >> 5) * 1024 +
         (size_t)((rl & 31) + 32 * ((kk & 31) >> 4)) * 16 + (size_t)(kk & 15);
}
__device__ __forceinline__ int w8_code(const uint8_t* wp, int64_t n, int nk, int k) {  // k < 64 nk
  const int kp = k >> 6, kk = k & 63, seg = kk >> 4, i = kk & 15, j = i & 7, pos = j < 4 ? 2 * j : 2 * (j - 4) + 1;
  int m = 0;
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    const uint8_t* pnl = wp + ((n / 16) * (3 * (int64_t)nk) + (int64_t)l * nk + kp) * LQER_PANEL_BYTES;
    const uint32_t word = *(const uint32_t*)(pnl + (n & 15) * 32 + (i < 8 ? 0 : 16) + seg * 4);
    const uint32_t nib = (word >> (4 * pos)) & 0xfu;
    const int d = (nib & 8u) ? -(int)(nib & 7u) : (int)nib;
    m = m * 8 + d;
  }
  return m;
}

// per weight row: one exponent for the whole row (else flags[0]: not eligible - a row with several block exponents keeps the
// limb route), the i32 bound sum_k |code| 127 < 2^31, the row scale
__global__ __launch_bounds__(256) void k_i8_rows8(const uint8_t* __restrict__ wp, int64_t N, int64_t Np, int nk, int nk8,
                                                   uint8_t* __restrict__ img, int32_t* __restrict__ flags) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float* wscale = (float*)(img + (size_t)(Np / 256) * nk8 * 2 * (256 * 64));
  int bb = 0;
  bool bad = false;
  unsigned long long asum = 0;
  if (n < N) {
    for (int kp = 0; kp < nk; ++kp) {
      const uint8_t* p2 = wp + ((n / 16) * (3 * (int64_t)nk) + 2 * (int64_t)nk + kp) * LQER_PANEL_BYTES;  // limb 2: exponent e - mbits
      for (int seg = 0; seg < 4; ++seg) {
        unsigned a16 = 0;
        for (int i = 0; i < 16; ++i) {
          const int c = w8_code(wp, n, nk, kp * 64 + seg * 16 + i);
          a16 += (unsigned)(c < 0 ? -c : c);
        }
        if (!a16) continue;
        asum += a16;
        const int eb = p2[512 + (n & 15) * 4 + seg];
        if (bb == 0) bb = eb;
        else if (bb != eb) bad = true;
      }
    }
  }
  if (asum * 127ull >= (1ull << 31)) bad = true;
  if (bb && bb < 2) bad = true;  // (the scale must be a normal float)
  wscale[n] = bb ? __uint_as_float((uint32_t)bb << 23) : 0.0f;
  if (bad) atomicOr(flags, 1);
}

// one thread per (row, 16-k chunk): 16 codes -> 16 bytes
__global__ __launch_bounds__(256) void k_i8_codes8(const uint8_t* __restrict__ wp, int64_t N, int64_t Np, int nk, int nk8,
                                                    uint8_t* __restrict__ img) {
  const int nh = 2 * nk8;  // half-steps of 64 k in the image (>= nk: the image is padded to 128 k)
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= Np * nh * 4) return;
  const int64_t n = idx / (nh * 4);
  const int rem = (int)(idx - n * (nh * 4)), h = rem >> 2, c = rem & 3;
  const int64_t tn = n / 256;
  const int rl = (int)(n - tn * 256);
  u32x4 out = {0, 0, 0, 0};
  if (n < N && h < nk) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const uint32_t b = (uint32_t)(uint8_t)(int8_t)w8_code(wp, n, nk, h * 64 + c * 16 + i);
      out[i >> 2] |= b << (8 * (i & 3));
    }
  }
  (void)tn, (void)rl;
  *(u32x4*)(img + w8_image_offset(n, (int64_t)h * 64 + c * 16, nk8)) = out;
}

// test hook: the image back to dequantized fp32 [N,K]
__global__ __launch_bounds__(256) void k_i8_unpack8(const uint8_t* __restrict__ img, int64_t N, int64_t K, int64_t Np, int nk8,
                                                     float* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * K) return;
  const int64_t n = idx / K, k = idx - n * K;
  const int nh = 2 * nk8;
  const float* wscale = (const float*)(img + (size_t)(Np / 256) * nh * (256 * 64));
  const int8_t code = (int8_t)img[w8_image_offset(n, k, nk8)];
  out[idx] = (float)code * wscale[n];
}

// ---- the GEMM -----------------------------------------------------------------------------------------------------------
// BOUT: 0 pass-through, 2 one block per row (exponent from the k_bout_amax pre-pass).  SHIFT: the weight's blocks are shorter
// than its rows - every n tile then says in its MODE byte how its group exponents travel (k_i8_rows): PRESHIFT (the lane
// cw << (4 - q) carries them: 11 vector instructions per 8 weights - 7 when no row of the tile spreads over more than two
// binades, PRESHIFT1 -, no folds), FOLD (one v_lshl_add_u32 per output element and group), or NONE.  The main loops live in one
// kernel; the choice is uniform per tile.
// W8 (no SHIFT): 8-bit weights (the reference's W8A8 baseline, sweep_baseline_no_lqer.sh:73-76) - the image holds the int8 codes
// themselves, fragment-major, one exponent per row: no expand at all.  A 128-k step of codes is 32 KiB per tile - twice the nibbles.
//   NT = 8: the weight ring runs at HALF-step granularity - three slots of 256 rows x 64 k (16 KiB, in the 48 KiB the nibble ring
//           occupies), the half-step h = 2 kt + P in slot h % 3, requested two half-steps ahead; the weight fragment of a slice
//           is one 16-byte LDS read; the activation ring stays as it is (three 128-k slots, two steps ahead).
//   NT = 4: a wave's weight rows are nobody else's - its codes skip LDS: four coalesced 16-byte loads per lane and step into one of
//           four register sets, three steps ahead beside the activation ring (step4_w8).  35.8 us per round of 4096-k tiles on
//           all 256 CUs where 256-row tiles fill half of them in 63 us (M = 2048, 4096 x 4096); 1,500 cycles per step.
// XCH (round 6): the instantiation for ONE round of 128-row tiles that exchanges the B_out row maxima inside the launch (what used to be
// the runtime flag g.bout_xch) - no tile loop, every operand of the epilogue requested in the prologue (the xAq panel in an LDS region
// of its own behind the row tables, the wave's B^T fragments, column scale and bias in registers across the main loop), the first ring
// step requested before anything else.
// MRX (round 6): the same exchange for grids of SEVERAL rounds of 128-row tiles (2048 x 4096 -> 11008: 688 tiles on 256 resident
// workgroups), where the maxima of a row band are needed by its first epilogue but its tiles run in different rounds.  Every workgroup
// starts with ONE item of the pre-pass - (row band, sixteenth of the columns): the band's xAq panel into LDS, xch_load_s / xch_compute_s
// over its two or three column tiles, the waves' maxima through LDS - and publishes {maximum, tag} granules [row][16 segments] (sc1 stores;
// the launch's tag as in the one-round exchange: nothing is zeroed).  A tile's row constants are folded from its band's 16 granules per
// row (sc1 loads, a tile ahead as before; the first tile's behind its main loop); where a tag is missing the workgroup polls, bounded,
// and then computes the band's sixteen items itself (same routine, same bits).  Replaces the k_bout_amax launch (7.7 us + its zero fill).
template <int DT, bool LOWRANK, int BOUT, bool SHIFT, int NT, bool W8 = false, bool XCH = false, bool MRX = false>
__global__ __launch_bounds__(512) void k_lqer_gemm_i8(GemmArgs g) {
  static_assert(!MRX || (!XCH && !W8 && LOWRANK && BOUT == 2 && NT == 4), "multi-round exchange: 4-bit weights on 128-row tiles, one B_out block per row");
  static_assert(!W8 || !SHIFT, "8-bit weight codes: one exponent per weight row");
  static_assert(!XCH || (NT == 4 && LOWRANK && BOUT == 2), "the in-launch exchange: 128-row tiles, one B_out block per row");
  constexpr bool W8D = W8 && NT == 4;  // ... on 128-row tiles: the codes go straight from global memory into registers (step4_w8)
  constexpr int W8_SLOT = 256 * 64;  // one half-step of int8 codes
  using G = Geo<NT>;
  constexpr int BM = G::BM, DEPTH = G::DEPTH, NSLOT = G::NSLOT, A_SLOT = G::A_SLOT, NPA = G::NPA, OFF_A = G::OFF_A, OFF_W = G::OFF_W;
  constexpr int PANEL = G::PANEL, EP_STAGE = G::EP_STAGE, EP_OUT = G::EP_OUT, EP_TAB = G::EP_TAB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane_k = threadIdx.x & 63;
  const int wave_k = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef LQER_CLOCKPROBE
  unsigned long long cp_c[4], cp_r[4], cp_e1 = 0, cp_e1r = 0, cp_e2 = 0, cp_e2r = 0, cp_a = 0, cp_b = 0, cp_cc = 0, cp_x = 0;
  unsigned long long cp_p[4] = {0, 0, 0, 0}, cp_tries = 0;  // round 6: prologue sections (first request out, all requests out, row maxima computed, published)
  I8_STAMP(cp_c[0], cp_r[0]);
#endif

  if constexpr (XCH || MRX) {
    // every kernel argument the prologue and the epilogue read, asked for in ONE batch of scalar loads (an empty asm that names them all:
    // they must be in registers here, so the loads go out together and are awaited once) - left to itself hipcc requested them where
    // first used, five dependent scalar round trips in front of the first ring request
    asm volatile("" ::"s"(g.xq), "s"(g.w8), "s"(g.xaq), "s"(g.bt), "s"(g.bias), "s"(g.xscale), "s"(g.bout_amax), "s"(g.y), "s"(g.ldy), "s"(g.M),
                 "s"(g.N), "s"(g.Np), "s"(g.Kp), "s"(g.rp), "s"(g.b_limbs), "s"(g.xaq_ld), "s"(g.tiles_m), "s"(g.tiles_n), "s"(g.xch_nonce),
                 "s"(g.tuning), "s"(g.bout.mbits), "s"(g.bout.emin), "s"(g.bout.emax));
  }
  const int nt = g.tiles_m * g.tiles_n;
  // XCD-aware tile order of a virtual block id (blocks b, b + 8, ... share an XCD; the grid is a multiple of 8 or covers nt)
  // g.xcd_bm > 0 (round 6, LQER_TUNE_XCD_BLOCK; host-checked: nt % 8 == 0 and the grid divides): an XCD's nt / 8 tiles form a BLOCK of
  // xcd_bm token tiles x (nt / 8 / xcd_bm) weight tiles instead of whole rows of weight tiles - 16 x 16 tiles as 4 x 8 blocks read
  // 2 + 4.2 MB per XCD instead of 1 + 8.5 (K = 4096)
  auto tile_of = [&](int b) {
    const int xcd = b & 7, q8 = nt >> 3, r8 = nt & 7;
    if (g.xcd_bm > 0) {
      const int bn = q8 / g.xcd_bm, gx = g.tiles_n / bn;  // block width in weight tiles; XCD blocks per row of blocks
      const int xr = xcd / gx, xc = xcd - xr * gx, i = b >> 3, r = i / bn, c = i - r * bn;
      return (xr * g.xcd_bm + r) * g.tiles_n + xc * bn + c;
    }
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
  };
  const int Kp8 = g.Kp;  // (the int8 image's row stride)
  const int nk = Kp8 / I8_BK;
  const uint8_t* const xq8 = (const uint8_t*)g.xq;
  int vb = blockIdx.x;
  int tile = tile_of(vb);
  int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  int m0 = tm * BM, n0 = tn * BN;

  // with ONE 64-column panel of xAq the epilogue lives in ring slots 1 and 2 (stage: activation slot 1; output transposes:
  // activation slot 2, waves 6 and 7 in weight slot 2), so that slot 0 can take the next tile's first step meanwhile
  const bool one_panel = XCH || !LOWRANK || g.rp <= 64;  // (wave-uniform)
  // ---- exchange of the B_out row maxima inside the launch (one round of 128-row tiles, one block per row: no pre-pass) ------------
  // Every workgroup computes, in its prologue (under the ring fill), the row maxima of ITS tile's side product - k_bout_amax's
  // arithmetic and order for these 256 columns - and publishes their block exponents, four rows per {bytes, tag} granule [row quad][tn]
  // (one 8-byte sc1 store: bytes and tag arrive together, nothing to zero, nothing to fence); at the epilogue the 128 row threads
  // gather the tiles_n granules of their quad (sc1, requested before the conversion pass, polled until the tags match) and take the
  // largest exponent - the exponent of a maximum is the maximum of the exponents: the bits of the pre-pass.  The tag is the launch's nonce (host counter + dispatch id + queue: a replayed graph node
  // gets a fresh one).  Nobody has to wait for anybody: a workgroup that does not see a neighbour's granules in time (a grid that is
  // not resident at once: another stream's kernel on the CUs) computes the whole band's maxima itself, same routine.
  constexpr bool XCH_OK = XCH || MRX;  // (the pre-phase routines: xch_load_s / xch_compute_s / xch_reduce)
  constexpr bool xch = XCH;
  uint32_t xtag = 0;
  if constexpr (XCH || MRX) {
    // host call counter, dispatch id and queue are mixed INDEPENDENTLY (distinct odd multipliers, then a murmur-style finaliser), with
    // the launch's shape on top: two launches share a tag only by a 2^-32 accident, never by calls + dispatch id adding up alike
    uint32_t h = g.xch_nonce ^ ((uint32_t)lqer_dispatch_id() * 0xC2B2AE35u) ^
                 ((uint32_t)((unsigned long long)__builtin_amdgcn_queue_ptr() >> 6) * 0x85EBCA6Bu) ^ ((uint32_t)g.M * 0x27D4EB2Fu) ^
                 ((uint32_t)g.Np * 0x165667B1u);
    h ^= h >> 16, h *= 0x7FEB352Du, h ^= h >> 15, h *= 0x846CA68Bu, h ^= h >> 16;
    xtag = h;
  }
  const int64_t xch_Mp = (int64_t)(g.M + LQER_M_ALIGN - 1) / LQER_M_ALIGN * LQER_M_ALIGN;
  const int ep_stage = XCH ? G::EP_XAQ : (one_panel ? A_SLOT : EP_STAGE);
  float t_xs = 0.f, t_amax = 0.f;  // this lane's row constants of the tile whose tables are written next
  uint32_t t_bad = 0;              // MRX: some granule of that row did not carry this launch's tag
  int mrx_epoch = 0;               // MRX: votes of the workgroup (wg_any)
  bool first = true;
  // the tile's MODE byte travels as the aligned dword around it, requested a tile ahead (here: the first tile's, a scalar load - nothing
  // has been stored yet; the next tile's where the epilogue requests its row constants) and looked at only where the main loop is
  // chosen: a byte load at the head of the tile sat, with its vmcnt(0), in front of the ring fill
  // XCH: this wave's B^T fragments (the side product's B operand, <= 2 limbs x 4 slices), its column's scale and bias: requested once, in
  // the prologue, for the row-maxima pre-phase AND the epilogue
  bf16x8 sbx[XCH_OK ? 8 : 1];
  bf16x8 mfr[MRX ? 12 : 1];  // MRX: the B^T fragments of one batch of an item's column tiles (3 tiles x <= 4 fragments, or 1 x 8)
  float ws_x = 0.f, bv_x = 0.f;
  uint32_t mode_word = 0, mode_sh = 0;
  auto load_mode = [&](int tn_) {
    if constexpr (SHIFT) {
      // (the table starts on a 4-byte boundary: pointer arithmetic on the kernel argument keeps the load a GLOBAL one - through an
      // integer cast it becomes a flat load, behind which hipcc's waitcnt pass drains everything)
      mode_word = ((const uint32_t*)(g.w8 + i8_weight_mode_offset(g.Np, nk)))[tn_ >> 2];
      mode_sh = 8u * (uint32_t)(tn_ & 3);
    }
  };
  load_mode(tn);

  for (;;) {  // ---- tiles of this workgroup ----------------------------------------------------------------------------
  // Per-lane constants are re-derived for every tile from laundered ids (an empty asm the optimiser cannot look through):
  // kept in registers across the epilogue - which needs all 256 - they would be spilled to scratch.
  int lane_l = lane_k, wave_l = wave_k;
  asm volatile("" : "+v"(lane_l));
  asm volatile("" : "+s"(wave_l));
  const int lane = lane_l, wave = wave_l, tid = wave * 64 + lane;
  const int l31 = lane & 31, lh = lane >> 5;
  // ---- staging addresses (per lane, fixed for the kernel) ------------------------------------------------------------
  // activations: wave w stages rows [8 NPA w, 8 NPA (w + 1)) of the tile as NPA pieces of 8 rows x 128 B (NT = 8: 4 pieces, 32 rows;
  // NT = 4: 2 pieces, 16 rows), chunk-swizzled on the source side
  int a_voff[4] = {0, 0, 0, 0};  // (a fixed extent: a template-dependent one, captured by the lambdas below, loses hipcc the host stub)
#pragma unroll
  for (int i = 0; i < NPA; ++i) {
    const int row = wave * (8 * NPA) + i * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    a_voff[i] = row * Kp8 + chunk * 16;
  }
  // weights: the block's 16 KiB of nibbles are 16 contiguous 1-KiB pieces: wave w stages pieces 2w and 2w+1; wave 0 also the
  // 256 shift bytes (4 B per lane)
  // (W8: the wave's own 4 KiB of the step's fragment-major block - slices 2 P and 2 P + 1 of half-step P are its pieces 2 P, 2 P + 1)
  const int w_voff0 = (W8 ? wave * 4096 : (2 * wave) * 1024) + lane * 16, w_voff1 = w_voff0 + 1024, s_voff = 256 * 64 + lane * 4;
  const uint8_t* const a_base = xq8 + (int64_t)m0 * Kp8;
  const uint8_t* const w_base = g.w8 + (size_t)tn * nk * (W8 ? 2 * W8_SLOT : I8_WBLOCK);
  auto make_rs = [](const uint8_t* base, uint32_t range) {
    const unsigned long long b64 = (unsigned long long)base;
    return (u32x4){(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b64),
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b64 >> 32)) & 0xffffu, range, 0x00020000u};
  };
  const u32x4 a_rs = make_rs(a_base, (uint32_t)(BM * Kp8)), w_rs = make_rs(w_base, (uint32_t)(nk * (W8 ? 2 * W8_SLOT : I8_WBLOCK)));
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem;
  const uint32_t m0_a = lds0 + OFF_A + wave * (8 * NPA) * 128;  // + slot * A_SLOT + piece * 1024
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
  // (fragment-major image: the wave reads back, lane by lane, the two 1-KiB pieces it copied itself)
  const uint32_t fw8_0 = lds0 + OFF_W + (2 * wave) * 1024 + lane * 16;
  const uint32_t fw8_1 = fw8_0 + 1024;

  // one LDS-DMA batch = the operands of one step: NPA + 2 loads per wave (wave 0: one more)
  auto issue_step = [&](const uint8_t* ab, const uint8_t* wb, int kt, int slot) {
    const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ab, 0, BM * Kp8, 0x00020000);
    const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, nk * I8_WBLOCK, 0x00020000);
#pragma unroll
    for (int i = 0; i < NPA; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_void*)(smem + OFF_A + slot * A_SLOT + wave * (8 * NPA) * 128 + i * 1024), 16,
                                               a_voff[i], kt * I8_BK, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(smem + OFF_W + slot * W_SLOT + (2 * wave) * 1024), 16, w_voff0,
                                             kt * I8_WBLOCK, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(smem + OFF_W + slot * W_SLOT + (2 * wave + 1) * 1024), 16, w_voff1,
                                             kt * I8_WBLOCK, 0, 0);
    // the 256 shift bytes are wave 0's; the other waves issue the same request against an EMPTY range into a dump area of their own.
    // No branch: hipcc's waitcnt pass counts the requests behind a load only back to the last control-flow join - with a branch per step
    // it waited for the whole ring fill (vmcnt(0)) wherever the prologue uses anything it has loaded
    const auto s_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, wave == 0 ? nk * I8_WBLOCK : 0, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(s_rsrc, (lds_void*)(smem + (wave == 0 ? OFF_W + slot * W_SLOT + 256 * 64 : G::EP_DUMP + wave * 256)), 4,
                                             s_voff, kt * I8_WBLOCK, 0, 0);
  };
  // W8: the two operands have their own cadence
  auto issue_a8 = [&](const uint8_t* ab, int kt, int slot) {
    const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ab, 0, BM * Kp8, 0x00020000);
#pragma unroll
    for (int i = 0; i < NPA; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_void*)(smem + OFF_A + slot * A_SLOT + wave * (8 * NPA) * 128 + i * 1024), 16,
                                               a_voff[i], kt * I8_BK, 0, 0);
  };
  auto issue_w8 = [&](const uint8_t* wb, int h, int ws) {  // half-step h of the tile's codes -> weight slot ws
    const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, nk * 2 * W8_SLOT, 0x00020000);
    const int soff = (h >> 1) * (2 * W8_SLOT) + (h & 1) * 2048;  // step h / 2, the wave's pieces 2 (h % 2) and 2 (h % 2) + 1
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(smem + OFF_W + ws * W8_SLOT + (2 * wave) * 1024), 16, w_voff0, soff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(smem + OFF_W + ws * W8_SLOT + (2 * wave + 1) * 1024), 16, w_voff1, soff, 0, 0);
  };
  // W8 on 128-row tiles (W8D): a wave's weight rows are nobody else's, so its codes skip LDS - four 16-byte loads per lane and step
  // from the fragment-major image (1 KiB per instruction, fully coalesced) into one of four register sets, three steps ahead.  These
  // loads and the activation LDS-DMA of the same loop are BUILTINS, not asm: hipcc's waitcnt pass then counts every request and its
  // own wait in front of the MFMAs that read a set is exact (loads return in issue order; the manual vmcnt at the end of LOAD, for
  // the activations everybody reads, has retired the set by then anyway).
  i32x4 wq[W8D ? 4 : 1][4];
  const int wq_voff = wave * 4096 + lane * 16;
  auto issue_wq = [&](const uint8_t* wb, int kt, auto set_c) {
    if constexpr (W8D) {
      constexpr int S = decltype(set_c)::value;
      const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, nk * 2 * W8_SLOT, 0x00020000);
#pragma unroll
      for (int sl = 0; sl < 4; ++sl)
        wq[S][sl] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wq_voff + sl * 1024, kt * (2 * W8_SLOT), 0));
    }
  };
  // per-row constants of the epilogue: requested ahead, written to LDS (asm: invisible to hipcc's waitcnt pass, which would
  // drain the ring fill in front of a visible LDS store) once the ring fill has been issued
  auto load_tables = [&](int m0_) {
    if (tid < BM) {
      t_xs = g.xscale[m0_ + tid];
      if constexpr (LOWRANK && BOUT == 2) {
        if (xch) {  // (gathered at the epilogue)
          t_amax = 1.0f;
        } else if constexpr (MRX) {
          // the row's 16 {maximum, tag} granules: 128 contiguous bytes, agent scope (published inside this launch by other workgroups)
          const int64_t Mp = (int64_t)(g.M + LQER_M_ALIGN - 1) / LQER_M_ALIGN * LQER_M_ALIGN;
          const auto m_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g.bout_amax, 0, (int)(Mp * LQER_AMAX_NSEG * 8), 0x00020000);
          i32x4 q[LQER_AMAX_NSEG / 2];
#pragma unroll
          for (int j = 0; j < LQER_AMAX_NSEG / 2; ++j)
            q[j] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(m_rsrc, (int)(((m0_ + tid) * LQER_AMAX_NSEG + 2 * j) * 8), 0, 16));
          float m = 0.f;
          uint32_t bad = 0;
#pragma unroll
          for (int j = 0; j < LQER_AMAX_NSEG / 2; ++j) {
            m = fmaxf(m, fmaxf(__int_as_float(q[j][0]), __int_as_float(q[j][2])));
            bad |= ((uint32_t)q[j][1] ^ xtag) | ((uint32_t)q[j][3] ^ xtag);
          }
          t_amax = m, t_bad = bad;
        } else if (g.bout_nseg > 0) {  // the pre-pass left one partial per column segment (no atomics, no zero-fill): fold them
          const int64_t Mp = (int64_t)(g.M + LQER_M_ALIGN - 1) / LQER_M_ALIGN * LQER_M_ALIGN;
          // (all cells requested at once - a runtime loop waits for every load in turn, 16 round trips in front of the ring
          // fill; segments past the last one re-read it: max does not mind)
          const float* const cell = g.bout_amax + m0_ + tid;
          const int last = g.bout_nseg - 1;
          auto fold = [&](auto n_c) {
            constexpr int NC = decltype(n_c)::value;
            float v[NC];
#pragma unroll
            for (int sgi = 0; sgi < NC; ++sgi) v[sgi] = cell[(int64_t)(sgi < last ? sgi : last) * Mp];
            float m = v[0];
#pragma unroll
            for (int sgi = 1; sgi < NC; ++sgi) m = fmaxf(m, v[sgi]);
            return m;
          };
          const float m = g.bout_nseg > LQER_AMAX_NSEG ? fold(std::integral_constant<int, LQER_AMAX_NSEG_WIDE>{})
                                                       : fold(std::integral_constant<int, LQER_AMAX_NSEG>{});
          t_amax = m;
        } else {
          t_amax = g.bout_amax[(int64_t)(m0_ + tid) * g.bout_nblk];
        }
      }
    }
  };
  auto write_tables = [&]() {
    if (tid < BM) {
      const uint32_t ta = lds0 + EP_TAB + 4 * tid;
      asm volatile("ds_write_b32 %0, %1" ::"v"(ta), "v"(t_xs) : "memory");
      if constexpr (LOWRANK && BOUT == 2) {
        // both B_out scales as normal floats: mbits - e lies in [-121, 134] for an 8-bit exponent field; rows with
        // e < mbits - 126 have |s| < 2^-119 <= 1e-8 everywhere (a zero row: e = -127), i.e. every element takes the
        // pass-through whatever the scale - clamp, the result does not change
        int up = g.bout.mbits - block_exponent(t_amax, g.bout);
        up = up > 126 ? 126 : (up < -126 ? -126 : up);
        const uint32_t upb = (uint32_t)(127 + up) << 23;
        asm volatile("ds_write_b32 %0, %1 offset:1024\n\tds_write_b32 %0, %2 offset:2048\n\tds_write_b32 %0, %3 offset:3072" ::"v"(ta),
                     "v"(upb), "v"((uint32_t)(127 - up) << 23), "v"(1e-9f * __uint_as_float(upb))
                     : "memory");
      }
    }
  };
  // exchange: the side product of column tile `tnx` against this workgroup's 128 rows, this wave's 32 columns - operands straight from
  // global memory in k_bout_amax's layout and summation order (limb-major, slices ascending; B^T fragment as the A operand, so that
  // every register of a lane's accumulator is the same token row), folded into the running row maxima mx[u] (rows 32 u + l31)
  const int xch_nsl = XCH_OK ? g.rp / 16 : 0;
  auto xch_load_s = [&](int tnx, auto nl_c, auto nsl_c) {  // static (limbs, slices): straight-line requests
    if constexpr (XCH_OK) {
      constexpr int NL = decltype(nl_c)::value, NSL = decltype(nsl_c)::value;
      const bf16_t* const bl = g.bt + (int64_t)(tnx * BN + wave * 32 + l31) * g.rp + 8 * lh;
      const int64_t limb = (int64_t)g.Np * g.rp;
#pragma unroll
      for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int ks = 0; ks < NSL; ++ks) sbx[l * NSL + ks] = *(const bf16x8*)(bl + l * limb + ks * 16);
    }
  };
  auto xch_compute_s = [&](float (&mx)[4], auto nl_c, auto nsl_c) {
    if constexpr (XCH_OK) {
      constexpr int NL = decltype(nl_c)::value, NSL = decltype(nsl_c)::value;
      // the xAq fragments of the 4 row groups come from the tile's LDS panel (requested once per workgroup, 16 KiB, where round 5 had
      // every wave fetch all 128 rows itself - 64 KiB per workgroup through a texture path that the ring fill needs); asm reads: an LDS
      // access hipcc can see waits for every LDS-DMA in flight
      bf16x8 pxf[4][NSL];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int ks = 0; ks < NSL; ++ks)
          asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(pxf[u][ks]) : "v"(lds0 + G::EP_XAQ + swz(l31, 2 * ks + lh)), "i"(u * 4096) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pxf[0][0])::"memory");
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int ks = 0; ks < NSL; ++ks) asm volatile("" : "+v"(pxf[u][ks]));
      // (the four row groups' chains of NL x NSL dependent MFMAs run interleaved - each accumulator keeps k_bout_amax's order, limb-major,
      // slices ascending -: one chain after the other left the matrix pipe waiting out every MFMA's latency, ~1,300 cycles in front of
      // the main loop)
      f32x16 acc4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc4[u] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int ks = 0; ks < NSL; ++ks)
#pragma unroll
          for (int u = 0; u < 4; ++u) acc4[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sbx[l * NSL + ks], pxf[u][ks], acc4[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const f32x16 acc = acc4[u];
        float m = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k += 2) m = fmaxf(fmaxf(m, fabsf(acc[k])), fabsf(acc[k + 1]));
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);  // lanes l and l ^ 32
        mx[u] = fmaxf(mx[u], fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])));
      }
    }
  };
  const int xch_key = XCH_OK ? g.b_limbs * 16 + xch_nsl : 0;  // (wave-uniform; i8_amax_exchange_ok admits exactly six cases)
  // ... the waves' maxima through LDS (asm: an LDS access hipcc can see would wait for every LDS-DMA in flight): [wave][row] fp32 in
  // activation slot 3 - free in the prologue until the first LOAD requests step 3, and at the epilogue until the output transposes
  const uint32_t xch_red = lds0 + OFF_A + 3 * A_SLOT;
  auto xch_reduce = [&](const float (&mx)[4]) -> float {  // -> the row maximum of row `tid` (threads tid < BM)
    float r = 0.f;
    if constexpr (XCH_OK) {
      if (lh == 0) {
        const uint32_t a = xch_red + (wave * BM + l31) * 4;
        asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:128\n\tds_write_b32 %0, %3 offset:256\n\tds_write_b32 %0, %4 offset:384" ::"v"(a),
                     "v"(mx[0]), "v"(mx[1]), "v"(mx[2]), "v"(mx[3])
                     : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (tid < BM) {
        const uint32_t a = xch_red + tid * 4;
        float v0, v1, v2, v3, v4, v5, v6, v7;
        asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:512\n\tds_read_b32 %2, %8 offset:1024\n\tds_read_b32 %3, %8 offset:1536\n\t"
                     "ds_read_b32 %4, %8 offset:2048\n\tds_read_b32 %5, %8 offset:2560\n\tds_read_b32 %6, %8 offset:3072\n\t"
                     "ds_read_b32 %7, %8 offset:3584\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
                     : "v"(a)
                     : "memory");
        r = fmaxf(fmaxf(fmaxf(v0, v1), fmaxf(v2, v3)), fmaxf(fmaxf(v4, v5), fmaxf(v6, v7)));
      }
    }
    return r;
  };
  // MRX: one item of the pre-pass - row band item / 16, column tiles [seg tiles_n / 16, (seg + 1) tiles_n / 16) - by the whole workgroup:
  // the band's xAq panel into the exchange's LDS region, the side product of each column tile against it (this wave's 32 columns; B^T
  // fragments of up to three tiles requested together: one cold round trip), the waves' row maxima through LDS, then {maximum, tag} of
  // row r as granule [r][seg] (one 8-byte sc1 store: value and tag arrive together).  In three steps, so that a workgroup's OWN item can
  // put its requests in front of the first tile's ring fill and compute behind it (mrx_issue - ring_fill - mrx_finish with a counted
  // wait); the fall-back (mrx_item) runs them back to back.
  auto mrx_span = [&](int item, int& t_lo, int& t_hi) {
    const int seg = item & (LQER_AMAX_NSEG - 1);
    t_lo = (seg * g.tiles_n) >> 4, t_hi = ((seg + 1) * g.tiles_n) >> 4;
  };
  auto mrx_load_batch = [&](int t0, int t_hi, auto nl_c, auto nsl_c) {  // tiles t0 .. of the item -> mfr
    if constexpr (MRX) {
      constexpr int NL = decltype(nl_c)::value, NSL = decltype(nsl_c)::value, TP = NL * NSL <= 4 ? 3 : 1;
      const int64_t limb = (int64_t)g.Np * g.rp;
#pragma unroll
      for (int j = 0; j < TP; ++j) {
        const int tt = t0 + j < t_hi ? t0 + j : t_hi - 1;  // (past the segment: a tile it has, computed twice - max does not mind)
        const bf16_t* const bl = g.bt + (int64_t)(tt * BN + wave * 32 + l31) * g.rp + 8 * lh;
#pragma unroll
        for (int l = 0; l < NL; ++l)
#pragma unroll
          for (int ks = 0; ks < NSL; ++ks) mfr[j * NL * NSL + l * NSL + ks] = *(const bf16x8*)(bl + l * limb + ks * 16);
      }
    }
  };
  auto mrx_issue = [&](int item) {  // the band's panel (LDS-DMA: NPA requests per wave) and the first batch of fragments
    if constexpr (MRX) {
      using std::integral_constant;
      int t_lo, t_hi;
      mrx_span(item, t_lo, t_hi);
      const int mb = (item >> 4) * BM;
      const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(g.xaq + (int64_t)mb * g.xaq_ld), 0, BM * g.xaq_ld * 2, 0x00020000);
#pragma unroll
      for (int i = 0; i < NPA; ++i) {
        const int row = wave * (8 * NPA) + i * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void*)(smem + G::EP_XAQ + wave * (8 * NPA) * 128 + i * 1024), 16,
                                                 row * g.xaq_ld * 2 + chunk * 16, 0, 0, 0);
      }
      if (t_hi > t_lo) {
        switch (xch_key) {
          case 16 + 1: mrx_load_batch(t_lo, t_hi, integral_constant<int, 1>{}, integral_constant<int, 1>{}); break;
          case 16 + 2: mrx_load_batch(t_lo, t_hi, integral_constant<int, 1>{}, integral_constant<int, 2>{}); break;
          case 16 + 4: mrx_load_batch(t_lo, t_hi, integral_constant<int, 1>{}, integral_constant<int, 4>{}); break;
          case 32 + 1: mrx_load_batch(t_lo, t_hi, integral_constant<int, 2>{}, integral_constant<int, 1>{}); break;
          case 32 + 2: mrx_load_batch(t_lo, t_hi, integral_constant<int, 2>{}, integral_constant<int, 2>{}); break;
          default: mrx_load_batch(t_lo, t_hi, integral_constant<int, 2>{}, integral_constant<int, 4>{}); break;
        }
      }
    }
  };
  auto mrx_finish = [&](int item) {  // (the panel has landed for every wave: the caller's wait + barrier)
    if constexpr (MRX) {
      using std::integral_constant;
      int t_lo, t_hi;
      mrx_span(item, t_lo, t_hi);
      const int seg = item & (LQER_AMAX_NSEG - 1), mb = (item >> 4) * BM;
      float mx[4] = {0.f, 0.f, 0.f, 0.f};
      auto batches = [&](auto nl_c, auto nsl_c) {
        constexpr int NL = decltype(nl_c)::value, NSL = decltype(nsl_c)::value, TP = NL * NSL <= 4 ? 3 : 1;
        for (int t0 = t_lo; t0 < t_hi; t0 += TP) {
          if (t0 > t_lo) mrx_load_batch(t0, t_hi, nl_c, nsl_c);  // (the first batch: requested by mrx_issue)
          // two row groups at a time (their xAq fragments from the panel - asm reads, see xch_compute_s -, two interleaved accumulator
          // chains per tile): four at once, beside three tiles' fragments, spill
#pragma unroll
          for (int up = 0; up < 2; ++up) {
            bf16x8 pxf[2][NSL];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
              for (int ks = 0; ks < NSL; ++ks) {
                if (up == 0) asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(pxf[u][ks]) : "v"(lds0 + G::EP_XAQ + swz(l31, 2 * ks + lh)), "i"(u * 4096) : "memory");
                else asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(pxf[u][ks]) : "v"(lds0 + G::EP_XAQ + swz(l31, 2 * ks + lh)), "i"((u + 2) * 4096) : "memory");
              }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pxf[0][0])::"memory");
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
              for (int ks = 0; ks < NSL; ++ks) asm volatile("" : "+v"(pxf[u][ks]));
#pragma unroll
            for (int j = 0; j < TP; ++j) {
              f32x16 acc2[2];
#pragma unroll
              for (int u = 0; u < 2; ++u) acc2[u] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
              for (int l = 0; l < NL; ++l)  // (k_bout_amax's order per accumulator: limb-major, slices ascending)
#pragma unroll
                for (int ks = 0; ks < NSL; ++ks)
#pragma unroll
                  for (int u = 0; u < 2; ++u)
                    acc2[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mfr[j * NL * NSL + l * NSL + ks], pxf[u][ks], acc2[u], 0, 0, 0);
#pragma unroll
              for (int u = 0; u < 2; ++u) {
                float m = 0.f;
#pragma unroll
                for (int k = 0; k < 16; k += 2) m = fmaxf(fmaxf(m, fabsf(acc2[u][k])), fabsf(acc2[u][k + 1]));
                auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);  // lanes l and l ^ 32
                mx[2 * up + u] = fmaxf(mx[2 * up + u], fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])));
              }
            }
          }
        }
      };
      if (t_hi > t_lo) {
        switch (xch_key) {
          case 16 + 1: batches(integral_constant<int, 1>{}, integral_constant<int, 1>{}); break;
          case 16 + 2: batches(integral_constant<int, 1>{}, integral_constant<int, 2>{}); break;
          case 16 + 4: batches(integral_constant<int, 1>{}, integral_constant<int, 4>{}); break;
          case 32 + 1: batches(integral_constant<int, 2>{}, integral_constant<int, 1>{}); break;
          case 32 + 2: batches(integral_constant<int, 2>{}, integral_constant<int, 2>{}); break;
          default: batches(integral_constant<int, 2>{}, integral_constant<int, 4>{}); break;
        }
      }
      const float r = xch_reduce(mx);  // (its barrier: every wave is done with the panel)
      if (tid < BM) {
        const int64_t Mp = (int64_t)(g.M + LQER_M_ALIGN - 1) / LQER_M_ALIGN * LQER_M_ALIGN;
        const auto m_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g.bout_amax, 0, (int)(Mp * LQER_AMAX_NSEG * 8), 0x00020000);
        const u32x2_g gv = {__float_as_uint(r), xtag};
        __builtin_amdgcn_raw_buffer_store_b64(gv, m_rsrc, (int)((((mb + tid) * LQER_AMAX_NSEG) + seg) * 8), 0, 16);  // sc1
      }
    }
  };
  // the fall-back (a band whose producers did not publish in time): the same item, slowly - one row group, one fragment at a time, runtime
  // loops - so that it fits beside the live accumulators behind the main loop, where it is called (the fast form there spills)
  auto mrx_item = [&](int item) {
    if constexpr (MRX) {
      int t_lo, t_hi;
      mrx_span(item, t_lo, t_hi);
      const int seg = item & (LQER_AMAX_NSEG - 1), mb = (item >> 4) * BM;
      {
        const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(g.xaq + (int64_t)mb * g.xaq_ld), 0, BM * g.xaq_ld * 2, 0x00020000);
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
          const int row = wave * (8 * NPA) + i * 8 + (lane >> 3);
          const int chunk = (lane & 7) ^ ((row >> 1) & 7);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void*)(smem + G::EP_XAQ + wave * (8 * NPA) * 128 + i * 1024), 16,
                                                   row * g.xaq_ld * 2 + chunk * 16, 0, 0, 0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      float mx[4] = {0.f, 0.f, 0.f, 0.f};
      const int64_t limb = (int64_t)g.Np * g.rp;
      for (int t = t_lo; t < t_hi; ++t) {
        const bf16_t* const bl = g.bt + (int64_t)(t * BN + wave * 32 + l31) * g.rp + 8 * lh;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          f32x16 acc = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
          for (int l = 0; l < g.b_limbs; ++l)
            for (int ks = 0; ks < xch_nsl; ++ks) {
              const bf16x8 fr = *(const bf16x8*)(bl + l * limb + ks * 16);
              bf16x8 px;
              asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(px) : "v"(lds0 + G::EP_XAQ + u * 4096 + swz(l31, 2 * ks + lh)) : "memory");
              acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr, px, acc, 0, 0, 0);
            }
          float m = 0.f;
#pragma unroll
          for (int k = 0; k < 16; k += 2) m = fmaxf(fmaxf(m, fabsf(acc[k])), fabsf(acc[k + 1]));
          auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
          mx[u] = fmaxf(mx[u], fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])));
        }
      }
      const float r = xch_reduce(mx);
      if (tid < BM) {
        const int64_t Mp = (int64_t)(g.M + LQER_M_ALIGN - 1) / LQER_M_ALIGN * LQER_M_ALIGN;
        const auto m_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g.bout_amax, 0, (int)(Mp * LQER_AMAX_NSEG * 8), 0x00020000);
        const u32x2_g gv = {__float_as_uint(r), xtag};
        __builtin_amdgcn_raw_buffer_store_b64(gv, m_rsrc, (int)((((mb + tid) * LQER_AMAX_NSEG) + seg) * 8), 0, 16);  // sc1
      }
    }
  };
  // MRX: a vote of the workgroup - true when `c` holds for some lane of waves 0-1 (the row threads); two barriers.  The word carries the
  // vote's number, so nothing has to be reset (a reset could overtake the next vote's write)
  auto wg_any = [&](bool c) -> bool {
    bool any = false;
    if constexpr (MRX) {
      ++mrx_epoch;
      const uint32_t wa = lds0 + EP_TAB + 1020;
      if (wave < 2 && __builtin_amdgcn_ballot_w64(c) != 0 && lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(wa), "v"(mrx_epoch) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      int f;
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(f) : "v"(wa) : "memory");
      asm volatile("s_barrier" ::: "memory");
      any = __builtin_amdgcn_readfirstlane(f) == mrx_epoch;
    }
    return any;
  };
  // granules [Mp / 4][LQER_AMAX_NSEG] x {exponent bytes of 4 rows, tag}: a row quad's 16 granules are 128 contiguous bytes
  const auto xch_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g.bout_amax, 0, XCH_OK && xch ? (int)(xch_Mp / 4 * LQER_AMAX_NSEG * 8) : 0, 0x00020000);
  // The ring fill in two halves (round 6).  HEAD: the tile's first step - the request everything waits for - goes out before anything else
  // (a later tile's step 0 is already in slot 0 when the epilogue before it kept out of that slot - one panel of xAq -, its row constants
  // were requested there too).  Between the halves: whatever else the prologue reads (the first tile's row constants; XCH: the xAq panel,
  // the side product's fragments, column scale and bias) - loads return in issue order, so the counted waits below, which leave only the
  // TAIL's requests in flight, cover it.  TAIL: the steps behind the first (past the end of K: dropped by the buffer range check) and the
  // tables' LDS writes.
  // The gather of the epilogue, requested EARLY (round 6): the main loop's last three LOADs request steps past the end of K - dead
  // requests that only keep the counted waits uniform.  In the FIRST of them (step nk) waves 0 and 1 send their two activation pieces'
  // requests to the granules instead: same instruction, same count, other descriptor / offset / LDS address - the agent-scope round trip
  // (~3.9 k cycles, which waves 0-1 used to wait out behind the conversion pass with the other six at the barrier) passes under the last
  // three steps.  By then (K >= 2048: > 15 steps = 10 us after the prologue's publish) every resident workgroup has published; shorter K
  // keep the request at the epilogue.
#ifndef LQER_XCH_EARLY_GATHER
#define LQER_XCH_EARLY_GATHER 0  // 1: the form above (measured, round 6: the epilogue gains nothing - what waves 0-1 spend behind the
                                   // conversion pass is the tag test's instructions, not the round trip - and the operand selects cost the
                                   // main loop 5-6 %: 24.7 vs 23.4 us at K = 4096, profiles/r06_i8_timeline.txt); 0: requested at the epilogue
#endif
  const bool xg_early = LQER_XCH_EARLY_GATHER && XCH && !W8 && nk >= 16;  // (uniform)
  const u32x4 xg_rs = make_rs((const uint8_t*)g.bout_amax, XCH ? (uint32_t)(xch_Mp / 4 * LQER_AMAX_NSEG * 8) : 0u);
  int xg_voff[2] = {0, 0};
  if constexpr (XCH) {
#pragma unroll
    for (int j = 0; j < 2; ++j) xg_voff[j] = (int)(((m0 >> 2) + 16 * wave + 8 * j + (lane >> 3)) * (LQER_AMAX_NSEG * 8)) + (lane & 7) * 16;
  }
  auto ring_fill_head = [&]() {
    using std::integral_constant;
    if constexpr (W8D) {
      // request order per step [A x 2, W x 4] (a compiler barrier between the two keeps it): the vmcnt(12) below retires step 0 whole,
      // the vmcnt(16) at the end of every LOAD the activations of the step after it
      if (first || !one_panel) issue_a8(a_base, 0, 0);
      asm volatile("" ::: "memory");
      issue_wq(w_base, 0, integral_constant<int, 0>{});
      asm volatile("" ::: "memory");
    } else if constexpr (W8) {
      // request order [A(0) x 4, W(0) x 2, W(1) x 2, A(1) x 4]: the vmcnt(6) below leaves W(1) and A(1) in flight; the first LOAD's
      // vmcnt(6) - its own four requests and two more - then retires W(1), which the second LOAD reads
      if (first || !one_panel) {
        issue_a8(a_base, 0, 0);
        issue_w8(w_base, 0, 0);
      }
    } else {
      if (first || !one_panel) issue_step(a_base, w_base, 0, 0);
    }
  };
  auto ring_fill_tail = [&]() {
    using std::integral_constant;
    if constexpr (W8D) {
      issue_a8(a_base, 1, 1);
      asm volatile("" ::: "memory");
      issue_wq(w_base, 1, integral_constant<int, 1>{});
      asm volatile("" ::: "memory");
      issue_a8(a_base, 2, 2);
      asm volatile("" ::: "memory");
      issue_wq(w_base, 2, integral_constant<int, 2>{});
      asm volatile("" ::: "memory");
    } else if constexpr (W8) {
      issue_w8(w_base, 1, 1);
      issue_a8(a_base, 1, 1);
    } else {
      issue_step(a_base, w_base, 1, 1);
      if constexpr (DEPTH == 3) issue_step(a_base, w_base, 2, 2);
    }
    write_tables();
  };
  auto ring_fill = [&]() {
    ring_fill_head();
    if (first && !MRX) load_tables(m0);  // (MRX: nothing is published yet - the first tile's tables are written behind its main loop)
    ring_fill_tail();
  };
  // exchange: the tile's first ring step first, then - in this order, all of them L2 hits - the xAq panel (LDS-DMA into its own region:
  // the epilogue's side product reads it from there), this tile's side-product operands (B^T fragments: kept for the epilogue; xAq
  // fragments in k_bout_amax's layout), the row and column constants; then the ring's other steps.  The products run behind the requests -
  // one switch around both halves, so that only the registers of the (limbs, slices) case at hand are live across the fill
  float xmx[4] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (XCH) {
    using std::integral_constant;
    // request order: [the xAq panel, the B^T fragments, row / column constants] [ring steps 0, 1, 2 - 15 requests per wave, no branch].
    // The first group is small (48 KiB per workgroup) and lands a ring step's transfer time ahead of step 0: the row maxima are computed,
    // reduced and published while the ring fills.
    {
      const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(g.xaq + (int64_t)m0 * g.xaq_ld), 0, BM * g.xaq_ld * 2, 0x00020000);
#pragma unroll
      for (int i = 0; i < NPA; ++i) {
        const int row = wave * (8 * NPA) + i * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void*)(smem + G::EP_XAQ + wave * (8 * NPA) * 128 + i * 1024), 16,
                                                 row * g.xaq_ld * 2 + chunk * 16, 0, 0, 0);
      }
    }
    // (fragments the (limbs, slices) case at hand does not have stay undefined - nothing reads them; a zero fill would have to wait for
    // the requests in flight before it may write their registers)
    switch (xch_key) {
      case 16 + 1: xch_load_s(tn, integral_constant<int, 1>{}, integral_constant<int, 1>{}); break;
      case 16 + 2: xch_load_s(tn, integral_constant<int, 1>{}, integral_constant<int, 2>{}); break;
      case 16 + 4: xch_load_s(tn, integral_constant<int, 1>{}, integral_constant<int, 4>{}); break;
      case 32 + 1: xch_load_s(tn, integral_constant<int, 2>{}, integral_constant<int, 1>{}); break;
      case 32 + 2: xch_load_s(tn, integral_constant<int, 2>{}, integral_constant<int, 2>{}); break;
      default: xch_load_s(tn, integral_constant<int, 2>{}, integral_constant<int, 4>{}); break;
    }
    load_tables(m0);
    {
      const int n_ = n0 + wave * 32 + l31;
      ws_x = ((const float*)(g.w8 + (size_t)g.tiles_n * nk * (W8 ? 2 * W8_SLOT : I8_WBLOCK)))[n_];
      bv_x = g.bias ? g.bias[n_] : 0.f;
    }
#ifdef LQER_CLOCKPROBE
    I8_STAMP(cp_p[0], cp_x);
#endif
    ring_fill_head();
    ring_fill_tail();  // (its table writes wait for this lane's row constant: the youngest request of the first group)
    // the first group has landed for this wave (the ring's requests - W4: 15, W8 codes into registers: 18 - may stay in flight), then for all
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(W8D ? 18 : 15) : "memory");
#ifdef LQER_CLOCKPROBE
    I8_STAMP(cp_p[1], cp_x);
#endif
    switch (xch_key) {
      case 16 + 1: xch_compute_s(xmx, integral_constant<int, 1>{}, integral_constant<int, 1>{}); break;
      case 16 + 2: xch_compute_s(xmx, integral_constant<int, 1>{}, integral_constant<int, 2>{}); break;
      case 16 + 4: xch_compute_s(xmx, integral_constant<int, 1>{}, integral_constant<int, 4>{}); break;
      case 32 + 1: xch_compute_s(xmx, integral_constant<int, 2>{}, integral_constant<int, 1>{}); break;
      case 32 + 2: xch_compute_s(xmx, integral_constant<int, 2>{}, integral_constant<int, 2>{}); break;
      default: xch_compute_s(xmx, integral_constant<int, 2>{}, integral_constant<int, 4>{}); break;
    }
#ifdef LQER_CLOCKPROBE
    asm volatile("" ::"v"(xmx[0]), "v"(xmx[3]));
    I8_STAMP(cp_p[2], cp_x);
#endif
  } else {
    // MRX, first tile: this workgroup's item of the pre-pass - its requests (the band's panel, the first fragments) in FRONT of the tile's
    // ring fill, its arithmetic behind it (loads return in issue order: the counted wait leaves the ring's 15 requests in flight)
    const bool mrx_mine = MRX && first && (int)blockIdx.x < g.tiles_m * LQER_AMAX_NSEG;  // (workgroup-uniform)
    if constexpr (MRX) {
      if (first && tid == 0) asm volatile("ds_write_b32 %0, %1 offset:1020" ::"v"(lds0 + EP_TAB), "v"(0u) : "memory");  // (the vote word)
      if (mrx_mine) mrx_issue((int)blockIdx.x);
    }
    ring_fill();
    if constexpr (MRX) {
#ifdef LQER_CLOCKPROBE
      if (first) I8_STAMP(cp_p[0], cp_x);  // the item's and the first tile's requests are out
#endif
      if (mrx_mine) {
        asm volatile("s_waitcnt vmcnt(15)\n\ts_barrier" ::: "memory");
        mrx_finish((int)blockIdx.x);
      }
#ifdef LQER_CLOCKPROBE
      if (first) I8_STAMP(cp_p[1], cp_x);  // the item is published
#endif
    }
  }
  if constexpr (XCH_OK) {
    if (xch) {  // this tile's row maxima -> granules [tn][m0 + row]; the miss vote of the epilogue starts clean
      const float r = xch_reduce(xmx);
      if (tid < BM) {  // (waves 0 and 1, whole)
        // what travels is the row's block EXPONENT (the exponent of a maximum is the maximum of the exponents: block_exponent is
        // monotone), one byte - four rows per granule: {4 exponent bytes, tag}, a quarter of the requests and bytes at the gather
        uint32_t wv = (uint32_t)(block_exponent(r, g.bout) - g.bout.emin) << (8 * (tid & 3));
        wv |= __shfl_xor(wv, 1, 64);
        wv |= __shfl_xor(wv, 2, 64);
        if ((tid & 3) == 0) {
          const u32x2_g gv = {wv, xtag};
          __builtin_amdgcn_raw_buffer_store_b64(gv, xch_rsrc, (int)((((m0 + tid) >> 2) * LQER_AMAX_NSEG + tn) * 8), 0, 16);  // sc1
        }
      }
      if (tid == 0) asm volatile("ds_write_b32 %0, %1 offset:1020" ::"v"(lds0 + EP_TAB), "v"(0u) : "memory");
#ifdef LQER_CLOCKPROBE
      I8_STAMP(cp_p[3], cp_x);
#endif
    }
  }

  i32x16 R[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) R[i][j] = 0;

  // ---- main loop: half-steps h = 2 kt + P (gemm_w4a8_m256.hip has the barrier / RAW / WAR argument) ----------------------
  //   waves 0-3:    ... | LOAD(h)  | COMPUTE(h) | LOAD(h+1) | ...
  //   waves 4-7:    ... | COMP(h-1)| LOAD(h)    | COMPUTE(h)| ...
  // LOAD(kt, 0): the step's weight words (2 x 16 B) and shift byte, the activation fragments of token tiles 0-3 (16 x 16 B),
  // half of the prefetch of step kt+2; LOAD(kt, 1): tiles 4-7, the other half.  A wave ends LOAD(kt, 1) with vmcnt(6): its own
  // batch of step kt+1 has landed (the batch of kt+2 - 6 loads, wave 0: 7 - may stay in flight), then passes a barrier before
  // anyone reads step kt+1.
  // NT = 4 (128-row tiles): one phase per step - LOAD(kt) reads all 16 activation fragments of the 4 token tiles and issues the
  // whole batch of step kt+3 (4 loads, wave 0: 5), ends with vmcnt(8): the batches of kt+2 and kt+3 may stay in flight.
  const bool late = wave >= 4;
  if constexpr (NT == 8) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // step 0 landed, the tables written
  else if constexpr (W8D) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (late) asm volatile("s_barrier" ::: "memory");
#ifdef LQER_CLOCKPROBE
  I8_STAMP(cp_c[1], cp_r[1]);
  if constexpr (MRX) { if (first) cp_p[3] = cp_c[1]; }
#endif
  i32x4 wf[4];     // the step's expanded weight fragments (slices 0..3): live across both half-steps
  uint32_t sv = 0;  // this lane's (column's) shift of the step's 128-k group
  // SHIFT: the group sums of a half-step's LAST token tile are folded at the head of the next half-step's LOAD section - the
  // partner wave of the SIMD is computing then and the vector ALU is idle -, those of the other three tiles under the MFMAs
  // of the tile after them (4 folds per MFMA slot, from two slots behind the tile's last MFMA: its results have landed)
  i32x16 Gd = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  auto half_step = [&](int kt, auto slot_c, auto half_c, auto mode_c) {
    constexpr int SLOT = decltype(slot_c)::value;
    constexpr int P = decltype(half_c)::value;
    constexpr int MODE = decltype(mode_c)::value;  // I8_MODE_* of this tile
    constexpr int slot_new = (SLOT + DEPTH) % NSLOT;
    constexpr int A_IMM = (SLOT == 2 ? 0 : SLOT * A_SLOT) + 4 * P * 4096;  // tile t of this half: + 4096 t
    __builtin_amdgcn_s_setprio(1);
    if constexpr (MODE == I8_MODE_FOLD) {  // the previous half-step's last tile (sv still holds that step's shift: the asm below updates it)
      constexpr int prev = 4 * (1 - P) + 3;
#pragma unroll
      for (int j = 0; j < 16; ++j) R[prev][j] = (int)(((uint32_t)Gd[j] << sv) + (uint32_t)R[prev][j]);
      __builtin_amdgcn_sched_barrier(0);
    }
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
    // PRESHIFT: the lane is the nibble's value arithmetically shifted: (cw << 4) >> q per byte.  The logical shift moves only
    // zeros across byte borders (the low nibble of every byte of x is 0, q <= 4); the sign fill comes from v_perm_b32's
    // sign selectors (a byte of 0x00 / 0xFF per odd byte of its sources: w and w << 8 give the four high-nibble signs, w << 4
    // and w << 12 the low-nibble ones) masked to the q vacated bits.  11 vector instructions per word of 8 weights.
    uint32_t hq4 = 0;  // per byte: the q high bits
    if constexpr (MODE == I8_MODE_PRESHIFT && P == 0) hq4 = ((0xFF00u >> sv) & 0xFFu) * 0x01010101u;
    auto expand = [&](uint32_t w0, uint32_t w1) {
      if constexpr (MODE == I8_MODE_PRESHIFT1) {  // q <= 1: (x >> q) | (x & 0x80808080) - the one vacated bit is the sign bit
        const uint32_t a0 = (w0 << 4) & 0xF0F0F0F0u, b0 = w0 & 0xF0F0F0F0u, a1 = (w1 << 4) & 0xF0F0F0F0u, b1 = w1 & 0xF0F0F0F0u;
        return (i32x4){(int)((a0 & 0x80808080u) | (a0 >> sv)), (int)((b0 & 0x80808080u) | (b0 >> sv)),
                       (int)((a1 & 0x80808080u) | (a1 >> sv)), (int)((b1 & 0x80808080u) | (b1 >> sv))};
      } else if constexpr (MODE == I8_MODE_PRESHIFT) {
        auto one = [&](uint32_t w, uint32_t& lo, uint32_t& hi) {
          const uint32_t t = w << 4;
          const uint32_t m_hi = __builtin_amdgcn_perm(w, w << 8, 0x0B090A08u), m_lo = __builtin_amdgcn_perm(t, w << 12, 0x0B090A08u);
          lo = (m_lo & hq4) | ((t & 0xF0F0F0F0u) >> sv);
          hi = (m_hi & hq4) | ((w & 0xF0F0F0F0u) >> sv);
        };
        uint32_t l0, h0, l1, h1;
        one(w0, l0, h0);
        one(w1, l1, h1);
        return (i32x4){(int)l0, (int)h0, (int)l1, (int)h1};
      } else {
        return (i32x4){(int)((w0 << 4) & 0xF0F0F0F0u), (int)(w0 & 0xF0F0F0F0u), (int)((w1 << 4) & 0xF0F0F0F0u), (int)(w1 & 0xF0F0F0F0u)};
      }
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
    if constexpr (MODE != I8_MODE_FOLD) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int t = 0; t < 4; ++t) R[4 * P + t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa[t][ks], wf[ks], R[4 * P + t], 0, 0, 0);
      if constexpr ((MODE == I8_MODE_PRESHIFT || MODE == I8_MODE_PRESHIFT1) && P == 0) {
        // issue order: the expand of slice ks + 1 (22 vector instructions) in the shadow of the four MFMAs of slice ks
#define I8_SLOT(NV)                                       \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  \
        if constexpr ((NV) > 0) __builtin_amdgcn_sched_group_barrier(0x002, (NV), 0);
        I8_SLOT(6) I8_SLOT(6) I8_SLOT(5) I8_SLOT(5)
        I8_SLOT(6) I8_SLOT(6) I8_SLOT(5) I8_SLOT(5)
        I8_SLOT(6) I8_SLOT(6) I8_SLOT(5) I8_SLOT(5)
        I8_SLOT(0) I8_SLOT(0) I8_SLOT(0) I8_SLOT(0)
#undef I8_SLOT
      }
    } else {
      const i32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      i32x16 G[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        G[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa[t][0], wf[0], z, 0, 0, 0);
#pragma unroll
        for (int ks = 1; ks < 4; ++ks) G[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa[t][ks], wf[ks], G[t], 0, 0, 0);
        if (t > 0) {
#pragma unroll
          for (int j = 0; j < 16; ++j) R[4 * P + t - 1][j] = (int)(((uint32_t)G[t - 1][j] << sv) + (uint32_t)R[4 * P + t - 1][j]);
        }
      }
      Gd = G[3];
      // issue order: one MFMA, then the vector instructions that fit its shadow - the expands of the step's remaining weight
      // fragments first (P = 0: needed by the MFMAs of slots 1..3), the folds of tile t from slot 4 t + 5 on
#define I8_SLOT(NV)                                       \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  \
      if constexpr ((NV) > 0) __builtin_amdgcn_sched_group_barrier(0x002, (NV), 0);
      I8_SLOT(P == 0 ? 6 : 0) I8_SLOT(P == 0 ? 6 : 0) I8_SLOT(P == 0 ? 6 : 0) I8_SLOT(0)
      I8_SLOT(0) I8_SLOT(4) I8_SLOT(4) I8_SLOT(4)
      I8_SLOT(4) I8_SLOT(4) I8_SLOT(4) I8_SLOT(4)
      I8_SLOT(4) I8_SLOT(5) I8_SLOT(5) I8_SLOT(6)
#undef I8_SLOT
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- the same half-step with the step split along K instead of along the token tiles (NONE / PRESHIFT: no group tile is
  // needed): half P multiplies the 32-k slices 2P and 2P + 1 of ALL 8 token tiles.  The weight words of a half are one 16-byte
  // read and their expansion - 22 vector instructions per slice with pre-shifted lanes - is spread evenly: one slice before
  // the barrier, one in the shadow of the first eight MFMAs, in BOTH halves (the token-tile split needs all four slices in
  // its first half and none in its second).  Ring, prefetch pieces, waits and barriers are those of half_step.
  uint32_t hq4_k = 0;  // PRESHIFT: per byte the q high bits (from the step's shift byte, read in half 0)
  auto half_step_k = [&](int kt, auto slot_c, auto half_c, auto mode_c) {
    constexpr int SLOT = decltype(slot_c)::value;
    constexpr int P = decltype(half_c)::value;
    constexpr int MODE = decltype(mode_c)::value;
    constexpr int slot_new = (SLOT + DEPTH) % NSLOT;
    constexpr int A_IMM = (SLOT == 2 ? 0 : SLOT * A_SLOT);  // tile t: + 4096 t
    __builtin_amdgcn_s_setprio(1);
    const int ktn = __builtin_amdgcn_readfirstlane(kt + DEPTH);
    const int a_soff = ktn * I8_BK, w_soff = ktn * I8_WBLOCK;
    const uint32_t m0a0 = m0_a + slot_new * A_SLOT + (2 * P) * 1024, m0a1 = m0a0 + 1024;
    const uint32_t m0w = m0_w + slot_new * W_SLOT + P * 1024, m0s = m0_s + slot_new * W_SLOT;
    i32x4 xk[8][2];  // [token tile][slice of this half]
    u32x4 wr;        // the half's weight words: slices 2P, 2P + 1
#define I8K_READS                                                                                                      \
      "ds_read_b128 %[x00], %[fa0] offset:%c[aimm]\n\tds_read_b128 %[x01], %[fa1] offset:%c[aimm]\n\t"  \
      "ds_read_b128 %[x10], %[fa0] offset:%c[aimm]+4096\n\tds_read_b128 %[x11], %[fa1] offset:%c[aimm]+4096\n\t"  \
      "ds_read_b128 %[x20], %[fa0] offset:%c[aimm]+8192\n\tds_read_b128 %[x21], %[fa1] offset:%c[aimm]+8192\n\t"  \
      "ds_read_b128 %[x30], %[fa0] offset:%c[aimm]+12288\n\tds_read_b128 %[x31], %[fa1] offset:%c[aimm]+12288\n\t"  \
      "ds_read_b128 %[x40], %[fa0] offset:%c[aimm]+16384\n\tds_read_b128 %[x41], %[fa1] offset:%c[aimm]+16384\n\t"  \
      "ds_read_b128 %[x50], %[fa0] offset:%c[aimm]+20480\n\tds_read_b128 %[x51], %[fa1] offset:%c[aimm]+20480\n\t"  \
      "ds_read_b128 %[x60], %[fa0] offset:%c[aimm]+24576\n\tds_read_b128 %[x61], %[fa1] offset:%c[aimm]+24576\n\t"  \
      "ds_read_b128 %[x70], %[fa0] offset:%c[aimm]+28672\n\tds_read_b128 %[x71], %[fa1] offset:%c[aimm]+28672\n\t"  \
      "ds_read_b128 %[wr], %[fw] offset:%c[wimm]\n\t"
#define I8K_DMA                                                                                                        \
      "s_mov_b32 m0, %[m0a0]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[av0], %[ars], %[asoff] offen lds\n\t"                   \
      "s_mov_b32 m0, %[m0a1]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[av1], %[ars], %[asoff] offen lds\n\t"                   \
      "s_mov_b32 m0, %[m0w]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[wv], %[wrs], %[wsoff] offen lds\n\t"
#define I8K_OUTS [x00] "=&v"(xk[0][0]), [x01] "=&v"(xk[0][1]), [x10] "=&v"(xk[1][0]), [x11] "=&v"(xk[1][1]), [x20] "=&v"(xk[2][0]), [x21] "=&v"(xk[2][1]), [x30] "=&v"(xk[3][0]), [x31] "=&v"(xk[3][1]), [x40] "=&v"(xk[4][0]), [x41] "=&v"(xk[4][1]), [x50] "=&v"(xk[5][0]), [x51] "=&v"(xk[5][1]), [x60] "=&v"(xk[6][0]), [x61] "=&v"(xk[6][1]), [x70] "=&v"(xk[7][0]), [x71] "=&v"(xk[7][1]), [wr] "=&v"(wr)
#define I8K_INS                                                                                                        \
      [fa0] "v"(SLOT == 2 ? fa_hi[2 * P] : fa_lo[2 * P]), [fa1] "v"(SLOT == 2 ? fa_hi[2 * P + 1] : fa_lo[2 * P + 1]),         \
      [aimm] "i"(A_IMM), [fw] "v"(P == 0 ? fw_a : fw_b), [wimm] "i"(SLOT * W_SLOT), [av0] "v"(a_voff[2 * P]),                 \
      [av1] "v"(a_voff[2 * P + 1]), [wv] "v"(P == 0 ? w_voff0 : w_voff1), [ars] "s"(a_rs), [wrs] "s"(w_rs), [m0a0] "s"(m0a0), \
      [m0a1] "s"(m0a1), [asoff] "s"(a_soff), [m0w] "s"(m0w), [wsoff] "s"(w_soff)
    if constexpr (P == 0) {
      asm volatile(I8K_READS "ds_read_u8 %[sv], %[fs] offset:%c[wimm]\n\t" I8K_DMA "s_waitcnt lgkmcnt(0)"
                   : I8K_OUTS, [sv] "=&v"(sv)
                   : I8K_INS, [fs] "v"(fs_addr)
                   : "memory");
    } else {
      asm volatile(I8K_READS I8K_DMA
                   "s_cmp_lg_u32 %[wave], 0\n\ts_cbranch_scc1 1f\n\t"
                   "s_mov_b32 m0, %[m0s]\n\ts_nop 0\n\tbuffer_load_dword %[sv4], %[wrs], %[wsoff] offen lds\n\t"
                   "1:\n\ts_waitcnt vmcnt(6) lgkmcnt(0)"
                   : I8K_OUTS
                   : I8K_INS, [wave] "s"(wave), [sv4] "v"(s_voff), [m0s] "s"(m0s)
                   : "memory", "scc");
    }
#undef I8K_READS
#undef I8K_DMA
#undef I8K_OUTS
#undef I8K_INS
    if constexpr (MODE == I8_MODE_PRESHIFT && P == 0) hq4_k = ((0xFF00u >> sv) & 0xFFu) * 0x01010101u;
    auto expand = [&](uint32_t w0, uint32_t w1) {
      if constexpr (MODE == I8_MODE_PRESHIFT1) {  // q <= 1: (x >> q) | (x & 0x80808080) - the one vacated bit is the sign bit
        const uint32_t a0 = (w0 << 4) & 0xF0F0F0F0u, b0 = w0 & 0xF0F0F0F0u, a1 = (w1 << 4) & 0xF0F0F0F0u, b1 = w1 & 0xF0F0F0F0u;
        return (i32x4){(int)((a0 & 0x80808080u) | (a0 >> sv)), (int)((b0 & 0x80808080u) | (b0 >> sv)),
                       (int)((a1 & 0x80808080u) | (a1 >> sv)), (int)((b1 & 0x80808080u) | (b1 >> sv))};
      } else if constexpr (MODE == I8_MODE_PRESHIFT) {  // (the lane cw << (4 - q): see half_step)
        auto one = [&](uint32_t w, uint32_t& lo, uint32_t& hi) {
          const uint32_t t = w << 4;
          const uint32_t m_hi = __builtin_amdgcn_perm(w, w << 8, 0x0B090A08u), m_lo = __builtin_amdgcn_perm(t, w << 12, 0x0B090A08u);
          lo = (m_lo & hq4_k) | ((t & 0xF0F0F0F0u) >> sv);
          hi = (m_hi & hq4_k) | ((w & 0xF0F0F0F0u) >> sv);
        };
        uint32_t l0, h0, l1, h1;
        one(w0, l0, h0);
        one(w1, l1, h1);
        return (i32x4){(int)l0, (int)h0, (int)l1, (int)h1};
      } else {
        return (i32x4){(int)((w0 << 4) & 0xF0F0F0F0u), (int)(w0 & 0xF0F0F0F0u), (int)((w1 << 4) & 0xF0F0F0F0u), (int)(w1 & 0xF0F0F0F0u)};
      }
    };
    i32x4 wa = expand(wr[0], wr[1]);
    asm volatile("s_barrier" : "+v"(wa)::"memory");
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    // ---- COMPUTE
    const i32x4 wb = expand(wr[2], wr[3]);
#pragma unroll
    for (int t = 0; t < 8; ++t) R[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xk[t][0], wa, R[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 8; ++t) R[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xk[t][1], wb, R[t], 0, 0, 0);
    if constexpr (MODE == I8_MODE_PRESHIFT || MODE == I8_MODE_PRESHIFT1) {  // the second slice's expand: <= 3 vector instructions per MFMA slot
#define I8_SLOT(NV)                                     \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  \
      if constexpr ((NV) > 0) __builtin_amdgcn_sched_group_barrier(0x002, (NV), 0);
      I8_SLOT(3) I8_SLOT(3) I8_SLOT(3) I8_SLOT(3) I8_SLOT(3) I8_SLOT(3) I8_SLOT(3) I8_SLOT(3)
      I8_SLOT(0) I8_SLOT(0) I8_SLOT(0) I8_SLOT(0) I8_SLOT(0) I8_SLOT(0) I8_SLOT(0) I8_SLOT(0)
#undef I8_SLOT
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- NT = 4 (128-row tiles): one phase per 128-k step.  LOAD: the 16 activation fragments of the 4 token tiles, the step's
  // weight words and shift byte, the whole LDS-DMA batch of step kt + 3; COMPUTE: 16 MFMAs, three of the four weight slices
  // expanded in their shadow (the first before the barrier) - half_step's first half, every phase.
  auto step4 = [&](int kt, auto slot_c, auto mode_c) {
    constexpr int SLOT = decltype(slot_c)::value;
    constexpr int MODE = decltype(mode_c)::value;
    constexpr int slot_new = (SLOT + DEPTH) % NSLOT;
    constexpr int A_IMM = SLOT * A_SLOT;  // tile t: + 4096 t (every slot within the DS offset field)
    __builtin_amdgcn_s_setprio(1);
    if constexpr (MODE == I8_MODE_FOLD) {  // the previous step's last tile (sv still holds that step's shift: the asm below updates it)
#pragma unroll
      for (int j = 0; j < 16; ++j) R[3][j] = (int)(((uint32_t)Gd[j] << sv) + (uint32_t)R[3][j]);
      __builtin_amdgcn_sched_barrier(0);
    }
    const int ktn = __builtin_amdgcn_readfirstlane(kt + DEPTH);
    int a_soff = ktn * I8_BK;
    const int w_soff = ktn * I8_WBLOCK;
    uint32_t m0a0 = m0_a + slot_new * A_SLOT, m0a1 = m0a0 + 1024;
    const uint32_t m0w0 = m0_w + slot_new * W_SLOT, m0w1 = m0w0 + 1024, m0s = m0_s + slot_new * W_SLOT;
    u32x4 ars_e = a_rs;
    int av0_e = a_voff[0], av1_e = a_voff[1];
    if constexpr (XCH && !W8) {
      if (xg_early && ktn == nk && wave < 2) {  // (uniform) the step past the end of K: the granules instead of nothing
        ars_e = xg_rs, a_soff = 0;
        m0a0 = lds0 + G::EP_GATHER + wave * 2048, m0a1 = m0a0 + 1024;
        av0_e = xg_voff[0], av1_e = xg_voff[1];
      }
    }
    i32x4 xa[4][4];  // [token tile][slice]
    u32x4 wr0, wr1;
    asm volatile(
        "ds_read_b128 %[x00], %[fa0] offset:%c[aimm]\n\tds_read_b128 %[x01], %[fa1] offset:%c[aimm]\n\t"
        "ds_read_b128 %[x02], %[fa2] offset:%c[aimm]\n\tds_read_b128 %[x03], %[fa3] offset:%c[aimm]\n\t"
        "ds_read_b128 %[x10], %[fa0] offset:%c[aimm]+4096\n\tds_read_b128 %[x11], %[fa1] offset:%c[aimm]+4096\n\t"
        "ds_read_b128 %[x12], %[fa2] offset:%c[aimm]+4096\n\tds_read_b128 %[x13], %[fa3] offset:%c[aimm]+4096\n\t"
        "ds_read_b128 %[x20], %[fa0] offset:%c[aimm]+8192\n\tds_read_b128 %[x21], %[fa1] offset:%c[aimm]+8192\n\t"
        "ds_read_b128 %[x22], %[fa2] offset:%c[aimm]+8192\n\tds_read_b128 %[x23], %[fa3] offset:%c[aimm]+8192\n\t"
        "ds_read_b128 %[x30], %[fa0] offset:%c[aimm]+12288\n\tds_read_b128 %[x31], %[fa1] offset:%c[aimm]+12288\n\t"
        "ds_read_b128 %[x32], %[fa2] offset:%c[aimm]+12288\n\tds_read_b128 %[x33], %[fa3] offset:%c[aimm]+12288\n\t"
        "ds_read_b128 %[wr0], %[fwa] offset:%c[wimm]\n\tds_read_b128 %[wr1], %[fwb] offset:%c[wimm]\n\t"
        "ds_read_u8 %[sv], %[fs] offset:%c[wimm]\n\t"
        // (no cache-policy bits on these two even when they carry the gather: the CU's L1 is invalidated at the launch's start and nobody
        // on this CU has read the granule lines since; sc0 on every activation piece cost the main loop 17 %)
        "s_mov_b32 m0, %[m0a0]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[av0], %[ars], %[asoff] offen lds\n\t"
        "s_mov_b32 m0, %[m0a1]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[av1], %[ars], %[asoff] offen lds\n\t"
        "s_mov_b32 m0, %[m0w0]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[wv0], %[wrs], %[wsoff] offen lds\n\t"
        "s_mov_b32 m0, %[m0w1]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[wv1], %[wrs], %[wsoff] offen lds\n\t"
        "s_cmp_lg_u32 %[wave], 0\n\ts_cbranch_scc1 1f\n\t"
        "s_mov_b32 m0, %[m0s]\n\ts_nop 0\n\tbuffer_load_dword %[sv4], %[wrs], %[wsoff] offen lds\n\t"
        "1:\n\ts_waitcnt vmcnt(8) lgkmcnt(0)"
        : [x00] "=&v"(xa[0][0]), [x01] "=&v"(xa[0][1]), [x02] "=&v"(xa[0][2]), [x03] "=&v"(xa[0][3]), [x10] "=&v"(xa[1][0]),
          [x11] "=&v"(xa[1][1]), [x12] "=&v"(xa[1][2]), [x13] "=&v"(xa[1][3]), [x20] "=&v"(xa[2][0]), [x21] "=&v"(xa[2][1]),
          [x22] "=&v"(xa[2][2]), [x23] "=&v"(xa[2][3]), [x30] "=&v"(xa[3][0]), [x31] "=&v"(xa[3][1]), [x32] "=&v"(xa[3][2]),
          [x33] "=&v"(xa[3][3]), [wr0] "=&v"(wr0), [wr1] "=&v"(wr1), [sv] "=&v"(sv)
        : [fa0] "v"(fa_lo[0]), [fa1] "v"(fa_lo[1]), [fa2] "v"(fa_lo[2]), [fa3] "v"(fa_lo[3]), [aimm] "i"(A_IMM), [fwa] "v"(fw_a),
          [fwb] "v"(fw_b), [fs] "v"(fs_addr), [wimm] "i"(SLOT * W_SLOT), [av0] "v"(av0_e), [av1] "v"(av1_e),
          [wv0] "v"(w_voff0), [wv1] "v"(w_voff1), [sv4] "v"(s_voff), [ars] "s"(ars_e), [wrs] "s"(w_rs), [m0a0] "s"(m0a0),
          [m0a1] "s"(m0a1), [m0w0] "s"(m0w0), [m0w1] "s"(m0w1), [m0s] "s"(m0s), [asoff] "s"(a_soff), [wsoff] "s"(w_soff),
          [wave] "s"(wave)
        : "memory", "scc");
    uint32_t hq4 = 0;  // PRESHIFT: per byte the q high bits (see half_step)
    if constexpr (MODE == I8_MODE_PRESHIFT) hq4 = ((0xFF00u >> sv) & 0xFFu) * 0x01010101u;
    auto expand = [&](uint32_t w0, uint32_t w1) {
      if constexpr (MODE == I8_MODE_PRESHIFT1) {
        const uint32_t a0 = (w0 << 4) & 0xF0F0F0F0u, b0 = w0 & 0xF0F0F0F0u, a1 = (w1 << 4) & 0xF0F0F0F0u, b1 = w1 & 0xF0F0F0F0u;
        return (i32x4){(int)((a0 & 0x80808080u) | (a0 >> sv)), (int)((b0 & 0x80808080u) | (b0 >> sv)),
                       (int)((a1 & 0x80808080u) | (a1 >> sv)), (int)((b1 & 0x80808080u) | (b1 >> sv))};
      } else if constexpr (MODE == I8_MODE_PRESHIFT) {
        auto one = [&](uint32_t w, uint32_t& lo, uint32_t& hi) {
          const uint32_t t = w << 4;
          const uint32_t m_hi = __builtin_amdgcn_perm(w, w << 8, 0x0B090A08u), m_lo = __builtin_amdgcn_perm(t, w << 12, 0x0B090A08u);
          lo = (m_lo & hq4) | ((t & 0xF0F0F0F0u) >> sv);
          hi = (m_hi & hq4) | ((w & 0xF0F0F0F0u) >> sv);
        };
        uint32_t l0, h0, l1, h1;
        one(w0, l0, h0);
        one(w1, l1, h1);
        return (i32x4){(int)l0, (int)h0, (int)l1, (int)h1};
      } else {
        return (i32x4){(int)((w0 << 4) & 0xF0F0F0F0u), (int)(w0 & 0xF0F0F0F0u), (int)((w1 << 4) & 0xF0F0F0F0u), (int)(w1 & 0xF0F0F0F0u)};
      }
    };
    wf[0] = expand(wr0[0], wr0[1]);
    asm volatile("s_barrier" : "+v"(wf[0])::"memory");
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    // ---- COMPUTE
    wf[1] = expand(wr0[2], wr0[3]);
    wf[2] = expand(wr1[0], wr1[1]);
    wf[3] = expand(wr1[2], wr1[3]);
#define I8_SLOT(NV)                                     \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  \
    if constexpr ((NV) > 0) __builtin_amdgcn_sched_group_barrier(0x002, (NV), 0);
    if constexpr (MODE != I8_MODE_FOLD) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int t = 0; t < 4; ++t) R[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa[t][ks], wf[ks], R[t], 0, 0, 0);
      if constexpr (MODE == I8_MODE_PRESHIFT || MODE == I8_MODE_PRESHIFT1) {
        // issue order: the expand of slice ks + 1 (<= 22 vector instructions) in the shadow of the four MFMAs of slice ks
        I8_SLOT(6) I8_SLOT(6) I8_SLOT(5) I8_SLOT(5)
        I8_SLOT(6) I8_SLOT(6) I8_SLOT(5) I8_SLOT(5)
        I8_SLOT(6) I8_SLOT(6) I8_SLOT(5) I8_SLOT(5)
        I8_SLOT(0) I8_SLOT(0) I8_SLOT(0) I8_SLOT(0)
      }
    } else {
      const i32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      i32x16 G4[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        G4[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa[t][0], wf[0], z, 0, 0, 0);
#pragma unroll
        for (int ks = 1; ks < 4; ++ks) G4[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa[t][ks], wf[ks], G4[t], 0, 0, 0);
        if (t > 0) {
#pragma unroll
          for (int j = 0; j < 16; ++j) R[t - 1][j] = (int)(((uint32_t)G4[t - 1][j] << sv) + (uint32_t)R[t - 1][j]);
        }
      }
      Gd = G4[3];
      I8_SLOT(6) I8_SLOT(6) I8_SLOT(6) I8_SLOT(0)
      I8_SLOT(0) I8_SLOT(4) I8_SLOT(4) I8_SLOT(4)
      I8_SLOT(4) I8_SLOT(4) I8_SLOT(4) I8_SLOT(4)
      I8_SLOT(4) I8_SLOT(5) I8_SLOT(5) I8_SLOT(6)
    }
#undef I8_SLOT
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- W8 on 128-row tiles: one phase per 128-k step.  LOAD: the requests of step kt + 3 first (two activation pieces by LDS-DMA,
  // then the wave's four 1-KiB code fragments into register set (kt + 3) % 4), then the 16 activation fragment reads of step kt and
  // vmcnt(16): of the three batches in flight only the activations of step kt + 1 - the oldest two requests - must have landed
  // before the barrier lets anyone read them; COMPUTE: 16 MFMAs on register set kt % 4, no expand.
  auto step4_w8 = [&](int kt, auto slot_c) {
    constexpr int SLOT = decltype(slot_c)::value;
    constexpr int slot_new = (SLOT + DEPTH) % NSLOT;
    constexpr int A_IMM = SLOT * A_SLOT;
    __builtin_amdgcn_s_setprio(1);
    const int ktn = __builtin_amdgcn_readfirstlane(kt + DEPTH);
    issue_a8(a_base, ktn, slot_new);
    asm volatile("" ::: "memory");
    issue_wq(w_base, ktn, std::integral_constant<int, slot_new>{});
    i32x4 xa[4][4];  // [token tile][slice]
    asm volatile(
        "ds_read_b128 %[x00], %[fa0] offset:%c[aimm]\n\tds_read_b128 %[x01], %[fa1] offset:%c[aimm]\n\t"
        "ds_read_b128 %[x02], %[fa2] offset:%c[aimm]\n\tds_read_b128 %[x03], %[fa3] offset:%c[aimm]\n\t"
        "ds_read_b128 %[x10], %[fa0] offset:%c[aimm]+4096\n\tds_read_b128 %[x11], %[fa1] offset:%c[aimm]+4096\n\t"
        "ds_read_b128 %[x12], %[fa2] offset:%c[aimm]+4096\n\tds_read_b128 %[x13], %[fa3] offset:%c[aimm]+4096\n\t"
        "ds_read_b128 %[x20], %[fa0] offset:%c[aimm]+8192\n\tds_read_b128 %[x21], %[fa1] offset:%c[aimm]+8192\n\t"
        "ds_read_b128 %[x22], %[fa2] offset:%c[aimm]+8192\n\tds_read_b128 %[x23], %[fa3] offset:%c[aimm]+8192\n\t"
        "ds_read_b128 %[x30], %[fa0] offset:%c[aimm]+12288\n\tds_read_b128 %[x31], %[fa1] offset:%c[aimm]+12288\n\t"
        "ds_read_b128 %[x32], %[fa2] offset:%c[aimm]+12288\n\tds_read_b128 %[x33], %[fa3] offset:%c[aimm]+12288\n\t"
        "s_waitcnt vmcnt(16) lgkmcnt(0)"
        : [x00] "=&v"(xa[0][0]), [x01] "=&v"(xa[0][1]), [x02] "=&v"(xa[0][2]), [x03] "=&v"(xa[0][3]), [x10] "=&v"(xa[1][0]),
          [x11] "=&v"(xa[1][1]), [x12] "=&v"(xa[1][2]), [x13] "=&v"(xa[1][3]), [x20] "=&v"(xa[2][0]), [x21] "=&v"(xa[2][1]),
          [x22] "=&v"(xa[2][2]), [x23] "=&v"(xa[2][3]), [x30] "=&v"(xa[3][0]), [x31] "=&v"(xa[3][1]), [x32] "=&v"(xa[3][2]),
          [x33] "=&v"(xa[3][3])
        : [fa0] "v"(fa_lo[0]), [fa1] "v"(fa_lo[1]), [fa2] "v"(fa_lo[2]), [fa3] "v"(fa_lo[3]), [aimm] "i"(A_IMM)
        : "memory");
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    // ---- COMPUTE
    if constexpr (W8D) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int t = 0; t < 4; ++t) R[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa[t][ks], wq[SLOT][ks], R[t], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- W8: half-step h = 2 kt + P - the 32-k slices 2 P and 2 P + 1 of all 8 token tiles against this wave's two 16-byte weight
  // fragments from weight slot h % 3.  LOAD: the 16 activation fragment reads, then the requests (the codes of half-step h + 2
  // FIRST, then this half's share of the activations of step kt + 2), and only THEN the wait for this half-step's own codes: a wave
  // reads the weight rows it requested itself, so their landing needs no barrier - just vmcnt(10) (this LOAD's four requests, the
  // previous LOAD's four and the two activation pieces before those may stay in flight) right in front of the two weight reads,
  // one phase later than a wait at the end of the previous LOAD would sit (the codes travel two half-steps ahead: that phase is a
  // quarter of their time budget; measured +-0 at 8192 x 4096 x 4096 - 127.0 vs 127.4 us per launch: the loop is paced by the 64 KiB
  // the two LDS-DMA streams bring in per step, ~38 GB/s per CU, not by this wait).  The activations, which every wave
  // reads, keep their wait in front of the barrier: at P = 1 vmcnt(8) retires every piece of step kt + 1.  COMPUTE: 16 MFMAs.
  auto half_step_w8 = [&](int kt, auto slot_c, auto half_c) {
    constexpr int SLOT = decltype(slot_c)::value;
    constexpr int P = decltype(half_c)::value;
    constexpr int slot_new = (SLOT + DEPTH) % NSLOT;
    constexpr int WS = (2 * SLOT + P) % 3, WS_NEW = (2 * SLOT + P + 2) % 3;
    constexpr int A_IMM = (SLOT == 2 ? 0 : SLOT * A_SLOT);  // tile t: + 4096 t
    __builtin_amdgcn_s_setprio(1);
    const int ktn = __builtin_amdgcn_readfirstlane(kt + DEPTH);
    const int a_soff = ktn * I8_BK, w_soff = (kt + 1) * (2 * W8_SLOT) + P * 2048;
    const uint32_t m0a0 = m0_a + slot_new * A_SLOT + (2 * P) * 1024, m0a1 = m0a0 + 1024;
    const uint32_t m0w0 = m0_w + WS_NEW * W8_SLOT, m0w1 = m0w0 + 1024;
    i32x4 xk[8][2];  // [token tile][slice of this half]
    i32x4 w0, w1;    // the half's weight fragments: slices 2P, 2P + 1
    asm volatile(
        "ds_read_b128 %[x00], %[fa0] offset:%c[aimm]\n\tds_read_b128 %[x01], %[fa1] offset:%c[aimm]\n\t"
        "ds_read_b128 %[x10], %[fa0] offset:%c[aimm]+4096\n\tds_read_b128 %[x11], %[fa1] offset:%c[aimm]+4096\n\t"
        "ds_read_b128 %[x20], %[fa0] offset:%c[aimm]+8192\n\tds_read_b128 %[x21], %[fa1] offset:%c[aimm]+8192\n\t"
        "ds_read_b128 %[x30], %[fa0] offset:%c[aimm]+12288\n\tds_read_b128 %[x31], %[fa1] offset:%c[aimm]+12288\n\t"
        "ds_read_b128 %[x40], %[fa0] offset:%c[aimm]+16384\n\tds_read_b128 %[x41], %[fa1] offset:%c[aimm]+16384\n\t"
        "ds_read_b128 %[x50], %[fa0] offset:%c[aimm]+20480\n\tds_read_b128 %[x51], %[fa1] offset:%c[aimm]+20480\n\t"
        "ds_read_b128 %[x60], %[fa0] offset:%c[aimm]+24576\n\tds_read_b128 %[x61], %[fa1] offset:%c[aimm]+24576\n\t"
        "ds_read_b128 %[x70], %[fa0] offset:%c[aimm]+28672\n\tds_read_b128 %[x71], %[fa1] offset:%c[aimm]+28672\n\t"
        "s_mov_b32 m0, %[m0w0]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[wv0], %[wrs], %[wsoff] offen lds\n\t"
        "s_mov_b32 m0, %[m0w1]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[wv1], %[wrs], %[wsoff] offen lds\n\t"
        "s_mov_b32 m0, %[m0a0]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[av0], %[ars], %[asoff] offen lds\n\t"
        "s_mov_b32 m0, %[m0a1]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[av1], %[ars], %[asoff] offen lds\n\t"
        "s_waitcnt vmcnt(10)\n\t"
        "ds_read_b128 %[w0], %[fw0] offset:%c[wimm]\n\tds_read_b128 %[w1], %[fw1] offset:%c[wimm]\n\t"
        "s_waitcnt vmcnt(%c[vmend]) lgkmcnt(0)"
        : [x00] "=&v"(xk[0][0]), [x01] "=&v"(xk[0][1]), [x10] "=&v"(xk[1][0]), [x11] "=&v"(xk[1][1]), [x20] "=&v"(xk[2][0]),
          [x21] "=&v"(xk[2][1]), [x30] "=&v"(xk[3][0]), [x31] "=&v"(xk[3][1]), [x40] "=&v"(xk[4][0]), [x41] "=&v"(xk[4][1]),
          [x50] "=&v"(xk[5][0]), [x51] "=&v"(xk[5][1]), [x60] "=&v"(xk[6][0]), [x61] "=&v"(xk[6][1]), [x70] "=&v"(xk[7][0]),
          [x71] "=&v"(xk[7][1]), [w0] "=&v"(w0), [w1] "=&v"(w1)
        : [fa0] "v"(SLOT == 2 ? fa_hi[2 * P] : fa_lo[2 * P]), [fa1] "v"(SLOT == 2 ? fa_hi[2 * P + 1] : fa_lo[2 * P + 1]), [aimm] "i"(A_IMM),
          [fw0] "v"(fw8_0), [fw1] "v"(fw8_1), [wimm] "i"(WS * W8_SLOT), [av0] "v"(a_voff[2 * P]), [av1] "v"(a_voff[2 * P + 1]),
          [wv0] "v"(w_voff0), [wv1] "v"(w_voff1), [ars] "s"(a_rs), [wrs] "s"(w_rs), [m0a0] "s"(m0a0), [m0a1] "s"(m0a1), [m0w0] "s"(m0w0),
          [m0w1] "s"(m0w1), [asoff] "s"(a_soff), [wsoff] "s"(w_soff), [vmend] "i"(P == 1 ? 8 : 10)
        : "memory");
    asm volatile("s_barrier" : "+v"(w0)::"memory");
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    // ---- COMPUTE
#pragma unroll
    for (int t = 0; t < 8; ++t) R[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xk[t][0], w0, R[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 8; ++t) R[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xk[t][1], w1, R[t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  using std::integral_constant;
#ifndef LQER_I8_KSPLIT
#define LQER_I8_KSPLIT 1  // 0: the token-tile split for every mode (A/B builds)
#endif
  auto main_loop = [&](auto mode_c) {
    if constexpr (W8D) {
      // unrolled by the ring size (activation slot and register set are compile-time constants) with NO exit inside the four steps:
      // K is walked in multiples of 512 - a step past its end multiplies whatever the activation slot holds (int8 of an earlier
      // step) with the zeros the range-checked weight loads return: exactly nothing (at most 3 steps per tile; none at K = 4096).
      // Why: hipcc structurizes `if (kt + i < nk)` / `break` into flags tested at the loop latch, its waitcnt pass then sees a
      // path from step 1 straight back to step 0 and waits for all but 6-9 requests in front of step 0's MFMAs - the prefetch
      // distance would collapse to one step every fourth step.
      for (int kt = 0; kt < nk; kt += NSLOT) {
        step4_w8(kt, integral_constant<int, 0>{});
        step4_w8(kt + 1, integral_constant<int, 1>{});
        step4_w8(kt + 2, integral_constant<int, 2>{});
        step4_w8(kt + 3, integral_constant<int, 3 % NSLOT>{});
      }
    } else if constexpr (W8) {
      for (int kt = 0; kt < nk; kt += NSLOT) {  // (three steps = six half-steps: activation slot kt % 3, weight slot h % 3)
        half_step_w8(kt, integral_constant<int, 0>{}, integral_constant<int, 0>{});
        half_step_w8(kt, integral_constant<int, 0>{}, integral_constant<int, 1>{});
        if (kt + 1 < nk) {
          half_step_w8(kt + 1, integral_constant<int, 1>{}, integral_constant<int, 0>{});
          half_step_w8(kt + 1, integral_constant<int, 1>{}, integral_constant<int, 1>{});
        }
        if (kt + 2 < nk) {
          half_step_w8(kt + 2, integral_constant<int, 2>{}, integral_constant<int, 0>{});
          half_step_w8(kt + 2, integral_constant<int, 2>{}, integral_constant<int, 1>{});
        }
      }
    } else if constexpr (NT == 4) {
      for (int kt = 0; kt < nk; kt += NSLOT) {  // unrolled by the ring size: slots are compile-time constants
        step4(kt, integral_constant<int, 0>{}, mode_c);
        if (kt + 1 < nk) step4(kt + 1, integral_constant<int, 1>{}, mode_c);
        if (kt + 2 < nk) step4(kt + 2, integral_constant<int, 2>{}, mode_c);
        if (kt + 3 < nk) step4(kt + 3, integral_constant<int, 3 % NSLOT>{}, mode_c);
      }
    } else {
      auto hs = [&](int kt, auto slot_c, auto half_c) {
        if constexpr (LQER_I8_KSPLIT && decltype(mode_c)::value != I8_MODE_FOLD) half_step_k(kt, slot_c, half_c, mode_c);
        else half_step(kt, slot_c, half_c, mode_c);
      };
      for (int kt = 0; kt < nk; kt += NSLOT) {  // unrolled by the ring size: slots are compile-time constants
        hs(kt, integral_constant<int, 0>{}, integral_constant<int, 0>{});
        hs(kt, integral_constant<int, 0>{}, integral_constant<int, 1>{});
        if (kt + 1 < nk) {
          hs(kt + 1, integral_constant<int, 1>{}, integral_constant<int, 0>{});
          hs(kt + 1, integral_constant<int, 1>{}, integral_constant<int, 1>{});
        }
        if (kt + 2 < nk) {
          hs(kt + 2, integral_constant<int, 2>{}, integral_constant<int, 0>{});
          hs(kt + 2, integral_constant<int, 2>{}, integral_constant<int, 1>{});
        }
      }
    }
  };
  int tile_mode = I8_MODE_NONE;  // (workgroup-uniform)
  if constexpr (SHIFT) tile_mode = __builtin_amdgcn_readfirstlane((int)((mode_word >> mode_sh) & 0xffu));
  if constexpr (!SHIFT) {
    main_loop(integral_constant<int, I8_MODE_NONE>{});
  } else {
    if (tile_mode == I8_MODE_PRESHIFT1) main_loop(integral_constant<int, I8_MODE_PRESHIFT1>{});
    else if (tile_mode == I8_MODE_PRESHIFT) main_loop(integral_constant<int, I8_MODE_PRESHIFT>{});
    else if (tile_mode == I8_MODE_FOLD) main_loop(integral_constant<int, I8_MODE_FOLD>{});
    else main_loop(integral_constant<int, I8_MODE_NONE>{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the prefetches issued past the end of K have drained
  if (!late) asm volatile("s_barrier" ::: "memory");
#ifdef LQER_CLOCKPROBE
  I8_STAMP(cp_c[2], cp_r[2]);
#endif
  if constexpr (SHIFT) {  // FOLD: the last half-step's last tile
    if (tile_mode == I8_MODE_FOLD) {
#pragma unroll
      for (int j = 0; j < 16; ++j) R[NT - 1][j] = (int)(((uint32_t)Gd[j] << sv) + (uint32_t)R[NT - 1][j]);
    }
  }
  // every wave is past its last LDS read of the ring: the epilogue may overwrite it after one more barrier
  asm volatile("s_barrier" ::: "memory");
  // the next tile of this workgroup: its first step goes into ring slot 0 NOW (one panel of xAq: the epilogue keeps out of
  // slot 0), its row constants are requested; the second step follows when the epilogue has released slot 1
  if constexpr (MRX) {
    // the tile's row constants: folded a tile ahead from the band's granules - unless some tag was missing then (the first tile: nothing
    // had been published when its tables were written).  Then: read again, bounded; a band that stays incomplete (its producers are not
    // resident: another stream holds their CUs) is computed here, all sixteen items - same routine, same bits.  (Behind the main
    // loop: the first tile's wait passes under it; the fall-back there is the register-light mrx_item.)
    if (wg_any(first || t_bad != 0)) {
      int sweeps = 0;
      for (;;) {
        load_tables(m0);
        const bool miss = (g.tuning & LQER_TUNE_AMAX_XCH_MISS) != 0;  // (test knob: take the fall-back)
        if (!wg_any(t_bad != 0) && !miss) break;
        if (miss || ++sweeps > LQER_XCH_SWEEPS) {
          for (int sg = 0; sg < LQER_AMAX_NSEG; ++sg) mrx_item((m0 / BM) * LQER_AMAX_NSEG + sg);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          load_tables(m0);
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
      write_tables();
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef LQER_CLOCKPROBE
      if (first) {
        I8_STAMP(cp_p[2], cp_x);  // the band's granules seen, tables written
        cp_tries = (unsigned long long)sweeps;
      }
#endif
    }
  }
  const int vb_next = vb + (int)gridDim.x;
  const bool has_next = !XCH && vb_next < nt;  // (workgroup-uniform; XCH: one round by construction)
  int m0_next = 0, n0_next = 0, tn_next = 0;
  if (has_next) {
    const int tile_n = tile_of(vb_next);
    const int tm_n = tile_n / g.tiles_n;
    tn_next = tile_n - tm_n * g.tiles_n;
    m0_next = tm_n * BM, n0_next = tn_next * BN;
    if constexpr (W8) {
      if (one_panel) {
        issue_a8(xq8 + (int64_t)m0_next * Kp8, 0, 0);
        if constexpr (!W8D) issue_w8(g.w8 + (size_t)tn_next * nk * 2 * W8_SLOT, 0, 0);  // (W8D: registers - requested at the tile's head)
      }
    } else {
      if (one_panel) issue_step(xq8 + (int64_t)m0_next * Kp8, g.w8 + (size_t)tn_next * nk * I8_WBLOCK, 0, 0);
    }
    load_tables(m0_next);
    load_mode(tn_next);
  }

  // ---- epilogue ---------------------------------------------------------------------------------------------------------
  {  // (its own per-lane constants from freshly laundered ids: see the head of the tile loop)
  int lane_e = lane_k, wave_e = wave_k;
  asm volatile("" : "+v"(lane_e));
  asm volatile("" : "+s"(wave_e));
  const int lane = lane_e, wave = wave_e;
  const int l31 = lane & 31, lh = lane >> 5;
  const int ep_out = one_panel ? (wave < 6 ? 2 * A_SLOT + wave * EP_OUT_WAVE : OFF_W + 2 * (W8 ? W8_SLOT : W_SLOT) + (wave - 6) * EP_OUT_WAVE)
                               : EP_OUT + wave * EP_OUT_WAVE;
  // lane: output column n = n0 + 32 wave + (lane & 31); register j of tile i: token row m0 + 32 i + (j&3) + 8 (j>>2) + 4 lh.
  // Everything per row is tabulated in LDS once (row scale, B_out exponent differences), everything per column is a lane
  // constant; the side product's code is fully static per (limbs, slices) pair.
  const int n = n0 + wave * 32 + l31;
  const float* const wscale = (const float*)(g.w8 + (size_t)g.tiles_n * nk * (W8 ? 2 * W8_SLOT : I8_WBLOCK));
  // (EP_TAB + 0: 2^(ex[m] - mbits), read in the conversion pass below)
  const float* const tab_up = (const float*)(smem + EP_TAB + 1024);  // B_out: 2^(mbits - e[m]) ...
  const float* const tab_dn = (const float*)(smem + EP_TAB + 2048);  // ... and 2^(e[m] - mbits)
  const float* const tab_es = (const float*)(smem + EP_TAB + 3072);  // ... and 1e-9 * 2^(mbits - e[m])
  // the side product's operands: the tile's rows of xAq by LDS-DMA into the ring's place (panels of 64 columns, the
  // activation tile's row pitch and swizzle); this wave's B^T fragments in registers when there are at most 8 (limb,
  // 16-deep slice) pairs - rank 64 with fp16 A / B, rank 128 with 8-bit A / B -, else re-fetched from L2 for every token
  // tile.  Their latency passes under the conversion of the integer tile (below).
  // exchange: the granules of this tile's 32 row quads x every column tile (4 KiB), requested now by LDS-DMA (no registers held across
  // the conversion pass; first read at sc0 - XCH_GATHER_AUX - or, with XCD tile blocks, sc1) into the gather region behind the xAq panel
  // (round 6; it was activation slot 3): wave w (0, 1) brings quads 16 w .. 16 w + 15 with two requests - request j, lane l:
  // quad 8 j + l / 8, piece l % 8 (16 B = the granules of column tiles 2 p, 2 p + 1) - and reads back what it requested itself
  // (vmcnt, no barrier).  The round trip passes under the staging and the conversion.
  const int tid_e = wave * 64 + lane;
  if constexpr (XCH_OK) {
    if (xch && wave < 2 && !xg_early) {
      if (g.xcd_bm > 0) {  // (XCD blocks: a row band's column tiles sit on several XCDs - agent scope from the first read on)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xch_rsrc, (lds_void*)(smem + G::EP_GATHER + wave * 2048 + j * 1024), 16,
                                                   (int)(((m0 >> 2) + 16 * wave + 8 * j + (lane >> 3)) * (LQER_AMAX_NSEG * 8)) + (lane & 7) * 16, 0, 0, 16);
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xch_rsrc, (lds_void*)(smem + G::EP_GATHER + wave * 2048 + j * 1024), 16,
                                                   (int)(((m0 >> 2) + 16 * wave + 8 * j + (lane >> 3)) * (LQER_AMAX_NSEG * 8)) + (lane & 7) * 16, 0, 0,
                                                   XCH_GATHER_AUX);
      }
    }
  }
  bf16x8 sb[LOWRANK ? 8 : 1];
#pragma unroll
  for (int i = 0; i < (LOWRANK ? 8 : 1); ++i) sb[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};  // (defined on every path: not carried around the tile loop)
  const int nslices = LOWRANK ? g.rp / 16 : 0;  // 16-deep slices per limb
  const bf16_t* const bt_lane = LOWRANK ? g.bt + (int64_t)n * g.rp + 8 * lh : nullptr;  // + l * Np * rp + 16 ks
  const int64_t bt_limb = (int64_t)g.Np * g.rp;
  if constexpr (LOWRANK && !XCH) {
    // (exact range: the 128-byte pieces of a narrow xAq run into the next row, past the last row of the buffer they read 0)
    const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(g.xaq + (int64_t)m0 * g.xaq_ld), 0, BM * g.xaq_ld * 2, 0x00020000);
    const int npanel = (g.rp + 63) >> 6;
    for (int pn = 0; pn < npanel; ++pn)
#pragma unroll
      for (int i = 0; i < NPA; ++i) {
        const int row = wave * (8 * NPA) + i * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void*)(smem + ep_stage + pn * PANEL + wave * (8 * NPA) * 128 + i * 1024), 16,
                                                 row * g.xaq_ld * 2 + chunk * 16, pn * 128, 0, 0);
      }
  }
  // B^T fragments of the common (limbs, slices per limb) pairs: static code (no runtime divisions, no exec masks)
  auto load_sb = [&](auto nl_c, auto nsl_c) {
    constexpr int NL = decltype(nl_c)::value, NSL = decltype(nsl_c)::value;
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
      for (int ks = 0; ks < NSL; ++ks) sb[l * NSL + ks] = *(const bf16x8*)(bt_lane + l * bt_limb + ks * 16);
  };
  const int side_key = LOWRANK ? g.b_limbs * 16 + nslices : 0;  // (wave-uniform)
  if constexpr (LOWRANK && !XCH) {
    using std::integral_constant;
    switch (side_key) {
      case 16 + 1: load_sb(integral_constant<int, 1>{}, integral_constant<int, 1>{}); break;
      case 16 + 2: load_sb(integral_constant<int, 1>{}, integral_constant<int, 2>{}); break;
      case 16 + 4: load_sb(integral_constant<int, 1>{}, integral_constant<int, 4>{}); break;
      case 16 + 8: load_sb(integral_constant<int, 1>{}, integral_constant<int, 8>{}); break;
      case 32 + 1: load_sb(integral_constant<int, 2>{}, integral_constant<int, 1>{}); break;
      case 32 + 2: load_sb(integral_constant<int, 2>{}, integral_constant<int, 2>{}); break;
      case 32 + 4: load_sb(integral_constant<int, 2>{}, integral_constant<int, 4>{}); break;
      default: break;
    }
  }
  const float ws = XCH ? ws_x : wscale[n];
  const float bv = XCH ? bv_x : (g.bias ? g.bias[n] : 0.f);
#ifdef LQER_CLOCKPROBE
  I8_STAMP(cp_a, cp_x);
#endif
  // the integer tile -> v = float(R) * xs[m] * ws[n] + bias[n], in place, two elements per packed fp32 instruction
  typedef __attribute__((ext_vector_type(2))) float f2;
  {
    const f2 ws2 = {ws, ws}, bv2 = {bv, bv};
    const uint32_t txs = lds0 + EP_TAB + 16 * lh;  // row 32 i + 8 q + 4 lh: + 128 i + 32 q bytes
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      // (asm: a table read hipcc can see would be preceded by a vmcnt(0), i.e. wait for the DMA and the loads just issued)
      f32x4 xs[4];
      asm volatile("ds_read_b128 %0, %4 offset:%c5\n\tds_read_b128 %1, %4 offset:%c5+32\n\tds_read_b128 %2, %4 offset:%c5+64\n\t"
                   "ds_read_b128 %3, %4 offset:%c5+96\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(xs[0]), "=&v"(xs[1]), "=&v"(xs[2]), "=&v"(xs[3])
                   : "v"(txs), "i"(128 * i)
                   : "memory");
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < 4; t += 2) {
          const int j = 4 * q + t;
          const f2 rf = {(float)R[i][j], (float)R[i][j + 1]};
          const f2 v = __builtin_elementwise_fma(rf * (f2){xs[q][t], xs[q][t + 1]}, ws2, bv2);
          R[i][j] = __float_as_int(v[0]), R[i][j + 1] = __float_as_int(v[1]);
        }
      // (pinned here: left to itself the optimiser sinks a tile's conversion to its use in the store phase and keeps the 16
      // table registers of every tile alive until then - scratch spills inside the tile loop)
      asm volatile("" : "+v"(R[i]));
    }
  }
#ifdef LQER_CLOCKPROBE
  asm volatile("" ::"v"(R[0][0]), "v"(R[NT - 1][15]));
  I8_STAMP(cp_b, cp_x);
#endif
  // exchange: the row's B_out scales from the gathered maximum (write_tables' arithmetic), in front of the barrier below
  auto xch_tables = [&](int e_row) {  // (e_row = block_exponent of the row's maximum)
    if constexpr (XCH_OK) {
      int up = g.bout.mbits - e_row;
      up = up > 126 ? 126 : (up < -126 ? -126 : up);
      const uint32_t upb = (uint32_t)(127 + up) << 23;
      asm volatile("ds_write_b32 %0, %1 offset:1024\n\tds_write_b32 %0, %2 offset:2048\n\tds_write_b32 %0, %3 offset:3072" ::"v"(
                       lds0 + EP_TAB + 4 * tid_e),
                   "v"(upb), "v"((uint32_t)(127 - up) << 23), "v"(1e-9f * __uint_as_float(upb))
                   : "memory");
    }
  };
  if constexpr (XCH_OK) {
    if (xch && wave < 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (this wave's requests: the granules - and the xAq pieces, awaited below anyway)
      const int qw = lane >> 2;  // this row's quad within the wave's 16
      const uint32_t ga = lds0 + G::EP_GATHER + wave * 2048 + (qw >> 3) * 1024 + (qw & 7) * 128;
      u32x4 gq[LQER_AMAX_NSEG / 2];  // piece p: {bytes, tag} of column tiles 2 p and 2 p + 1
      asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:32\n\tds_read_b128 %3, %8 offset:48\n\t"
                   "ds_read_b128 %4, %8 offset:64\n\tds_read_b128 %5, %8 offset:80\n\tds_read_b128 %6, %8 offset:96\n\t"
                   "ds_read_b128 %7, %8 offset:112\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(gq[0]), "=&v"(gq[1]), "=&v"(gq[2]), "=&v"(gq[3]), "=&v"(gq[4]), "=&v"(gq[5]), "=&v"(gq[6]), "=&v"(gq[7])
                   : "v"(ga)
                   : "memory");
      const int ntile = g.tiles_n;  // (granules of column tiles that do not exist are never looked at)
      const int goff = (int)(((m0 >> 2) + 16 * wave + qw) * (LQER_AMAX_NSEG * 8));
      bool ok = !(g.tuning & LQER_TUNE_AMAX_XCH_MISS);  // (test knob: take the fall-back)
      // Round 6: the tag test is BRANCH-FREE (one xor / and / or per granule under wave-uniform masks) and the poll loop sits behind one
      // wave-uniform ballot - the short-circuit form compiled to a chain of exec-mask branches that cost waves 0-1 (and, at the barrier
      // below, everybody) ~2,000 cycles per tile with every granule present.
      auto stale = [&]() {
        uint32_t bad = 0;
#pragma unroll
        for (int j = 0; j < LQER_AMAX_NSEG / 2; ++j) {
          const uint32_t v0 = 2 * j < ntile ? ~0u : 0u, v1 = 2 * j + 1 < ntile ? ~0u : 0u;  // (uniform)
          bad |= ((gq[j][1] ^ xtag) & v0) | ((gq[j][3] ^ xtag) & v1);
        }
        return bad;
      };
      uint32_t bad = stale();
      if (ok && __builtin_amdgcn_ballot_w64(bad != 0) != 0) {  // (rare: some granule of this wave's 64 rows is not there yet)
        for (int tries = 0; tries < XCH_SWEEPS && __builtin_amdgcn_ballot_w64(bad != 0) != 0; ++tries) {
#ifdef LQER_CLOCKPROBE
          cp_tries = (unsigned long long)(tries + 1);
#endif
          __builtin_amdgcn_s_sleep(8);
          if (bad != 0) {
#pragma unroll
            for (int j = 0; j < LQER_AMAX_NSEG / 2; ++j)
              if ((2 * j < ntile && gq[j][1] != xtag) || (2 * j + 1 < ntile && gq[j][3] != xtag))
                gq[j] = __builtin_amdgcn_raw_buffer_load_b128(xch_rsrc, goff + j * 16, 0, 16);  // sc1
          }
          bad = stale();
        }
        ok = bad == 0;
      }
      if (ok) {
        const int sh = 8 * (lane & 3);
        uint32_t eb = 0;  // (exponent bytes are >= 0: e - emin)
#pragma unroll
        for (int j = 0; j < LQER_AMAX_NSEG / 2; ++j) {  // (uniform masks again: no branches)
          const uint32_t v0 = 2 * j < ntile ? 0xffu : 0u, v1 = 2 * j + 1 < ntile ? 0xffu : 0u;
          eb = max(eb, max((gq[j][0] >> sh) & v0, (gq[j][2] >> sh) & v1));
        }
        xch_tables((int)eb + g.bout.emin);
      } else {
        asm volatile("ds_write_b32 %0, %1 offset:1020" ::"v"(lds0 + EP_TAB), "v"(1u) : "memory");  // the workgroup's vote
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifdef LQER_CLOCKPROBE
  I8_STAMP(cp_cc, cp_x);
#endif
  __syncthreads();  // the xAq tile has landed for every wave
  if constexpr (XCH_OK) {
    if (xch) {
      uint32_t vote;
      asm volatile("ds_read_b32 %0, %1 offset:1020\n\ts_waitcnt lgkmcnt(0)" : "=v"(vote) : "v"(lds0 + EP_TAB) : "memory");
      if (__builtin_amdgcn_readfirstlane((int)vote) != 0) {
        // fall-back (workgroup-uniform): some row did not see a neighbour's granules - every column tile's maxima are computed here,
        // same routine, same bits (the output regions are not in use yet: the reduce buffer is free)
        // (one row group at a time: the accumulators of the tile are live here - 16 fragment registers instead of 64)
        float mx[4] = {0.f, 0.f, 0.f, 0.f};
        for (int tnx = 0; tnx < g.tiles_n; ++tnx) {
          const bf16_t* const bl = g.bt + (int64_t)(tnx * BN + wave * 32 + l31) * g.rp + 8 * lh;
#pragma unroll 1
          for (int u = 0; u < 4; ++u) {
            f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int l = 0; l < g.b_limbs; ++l)
              for (int ks = 0; ks < xch_nsl; ++ks) {
                const bf16x8 bfr = *(const bf16x8*)(bl + l * bt_limb + ks * 16);
                const bf16x8 xfr = *(const bf16x8*)(g.xaq + (int64_t)(m0 + 32 * u + l31) * g.xaq_ld + ks * 16 + 8 * lh);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr, xfr, acc, 0, 0, 0);
              }
            float m = 0.f;
#pragma unroll
            for (int k = 0; k < 16; k += 2) m = fmaxf(fmaxf(m, fabsf(acc[k])), fabsf(acc[k + 1]));
            auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
            m = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            if (u == 0) mx[0] = fmaxf(mx[0], m);
            else if (u == 1) mx[1] = fmaxf(mx[1], m);
            else if (u == 2) mx[2] = fmaxf(mx[2], m);
            else mx[3] = fmaxf(mx[3], m);
          }
        }
        const float r = xch_reduce(mx);
        if (tid_e < BM) xch_tables(block_exponent(r, g.bout));
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
    }
  }
#ifdef LQER_CLOCKPROBE
  I8_STAMP(cp_e1, cp_e1r);
#endif
  // xAq fragment addresses: row l31 (+ 32 i: + 4096 B), chunk 2 (ks & 3) + lh, panel ks >> 2
  uint32_t xaddr[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) xaddr[c] = ep_stage + swz(l31, 2 * c + lh);
  // the side product of token tile i: static code for the common (limbs, slices per limb) pairs
  auto sbf = [&](int i) -> bf16x8 {  // (XCH: the fragments the prologue requested)
    if constexpr (XCH) return sbx[i];
    else return sb[i];
  };
  auto side_static = [&](int i, auto nl_c, auto nsl_c) {
    constexpr int NL = decltype(nl_c)::value, NSL = decltype(nsl_c)::value;
    f32x16 sp;
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
      for (int ks = 0; ks < NSL; ++ks) {
        const bf16x8 xf = *(const bf16x8*)(smem + xaddr[ks & 3] + (ks >> 2) * PANEL + i * 4096);
        if (l == 0 && ks == 0) {
          const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
          sp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, sbf(0), z, 0, 0, 0);
        } else {
          sp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, sbf(l * NSL + ks), sp, 0, 0, 0);
        }
      }
    return sp;
  };
  auto side = [&](int i) -> f32x16 {
    using std::integral_constant;
    switch (side_key) {
      case 16 + 1: return side_static(i, integral_constant<int, 1>{}, integral_constant<int, 1>{});
      case 16 + 2: return side_static(i, integral_constant<int, 1>{}, integral_constant<int, 2>{});
      case 16 + 4: return side_static(i, integral_constant<int, 1>{}, integral_constant<int, 4>{});
      case 16 + 8: return side_static(i, integral_constant<int, 1>{}, integral_constant<int, 8>{});
      case 32 + 1: return side_static(i, integral_constant<int, 2>{}, integral_constant<int, 1>{});
      case 32 + 2: return side_static(i, integral_constant<int, 2>{}, integral_constant<int, 2>{});
      case 32 + 4: return side_static(i, integral_constant<int, 2>{}, integral_constant<int, 4>{});
      default: break;
    }
    f32x16 sp = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int l = 0; l < g.b_limbs; ++l)  // (the same order: limb-major, slices ascending)
      for (int ks = 0; ks < nslices; ++ks) {
        const bf16x8 bf = *(const bf16x8*)(bt_lane + l * bt_limb + ks * 16);
        const bf16x8 xf = *(const bf16x8*)(smem + ep_stage + (ks >> 2) * PANEL + swz(l31 + 32 * i, 2 * (ks & 3) + lh));
        sp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, bf, sp, 0, 0, 0);
      }
    return sp;
  };
  unsigned char* const out_w = smem + ep_out;
  const int nb = n0 + wave * 32;
  const bool wide = DT != LQER_F32 && (g.ldy & 7) == 0 && nb + 32 <= g.N && (((uintptr_t)g.y) & 15) == 0;  // wave-uniform
  // 16-byte stores of whole 8-column pieces through a buffer descriptor whose range ends with row M - 1: rows of the
  // tile beyond M are dropped by the range check, no per-row branch
  const int rows_left = g.M - m0 < BM ? g.M - m0 : BM;
  const auto y_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((bf16_t*)g.y + (int64_t)m0 * g.ldy + nb), 0,
                                                        (int)((int64_t)(rows_left - 1) * g.ldy * 2 + 64), 0x00020000);
  // per element, two at a time in packed fp32 (v_pk_mul / v_pk_add / v_pk_fma_f32):
  //   v = float(R) * xs[m] * ws[n] + bias[n]
  //   B_out (block_fp.py:55-65 on the signed value - every step is odd-symmetric): t = (s +- 1e-9) * 2^(mbits-e);
  //   r = rne(t) = (t + 1.5 * 2^23) - 1.5 * 2^23 (|t| <= 256); q = clamp(r, +-mmax) * 2^(e-mbits); |s| <= 1e-8 keeps s
  const float mmax = g.bout.mmax;
  const f2 magic = {12582912.0f, 12582912.0f};
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    float yv[16];
    f32x16 sp;
    if constexpr (LOWRANK) sp = side(i);
    // B_out with one block per row, FAST form: r = rne(fma(s, 2^(mbits-e), +-1e-9 2^(mbits-e))) - the same number as
    // rne((s +- 1e-9) 2^(mbits-e)): scaling by a power of two commutes with the rounding of the sum -, y = fma(r, 2^(e-mbits), v)
    // (r 2^(e-mbits) is exact).  Six packed instructions per pair of elements instead of fourteen; what it leaves out - the clamp
    // at +-mmax (only a row's largest element can round up to 2^mbits) and the |s| <= 1e-8 pass-through - is detected through a
    // running max |r| / min |s| and sends the WAVE's whole token tile through the general form below (rare: ~1e-3 of the tiles).
    bool general = !(LOWRANK && BOUT == 2);
    if constexpr (LOWRANK && BOUT == 2) {
      float rmax = 0.f, smin = 3.0e38f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rloc = 32 * i + 8 * q + 4 * lh;  // + (j & 3)
        const f32x4 up4 = *(const f32x4*)(tab_up + rloc), dn4 = *(const f32x4*)(tab_dn + rloc), es4 = *(const f32x4*)(tab_es + rloc);
#pragma unroll
        for (int t = 0; t < 4; t += 2) {
          const int j = 4 * q + t;
          const f2 s2 = {sp[j], sp[j + 1]};
          const f2 c = {copysignf(es4[t], s2[0]), copysignf(es4[t + 1], s2[1])};
          // (v_rndne_f32 per element: 8 cycles per pair where the two packed magic-number adds take 16; same integers)
          const f2 u2 = __builtin_elementwise_fma(s2, (f2){up4[t], up4[t + 1]}, c);
          const f2 r = {__builtin_rintf(u2[0]), __builtin_rintf(u2[1])};
          rmax = fmaxf(fmaxf(fabsf(r[0]), fabsf(r[1])), rmax);
          smin = fminf(fminf(fabsf(s2[0]), fabsf(s2[1])), smin);
          const f2 y = __builtin_elementwise_fma(r, (f2){dn4[t], dn4[t + 1]}, (f2){__int_as_float(R[i][j]), __int_as_float(R[i][j + 1])});
          yv[j] = y[0], yv[j + 1] = y[1];
        }
      }
      general = __builtin_amdgcn_ballot_w64(rmax > mmax || smin <= 1e-8f) != 0;  // (wave-uniform)
    }
    if (general)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rloc = 32 * i + 8 * q + 4 * lh;  // + (j & 3)
      f32x4 up4 = {0, 0, 0, 0}, dn4 = {0, 0, 0, 0};
      if constexpr (LOWRANK && BOUT == 2) up4 = *(const f32x4*)(tab_up + rloc), dn4 = *(const f32x4*)(tab_dn + rloc);
#pragma unroll
      for (int t = 0; t < 4; t += 2) {
        const int j = 4 * q + t;
        f2 v = {__int_as_float(R[i][j]), __int_as_float(R[i][j + 1])};
        if constexpr (LOWRANK) {
          f2 sv2 = {sp[j], sp[j + 1]};
          if constexpr (BOUT == 2) {
            const f2 eps = {copysignf(1e-9f, sv2[0]), copysignf(1e-9f, sv2[1])};
            const f2 tt = (sv2 + eps) * (f2){up4[t], up4[t + 1]};
            f2 r = (tt + magic) - magic;
            r[0] = __builtin_amdgcn_fmed3f(r[0], -mmax, mmax);
            r[1] = __builtin_amdgcn_fmed3f(r[1], -mmax, mmax);
            const f2 qv = r * (f2){dn4[t], dn4[t + 1]};
            sv2[0] = fabsf(sv2[0]) <= 1e-8f ? sv2[0] : qv[0];
            sv2[1] = fabsf(sv2[1]) <= 1e-8f ? sv2[1] : qv[1];
          }
          v += sv2;
        }
        yv[j] = v[0], yv[j + 1] = v[1];
      }
    }
#ifdef LQER_CLOCKPROBE
    if (i == 1) {  // tiles 0 and 1 computed (not yet stored)
      asm volatile("" ::"v"(yv[0]), "v"(yv[15]));
      I8_STAMP(cp_e2, cp_e2r);
    }
#endif
    if constexpr (DT == LQER_F32) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int m = m0 + 32 * i + (j & 3) + 8 * (j >> 2) + 4 * lh;
        if (m < g.M && n < g.N) ((float*)g.y)[(int64_t)m * g.ldy + n] = yv[j];
      }
    } else {
      // 16-bit outputs: two tiles (64 rows x 32 columns) at a time through this wave's LDS region, then 16-byte stores
      unsigned char* const dst = out_w + (i & 1) * 32 * 80 + l31 * 2 + 4 * lh * 80;
      // (pairs of rows: one v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32 per two elements, the halves stored by ds_write_b16 and
      // ds_write_b16_d16_hi; the packed bf16 conversion keeps a NaN a NaN - MI355X_MICROARCH.md, correctness boundaries)
#pragma unroll
      for (int j = 0; j < 16; j += 2) {
        uint32_t pk;  // (asm: left to itself the compiler splits the pair back into two scalar conversions)
        if constexpr (DT == LQER_BF16) asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(yv[j]), "v"(yv[j + 1]));
        else asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(yv[j]), "v"(yv[j + 1]));
        *(uint16_t*)(dst + ((j & 3) + 8 * (j >> 2)) * 80) = (uint16_t)pk;
        *(uint16_t*)(dst + ((j & 3) + 1 + 8 * (j >> 2)) * 80) = (uint16_t)(pk >> 16);
      }
      if (i & 1) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int row = u * 16 + (lane >> 2), ch = lane & 3;
          const u32x4 v = *(const u32x4*)(out_w + row * 80 + ch * 16);
          const int mrow = 32 * (i - 1) + row;  // row within the tile
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
  }  // epilogue
  if (!has_next) break;
  // every wave is done with the xAq stage, the row tables and its output region: the next tile may take them over
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  vb = vb_next, m0 = m0_next, n0 = n0_next, tn = tn_next;
  first = false;
  }  // tiles of this workgroup
#ifdef LQER_CLOCKPROBE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  I8_STAMP(cp_c[3], cp_r[3]);
  if (g_i8_stamp_buf && lane_k == 0) {  // (the workgroup's LAST tile)
    unsigned long long* o = g_i8_stamp_buf + ((size_t)blockIdx.x * 8 + wave_k) * 16;
    o[8] = cp_p[0] - cp_c[0], o[9] = cp_p[1] - cp_c[0], o[10] = cp_p[2] - cp_c[0], o[11] = cp_p[3] - cp_c[0];  // prologue sections (XCH)
    o[12] = cp_tries, o[13] = cp_r[0], o[14] = cp_r[3], o[15] = cp_c[0];  // polls of the gather; absolute 100 MHz ticks at start / end
    o[0] = cp_c[2] - cp_c[1], o[1] = cp_r[2] - cp_r[1];  // main loop: cycles, 100 MHz ticks
    o[2] = cp_c[1] - cp_c[0];                            // prologue (ring fill)
    o[3] = ((cp_a - cp_c[2]) & 0xffff) | (((cp_b - cp_a) & 0xffff) << 16) | (((cp_cc - cp_b) & 0xffff) << 32) | (((cp_e1 - cp_cc) & 0xffff) << 48);
    o[4] = cp_c[3] - cp_c[2], o[5] = cp_r[3] - cp_r[2];  // epilogue
    o[6] = cp_e1 - cp_c[2], o[7] = cp_e2 - cp_e1;         // epilogue: staging + barrier; the first two tiles' math
  }
#endif
}

template <int DT, int NT>
static int launch(GemmArgs g, bool lowrank, int bout, hipStream_t st);

// LQER_TUNE_XCD_BLOCK(t) for the int8 kernel: t token tiles per XCD block, applied only where the tile grid divides - every XCD gets
// nt / 8 tiles (nt % 8 == 0), a block is t x (nt / 8 / t) tiles, blocks tile the grid (tiles_m % t == 0, tiles_n % bn == 0) and there
// are exactly 8 of them.  0 = rows of weight tiles (the default map).
static int i8_xcd_block(const GemmArgs& g, int nt) {
  const int t = (g.tuning >> 4) & 0x3f;
  if (t <= 0 || nt % 8 != 0 || (nt / 8) % t != 0 || g.tiles_m % t != 0) return 0;
  const int bn = (nt / 8) / t;
  if (bn <= 0 || g.tiles_n % bn != 0 || (g.tiles_m / t) * (g.tiles_n / bn) != 8) return 0;
  return t;
}

// 8-bit weight codes: 128-row tiles with the codes straight into registers, or 256-row tiles with a half-step weight ring in LDS
template <int DT, int NT>
static int launch_w8(GemmArgs g, bool lowrank, int bout, hipStream_t st) {
  constexpr int BM = Geo<NT>::BM, KERNEL_LDS = Geo<NT>::KERNEL_LDS;
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = g.Np / BN;
  constexpr int CUS = 256;
  const int nt_all = g.tiles_m * g.tiles_n;
  const unsigned grid = (unsigned)(nt_all < CUS ? nt_all : CUS);
  g.xcd_bm = i8_xcd_block(g, nt_all);
#define LQER_I8_LAUNCH8(LR, BO)                                                                   \
  do {                                                                                            \
    static LdsLimitOnce lds_once;                                                                 \
    lds_once.set((const void*)k_lqer_gemm_i8<DT, LR, BO, false, NT, true>, KERNEL_LDS);             \
    k_lqer_gemm_i8<DT, LR, BO, false, NT, true><<<grid, 512, KERNEL_LDS, st>>>(g);                  \
  } while (0)
  if constexpr (NT == 4) {
    if (g.bout_xch) {  // one round, the B_out row maxima exchanged inside the launch: its own instantiation
      constexpr int LDS_X = Geo<NT>::KERNEL_LDS_XCH;
      static LdsLimitOnce lds_once;
      lds_once.set((const void*)k_lqer_gemm_i8<DT, true, 2, false, NT, true, true>, LDS_X);
      k_lqer_gemm_i8<DT, true, 2, false, NT, true, true><<<grid, 512, LDS_X, st>>>(g);
      return check_launch("lqer_gemm_i8 (8-bit weights, exchange)");
    }
  }
  if (!lowrank)
    LQER_I8_LAUNCH8(false, 0);
  else if (bout == 2)
    LQER_I8_LAUNCH8(true, 2);
  else
    LQER_I8_LAUNCH8(true, 0);
#undef LQER_I8_LAUNCH8
  return check_launch("lqer_gemm_i8 (8-bit weights)");
}

template <int DT, int NT>
static int launch(GemmArgs g, bool lowrank, int bout, hipStream_t st) {
  constexpr int BM = Geo<NT>::BM, KERNEL_LDS = Geo<NT>::KERNEL_LDS;
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = g.Np / BN;
  // persistent: at most one workgroup per CU (the LDS ring leaves room for one), each walks tiles b, b + grid, ...
  constexpr int CUS = 256;
  const int nt_all = g.tiles_m * g.tiles_n;
  const unsigned grid = (unsigned)(nt_all < CUS ? nt_all : CUS);
  g.xcd_bm = i8_xcd_block(g, nt_all);
#define LQER_I8_LAUNCH(LR, BO)                                                                    \
  do {                                                                                            \
    if (g.i8_shift) {                                                                             \
      static LdsLimitOnce lds_once;                                                               \
      lds_once.set((const void*)k_lqer_gemm_i8<DT, LR, BO, true, NT>, KERNEL_LDS);                  \
      k_lqer_gemm_i8<DT, LR, BO, true, NT><<<grid, 512, KERNEL_LDS, st>>>(g);                       \
    } else {                                                                                      \
      static LdsLimitOnce lds_once;                                                               \
      lds_once.set((const void*)k_lqer_gemm_i8<DT, LR, BO, false, NT>, KERNEL_LDS);                 \
      k_lqer_gemm_i8<DT, LR, BO, false, NT><<<grid, 512, KERNEL_LDS, st>>>(g);                      \
    }                                                                                             \
  } while (0)
  if constexpr (NT == 4) {
    if (g.bout_xch == 2) {  // several rounds, the row maxima computed and exchanged by the GEMM's own workgroups (MRX)
      constexpr int LDS_X = Geo<NT>::KERNEL_LDS_XCH;
      if (g.i8_shift) {
        static LdsLimitOnce lds_once;
        lds_once.set((const void*)k_lqer_gemm_i8<DT, true, 2, true, NT, false, false, true>, LDS_X);
        k_lqer_gemm_i8<DT, true, 2, true, NT, false, false, true><<<grid, 512, LDS_X, st>>>(g);
      } else {
        static LdsLimitOnce lds_once;
        lds_once.set((const void*)k_lqer_gemm_i8<DT, true, 2, false, NT, false, false, true>, LDS_X);
        k_lqer_gemm_i8<DT, true, 2, false, NT, false, false, true><<<grid, 512, LDS_X, st>>>(g);
      }
      return check_launch("lqer_gemm_i8 (multi-round exchange)");
    }
    if (g.bout_xch) {  // one round, the B_out row maxima exchanged inside the launch: its own instantiation
      constexpr int LDS_X = Geo<NT>::KERNEL_LDS_XCH;
      if (g.i8_shift) {
        static LdsLimitOnce lds_once;
        lds_once.set((const void*)k_lqer_gemm_i8<DT, true, 2, true, NT, false, true>, LDS_X);
        k_lqer_gemm_i8<DT, true, 2, true, NT, false, true><<<grid, 512, LDS_X, st>>>(g);
      } else {
        static LdsLimitOnce lds_once;
        lds_once.set((const void*)k_lqer_gemm_i8<DT, true, 2, false, NT, false, true>, LDS_X);
        k_lqer_gemm_i8<DT, true, 2, false, NT, false, true><<<grid, 512, LDS_X, st>>>(g);
      }
      return check_launch("lqer_gemm_i8 (exchange)");
    }
  }
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

#ifdef LQER_CLOCKPROBE
extern "C" int lqer_debug_set_i8_stamp_buffer(void* p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(i8::g_i8_stamp_buf), &p, sizeof(p));
}
#endif

// Rows of a tile of the int8 kernel: rounds of one tile per CU (the grid is persistent: a CU walks ceil(tiles / 256) tiles), a
// 128-row tile priced at 0.56 of a 256-row one (half the main loop and epilogue, the same ring fill and launch ramp; the weight
// expand per MFMA doubles).  Llama-7B projections at M = 2048: 4096 x 4096 fills 128 CUs with 256-row tiles and all 256 with
// 128-row ones; N = 11008: 2 rounds of 256 rows against 3 x 0.56.  LQER_TUNE_I8_ROWS_* pins the choice (tests: same bits).
int i8_tile_rows(const GemmArgs& g) {
  if (g.tuning & LQER_TUNE_I8_ROWS_128) return 128;
  if (g.tuning & LQER_TUNE_I8_ROWS_256) return 256;
  // (8-bit weight codes: the same rule - the 128-row kernel, codes straight into registers, takes 35.8 us per round of 4096-k tiles
  // against 63.5 us of the 256-row kernel's half-step LDS ring: 0.56 again.  M = 2048 x 4096 x 4096: 128 rows, all 256 CUs,
  // 35.8 us against 63 us on half of them; M = 8192: 256 rows, 127 us against 137 us)
  constexpr int64_t CUS = 256;
  const int64_t tn = g.Np / i8::BN;
  const int64_t r256 = (((g.M + 255) / 256) * tn + CUS - 1) / CUS, r128 = (((g.M + 127) / 128) * tn + CUS - 1) / CUS;
  return r128 * 56 < r256 * 100 ? 128 : 256;
}

// The int8 kernel exchanges the B_out row maxima itself (no pre-pass launch) when every tile is resident at once - one round of
// 128-row tiles, at most one per CU -, the row band has at most LQER_AMAX_NSEG column tiles (one granule each per row) and the side
// product is at most 2 limbs x 4 slices (its operands wait in registers under the ring fill).
// (the CUs this device really has - a partitioned part shows 32 of them: a grid that runs in rounds there would send every workgroup
// through its polls and its fall-back; queried once per device)
static int device_cus() {
  static std::atomic<int> cus[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  int c = cus[dev & 63].load(std::memory_order_relaxed);
  if (c == 0) {
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 1;
    cus[dev & 63].store(c, std::memory_order_relaxed);
  }
  return c;
}

bool i8_amax_exchange_ok(const GemmArgs& g, bool lowrank, int bout) {
  if (!lowrank || bout != 2 || g.bout_nblk != 1) return false;
  if (i8_tile_rows(g) != 128) return false;
  const int64_t tn = g.Np / i8::BN, tm = (g.M + 127) / 128;
  const int cus = device_cus();
  if (tn > LQER_AMAX_NSEG || tm * tn > (cus < 256 ? cus : 256)) return false;
  const int nsl = g.rp / 16;
  return (g.b_limbs == 1 || g.b_limbs == 2) && (nsl == 1 || nsl == 2 || nsl == 4);
}

// ... and over SEVERAL rounds of 128-row tiles (MRX: 4-bit weights): the persistent grid of min(tiles, 256) workgroups is resident at
// once, and every (row band, sixteenth of the columns) item of the pre-pass has a workgroup of its own.
bool i8_amax_mrx_ok(const GemmArgs& g, bool lowrank, int bout) {
  if (!lowrank || bout != 2 || g.bout_nblk != 1 || g.w_i8codes || g.rp > 64) return false;
  if (i8_tile_rows(g) != 128) return false;
  const int64_t tn = g.Np / i8::BN, tm = (g.M + 127) / 128;
  const int cus = device_cus();
  const int64_t grid = tm * tn < 256 ? tm * tn : 256;
  if (tm * tn <= (cus < 256 ? cus : 256) && tn <= LQER_AMAX_NSEG) return false;  // (one round: the exchange instantiation)
  if (grid > cus || tm * LQER_AMAX_NSEG > grid) return false;
  const int nsl = g.rp / 16;
  return (g.b_limbs == 1 || g.b_limbs == 2) && (nsl == 1 || nsl == 2 || nsl == 4);
}

// The int8 main loop needs: the int8 images (g.w8 set by the caller for an LQER_Q_MXINT_I8 descriptor), a token count of the
// tile kernels (M >= 128; below, the sign-magnitude image serves the weight-streaming and 64-row kernels), B_out pass-through
// or one block per row, at most two 64-column panels of xAq.  An int8 tile costs 0.58 (256 rows) / 0.65 (128 rows) of the bf16
// kernel's tiles over the same rows, so there is no token count from which the bf16 tile kernel would be the better choice.
bool i8_eligible(const GemmArgs& g, int bout) {
  if (!g.w8 || g.M < 128) return false;
  if (!(bout == 0 || (bout == 2 && g.bout_nblk == 1))) return false;
  if (g.rp > 128) return false;
  return true;
}

int i8_dispatch(const GemmArgs& g, int dtype, bool lowrank, int bout, hipStream_t st) {
  if (g.w_i8codes) {
    const bool w128 = i8_tile_rows(g) == 128;
    switch (dtype) {
      case LQER_F32: return w128 ? i8::launch_w8<LQER_F32, 4>(g, lowrank, bout, st) : i8::launch_w8<LQER_F32, 8>(g, lowrank, bout, st);
      case LQER_F16: return w128 ? i8::launch_w8<LQER_F16, 4>(g, lowrank, bout, st) : i8::launch_w8<LQER_F16, 8>(g, lowrank, bout, st);
      case LQER_BF16: return w128 ? i8::launch_w8<LQER_BF16, 4>(g, lowrank, bout, st) : i8::launch_w8<LQER_BF16, 8>(g, lowrank, bout, st);
    }
    set_error("unknown dtype %d", dtype);
    return LQER_E_INVALID;
  }
  const bool t128 = i8_tile_rows(g) == 128;
  switch (dtype) {
    case LQER_F32: return t128 ? i8::launch<LQER_F32, 4>(g, lowrank, bout, st) : i8::launch<LQER_F32, 8>(g, lowrank, bout, st);
    case LQER_F16: return t128 ? i8::launch<LQER_F16, 4>(g, lowrank, bout, st) : i8::launch<LQER_F16, 8>(g, lowrank, bout, st);
    case LQER_BF16: return t128 ? i8::launch<LQER_BF16, 4>(g, lowrank, bout, st) : i8::launch<LQER_BF16, 8>(g, lowrank, bout, st);
  }
  set_error("unknown dtype %d", dtype);
  return LQER_E_INVALID;
}

int i8_prepare_dispatch(const void* w_packed, int64_t N, int64_t K, int mbits, void* w_i8, int32_t* flags, hipStream_t st) {
  const int64_t Np = lqer_padded_n(N);
  const int nk = (int)(lqer_padded_k(K) / 64), nk8 = (int)(padded_k8(K) / I8_BK);
  (void)hipMemsetAsync(flags, 0, 2 * sizeof(int32_t), st);
  if (mbits > 3) {  // 8-bit codes from the three limb images
    i8::k_i8_rows8<<<(unsigned)(Np / 256), 256, 0, st>>>((const uint8_t*)w_packed, N, Np, nk, nk8, (uint8_t*)w_i8, flags);
    const int64_t items8 = Np * (2 * nk8) * 4;
    i8::k_i8_codes8<<<(unsigned)((items8 + 255) / 256), 256, 0, st>>>((const uint8_t*)w_packed, N, Np, nk, nk8, (uint8_t*)w_i8);
    return check_launch("lqer_i8_prepare (8-bit weights)");
  }
  // (the exponent bytes of the sign-magnitude image are already biased by the mantissa width)
  i8::k_i8_rows<<<(unsigned)((Np + 255) / 256), 256, 0, st>>>((const uint8_t*)w_packed, N, Np, nk, nk8, (uint8_t*)w_i8, flags);
  const int64_t items = Np * nk8;
  i8::k_i8_codes<<<(unsigned)((items + 255) / 256), 256, 0, st>>>((const uint8_t*)w_packed, N, Np, nk, nk8, (uint8_t*)w_i8);
  return check_launch("lqer_i8_prepare");
}

int i8_unpack_dispatch(const void* w_i8, int64_t N, int64_t K, float* out, hipStream_t st, bool codes8) {
  const int64_t Np = lqer_padded_n(N);
  const int nk8 = (int)(padded_k8(K) / I8_BK);
  if (codes8) {
    i8::k_i8_unpack8<<<(unsigned)((N * K + 255) / 256), 256, 0, st>>>((const uint8_t*)w_i8, N, K, Np, nk8, out);
    return check_launch("lqer_unpack_weight_i8 (8-bit weights)");
  }
  i8::k_i8_unpack<<<(unsigned)((N * K + 255) / 256), 256, 0, st>>>((const uint8_t*)w_i8, N, K, Np, nk8, out);
  return check_launch("lqer_unpack_weight_i8");
}

}  // namespace lqer
