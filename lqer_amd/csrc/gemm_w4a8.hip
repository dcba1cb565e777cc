// The fused W4 x A8 Linear kernel:
//
//   y[m,n] = sum_k xq[m,k] * Wq[n,k]  +  bq[n]  +  Q_Bout( sum_j xAq[m,j] * B[j,n] )
//
// replaces reference quantized_layers/linear.py:155-156 (torch.matmul(xA, B), B_out_quantizer,
// F.linear, add).  One workgroup = one 128(m) x 256(n) output tile (wide in n: a weight costs 0.56 B
// of L2->LDS traffic per element, an activation 2 B), 8 waves side by side along n, each
// wave 128 x 32 = 4 x 1 tiles of v_mfma_f32_32x32x16_bf16 (bf16 holds every MXINT value
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
//  * main loop, BK = 64: every global access is an LDS-DMA (buffer_load ... lds, 16 B/lane) into a
//    4-slot ring, three k-steps ahead, retired with a COUNTED s_waitcnt vmcnt so that loads stay in
//    flight across barriers.  The activation tile is staged as bf16 (XOR swizzle on the source
//    address); the weights stay PACKED in LDS (4-bit codes + block exponents, 0.56 B per weight) and
//    each wave expands the fragments it needs in registers, in the shadow of its MFMAs.
//  * the two waves of a SIMD run half a k-step apart (LOAD / COMPUTE ping-pong, see the main loop).
//  * tiles are numbered so that each of the 8 XCDs works on a contiguous run of tiles (same token
//    rows -> the activation slab stays in that XCD's L2).
#include <atomic>
#include <type_traits>

#include "common.h"

namespace lqer {

constexpr int BM = 128, BN = 256, BK = 64;
#ifndef LQER_DEPTH
#define LQER_DEPTH 3
#endif
constexpr int DEPTH = LQER_DEPTH;                     // k-steps of prefetch in flight
constexpr int NSLOT = DEPTH + 1;                      // LDS ring slots
constexpr int R_SLOT = (BN / 16) * LQER_PANEL_BYTES;  // 9216 B  packed weight panels (4-bit codes + exponents)
constexpr int OFF_A = 0;
// per tile height (MT 32-row tiles): activation slot 16 KiB (128 rows) or 8 KiB (64 rows), bf16; the panels behind the ring
constexpr int a_slot_bytes(int mt) { return 32 * mt * BK * 2; }
#ifdef LQER_ABL_E4M3SIM
// Timing-only experiment (profiles/r05_e4m3_sim.txt): what the main loop would cost with a second, prefill-only
// weight image of one e4m3 byte per weight - one more 1-KiB LDS-DMA per wave and step into a shadow region behind the ring (bytes of
// a neighbouring k-step), one more 16-byte fragment read, and an expand of four v_cvt_scalef32_pk_bf16_fp8 per 8 weights.  The
// results of this build are garbage; 128-row tiles only.
constexpr int SH_SLOT = 8192, SH_PAD = 4096;
constexpr int gemm_lds_bytes(int mt) { return NSLOT * a_slot_bytes(mt) + NSLOT * R_SLOT + NSLOT * SH_SLOT + SH_PAD; }
#else
constexpr int gemm_lds_bytes(int mt) { return NSLOT * a_slot_bytes(mt) + NSLOT * R_SLOT; }
#endif  // 102400 B / 69632 B (two workgroups per CU)

// byte offset of 16-byte chunk `c` (8 bf16 along k) of tile row `r`; rows are 128 B.
// chunk ^ ((row >> 1) & 7): the 16 lanes of a ds_read_b128 group then hit 16 distinct 16-B slots.
__device__ __forceinline__ int swz(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

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
template <int OFF>
__device__ __forceinline__ bf16x8 lds_read128(uint32_t addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
  return v;
}
__device__ __forceinline__ u32x4 lds_read128u(uint32_t addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
__device__ __forceinline__ uint32_t lds_read32_512(uint32_t addr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1 offset:512" : "=v"(v) : "v"(addr));
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
__device__ __forceinline__ void lds_wait(bf16x8& a, bf16x8& b) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void lds_wait(u32x2& a, int& b) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b));
}

#ifndef LQER_DEFER_WHERE
#define LQER_DEFER_WHERE 0  // DEFER pieces: 0 inside COMPUTE (between its MFMAs), 1 at the head of LOAD
#endif
#ifdef LQER_STAMPS
#define LQER_LOAD_BARRIER ""  // the diagnostic build stamps between the waits and the barrier
#else
#define LQER_LOAD_BARRIER "\n\ts_barrier"
#endif
#if defined(LQER_STAMPS) || defined(LQER_CLOCKPROBE)
__device__ unsigned long long* g_stamp_buf = nullptr;  // diagnostic builds only
#endif
#ifdef LQER_STAMPS
// Diagnostic build only: per-section cycle sums (s_memtime) of the main loop, written to a buffer that
// nothing else reads.  Never quote this build's run time (the stamps serialise the sections).
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

// BOUT: 0 pass-through, 1 blocks of 16 (max in registers), 2 any block (max from the pre-pass).  STAGED: the side product's
// operands go through LDS (below) - a separate instantiation, so that the direct route keeps its own register allocation.
// MT: 32-row tiles per wave = tile height / 32.  4: the 128 x 256 tile.  2: a 64 x 256 tile for token counts whose 128-row
// grid leaves most of the chip idle (half the MFMA work per expanded weight fragment, twice the workgroups).
// WTWOS: the packed weight holds two's-complement nibbles (w_quantizer = integer, codes -8 .. 7): the second expand of common.h
// (128-row tiles, staged side path and 16-bit / fp32 tensors only: a format no template configuration uses gets a working
// kernel, not a tuned one).
// XAPART: there is no xAq yet - the fused quantize kernel left the split-K partial tiles of x A (g.xa_part: part[c][m][rp] fp32) and
// the workgroup sums them in ascending chunk order and applies A_out (blocks of 16) to its 128 / 64 rows on the way into the
// LDS stage: k_xa_reduce4's arithmetic, item by item, and one dependent launch less per Linear.  Selectable only
// (LQER_TUNE_XA_REDUCE_IN_GEMM): at C2 the launch it saves took 4.9 us and this kernel grows by 5.3 us (1.0 of it the staged
// route's barriers) - the quantizer leaves one partial tile per 256 k (16 at K = 4096: 256 KB per workgroup, four round trips
// of four chunks), and the sum sits in front of the main loop, whose accumulators it opens, with nothing to hide behind.
// DEFER (round 6): the B_out re-quantization of the side product leaves the prologue.  There it was ~800 vector instructions per wave
// between the side product's MFMAs and the first k-step with nothing to hide behind: 3.5 of the 5.4 us in front of the main loop at
// 2048 x 4096 x 4096 (tools/clock_probe.py, "prologue split").  Here the side product stays in registers of its own (sp), the main
// loop opens on bias alone, its first 16 k-steps each carry one sixteenth of the re-quantization in the vector slots between their
// MFMAs (even piece: a block's maximum, exponent and scale factors; odd piece: its eight values), and sp is added behind the last
// k-step - as gemm_smallm.hip and decode1.hip always did.  Same formula (the scale exponent clamped to normal floats, exact while
// |x| <= 1e-8 passes through: gemm_w4a8_i8.hip's epilogue); blocks of 16, clamps up to 2^22 and K >= 1024 (launch_gemm).
template <int DT, bool LOWRANK, int BOUT, bool STAGED = false, int MT = 4, bool WTWOS = false, bool XAPART = false, bool DEFER = false>
__global__ __launch_bounds__(512) void k_lqer_gemm(GemmArgs g) {
  static_assert(MT == 4 || MT == 2, "128- or 64-row tiles");
  static_assert(!DEFER || (LOWRANK && BOUT == 1 && MT == 4 && !XAPART), "deferred B_out: blocks of 16 on 128-row tiles");
  static_assert(!XAPART || (LOWRANK && STAGED), "the partial tiles of x A enter through the LDS stage");
  static_assert(!WTWOS || DT != LQER_F16X, "integer weights: no fp16 main loop");
  constexpr int BMk = 32 * MT;   // tile rows
  constexpr int AP = MT / 2;     // 8-row LDS-DMA pieces of the activation tile per wave and k-step
  constexpr bool XF16 = DT == LQER_F16X;  // fp16 activation image, weights expanded to fp16, v_mfma_f32_32x32x16_f16
  constexpr int A_SLOT = a_slot_bytes(MT), OFF_R = NSLOT * A_SLOT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave;  // 8 waves side by side along n: each owns all 128 token rows x 32 output columns
  const int l31 = lane & 31, lh = lane >> 5;
#ifdef LQER_CLOCKPROBE
  unsigned long long cp_rin, cp_rq = 0;  // diagnostic build: the chip-wide 100 MHz counter at the wave's entry / in front of the B_out re-quantization
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(cp_rin)::"memory");
#endif

  // XCD-aware tile order: blocks b, b+8, ... share an XCD; give each XCD a contiguous tile range.
  const int nt = g.tiles_m * g.tiles_n;
  int tile;
  {
    const int b = blockIdx.x, xcd = b & 7, q8 = nt >> 3, r8 = nt & 7;
    tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
  }
  int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  if (g.xcd_bm > 0) {
    // XCD-local tile BLOCKS (the host checked divisibility): xcd_bm token tiles x nt / 8 / xcd_bm weight tiles per XCD, the XCDs
    // as a (tiles_m / xcd_bm) x (rest) grid - the bytes an XCD pulls through its L2 become W / gn + x / gm instead of W + x / 8
    const int b = blockIdx.x, xcd = b & 7, l = b >> 3;
    const int bn = (nt >> 3) / g.xcd_bm, gm = g.tiles_m / g.xcd_bm;
    tm = (xcd % gm) * g.xcd_bm + l % g.xcd_bm, tn = (xcd / gm) * bn + l / g.xcd_bm;
  }
  const int m0 = tm * BMk, n0 = tn * BN;
  const int nk = g.Kp / BK;

  // ---- staging ------------------------------------------------------------------------------
  // activation: wave w stages tile rows [16w, 16w+16): 2 x LDS-DMA of 8 rows x 128 B.  Buffer
  // addressing: wave-uniform descriptor + per-lane byte offset fixed for the whole kernel + the
  // k-step as scalar offset, so a prefetch costs no vector ALU work.
  const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(g.xq + (int64_t)m0 * g.Kp), 0, BMk * g.Kp * 2, 0x00020000);
  int a_voff[2] = {0, 0};  // (AP entries used; a dependent-size array here makes hipcc's HOST pass drop the kernel's stub silently)
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int row = wave * (8 * AP) + i * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    a_voff[i] = (row * g.Kp + chunk * 8) * 2;
  }
  // weights: the 16 panels of a k-step (16 x 576 B) are contiguous in the ring slot and are filled by nine 1-KiB
  // LDS-DMAs with per-lane source offsets (576 = 36 x 16: a lane's 16 bytes lie inside one panel): wave w issues
  // piece w, wave 0 also piece 8 - 25 LDS-DMA instructions per k-step and workgroup, all with full EXEC.
  const uint8_t* w_base = g.wp + ((int64_t)(n0 / 16) * nk) * LQER_PANEL_BYTES;
  const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)w_base, 0, 16 * nk * LQER_PANEL_BYTES, 0x00020000);
  auto w_piece_voff = [&](int piece) {
    const int byte = piece * 1024 + lane * 16;
    const int pnl = byte / LQER_PANEL_BYTES;
    return pnl * nk * LQER_PANEL_BYTES + (byte - pnl * LQER_PANEL_BYTES);
  };
  const int w_voff = w_piece_voff(wave), w_voff8 = w_piece_voff(8);
  unsigned char* const a_dst0 = smem + OFF_A + wave * (8 * AP) * 128;  // + slot * A_SLOT + piece * 1024
  unsigned char* const w_dst0 = smem + OFF_R + wave * 1024;      // + slot * R_SLOT
  auto issue_loads = [&](int kt, int slot) {  // 3 LDS-DMA instructions per wave (wave 0: 4)
#ifndef LQER_ABL_NO_A_LOAD
#pragma unroll
    for (int i = 0; i < AP; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_void*)(a_dst0 + slot * A_SLOT + i * 1024), 16, a_voff[i],
                                               kt * (BK * 2), 0, 0);
#endif
#ifndef LQER_ABL_NO_W_LOAD
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(w_dst0 + slot * R_SLOT), 16, w_voff, kt * LQER_PANEL_BYTES, 0, 0);
    if (wave == 0)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(smem + OFF_R + 8192 + slot * R_SLOT), 16, w_voff8,
                                               kt * LQER_PANEL_BYTES, 0, 0);
#endif
#ifdef LQER_ABL_E4M3SIM
    if constexpr (MT == 4)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(smem + OFF_R + NSLOT * R_SLOT + slot * SH_SLOT + wave * 1024), 16, w_voff,
                                               (kt ^ 1) * LQER_PANEL_BYTES, 0, 0);
#endif
  };
  // fragment read addresses (slot 0).  Activation: row = wave tile row + lane & 31, chunk 2 ks + (lane >> 5),
  // swizzled; the second m tile is +32 rows = +4096 B (row + 32 keeps (row >> 1) & 7).
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem;
  uint32_t fa_addr[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) fa_addr[ks] = lds0 + OFF_A + swz(l31, 2 * ks + lh);  // m tile i: + i * 4096
  // Packed weights: weight row n = wave tile column + lane & 31 lives in panel n / 16 at row n % 16; its
  // 32 B of codes hold the words of chunks {0,2,4,6} then {1,3,5,7}, so the lane's 4 words (chunk
  // 2 ks + (lane >> 5), ks = 0..3) are one 16-byte read; its 4 biased block exponents are one 4-byte read.
  const int nw = wn * 32 + l31;
  const uint32_t fw_addr = lds0 + OFF_R + (nw >> 4) * LQER_PANEL_BYTES + (nw & 15) * 32 + lh * 16;
  const uint32_t fe_addr = lds0 + OFF_R + (nw >> 4) * LQER_PANEL_BYTES + (nw & 15) * 4;  // + 512

  // ---- side path, first half: fetch before the ring prefetch -----------------------------------------------------
  // acc = Q_Bout(xAq @ B) + bias opens the accumulators.  The tile's rows of xAq are staged through LDS - 64 rank
  // entries per pass, in the ring slot that is not yet a prefetch target (slot DEPTH), in the activation tile's own
  // swizzled layout - and this wave's B^T fragments of a pass are fetched in one batch: one memory latency per pass
  // instead of one per 16-deep slice, no 8-fold refetch of xAq by the 8 waves.  The loads of pass 0 are issued BEFORE
  // the ring prefetch (loads return in order: their results can then be awaited while the prefetch is in flight), and
  // the staging writes / fragment reads are asm statements, invisible to the waitcnt pass (see above).
  constexpr int STG = STAGED ? BMk * 8 / 512 : 1;  // staged 16-byte chunks per thread and pass
  u32x4 stg[STG];
  bf16x8 sb[STAGED ? 2 : 1][STAGED ? 4 : 1];  // B^T fragments of one limb of the pass, double-buffered
  const uint32_t stage = lds0 + OFF_A + (NSLOT - 1) * A_SLOT;
  const bf16_t* const bt_row = STAGED ? g.bt + (int64_t)(n0 + wn * 32 + l31) * g.rp + 8 * lh : nullptr;
  float4 pv[XAPART ? 8 : 1];                                    // XAPART: the thread's first four chunk reads (8 floats each)
  const int xa_rows = XAPART ? (int)(g.xa_cstride / g.rp) : 0;  // rows the partial tiles hold (a multiple of 32)
  auto side_fetch_b = [&](int p0, int l, auto buf_c) {  // limb l of this wave's B^T fragments -> sb[buf]
    constexpr int BUF = decltype(buf_c)::value;
    if constexpr (STAGED) {
      const int cols = g.rp - p0 < 64 ? g.rp - p0 : 64;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        if (l < g.b_limbs && ks * 16 < cols) sb[BUF][ks] = *(const bf16x8*)(bt_row + (int64_t)l * g.Np * g.rp + p0 + ks * 16);
    }
  };
  auto side_fetch = [&](int p0) {
    if constexpr (!STAGED) return;
    const int cols = g.rp - p0 < 64 ? g.rp - p0 : 64;  // a multiple of 16
    const int cpr = cols >> 3;                          // 16-byte chunks per row
    if constexpr (XAPART) {
      // one item = 8 consecutive rank entries of one token; a thread has one item (rank <= 32) or two.  The first four chunk
      // reads of the thread (4 chunks of its item, or 2 of each) are requested here, ahead of the ring prefetch; side_reduce
      // (behind it) sums, requests the rest and quantizes
      const int nit = BMk * cpr > 512 ? 2 : 1;  // (workgroup-uniform)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = nit == 2 ? q >> 1 : 0, u = nit == 2 ? q & 1 : q;
        const int c = tid + 512 * j;
        const int row = c / cpr, ch = c - row * cpr;
        // (branch-free: rows and chunks that do not exist read the last one that does and are dropped in side_reduce - a
        // predicated load would be waited for on the spot, one round trip per chunk)
        const int rowc = m0 + row < xa_rows ? m0 + row : xa_rows - 1, uc = u < g.xa_nchunk ? u : g.xa_nchunk - 1;
        const float* src = g.xa_part + (int64_t)rowc * g.rp + p0 + 8 * ch + uc * g.xa_cstride;
        pv[2 * q] = *(const float4*)src;
        pv[2 * q + 1] = *(const float4*)(src + 4);
      }
    } else {
#pragma unroll
    for (int j = 0; j < STG; ++j) {
      const int c = tid + 512 * j;
      if (c < BMk * cpr) {
        const int row = c / cpr, ch = c - row * cpr;
        stg[j] = *(const u32x4*)(g.xaq + (int64_t)(m0 + row) * g.xaq_ld + p0 + 8 * ch);
      }
    }
    }
    side_fetch_b(p0, 0, std::integral_constant<int, 0>{});
  };
  // XAPART: chunks summed in ascending order (k_xa_reduce4's order and arithmetic), A_out over the block of 16 = the items of
  // lanes l and l ^ 1 (cpr is even), bf16 bits into the stage registers
  auto side_reduce = [&](int p0) {
    if constexpr (XAPART) {
      const int cols = g.rp - p0 < 64 ? g.rp - p0 : 64;
      const int cpr = cols >> 3;
      auto items = [&](auto nit_c) {
        constexpr int NIT = decltype(nit_c)::value, FB = 4 / NIT;  // items per thread, chunks per item already requested
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
          const int c = tid + 512 * j;
          const int row = c / cpr, ch = c - row * cpr;
          const bool live = c < BMk * cpr && m0 + row < xa_rows;
          float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          auto add = [&](const float4& a, const float4& b, bool on) {  // (x + 0 = x: a dropped chunk leaves the sum's bits alone)
            s[0] += on ? a.x : 0.f, s[1] += on ? a.y : 0.f, s[2] += on ? a.z : 0.f, s[3] += on ? a.w : 0.f;
            s[4] += on ? b.x : 0.f, s[5] += on ? b.y : 0.f, s[6] += on ? b.z : 0.f, s[7] += on ? b.w : 0.f;
          };
#pragma unroll
          for (int u = 0; u < FB; ++u) add(pv[2 * (FB * j + u)], pv[2 * (FB * j + u) + 1], live && u < g.xa_nchunk);
          const int rowc = m0 + row < xa_rows ? m0 + row : xa_rows - 1;
          const float* src = g.xa_part + (int64_t)rowc * g.rp + p0 + 8 * ch;
          for (int cc = FB; cc < g.xa_nchunk; cc += 4) {  // one chunk per 256 k: four more chunks per round trip
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int uc = cc + u < g.xa_nchunk ? cc + u : g.xa_nchunk - 1;
              v[2 * u] = *(const float4*)(src + uc * g.xa_cstride), v[2 * u + 1] = *(const float4*)(src + uc * g.xa_cstride + 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) add(v[2 * u], v[2 * u + 1], live && cc + u < g.xa_nchunk);
          }
          float amax = 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) amax = fmaxf(amax, fabsf(s[k]));
          amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
          const bool any = amax > 0.f;
          const int e = any ? block_exponent(amax, g.aout) : 0;
          uint32_t w[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float v0 = any ? ldexpf(mxint_mantissa(s[2 * k], e, g.aout), e - g.aout.mbits) : 0.f;
            const float v1 = any ? ldexpf(mxint_mantissa(s[2 * k + 1], e, g.aout), e - g.aout.mbits) : 0.f;
            w[k] = exact_bf16_bits(v0) | (exact_bf16_bits(v1) << 16);
          }
          stg[j] = (u32x4){w[0], w[1], w[2], w[3]};
        }
      };
      if (BMk * cpr > 512) {
        if constexpr (STG == 2) items(std::integral_constant<int, 2>{});
      } else {
        items(std::integral_constant<int, 1>{});
      }
    }
  };
  // (a side product of at most two 16-deep slices - rank <= 32 with one limb - is cheaper fetched directly: the two
  // barriers of the staged route cost more than they save there; launch_gemm picks the instantiation)
  constexpr bool side_staged = LOWRANK && STAGED;
  if constexpr (side_staged) side_fetch(0);
  // direct route (at most two 16-deep slices: rank <= 32 with one limb, or rank 16 with two): both slices' operands
  // are requested here, ahead of the ring prefetch, instead of one slice at a time behind it
  constexpr bool side_direct = LOWRANK && !STAGED;
  bf16x8 db[side_direct ? 2 : 1], dx[side_direct ? 2 : 1][side_direct ? MT : 1];
  const bool two_limbs = g.b_limbs > 1;  // (then rank 16: slice s = limb s; else slice s = rank entries 16 s ..)
  if constexpr (side_direct) {
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
      const int l = two_limbs ? sl : 0, ks = two_limbs ? 0 : sl;
      if (l < g.b_limbs && ks * 16 < g.rp) {
        db[sl] = *(const bf16x8*)(g.bt + ((int64_t)l * g.Np + n0 + wn * 32 + l31) * g.rp + ks * 16 + 8 * lh);
#pragma unroll
        for (int i = 0; i < MT; ++i) dx[sl][i] = *(const bf16x8*)(g.xaq + (int64_t)(m0 + i * 32 + l31) * g.xaq_ld + ks * 16 + 8 * lh);
      }
    }
  }

  // prologue loads: steps 0 .. DEPTH-1
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue_loads(d, d);  // (past the end of K: dropped by the buffer range check)

  f32x16 acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
  f32x16 sp[DEFER ? MT : 1];  // DEFER: the side product, re-quantized in place during the first k-steps
  if constexpr (DEFER) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int k = 0; k < 16; ++k) sp[i][k] = 0.f;
  }
  auto side_acc = [&](int i) -> f32x16& {
    if constexpr (DEFER) return sp[i];
    else return acc[i];
  };

  // ---- low-rank prologue: acc = Q_Bout(xAq @ B) + bias ----------------------------------------
  if constexpr (LOWRANK) {
    if constexpr (side_direct) {
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const int l = two_limbs ? sl : 0, ks = two_limbs ? 0 : sl;
        if (l < g.b_limbs && ks * 16 < g.rp) {
#pragma unroll
          for (int i = 0; i < MT; ++i) side_acc(i) = __builtin_amdgcn_mfma_f32_32x32x16_bf16(db[sl], dx[sl][i], side_acc(i), 0, 0, 0);
        }
      }
    }
    if constexpr (side_staged)
    for (int p0 = 0; p0 < g.rp; p0 += 64) {
      const int cols = g.rp - p0 < 64 ? g.rp - p0 : 64;
      const int cpr = cols >> 3;
      if (p0) {
        asm volatile("s_barrier" ::: "memory");  // the previous pass's fragment reads are done (lgkmcnt(0) below)
        side_fetch(p0);
      }
      side_reduce(p0);
#pragma unroll
      for (int j = 0; j < STG; ++j) {
        const int c = tid + 512 * j;
        if (c < BMk * cpr) {
          const int row = c / cpr, ch = c - row * cpr;
          lds_write128(stage + swz(row, ch), stg[j]);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      auto limb = [&](int l, auto buf_c) {
        constexpr int BUF = decltype(buf_c)::value;
        if (l >= g.b_limbs) return;
        side_fetch_b(p0, l + 1, std::integral_constant<int, BUF ^ 1>{});  // the next limb's fragments, under this limb's MFMAs
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          if (ks * 16 < cols) {
            const uint32_t fa = stage + swz(l31, 2 * ks + lh);  // m tile i: + i * 4096 (row + 32 keeps the swizzle)
            if constexpr (MT == 4) {
              bf16x8 x0 = lds_read128<0>(fa), x1 = lds_read128<4096>(fa), x2 = lds_read128<8192>(fa), x3 = lds_read128<12288>(fa);
              lds_wait(x0, x1, x2, x3);
              side_acc(0) = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sb[BUF][ks], x0, side_acc(0), 0, 0, 0);
              side_acc(1) = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sb[BUF][ks], x1, side_acc(1), 0, 0, 0);
              side_acc(2) = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sb[BUF][ks], x2, side_acc(2), 0, 0, 0);
              side_acc(3) = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sb[BUF][ks], x3, side_acc(3), 0, 0, 0);
            } else {
              bf16x8 x0 = lds_read128<0>(fa), x1 = lds_read128<4096>(fa);
              lds_wait(x0, x1);
              acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sb[BUF][ks], x0, acc[0], 0, 0, 0);
              acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sb[BUF][ks], x1, acc[1], 0, 0, 0);
            }
          }
      };
      limb(0, std::integral_constant<int, 0>{});
      limb(1, std::integral_constant<int, 1>{});
      limb(2, std::integral_constant<int, 0>{});
    }
#ifdef LQER_CLOCKPROBE
    asm volatile("s_waitcnt vmcnt(6)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(cp_rq)::"memory");  // side operands landed, MFMAs issued
#endif
    if constexpr (BOUT != 0 && !DEFER) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          float amax;
          if constexpr (BOUT == 1) {
            amax = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) amax = fmaxf(amax, fabsf(acc[i][8 * b + k]));
            amax = pair32_max(amax);
          } else {
            // block of L columns (L a multiple of 16): its max was reduced by k_bout_amax.  (No buffer: the integer
            // quantizer - fixed point, its "exponent" is pinned to 0 by the QP's clamp whatever amax says)
            amax = g.bout_amax ? g.bout_amax[(int64_t)(m0 + i * 32 + l31) * g.bout_nblk + (n0 + wn * 32 + 16 * b) / g.bout_L] : 1.0f;
          }
          const int e = block_exponent(amax, g.bout);  // amax = 0: every element takes the pass-through
          float blk[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) blk[k] = acc[i][8 * b + k];
          mxint_requant_fast(blk, e, g.bout);  // two elements per packed fp32 instruction (common.h)
#pragma unroll
          for (int k = 0; k < 8; ++k) acc[i][8 * b + k] = blk[k];
        }
    }
  }
  if (g.bias) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float bv = g.bias[n0 + wn * 32 + (k & 3) + 8 * (k >> 2) + 4 * lh];
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[i][k] += bv;
    }
  }

  // ---- main loop ------------------------------------------------------------------------------
  // Ping-pong: the two waves that share a SIMD (w and w+4) run one barrier apart.  A k-step is
  //   LOAD(kt):    read this wave's operands of step kt from ring slot kt % 4 into registers (8 x 16 B of
  //                activation fragments, 2 x 16 B of weight codes, 2 x 4 B of block exponents), issue the
  //                loads of step kt+3 into slot (kt+3) % 4, wait, barrier;
  //   COMPUTE(kt): per 16-deep k slice expand two weight fragments (VALU) and issue 4 MFMAs; barrier.
  // While waves 0-3 compute, waves 4-7 load, and vice versa:
  //   barrier index      2kt-1        2kt          2kt+1         2kt+2
  //   waves 0-3:    ... | LOAD(kt)  | COMPUTE(kt) | LOAD(kt+1)  | ...
  //   waves 4-7:    ... | COMP(kt-1)| LOAD(kt)    | COMPUTE(kt) | ...
  // RAW: every wave ends LOAD(kt-1) with a counted vmcnt that retires its own loads of step kt (the
  // DEPTH-1 younger batches of 3 stay in flight) and then passes a barrier (2kt-2 or 2kt-1) before
  // anyone starts LOAD(kt).  WAR: slot (kt+3) % 4 held step kt-1, last read in LOAD(kt-1) of waves 4-7,
  // which ends (lgkmcnt(0)) before barrier 2kt-1; the overwriting loads are issued after it.
  const bool late = wave >= 4;
  // loads(0) landed; two batches of AP + 1 (3, or 2 with 64-row tiles) may stay in flight
#ifdef LQER_ABL_E4M3SIM
  if constexpr (MT == 4) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
#else
  if constexpr (MT == 4) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
#endif
  else asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
  if (late) asm volatile("s_barrier" ::: "memory");
#ifdef LQER_CLOCKPROBE
  // diagnostic build: shader cycles (s_memtime) and 100 MHz ticks (s_memrealtime) around the whole main loop
  unsigned long long cp_c0, cp_r0;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(cp_c0), "=s"(cp_r0)::"memory");
#endif
#ifdef LQER_STAMPS
  unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif
  // hand-built buffer descriptors for the in-asm LDS-DMA (wave-uniform words).  Exact ranges: every step issues its
  // prefetch, also past the end of K - those lanes read inside the tile's rows or are dropped by the range check
  const unsigned long long a_base64 = (unsigned long long)(g.xq + (int64_t)m0 * g.Kp);
  const unsigned long long w_base64 = (unsigned long long)w_base;
  const u32x4 a_rs = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a_base64),
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a_base64 >> 32)) & 0xffffu,
                      (uint32_t)(BMk * g.Kp * 2), 0x00020000u};
  const u32x4 w_rs = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)w_base64),
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(w_base64 >> 32)) & 0xffffu,
                      (uint32_t)(16 * nk * LQER_PANEL_BYTES), 0x00020000u};
  const uint32_t m0_a = lds0 + OFF_A + wave * (8 * AP) * 128;  // + slot * A_SLOT (+ 1024: second piece)
  const uint32_t m0_w = lds0 + OFF_R + wave * 1024;      // + slot * R_SLOT
  const uint32_t m0_w8 = lds0 + OFF_R + 8192;            // + slot * R_SLOT (piece 8, wave 0)
  // One k-step.  The ring slot is a compile-time constant (the loop below is unrolled by the ring size), so every
  // LDS address is a per-lane base register + an immediate offset and the LDS-DMA destinations are constants.
  // LOAD(kt) is ONE asm statement (a compiler-visible gap between LDS reads and their wait lets hipcc copy registers
  // that have not landed): the 18 LDS reads first, then the LDS-DMA prefetch of step kt+3 - the reads' latency passes
  // while the DMA instructions issue -, then the counted waits.  The step's first weight fragment is expanded in the
  // slack left before the barrier, so that COMPUTE opens with an MFMA.  The loading wave has issue priority over
  // its computing SIMD partner (whose MFMAs only need an issue slot every 32 cycles).
  // DEFER piece P of k-step P (0..15): block j = P / 2 = (m tile j / 2, column half j % 2) of the side product.  Straight-line vector
  // code only (no branch: it has to share the COMPUTE section's scheduling region with the MFMAs it hides behind).
  float dq_s = 1.f, dq_inv = 1.f;  // scale factors 2^(mbits - e), 2^(e - mbits) of the block between its two pieces
  auto defer_piece = [&](auto piece_c) {
    constexpr int P = decltype(piece_c)::value;
    if constexpr (DEFER && P >= 0) {
      typedef __attribute__((ext_vector_type(2))) float f2;
      constexpr int J = P >> 1, I = J >> 1, B8 = 8 * (J & 1);
      if constexpr ((P & 1) == 0) {
        float amax = 0.f;
#pragma unroll
        for (int k = 0; k < 8; k += 2) amax = fmaxf(fmaxf(amax, fabsf(sp[I][B8 + k])), fabsf(sp[I][B8 + k + 1]));
        amax = pair32_max(amax);
        // (mbits - e clamped to the exponents of normal floats: beyond, every element of the block is below 1e-8 and passes through)
        int up = g.bout.mbits - block_exponent(amax, g.bout);
        up = up > 126 ? 126 : (up < -126 ? -126 : up);
        dq_s = __uint_as_float((uint32_t)(127 + up) << 23);
        dq_inv = __uint_as_float((uint32_t)(127 - up) << 23);
      } else {
        const float es = g.bout.eps * dq_s, hi = g.bout.mmax, lo = -g.bout.mneg, tiny = g.bout.tiny;
        const f2 magic = {12582912.0f, 12582912.0f};
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
          const f2 x = {sp[I][B8 + k], sp[I][B8 + k + 1]};
          const f2 c = {copysignf(es, x[0]), copysignf(es, x[1])};
          f2 r = (__builtin_elementwise_fma(x, (f2){dq_s, dq_s}, c) + magic) - magic;
          r[0] = __builtin_amdgcn_fmed3f(r[0], lo, hi);
          r[1] = __builtin_amdgcn_fmed3f(r[1], lo, hi);
          const f2 val = r * (f2){dq_inv, dq_inv};
          sp[I][B8 + k] = fabsf(x[0]) <= tiny ? x[0] : val[0];
          sp[I][B8 + k + 1] = fabsf(x[1]) <= tiny ? x[1] : val[1];
        }
      }
    }
  };
  // (the piece's results are needed a k-step later, or behind the loop: without a use inside the section hipcc sinks the whole
  // computation to that use - all sixteen pieces ended up behind the main loop.  An empty asm that "modifies" them pins them here)
  auto defer_pin = [&](auto piece_c) {
    constexpr int P = decltype(piece_c)::value;
    (void)dq_s, (void)dq_inv, (void)sp;  // (captured here: operands of an asm statement inside a dependent branch do not make a capture)
    if constexpr (DEFER && P >= 0) {
      constexpr int J = P >> 1, I = J >> 1, B8 = 8 * (J & 1);
      if constexpr ((P & 1) == 0) asm volatile("" : "+v"(dq_s), "+v"(dq_inv));
      else
        asm volatile("" : "+v"(sp[I][B8]), "+v"(sp[I][B8 + 1]), "+v"(sp[I][B8 + 2]), "+v"(sp[I][B8 + 3]), "+v"(sp[I][B8 + 4]), "+v"(sp[I][B8 + 5]),
                     "+v"(sp[I][B8 + 6]), "+v"(sp[I][B8 + 7]));
    }
  };
  auto step = [&](int kt, auto slot_c, auto piece_c) {
    constexpr int SLOT = decltype(slot_c)::value;
    constexpr int slot_new = SLOT == 0 ? NSLOT - 1 : SLOT - 1;  // (slot + DEPTH) % NSLOT
    STAMP(7);
#if LQER_DEFER_WHERE == 1  // (experiment: the piece at the head of LOAD, under the partner wave's MFMAs)
    defer_piece(piece_c);
    defer_pin(piece_c);
    __builtin_amdgcn_sched_barrier(0);
#endif
    __builtin_amdgcn_s_setprio(1);
    const int ktn = __builtin_amdgcn_readfirstlane(kt + DEPTH);
    const int a_soff = ktn * (BK * 2), w_soff = ktn * LQER_PANEL_BYTES;
    const uint32_t m0a0 = m0_a + slot_new * A_SLOT, m0a1 = m0a0 + 1024;
    const uint32_t m0w = m0_w + slot_new * R_SLOT, m0w8 = m0_w8 + slot_new * R_SLOT;
    bf16x8 xa[4][MT];  // [ks][m tile]
    u32x4 wr;
    uint32_t we;
#ifdef LQER_ABL_E4M3SIM
    u32x4 wr2 = {0u, 0u, 0u, 0u};
    if constexpr (MT == 4) {  // the second half of the e4m3 fragment + its LDS-DMA (waited for by the statement below)
      const uint32_t m0sh = lds0 + OFF_R + NSLOT * R_SLOT + slot_new * SH_SLOT + wave * 1024;
      const int sh_soff = (ktn ^ 1) * LQER_PANEL_BYTES;
      asm volatile("ds_read_b128 %0, %1 offset:%c2\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %4, %6 offen lds"
                   : "=&v"(wr2)
                   : "v"(fw_addr), "i"(NSLOT * R_SLOT + SLOT * SH_SLOT), "v"(w_voff), "s"(w_rs), "s"(m0sh), "s"(sh_soff)
                   : "memory");
    }
#define LQER_I_VM "8"
#else
#define LQER_I_VM "6"
#endif
    if constexpr (MT == 4) {
      asm volatile(
          "ds_read_b128 %0, %18 offset:%c25\n\tds_read_b32 %1, %19 offset:%c25+512\n\t"
          "ds_read_b128 %2, %20 offset:%c24\n\tds_read_b128 %3, %20 offset:%c24+4096\n\t"
          "ds_read_b128 %4, %20 offset:%c24+8192\n\tds_read_b128 %5, %20 offset:%c24+12288\n\t"
          "ds_read_b128 %6, %21 offset:%c24\n\tds_read_b128 %7, %21 offset:%c24+4096\n\t"
          "ds_read_b128 %8, %21 offset:%c24+8192\n\tds_read_b128 %9, %21 offset:%c24+12288\n\t"
          "ds_read_b128 %10, %22 offset:%c24\n\tds_read_b128 %11, %22 offset:%c24+4096\n\t"
          "ds_read_b128 %12, %22 offset:%c24+8192\n\tds_read_b128 %13, %22 offset:%c24+12288\n\t"
          "ds_read_b128 %14, %23 offset:%c24\n\tds_read_b128 %15, %23 offset:%c24+4096\n\t"
          "ds_read_b128 %16, %23 offset:%c24+8192\n\tds_read_b128 %17, %23 offset:%c24+12288\n\t"
          "s_mov_b32 m0, %32\n\ts_nop 0\n\tbuffer_load_dwordx4 %26, %30, %36 offen lds\n\t"
          "s_mov_b32 m0, %33\n\ts_nop 0\n\tbuffer_load_dwordx4 %27, %30, %36 offen lds\n\t"
          "s_mov_b32 m0, %34\n\ts_nop 0\n\tbuffer_load_dwordx4 %28, %31, %37 offen lds\n\t"
          "s_cmp_lg_u32 %38, 0\n\ts_cbranch_scc1 1f\n\t"
          "s_mov_b32 m0, %35\n\ts_nop 0\n\tbuffer_load_dwordx4 %29, %31, %37 offen lds\n\t"
          // own loads of step kt+1 landed: the batches of kt+2 and kt+3 (3 loads each, wave 0: 4 - it waits a little
          // more than it must) may stay in flight
          "1:\n\ts_waitcnt vmcnt(" LQER_I_VM ") lgkmcnt(0)"
          : "=&v"(wr), "=&v"(we), "=&v"(xa[0][0]), "=&v"(xa[0][1]), "=&v"(xa[0][2]), "=&v"(xa[0][3]), "=&v"(xa[1][0]),
            "=&v"(xa[1][1]), "=&v"(xa[1][2]), "=&v"(xa[1][3]), "=&v"(xa[2][0]), "=&v"(xa[2][1]), "=&v"(xa[2][2]),
            "=&v"(xa[2][3]), "=&v"(xa[3][0]), "=&v"(xa[3][1]), "=&v"(xa[3][2]), "=&v"(xa[3][3])
          : "v"(fw_addr), "v"(fe_addr), "v"(fa_addr[0]), "v"(fa_addr[1]), "v"(fa_addr[2]), "v"(fa_addr[3]),  // 18..23
            "i"(SLOT * A_SLOT), "i"(SLOT * R_SLOT),                                                         // 24, 25
            "v"(a_voff[0]), "v"(a_voff[1]), "v"(w_voff), "v"(w_voff8),                                       // 26..29
            "s"(a_rs), "s"(w_rs), "s"(m0a0), "s"(m0a1), "s"(m0w), "s"(m0w8), "s"(a_soff), "s"(w_soff), "s"(wave)  // 30..38
          : "memory", "scc");  // (s_cmp inside)
    } else {  // 64-row tile: two m tiles per k slice, one activation piece per wave
      asm volatile(
          "ds_read_b128 %0, %10 offset:%c17\n\tds_read_b32 %1, %11 offset:%c17+512\n\t"
          "ds_read_b128 %2, %12 offset:%c16\n\tds_read_b128 %3, %12 offset:%c16+4096\n\t"
          "ds_read_b128 %4, %13 offset:%c16\n\tds_read_b128 %5, %13 offset:%c16+4096\n\t"
          "ds_read_b128 %6, %14 offset:%c16\n\tds_read_b128 %7, %14 offset:%c16+4096\n\t"
          "ds_read_b128 %8, %15 offset:%c16\n\tds_read_b128 %9, %15 offset:%c16+4096\n\t"
          "s_mov_b32 m0, %23\n\ts_nop 0\n\tbuffer_load_dwordx4 %18, %21, %26 offen lds\n\t"
          "s_mov_b32 m0, %24\n\ts_nop 0\n\tbuffer_load_dwordx4 %19, %22, %27 offen lds\n\t"
          "s_cmp_lg_u32 %28, 0\n\ts_cbranch_scc1 1f\n\t"
          "s_mov_b32 m0, %25\n\ts_nop 0\n\tbuffer_load_dwordx4 %20, %22, %27 offen lds\n\t"
          "1:\n\ts_waitcnt vmcnt(4) lgkmcnt(0)"
          : "=&v"(wr), "=&v"(we), "=&v"(xa[0][0]), "=&v"(xa[0][1]), "=&v"(xa[1][0]), "=&v"(xa[1][1]), "=&v"(xa[2][0]),
            "=&v"(xa[2][1]), "=&v"(xa[3][0]), "=&v"(xa[3][1])
          : "v"(fw_addr), "v"(fe_addr), "v"(fa_addr[0]), "v"(fa_addr[1]), "v"(fa_addr[2]), "v"(fa_addr[3]),  // 10..15
            "i"(SLOT * A_SLOT), "i"(SLOT * R_SLOT),                                                         // 16, 17
            "v"(a_voff[0]), "v"(w_voff), "v"(w_voff8),                                                       // 18..20
            "s"(a_rs), "s"(w_rs), "s"(m0a0), "s"(m0w), "s"(m0w8), "s"(a_soff), "s"(w_soff), "s"(wave)         // 21..28
          : "memory", "scc");
    }
    STAMP(1);  // LDS reads + DMA issue + waits
#ifdef LQER_ABL_E4M3SIM
    int sim_ks = 0;
    auto expand = [&](uint32_t word, uint32_t scale_bits) {  // 8 weights = 2 words of e4m3 bytes: four conversions, nothing else
      const float scale = __uint_as_float(scale_bits);
      const uint32_t w2 = wr2[sim_ks++ & 3];
      u32x4 r;
      r[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(word, scale, false));
      r[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(word, scale, true));
      r[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w2, scale, false));
      r[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w2, scale, true));
      return __builtin_bit_cast(bf16x8, r);
    };
#else
    auto expand = [](uint32_t word, uint32_t scale_bits) {
      if constexpr (WTWOS) return expand_frag_twos(word, scale_bits);
      else return expand_frag_t<XF16>(word, scale_bits);
    };
#endif
    bf16x8 wb_first = expand(wr[0], (we & 0xffu) << 23);
    asm volatile("s_barrier" : "+v"(wb_first)::"memory");
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    STAMP(4);  // expand of the first fragment + barrier
    // ---- COMPUTE(kt)
#if LQER_DEFER_WHERE == 0
    defer_piece(piece_c);
#endif
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      // biased exponent byte ks -> fp32 bits of the block scale 2^(e - mbits)
      const uint32_t sc = ((we >> (8 * ks)) & 0xffu) << 23;
      const bf16x8 wb = ks == 0 ? wb_first : expand(wr[ks], sc);
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[i] = mfma_32x32x16<XF16>(wb, xa[ks][i], acc[i]);
    }
#if LQER_DEFER_WHERE == 0
    defer_pin(piece_c);
#endif
    __builtin_amdgcn_sched_barrier(0);
    STAMP(5);  // COMPUTE section issue
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    STAMP(6);  // barrier after COMPUTE
  };
  using std::integral_constant;
  constexpr integral_constant<int, -1> no_piece{};
  int kt0 = 0;
  if constexpr (DEFER) {  // the first 16 k-steps (launch_gemm: nk >= 16), each with its piece of the re-quantization
#define LQER_DSTEP(KT) step(KT, integral_constant<int, (KT) % NSLOT>{}, integral_constant<int, KT>{})
    LQER_DSTEP(0); LQER_DSTEP(1); LQER_DSTEP(2); LQER_DSTEP(3); LQER_DSTEP(4); LQER_DSTEP(5); LQER_DSTEP(6); LQER_DSTEP(7);
    LQER_DSTEP(8); LQER_DSTEP(9); LQER_DSTEP(10); LQER_DSTEP(11); LQER_DSTEP(12); LQER_DSTEP(13); LQER_DSTEP(14); LQER_DSTEP(15);
#undef LQER_DSTEP
    kt0 = 16;
  }
  if (kt0 < nk)
  for (int kt = kt0;; kt += NSLOT) {  // 4 steps per trip, slots 0..3; every step is the same branch-free stream
    step(kt, integral_constant<int, 0>{}, no_piece);
    if (kt + 1 >= nk) break;
    step(kt + 1, integral_constant<int, 1>{}, no_piece);
    if (kt + 2 >= nk) break;
    step(kt + 2, integral_constant<int, 2>{}, no_piece);
    if (kt + 3 >= nk) break;
    step(kt + 3, integral_constant<int, 3>{}, no_piece);
    if (kt + 4 >= nk) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the prefetches issued past the end of K have drained
  if (!late) asm volatile("s_barrier" ::: "memory");
#ifdef LQER_CLOCKPROBE
  {
    unsigned long long cp_c1, cp_r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(cp_c1), "=s"(cp_r1)::"memory");
    if (g_stamp_buf && lane == 0) {
      g_stamp_buf[(blockIdx.x * 8 + wave) * 8 + 0] = cp_c1 - cp_c0;
      g_stamp_buf[(blockIdx.x * 8 + wave) * 8 + 1] = cp_r1 - cp_r0;
      g_stamp_buf[(blockIdx.x * 8 + wave) * 8 + 2] = cp_rin;  // absolute ticks: entry, main loop start, main loop end
      g_stamp_buf[(blockIdx.x * 8 + wave) * 8 + 3] = cp_r0;
      g_stamp_buf[(blockIdx.x * 8 + wave) * 8 + 4] = cp_r1;
      g_stamp_buf[(blockIdx.x * 8 + wave) * 8 + 7] = cp_rq;
    }
  }
#endif
#ifdef LQER_STAMPS
  if (g_stamp_buf && lane == 0)
    for (int i = 0; i < 8; ++i) g_stamp_buf[(blockIdx.x * 8 + wave) * 8 + i] = st_sum[i];
#endif

  if constexpr (DEFER) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[i][k] += sp[i][k];
  }
  // ---- store ----------------------------------------------------------------------------------
  // per 32x32 tile and quad q: regs 4q..4q+3 = columns n = nb + 8q + 4 lh + (0..3) of token row m
#ifdef LQER_ABL_NO_STORE
  if (g.M > 0) {
    for (int i = 0; i < MT; ++i) asm volatile("" ::"v"(acc[i]));
    return;
  }
#endif
  const bool aligned16 = (((uintptr_t)g.y) & 15) == 0;
  const int nb = n0 + wn * 32;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
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
      // 16-bit outputs: pack 4 columns into 8 B, then merge quads (q, q+1) of lanes l / l^32 into one
      // 16-B store: lanes 0-31 get columns 16p .. 16p+7, lanes 32-63 columns 16p+8 .. 16p+15
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
#ifdef LQER_CLOCKPROBE
  {
    unsigned long long cp_r2, cp_r3;  // stores issued; stores acknowledged
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(cp_r2)::"memory");
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(cp_r3)::"memory");
    if (g_stamp_buf && lane == 0) {
      g_stamp_buf[(blockIdx.x * 8 + wave) * 8 + 5] = cp_r2;
      g_stamp_buf[(blockIdx.x * 8 + wave) * 8 + 6] = cp_r3;
    }
  }
#endif
}

// Pre-pass for B_out blocks other than 16 columns: max |xAq @ B| over every (token row, block of L columns),
// L a multiple of 16.  One wave = one 32 x 32 tile of the product (same MFMA orientation as the GEMM
// prologue); the per-16-column maxima are folded into amax[m][n / L] with atomicMax on the fp32 bit pattern
// (non-negative floats order like unsigned integers; max is order-independent, so the result is
// reproducible).  The buffer is zeroed on the stream before this kernel.
// RG 32-row groups per wave (every B^T fragment feeds RG MFMAs), NKS 16-deep slices of the padded rank (exact: no per-slice branch).
// Round 3: the kernel was a chain of load -> vmcnt(0) -> 4 MFMAs per slice (a branch per slice kept hipcc from batching the
// loads; 228 registers: two waves per SIMD) - a third of its MFMA time.  Now a (column tile, limb) BATCH of NKS fragments is
// requested one batch ahead of the MFMAs that consume it (two named register sets), the first MFMA of a tile takes a literal
// zero accumulator, and a lane keeps one running maximum per row group (every register of its accumulator is the same token
// row), folded with v_max3_f32; the lane pair is combined once, at the commit.
#ifndef LQER_AMAX_WAVES
#define LQER_AMAX_WAVES 2048
#endif
template <int RG, int NKS>
__global__ __launch_bounds__(256) void k_bout_amax(GemmArgs g, int tiles_n32, int seg_tiles) {
  // One wave = 32 RG token rows x a run of `seg_tiles` 32-column tiles: the rows' xAq fragments stay in registers, the
  // running maximum of the current B_out block stays in a register and is committed (one atomicMax per row) when
  // the run leaves the block - a handful of atomics per row instead of one per 16 columns.
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t groups = ((g.M + 31) / 32 + RG - 1) / RG;
  const int nseg = (tiles_n32 + seg_tiles - 1) / seg_tiles;
  if (wid >= groups * nseg) return;
  const int tg = (int)(wid / nseg), sg = (int)(wid - (int64_t)tg * nseg);
  const int l31 = lane & 31, lh = lane >> 5;
  const int Mp = (g.M + LQER_M_ALIGN - 1) / LQER_M_ALIGN * LQER_M_ALIGN;  // rows of xaq / bout_amax that exist
  bf16x8 xa[RG][NKS];
  int rowv[RG];
#pragma unroll
  for (int u = 0; u < RG; ++u) {
    const int row = (tg * RG + u) * 32 + l31;
    rowv[u] = row < Mp ? row : -1;
    const int rc = row < Mp ? row : Mp - 1;  // (a clamped duplicate: computed, never committed)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) xa[u][ks] = *(const bf16x8*)(g.xaq + (int64_t)rc * g.xaq_ld + ks * 16 + 8 * lh);
  }
  const int t_begin = sg * seg_tiles;
  const int t_end = t_begin + seg_tiles < tiles_n32 ? t_begin + seg_tiles : tiles_n32;
  int cur_blk = (t_begin * 32) / g.bout_L;
  float cur[RG];
#pragma unroll
  for (int u = 0; u < RG; ++u) cur[u] = 0.f;
  auto commit = [&]() {
#pragma unroll
    for (int u = 0; u < RG; ++u) {
      const float m = pair32_max(cur[u]);  // lanes l and l ^ 32 hold the two column halves of the same token row
      if (lh == 0 && rowv[u] >= 0) {
        if (g.bout_nseg > 0) g.bout_amax[(int64_t)sg * Mp + rowv[u]] = m;  // one block per row: this segment's partial (plain store)
        else atomicMax((unsigned int*)g.bout_amax + (int64_t)rowv[u] * g.bout_nblk + cur_blk, __float_as_uint(m));
      }
    }
  };
  const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int nb = (t_end - t_begin) * g.b_limbs;  // batches: (tile, limb), limb fastest - the GEMM kernels' summation order
  const int64_t limb_stride = (int64_t)g.Np * g.rp;
  const bf16_t* const b_lane = g.bt + (int64_t)l31 * g.rp + 8 * lh;
  bf16x8 ba[NKS], bb[NKS];
  auto load = [&](int i, bf16x8 (&dst)[NKS]) {  // batch i (past the end: the last one again - never used)
    const int ii = i < nb ? i : nb - 1;
    const int tn = t_begin + ii / g.b_limbs, l = ii - (ii / g.b_limbs) * g.b_limbs;
    const bf16_t* p = b_lane + l * limb_stride + (int64_t)tn * 32 * g.rp;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) dst[ks] = *(const bf16x8*)(p + ks * 16);
  };
  f32x16 acc[RG];
  int tn = t_begin, l = 0;
  auto consume = [&](const bf16x8 (&src)[NKS]) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      if (l == 0 && ks == 0) {
#pragma unroll
        for (int u = 0; u < RG; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(src[0], xa[u][0], zero, 0, 0, 0);
      } else {
#pragma unroll
        for (int u = 0; u < RG; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(src[ks], xa[u][ks], acc[u], 0, 0, 0);
      }
    }
    if (++l == g.b_limbs) {  // the tile's product is complete: fold it into the running maxima
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int blk = (tn * 32 + 16 * b) / g.bout_L;  // wave-uniform
        if (blk != cur_blk) {
          commit();
          cur_blk = blk;
#pragma unroll
          for (int u = 0; u < RG; ++u) cur[u] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < RG; ++u) {
          float m = cur[u];
#pragma unroll
          for (int k = 0; k < 8; k += 2) m = fmaxf(fmaxf(m, fabsf(acc[u][8 * b + k])), fabsf(acc[u][8 * b + k + 1]));  // v_max3_f32 |.|
          cur[u] = m;
        }
      }
      l = 0, ++tn;
    }
  };
  if (nb > 0) load(0, ba);
  for (int i = 0; i < nb; i += 2) {
    load(i + 1, bb);
    consume(ba);
    if (i + 1 >= nb) break;
    load(i + 2, ba);
    consume(bb);
  }
  commit();
}

// The same pre-pass with the B^T fragments through LDS, for rank 64 (the W4A8 INT configurations).  Counters of k_bout_amax<4, 4>
// at C4 (tools/r03_sidepmc.sh): the texture addresser is busy 62 % of the kernel, the MFMA pipe 30 % - every wave fetches its own
// copy of the B^T run, 32 lanes x 128-byte rows per request.  Here the four waves of a workgroup take four different groups of
// 128 token rows over the SAME run of column tiles: a stage of AMX_SB (tile, limb) batches - 4 KB each, contiguous in the B^T
// image - is read once per workgroup with one 16-byte request per thread and batch (16 lanes per 256-byte line), swizzled into
// LDS (two stage buffers, one barrier per stage), and every wave reads its fragments from there.  Same maxima (max is
// order-independent), same commit.
constexpr int AMX_SB = 4;
__global__ __launch_bounds__(256) void k_bout_amax_lds(GemmArgs g, int tiles_n32, int seg_tiles) {
  constexpr int RG = 4, NKS = 4;
  __shared__ __attribute__((aligned(16))) unsigned char sbuf[2][AMX_SB][4096];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int64_t groups = ((g.M + 31) / 32 + RG - 1) / RG;
  const int nseg = (tiles_n32 + seg_tiles - 1) / seg_tiles;
  const int wgr = (int)(blockIdx.x / nseg), sg = (int)(blockIdx.x - (int64_t)wgr * nseg);
  const int64_t tg_raw = (int64_t)wgr * 4 + wave;
  const bool live = tg_raw < groups;               // (a wave past the last row group still stages and meets the barriers)
  const int tg = (int)(live ? tg_raw : groups - 1);
  const int Mp = (g.M + LQER_M_ALIGN - 1) / LQER_M_ALIGN * LQER_M_ALIGN;
  bf16x8 xa[RG][NKS];
  int rowv[RG];
#pragma unroll
  for (int u = 0; u < RG; ++u) {
    const int row = (tg * RG + u) * 32 + l31;
    rowv[u] = live && row < Mp ? row : -1;
    const int rc = row < Mp ? row : Mp - 1;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) xa[u][ks] = *(const bf16x8*)(g.xaq + (int64_t)rc * g.xaq_ld + ks * 16 + 8 * lh);
  }
  const int t_begin = sg * seg_tiles;
  const int t_end = t_begin + seg_tiles < tiles_n32 ? t_begin + seg_tiles : tiles_n32;
  int cur_blk = (t_begin * 32) / g.bout_L;
  float cur[RG];
#pragma unroll
  for (int u = 0; u < RG; ++u) cur[u] = 0.f;
  auto commit = [&]() {
#pragma unroll
    for (int u = 0; u < RG; ++u) {
      const float m = pair32_max(cur[u]);
      if (lh == 0 && rowv[u] >= 0) {
        if (g.bout_nseg > 0) g.bout_amax[(int64_t)sg * Mp + rowv[u]] = m;
        else atomicMax((unsigned int*)g.bout_amax + (int64_t)rowv[u] * g.bout_nblk + cur_blk, __float_as_uint(m));
      }
    }
  };
  const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int nb = (t_end - t_begin) * g.b_limbs;  // batches (tile, limb), limb fastest
  const int nst = (nb + AMX_SB - 1) / AMX_SB;
  const int64_t limb_stride = (int64_t)g.Np * g.rp;
  // staging: thread t moves 16-byte chunk t of a batch's 4 KB (row t / 8 of the tile, chunk t % 8 of its 128 B) to the swizzled slot
  const int srow = tid >> 3, sch = tid & 7;
  const int sdst = srow * 128 + ((sch ^ (srow & 7)) << 4);
  u32x4 st[AMX_SB];
  auto gload = [&](int s) {
#pragma unroll
    for (int u = 0; u < AMX_SB; ++u) {
      const int i = s * AMX_SB + u, ii = i < nb ? i : nb - 1;  // (past the end: the last batch again, never consumed)
      const int tn = t_begin + ii / g.b_limbs, l = ii - (ii / g.b_limbs) * g.b_limbs;
      st[u] = *(const u32x4*)(g.bt + l * limb_stride + (int64_t)tn * 32 * g.rp + tid * 8);
    }
  };
  auto lwrite = [&](int buf) {
#pragma unroll
    for (int u = 0; u < AMX_SB; ++u) *(u32x4*)(&sbuf[buf][u][sdst]) = st[u];
  };
  int fo[NKS];  // fragment offsets inside a batch: row l31, chunk 2 ks + lh
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) fo[ks] = l31 * 128 + (((2 * ks + lh) ^ (l31 & 7)) << 4);
  f32x16 acc[RG];
  int tn = t_begin, l = 0;
  gload(0);
  lwrite(0);
  __syncthreads();
  for (int s = 0; s < nst; ++s) {
    if (s + 1 < nst) gload(s + 1);
    const int buf = s & 1;
#pragma unroll
    for (int u2 = 0; u2 < AMX_SB; ++u2) {
      if (s * AMX_SB + u2 < nb) {
        bf16x8 fr[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) fr[ks] = *(const bf16x8*)(&sbuf[buf][u2][fo[ks]]);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          if (l == 0 && ks == 0) {
#pragma unroll
            for (int u = 0; u < RG; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[0], xa[u][0], zero, 0, 0, 0);
          } else {
#pragma unroll
            for (int u = 0; u < RG; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[ks], xa[u][ks], acc[u], 0, 0, 0);
          }
        }
        if (++l == g.b_limbs) {
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int blk = (tn * 32 + 16 * b) / g.bout_L;  // wave-uniform
            if (blk != cur_blk) {
              commit();
              cur_blk = blk;
#pragma unroll
              for (int u = 0; u < RG; ++u) cur[u] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < RG; ++u) {
              float m = cur[u];
#pragma unroll
              for (int k = 0; k < 8; k += 2) m = fmaxf(fmaxf(m, fabsf(acc[u][8 * b + k])), fabsf(acc[u][8 * b + k + 1]));
              cur[u] = m;
            }
          }
          l = 0, ++tn;
        }
      }
    }
    if (s + 1 < nst) lwrite(buf ^ 1);  // (everybody left that buffer before the previous barrier)
    __syncthreads();
  }
  commit();
}

template <int DT>
static int launch_gemm(const GemmArgs& g, bool lowrank, int bout, hipStream_t st) {
  const unsigned grid = (unsigned)(g.tiles_m * g.tiles_n);
#define LQER_GEMM_LAUNCH(LR, BO)                                                                                \
  do {                                                                                                          \
    static LdsLimitOnce lds_once;                                                                               \
    lds_once.set((const void*)k_lqer_gemm<DT, LR, BO>, gemm_lds_bytes(4));                                      \
    k_lqer_gemm<DT, LR, BO><<<grid, 512, gemm_lds_bytes(4), st>>>(g);                                           \
  } while (0)
#define LQER_GEMM_LAUNCH_STAGED(BO)                                                                             \
  do {                                                                                                          \
    static LdsLimitOnce lds_once;                                                                               \
    lds_once.set((const void*)k_lqer_gemm<DT, true, BO, true>, gemm_lds_bytes(4));                              \
    k_lqer_gemm<DT, true, BO, true><<<grid, 512, gemm_lds_bytes(4), st>>>(g);                                   \
  } while (0)
#ifndef LQER_STAGE_MIN
#define LQER_STAGE_MIN 32
#endif
  const bool staged = lowrank && g.rp * g.b_limbs > LQER_STAGE_MIN;  // more than two 16-deep slices of side product
  if (g.w_twos) {  // integer weights: one instantiation per (element type, side path, B_out) - 128-row tiles, staged side path
    if constexpr (DT == LQER_F16X) {
      set_error("linear_gemm: integer weights have no fp16 main loop (pass-through fp16 activations take the limb route)");
      return LQER_E_UNSUPPORTED;
    } else {
#define LQER_GEMM_LAUNCH_TWOS(LR, BO, ST)                                                                       \
  do {                                                                                                          \
    static LdsLimitOnce lds_once;                                                                               \
    lds_once.set((const void*)k_lqer_gemm<DT, LR, BO, ST, 4, true>, gemm_lds_bytes(4));                         \
    k_lqer_gemm<DT, LR, BO, ST, 4, true><<<grid, 512, gemm_lds_bytes(4), st>>>(g);                              \
  } while (0)
      if (!lowrank) LQER_GEMM_LAUNCH_TWOS(false, 0, false);
      else if (bout == 1) LQER_GEMM_LAUNCH_TWOS(true, 1, true);
      else if (bout == 2) LQER_GEMM_LAUNCH_TWOS(true, 2, true);
      else LQER_GEMM_LAUNCH_TWOS(true, 0, true);
#undef LQER_GEMM_LAUNCH_TWOS
      return check_launch("lqer_gemm");
    }
  }
  if (lowrank && g.xa_part) {  // the partial tiles of x A instead of xAq (api: lqer_tile_partials): staged side path, 16-bit tensors
    if constexpr (DT != LQER_F16 && DT != LQER_BF16) {
      set_error("linear_gemm: the partial-tile route takes fp16 / bf16 tensors");
      return LQER_E_UNSUPPORTED;
    } else {
#define LQER_GEMM_LAUNCH_XAP(BO, MTv)                                                                           \
  do {                                                                                                          \
    static LdsLimitOnce lds_once;                                                                               \
    lds_once.set((const void*)k_lqer_gemm<DT, true, BO, true, MTv, false, true>, gemm_lds_bytes(MTv));          \
    k_lqer_gemm<DT, true, BO, true, MTv, false, true><<<grid, 512, gemm_lds_bytes(MTv), st>>>(g);               \
  } while (0)
      if (bout == 2 || g.tiles_m_rows != BM) {  // (64-row tiles: two workgroups per CU leave the reduction no registers)
        set_error("linear_gemm: the partial-tile route serves 128-row tiles with B_out in blocks of 16 or pass-through");
        return LQER_E_UNSUPPORTED;
      }
      if (bout == 1) LQER_GEMM_LAUNCH_XAP(1, 4); else LQER_GEMM_LAUNCH_XAP(0, 4);
#undef LQER_GEMM_LAUNCH_XAP
      return check_launch("lqer_gemm");
    }
  }
#define LQER_GEMM_LAUNCH_H64(LR, BO, ST)                                                                        \
  do {                                                                                                          \
    static LdsLimitOnce lds_once;                                                                               \
    lds_once.set((const void*)k_lqer_gemm<DT, LR, BO, ST, 2>, gemm_lds_bytes(2));                               \
    k_lqer_gemm<DT, LR, BO, ST, 2><<<grid, 512, gemm_lds_bytes(2), st>>>(g);                                    \
  } while (0)
  if (g.tiles_m_rows == 64) {  // (gemm_dispatch: the 128-row grid would fill at most half of the CUs)
    if (!lowrank) LQER_GEMM_LAUNCH_H64(false, 0, false);
    else if (bout == 1) { if (staged) LQER_GEMM_LAUNCH_H64(true, 1, true); else LQER_GEMM_LAUNCH_H64(true, 1, false); }
    else if (bout == 2) { if (staged) LQER_GEMM_LAUNCH_H64(true, 2, true); else LQER_GEMM_LAUNCH_H64(true, 2, false); }
    else { if (staged) LQER_GEMM_LAUNCH_H64(true, 0, true); else LQER_GEMM_LAUNCH_H64(true, 0, false); }
    return check_launch("lqer_gemm");
  }
#undef LQER_GEMM_LAUNCH_H64
  // B_out in blocks of 16 re-quantized under the first 16 k-steps instead of in front of the main loop (DEFER above): K >= 1024,
  // clamps within the magic-number rounding, and the 1e-8 pass-through that makes the clamped scale exponent exact
  const bool defer = lowrank && bout == 1 && !(g.tuning & LQER_TUNE_BOUT_IN_PROLOGUE) && g.Kp / BK >= 16 && g.bout.kind == LQER_Q_MXINT &&
                     g.bout.mmax <= 4194304.0f && g.bout.mneg <= 4194304.0f && g.bout.tiny >= 1e-8f;
#define LQER_GEMM_LAUNCH_DEFER(ST)                                                                              \
  do {                                                                                                          \
    static LdsLimitOnce lds_once;                                                                               \
    lds_once.set((const void*)k_lqer_gemm<DT, true, 1, ST, 4, false, false, true>, gemm_lds_bytes(4));          \
    k_lqer_gemm<DT, true, 1, ST, 4, false, false, true><<<grid, 512, gemm_lds_bytes(4), st>>>(g);               \
  } while (0)
  if (!lowrank)
    LQER_GEMM_LAUNCH(false, 0);
  else if (bout == 1 && defer) {
    if (staged) LQER_GEMM_LAUNCH_DEFER(true); else LQER_GEMM_LAUNCH_DEFER(false);
  } else if (bout == 1) {
    if (staged) LQER_GEMM_LAUNCH_STAGED(1); else LQER_GEMM_LAUNCH(true, 1);
  } else if (bout == 2) {
    if (staged) LQER_GEMM_LAUNCH_STAGED(2); else LQER_GEMM_LAUNCH(true, 2);
  } else {
    if (staged) LQER_GEMM_LAUNCH_STAGED(0); else LQER_GEMM_LAUNCH(true, 0);
  }
#undef LQER_GEMM_LAUNCH_STAGED
#undef LQER_GEMM_LAUNCH
  return check_launch("lqer_gemm");
}

#if defined(LQER_STAMPS) || defined(LQER_CLOCKPROBE)
extern "C" int lqer_debug_set_stamp_buffer(void* p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &p, sizeof(p));
}
#endif

size_t gemm_scratch_bytes(int64_t m_max, int64_t N, const QP& bout) {
  if (bout.kind != LQER_Q_MXINT || bout.block == 16) return 0;
  const int64_t Np = lqer_padded_n(N);
  const int64_t L = (bout.block <= 0 || bout.block >= N) ? Np : bout.block;
  const int64_t nblk = (Np + L - 1) / L;
  // (one block per row on the int8 route: up to LQER_AMAX_NSEG column-segment partials per row instead of one atomic cell - or as
  // many {value, tag} granules of 8 bytes when the GEMM exchanges the maxima itself)
  return (size_t)lqer_padded_m(m_max) * (nblk == 1 ? 2 * LQER_AMAX_NSEG : nblk) * sizeof(float);
}

// B_out handling of a launch: 0 pass-through, 1 blocks of 16 (maxima in registers), 2 other blocks (pre-pass); < 0 error
static int bout_mode(const GemmArgs& g, bool lowrank, int* L_out) {
  if (lowrank && g.bout.kind == LQER_Q_MXINT) {
    if (g.bout.block == 16) return 1;
    const int L = (g.bout.block <= 0 || g.bout.block >= g.N) ? g.Np : g.bout.block;
    if (L % 16 != 0) {
      set_error("B_out_quantizer block %d: must be a multiple of 16 or cover the row", g.bout.block);
      return LQER_E_UNSUPPORTED;
    }
    if (L_out) *L_out = L;
    return 2;
  }
  if (lowrank && g.bout.kind == LQER_Q_INT) {  // fixed point: elementwise, no block maxima - the "any block" code without its pre-pass
    if (L_out) *L_out = 0;
    return 2;
  }
  if (lowrank && g.bout.kind != LQER_Q_PASSTHROUGH) {
    set_error("B_out_quantizer kind %d not implemented", g.bout.kind);
    return LQER_E_UNSUPPORTED;
  }
  return 0;
}

int gemm_route(const GemmArgs& g, bool lowrank) {
  const int bout = bout_mode(g, lowrank, nullptr);
  if (bout < 0) return bout;
  if (g.w8) {  // LQER_Q_MXINT_I8: the int8 kernel or nothing (the caller falls back to LQER_Q_MXINT on the same buffers)
    GemmArgs t = g;
    if (bout == 2 && g.bout.kind == LQER_Q_MXINT) {
      const int L = (g.bout.block <= 0 || g.bout.block >= g.N) ? g.Np : g.bout.block;
      t.bout_nblk = (g.Np + L - 1) / L;
    }
    if (i8_eligible(t, bout)) return LQER_ROUTE_I8;
    t.w8 = nullptr;
    return gemm_route(t, lowrank);
  }
  if (g.w_twos) return LQER_ROUTE_TILE128;  // integer weights (two's-complement nibbles): the 128-row tile kernel at every M
  if (smallm_eligible(g, bout)) return LQER_ROUTE_SMALLM;
  if (m256_eligible(g)) return LQER_ROUTE_TILE256;
  return LQER_ROUTE_TILE128;
}

// Rows of a tile of the 128-row kernel family.  Token counts whose 128-row grid covers at most half of the CUs: 64-row tiles
// (twice the workgroups, half the MFMA work per expanded weight fragment - the k-step is then paced by the weight expand,
// NOTEBOOK.md §4.1) as long as they still fit one round.
int gemm_tile_rows(const GemmArgs& g) {
#ifndef LQER_NO_H64
  constexpr int CUS = 256;
  const int64_t tn = g.Np / BN;
  const int64_t t128 = (int64_t)((g.M + BM - 1) / BM) * tn, t64 = (int64_t)((g.M + 63) / 64) * tn;
  const int pin = (g.tuning & LQER_TUNE_TILE_ROWS_128) ? 128 : ((g.tuning & LQER_TUNE_TILE_ROWS_64) ? 64 : 0);  // (tests)
  if (!g.w_twos && ((pin != 128 && 2 * t128 <= CUS && t64 > t128 && g.M > 64) || (pin == 64 && g.M > 64))) return 64;
#endif
  return BM;
}

__global__ void k_zero_cells(uint32_t* p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0u;
}

// How a launch with one or more B_out blocks per row (bout == 2, block_fp; g.bout_nblk set) gets its row-block maxima:
//   xch   - one round of the int8 kernel's 128-row tiles: exchanged inside the GEMM launch, no pre-pass;
//   parts - one block per row, consumed by the int8 kernel: every wave of the pre-pass leaves the maximum of ITS column segment in its
//           own cell [segment][row] (plain stores; at most LQER_AMAX_NSEG segments, the GEMM folds them when it reads a row's
//           constants) - no atomics, so no zero-fill launch in front (4.7 us of a 62-us step at M = 2048).  Taken while the segments stay
//           narrow (up to 8 column tiles: beyond N = 4096 wider segments mean half the waves, each twice as long - 2048 x 11008, rank 32:
//           111.9 us with partials against 104.4 with cells; tools/ab_i8.py --rows --amax), or where the pre-pass would not use more
//           than LQER_AMAX_NSEG segments anyway (token counts from ~16k: C4 - the same pre-pass grid, minus the zero fill);
//   else  - one atomicMax cell per (row, block), zeroed first: `need` bytes at the head of the scratch.
struct AmaxPlan {
  bool parts, xch, mrx;
  size_t need;
};
static AmaxPlan amax_plan(const GemmArgs& g, bool lowrank, int bout) {
  const int tiles_n32 = g.Np / 32;
  const bool one = g.w8 && g.bout_nblk == 1;
  // segments the pre-pass would use with no cap (its LDS variant at rank 64, else the register variant at its default row groups)
  int64_t uncapped;
  if (g.rp == 64) {
    const int64_t wgroups4 = ((((g.M + 31) / 32 + 3) / 4) + 3) / 4;
    uncapped = (LQER_AMAX_WAVES / 4) / wgroups4;
  } else {
    const int RG = g.rp <= 64 ? 4 : (g.rp <= 128 ? 2 : 1);
    uncapped = LQER_AMAX_WAVES / (((g.M + 31) / 32 + RG - 1) / RG);
  }
  AmaxPlan p;
  p.parts = one && !(g.tuning & LQER_TUNE_AMAX_ATOMIC) &&
            (tiles_n32 <= 8 * LQER_AMAX_NSEG || uncapped <= LQER_AMAX_NSEG || (g.tuning & LQER_TUNE_AMAX_PARTS));
  // ... or no pre-pass at all: one round of the int8 kernel's 128-row tiles exchanges the maxima inside the GEMM launch
  p.xch = one && !(g.tuning & (LQER_TUNE_AMAX_ATOMIC | LQER_TUNE_AMAX_PARTS)) && i8_eligible(g, bout) && i8_amax_exchange_ok(g, lowrank, bout);
  //   mrx - several rounds of 128-row tiles on a resident grid: the GEMM's workgroups each compute one item of the pre-pass at their start
  //         and exchange {maximum, tag} granules (gemm_w4a8_i8.hip, MRX): no pre-pass launch, nothing zeroed
  p.mrx = one && !p.xch && !(g.tuning & (LQER_TUNE_AMAX_NO_MRX | LQER_TUNE_AMAX_ATOMIC | LQER_TUNE_AMAX_PARTS)) && i8_eligible(g, bout) &&
          i8_amax_mrx_ok(g, lowrank, bout);
  if (p.mrx) p.parts = false;
  p.need = (size_t)lqer_padded_m(g.M) * ((p.xch || p.mrx) ? 2 * LQER_AMAX_NSEG : (p.parts ? LQER_AMAX_NSEG_WIDE : g.bout_nblk)) * sizeof(float);
  return p;
}

// Bytes at the head of the GEMM's scratch that the pre-pass of this launch needs zeroed before it runs (0: no pre-pass, segment
// partials or the in-launch exchange).  lqer_linear_forward lets the one-launch activation kernel of the same forward write the
// zeros (the two calls are handed the same scratch) and sets g.amax_zeroed - the memset launch in front of the pre-pass cost 4.8 us
// of the 116-us forward at 2048 x 4096 -> 11008.
size_t gemm_amax_zero_bytes(GemmArgs g, bool lowrank) {
  int L = 0;
  if (g.M == 0 || g.N == 0 || bout_mode(g, lowrank, &L) != 2 || g.bout.kind != LQER_Q_MXINT) return 0;
  g.bout_L = L, g.bout_nblk = (g.Np + L - 1) / L;
  // (asked before the GEMM call's limb count is known: the permissive answer - a launch that then cannot exchange the maxima itself
  // finds nothing prepared and zero-fills its cells as ever)
  if (g.b_limbs == 0) g.b_limbs = 1;
  const AmaxPlan p = amax_plan(g, lowrank, 2);
  return (p.parts || p.xch || p.mrx) ? 0 : p.need;
}

int gemm_dispatch(GemmArgs g, int dtype, bool lowrank, void* scratch, size_t scratch_bytes, hipStream_t st) {
  if (g.M == 0 || g.N == 0) return LQER_OK;
  int L = 0;
  const int bout = bout_mode(g, lowrank, &L);
  if (bout < 0) return bout;
  if (bout == 2 && g.bout.kind == LQER_Q_INT) {
    g.bout_L = 16, g.bout_nblk = 0, g.bout_amax = nullptr;
  } else if (bout == 2) {
      g.bout_L = L;
      g.bout_nblk = (g.Np + L - 1) / L;
      const int tiles_n32 = g.Np / 32;
      const AmaxPlan ap = amax_plan(g, lowrank, bout);
      const bool parts = ap.parts, xch = ap.xch || ap.mrx;
      const size_t need = ap.need;
      if (!scratch || scratch_bytes < need) {
        set_error("linear_gemm: scratch %zu B < %zu B for the B_out row-block maxima", scratch_bytes, need);
        return LQER_E_WORKSPACE;
      }
      g.bout_amax = (float*)scratch;
      g.bout_nseg = 0;
      g.bout_xch = 0;
      if (xch) {
        // the call's tag: a counter spread over all 32 bits (odd multiplier: a bijection); the kernel mixes in its dispatch id and queue
        static std::atomic<uint32_t> xch_calls{1};
        g.bout_xch = ap.mrx ? 2 : 1;
        g.xch_nonce = xch_calls.fetch_add(1, std::memory_order_relaxed) * 0x9E3779B1u;
      } else {
      // (amax_zeroed: the activation kernel of the same forward did it.  A kernel, not hipMemsetAsync: as a memset NODE of a captured graph
      // the fill left two of every four cells untouched on replay - ROCm 7.2, tools/graph_replay_gemm.py - while the eager call was fine)
      if (!parts && !g.amax_zeroed) k_zero_cells<<<(unsigned)((need / 4 + 255) / 256), 256, 0, st>>>((uint32_t*)scratch, (int64_t)(need / 4));
      // padded rank (x limbs of x A) -> 16-deep slices (a template parameter: exact, no per-slice branch) and row groups per wave
      const int nks = g.rp / 16;
      int RG = g.rp <= 64 ? 4 : (g.rp <= 128 ? 2 : 1);
      const int nseg_cap = parts ? (tiles_n32 <= 8 * LQER_AMAX_NSEG ? LQER_AMAX_NSEG : LQER_AMAX_NSEG_WIDE) : tiles_n32;
      // (segment partials cap the column split: fewer row groups per wave keep the grid at about a thousand waves)
      if (parts && nks <= 4)
        while (RG > 1 && (((g.M + 31) / 32 + RG - 1) / RG) * nseg_cap < 1024) RG >>= 1;
      const int64_t groups = ((g.M + 31) / 32 + RG - 1) / RG;
      int nseg = (int)(LQER_AMAX_WAVES / groups);  // one round of two waves per SIMD (the kernel holds 184-256 registers)
      nseg = nseg < 1 ? 1 : (nseg > tiles_n32 ? tiles_n32 : nseg);
      nseg = nseg > nseg_cap ? nseg_cap : nseg;
      const int seg_tiles = (tiles_n32 + nseg - 1) / nseg;
      const int nseg_used = (tiles_n32 + seg_tiles - 1) / seg_tiles;
      const int64_t waves = groups * nseg_used;
      const unsigned grid = (unsigned)((waves + 3) / 4);
      if (parts) g.bout_nseg = nseg_used;
#define LQER_AMAX(RGv, NKSv) k_bout_amax<RGv, NKSv><<<grid, 256, 0, st>>>(g, tiles_n32, seg_tiles)
#ifndef LQER_AMAX_NO_LDS
      const int64_t wgroups4 = ((((g.M + 31) / 32 + 3) / 4) + 3) / 4;  // workgroups of the LDS variant along the rows (4 waves x 4 row groups)
      if (nks == 4 && g.rp == 64 && (!parts || wgroups4 * LQER_AMAX_NSEG >= 256)) {  // rank 64: the B^T run through LDS, four row groups per workgroup
        int ns = (int)((LQER_AMAX_WAVES / 4) / wgroups4);
        ns = ns < 1 ? 1 : (ns > tiles_n32 ? tiles_n32 : ns);
        ns = ns > nseg_cap ? nseg_cap : ns;
        const int st_l = (tiles_n32 + ns - 1) / ns;
        const int ns_used = (tiles_n32 + st_l - 1) / st_l;
        const int64_t wgs = wgroups4 * ns_used;
        if (parts) g.bout_nseg = ns_used;
        k_bout_amax_lds<<<(unsigned)wgs, 256, 0, st>>>(g, tiles_n32, st_l);
      } else if (nks <= 4 && RG < 4) {
        switch (nks * 4 + RG) {
          case 4 + 1: LQER_AMAX(1, 1); break;
          case 4 + 2: LQER_AMAX(2, 1); break;
          case 8 + 1: LQER_AMAX(1, 2); break;
          case 8 + 2: LQER_AMAX(2, 2); break;
          case 12 + 1: LQER_AMAX(1, 3); break;
          case 12 + 2: LQER_AMAX(2, 3); break;
          case 16 + 1: LQER_AMAX(1, 4); break;
          default: LQER_AMAX(2, 4); break;
        }
      } else
#endif
      switch (nks) {
        case 1: LQER_AMAX(4, 1); break;
        case 2: LQER_AMAX(4, 2); break;
        case 3: LQER_AMAX(4, 3); break;
        case 4: LQER_AMAX(4, 4); break;
        case 5: LQER_AMAX(2, 5); break;
        case 6: LQER_AMAX(2, 6); break;
        case 7: LQER_AMAX(2, 7); break;
        case 8: LQER_AMAX(2, 8); break;
        case 9: LQER_AMAX(1, 9); break;
        case 10: LQER_AMAX(1, 10); break;
        case 11: LQER_AMAX(1, 11); break;
        case 12: LQER_AMAX(1, 12); break;
        case 13: LQER_AMAX(1, 13); break;
        case 14: LQER_AMAX(1, 14); break;
        case 15: LQER_AMAX(1, 15); break;
        case 16: LQER_AMAX(1, 16); break;
        default: set_error("B_out pre-pass: padded rank %d x limbs > 256", g.rp); return LQER_E_UNSUPPORTED;
      }
#undef LQER_AMAX
      }  // (pre-pass)
  }
  if (g.w8) {  // LQER_Q_MXINT_I8: xq is the int8 image - only the int8 kernel can read it
    if (!i8_eligible(g, bout)) {
      set_error("linear_gemm: LQER_Q_MXINT_I8 is not served for M=%d here (lqer_gemm_route != LQER_ROUTE_I8): call with "
                "LQER_Q_MXINT", g.M);
      return LQER_E_UNSUPPORTED;
    }
    return i8_dispatch(g, dtype, lowrank, bout, st);
  }
  if (!g.w_twos) {
    if (smallm_eligible(g, bout)) return smallm_dispatch(g, dtype, lowrank, bout, st);  // decode sizes: HBM-bound variant
    if (m256_eligible(g) && !g.xa_part) return m256_dispatch(g, dtype, lowrank, bout, st);  // large M: 256 x 256 tiles
  }
  g.tiles_n = g.Np / BN;
  g.tiles_m_rows = gemm_tile_rows(g);
  g.tiles_m = (g.M + g.tiles_m_rows - 1) / g.tiles_m_rows;
  {
    const int bm = (g.tuning >> 4) & 0x3f, nt = g.tiles_m * g.tiles_n;  // LQER_TUNE_XCD_BLOCK (measurements)
    g.xcd_bm = 0;
    if (bm > 0 && nt % 8 == 0 && (nt / 8) % bm == 0 && g.tiles_m % bm == 0 && 8 % (g.tiles_m / bm) == 0 &&
        g.tiles_n % (8 / (g.tiles_m / bm)) == 0 && (nt / 8) / bm == g.tiles_n / (8 / (g.tiles_m / bm)))
      g.xcd_bm = bm;
  }
  switch (dtype) {
    case LQER_F32: return launch_gemm<LQER_F32>(g, lowrank, bout, st);
    case LQER_F16: return g.x_f16 ? launch_gemm<LQER_F16X>(g, lowrank, bout, st) : launch_gemm<LQER_F16>(g, lowrank, bout, st);
    case LQER_BF16: return launch_gemm<LQER_BF16>(g, lowrank, bout, st);
  }
  set_error("unknown dtype %d", dtype);
  return LQER_E_INVALID;
}

}  // namespace lqer
