// The whole forward of a decode-size Linear (M <= 8 tokens) in ONE launch:
//
//   y = Q_x(x) Wq^T + bq + Q_Bout( Q_Aout( Q_x(x) A ) B )        (reference quantized_layers/linear.py:145-157)
//
// The two-launch decode route (k_quant_xa16 -> k_lqer_gemm_smallm) is bound by two kernel latencies and the boundary
// between them (11 us at M = 1 for 9.4 MB of weights).  Here one grid holds two kinds of workgroups:
//   * producers (blocks 0 .. np-1, one per 256 k of K): quantize their slab of x, multiply it with the slab of A
//     (v_mfma_f32_16x16x32_bf16: 4 waves x 2 k-slices, the waves' tiles summed in a fixed order) and publish the partial tile of x A as 8-byte
//     {value, tag} granules - one write-through (sc1) store each, the tag is this LAUNCH's nonce (host counter + dispatch id): no flag, no counter, nothing
//     to reset (MI355X_MICROARCH.md, hand-off price list: data-tagged granules);
//   * consumers (one per 16 output columns, as in gemm_smallm.hip): request their first weight panels, quantize ALL of x
//     into an LDS image themselves (M x K <= 8 x 4096 elements: cheaper than waiting for another kernel), stream their
//     packed weight rows through the MFMA, and only at the very end read the producers' granules (sc1 loads, requested as
//     the weight stream ends so that their round trip passes under the cross-wave combine, polled until the tags match),
//     sum them in a fixed slab order, apply A_out and run the side path epilogue of gemm_smallm.hip.
// The kernel is a chain of latencies (measured with -DLQER_D1_STAMPS, tools/d1_stamps.py, M = 1, K = N = 4096: first data
// 1.4 us after entry - every kernel starts behind cold caches -, producers acked at ~3 us, weight stream done at ~4-5,
// granule round trip 1.2, tail 1.8), so the tail is written for instruction COUNT: branch-free exponent rule, power-of-two
// scaling by exponent bits instead of ldexpf, DPP / v_permlane*_swap maxima instead of LDS shuffles, the workgroup's
// "all granules seen" vote carried by a barrier that is there anyway, bias and B fragments requested before the wait.
// GROUPS (lqer_linear_forward_group): Linears that are handed the SAME tokens - q/k/v, gate/up (reference llama_decoder.py:222-224,
// :104) - run as ONE launch: the producers multiply x with the concatenation of the members' A (rank tiles in chunks of four), the
// consumer workgroups of all members stream their own packed rows (the members' images stay where they are: a table of up
// to four members rides in the kernel arguments) and pick the rank columns of their member out of the shared granules.  Per
// member the arithmetic is that of its own launch: same bits.
// No workgroup ever waits for a consumer, producers wait for nobody, and the poll is bounded: if a granule has not arrived
// after QD1_SPIN sweeps (it has, in practice, long before a consumer asks - producers are the first blocks of the grid and
// finish in ~2 us) the consumer workgroup computes every partial tile itself with the producers' own routine (same bits)
// and goes on: every wave reaches the end of the kernel whatever the dispatch order.
#include <atomic>
#include <type_traits>

#include <cstdio>
#include <ctime>

#include "common.h"

namespace lqer {
namespace d1 {

constexpr int NW = 8;          // waves per workgroup
constexpr int MAXM = 8;        // token rows
constexpr int SLAB_K = 256;    // k per producer
constexpr int MAXNT = 4;       // rank tiles of 16 (padded rank <= 64)
#ifndef LQER_QD1_SPIN
#define LQER_QD1_SPIN 4096
#endif
// LDS (dynamic): [xs image: M x Kp bf16  |alias|  pslab 4 KiB, pred 28 KiB][red: cpw x 8 KiB][xaq 1 KiB][flag]
constexpr int MAXCPW = 4;                         // column blocks of 16 per consumer workgroup
constexpr int RED1_BYTES = NW * 4 * 64 * 4;       // one block's partial sums of the 8 waves
                                                  // (every block keeps its own: ONE barrier for all of a workgroup's blocks)
constexpr int XAQ_BYTES = MAXM * 64 * 2;
constexpr int PSLAB_BYTES = MAXM * SLAB_K * 2;
constexpr int PRED_BYTES = (NW - 1) * MAXNT * 4 * 64 * 4;
constexpr int FLAG_BYTES = 16;  // the workgroup's "a granule was missing" vote: its own word behind everything else

#ifndef LQER_D1_PREFETCH
#define LQER_D1_PREFETCH 1  // where the first granule batch is requested: 0 after the combine, 1 at the end of the weight stream, 2 inside its last iteration
#endif
#ifdef LQER_D1_STAMPS
__device__ unsigned long long* g_d1_stamps = nullptr;  // diagnostic build: s_memrealtime (100 MHz) at the phases of every workgroup
#define D1_STAMP(i)                                                                                            \
  do {                                                                                                         \
    unsigned long long t_;                                                                                     \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                            \
    if (g_d1_stamps && tid == 0) g_d1_stamps[(size_t)blockIdx.x * 16 + (i)] = t_;                                \
  } while (0)
// (after the instruction that produced `dep`: a stamp behind a load's first use)
#define D1_STAMP_AFTER(i, dep)                                                                                 \
  do {                                                                                                         \
    unsigned long long t_;                                                                                     \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "v"(dep) : "memory");                 \
    if (g_d1_stamps && tid == 0) g_d1_stamps[(size_t)blockIdx.x * 16 + (i)] = t_;                                \
  } while (0)
#else
#define D1_STAMP(i)
#define D1_STAMP_AFTER(i, dep)
#endif

// llvm.amdgcn.dispatch.id: the position of this launch's AQL packet in its queue - the same value in every workgroup of a
// launch, a new one for every launch, replayed hipGraph nodes included (clang has no builtin for it; the asm label binds
// the intrinsic, and the kernel descriptor then requests the two system SGPRs)
extern "C" __device__ unsigned long long lqer_dispatch_id() __asm("llvm.amdgcn.dispatch.id");

constexpr int MAXMEM = 4;  // Linears of one launch (q/k/v, gate/up)
struct Member {        // one Linear of the launch: its own packed operands and output
  const uint8_t* wp;   // packed panels
  const bf16_t* bt;    // [limbs][Np][rp]
  const float* bias;   // [Np] or null
  void* y;
  int64_t ldy;
  int N, Np, rp, b_limbs;
  int r_off;           // its first column in the concatenated x A (granules, A^T image)
  int cb0;             // its first consumer workgroup (consumer ids run member-major)
  int nblk;            // its column blocks of 16 (Np / 16); a consumer workgroup takes `cpw` consecutive ones
};
// The kernel-argument block is kept SMALL: on this stack the host writes it per launch at ~7 ns per byte (measured: 920 B against
// 616 B = +2.2 us per lqer_linear_forward call, and decode steps at M <= 4 are host-bound) - only what the kernel reads, and a
// single Linear's launch carries one table entry (NM = 1: 280 B), a group's four.
template <int NM>
struct Args {
  Member mem[NM];
  int M, Kp;           // tokens, padded K (shared by the members)
  QP aout, bout;       // A_out / B_out formats (shared)
  int nmem;
  int rp_all;          // padded ranks summed over the members: row pitch of the granules and rows of the A^T image
  const void* x;       // [M, K] tokens, row stride ldx
  int64_t ldx;
  int K;
  QP qx;
  const bf16_t* a_t;   // A^T bf16 image [rp_all][Kp] (one limb; the members' images concatenated along the rank)
  uint32_t* gran;      // granules [np][MAXM][rp_all] x {value, tag}
  uint32_t nonce;      // host part of the granule tag (a per-call counter); the kernel mixes in its dispatch id and queue
  int np;              // producers = ceil(Kp / 256)
  int spin;            // poll sweeps before a consumer computes the tiles itself
  int cpw;             // column blocks per consumer workgroup: the grid never exceeds what is resident at once (one round)
};

typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

struct SmPanel {
  u32x4 cw;
  uint32_t ex;
};

template <int DT>
__device__ __forceinline__ void load_block16(const void* x, int64_t off, float (&v)[16]) {
  if constexpr (DT == LQER_F32) {
    const float4* p = (const float4*)((const float*)x + off);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 t = p[i];
      v[4 * i] = t.x, v[4 * i + 1] = t.y, v[4 * i + 2] = t.z, v[4 * i + 3] = t.w;
    }
  } else {
    const uint4* p = (const uint4*)((const bf16_t*)x + off);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint4 t = p[i];
      const uint32_t w[4] = {t.x, t.y, t.z, t.w};
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

// the same block as the bytes it is stored in (a request that costs 8 registers for the 16-bit types, converted at its use)
template <int DT>
struct RawBlk {
  u32x4 r[DT == LQER_F32 ? 4 : 2];
};
template <int DT>
__device__ __forceinline__ void load_raw16(const void* x, int64_t off, RawBlk<DT>& b) {
  const u32x4* p = (const u32x4*)((const char*)x + off * (DT == LQER_F32 ? 4 : 2));
#pragma unroll
  for (int i = 0; i < (DT == LQER_F32 ? 4 : 2); ++i) b.r[i] = p[i];
}
template <int DT>
__device__ __forceinline__ void raw16_to_f32(const RawBlk<DT>& b, float (&v)[16]) {
  if constexpr (DT == LQER_F32) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = __uint_as_float(b.r[i >> 2][i & 3]);
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t w = b.r[i >> 2][i & 3];
      if constexpr (DT == LQER_F16) {
        typedef __attribute__((ext_vector_type(2))) _Float16 h2;
        const h2 h = __builtin_bit_cast(h2, w);
        v[2 * i] = (float)h[0], v[2 * i + 1] = (float)h[1];
      } else {
        v[2 * i] = __uint_as_float(w << 16), v[2 * i + 1] = __uint_as_float(w & 0xffff0000u);
      }
    }
  }
}

// one block of 16 -> its exact bf16 image (the arithmetic of k_quant_seg16 / k_quant_xa16: bit-identical images)
template <int DT>
__device__ __forceinline__ void quant_block16(const float (&v)[16], const QP& q, uint32_t (&w)[8]) {
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

// 16-byte chunk c of image row r (row pitch `pitch` bytes): chunks XOR-ed with 2 (r & 7), so that the rows one ds_read_b128
// lane group touches (8 rows x two neighbouring chunks) spread over the 16 chunk slots of a 256-byte bank row
__device__ __forceinline__ int img_off(int r, int c, int pitch) { return r * pitch + ((c ^ (2 * (r & 7))) << 4); }

// GROUP: more than one Linear in the launch (lqer_linear_forward_group) - a single Linear's instantiation reads its one table
// entry from static argument offsets only (the scan of the table's other entries costs its first weight request ~0.8 us:
// measured, rocprofv3 kernel durations 9.0 vs 8.1 us at M = 1).
template <int DT, int BOUT, bool GROUP>
// (two workgroups per CU: the grid is the N/16 consumers PLUS the producers - 272 for N = 4096 -, with one workgroup per CU the
// last 16 would wait for a whole round; 4 waves per SIMD caps the kernel at 128 registers.  Wider launches - a group of
// Linears, or M >= 6 where the x image leaves room for one workgroup per CU - give every consumer up to four column blocks, so
// that the whole launch is resident at once)
__global__ __launch_bounds__(64 * NW, 4) void k_decode1(Args<GROUP ? MAXMEM : 1> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const auto& g = a;  // (the shared formats and shapes: M, Kp, aout, bout)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int M = g.M, Kp = g.Kp, rp_all = a.rp_all;
  // The granule tag of THIS launch: the host's per-call counter, the dispatch id (x odd constant: consecutive launches of one
  // queue never share a tag; a replayed graph node gets a fresh one although its kernel arguments are frozen) and the queue's
  // address (two queues that take turns on one workspace).  Scalar: the same in every lane and workgroup of the launch.
  const uint32_t tag = (a.nonce + (uint32_t)lqer_dispatch_id() * 0x9E3779B1u) ^
                       ((uint32_t)((unsigned long long)__builtin_amdgcn_queue_ptr() >> 6) * 0x85EBCA6Bu);
  const int xpitch = (Kp * 2 + 255) / 256 * 256;  // (the chunk XOR stays inside a 256-byte group)
  // LDS: the producers' slab + reduction area ALIAS the consumers' x image - a consumer touches them (gather sums, the
  // fall-back's produce()) only behind the one barrier that follows its last weight stream, when the image is dead
  const int xs_bytes = M * xpitch;
  const int img_bytes = xs_bytes > PSLAB_BYTES + PRED_BYTES ? xs_bytes : PSLAB_BYTES + PRED_BYTES;
  unsigned char* const xs = smem;
  unsigned char* const pslab = smem;
  float* const pred = (float*)(pslab + PSLAB_BYTES);
  float* const red = (float*)(smem + img_bytes);
  const int red_bytes = a.cpw * RED1_BYTES;
  bf16_t* const xaq_l = (bf16_t*)(smem + img_bytes + red_bytes);
  volatile uint32_t* const miss_flag = (volatile uint32_t*)(smem + img_bytes + red_bytes + XAQ_BYTES);  // (its own word)
  if (tid == 0) *miss_flag = 0u;  // (ordered before its use by the barriers below)
  const auto gran_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.gran, 0, a.np * MAXM * rp_all * 8, 0x00020000);
  const int nt16_all = rp_all / 16;

  // ---- the partial tile of x A of slab p, published as granules (the producers' whole job; a consumer's fall-back)
  auto produce = [&](int p) {
    // Waves 0..3 multiply (2 k-slices of 32 each, accumulated in the MFMA in ascending k; total = ((w0 + w1) + w2) + w3): a
    // partial tile is a dependent chain - loads, quantize, MFMA, combine, publish - whose length the consumers wait for at
    // their very end, so it is kept short: three hand-overs through LDS instead of seven, only the live rank tiles.
    // Their A^T fragments come first: they do not depend on x, and their L2 / HBM latency then passes under the quantizer.
    // More than four rank tiles (a group's concatenated A): chunks of four against the same quantized slab.
    constexpr int PW = 4, PS = SLAB_K / 32 / PW;  // multiplying waves, k-slices of 32 per wave
    bf16x8 af[PS][MAXNT];
    const int64_t kw = (int64_t)p * SLAB_K + 32 * PS * wave;
    auto load_af = [&](int t0) {
      if (wave < PW) {
#pragma unroll
        for (int sl = 0; sl < PS; ++sl)
#pragma unroll
          for (int t = 0; t < MAXNT; ++t) {
            af[sl][t] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (t0 + t < nt16_all && kw + 32 * sl < Kp)
              af[sl][t] = *(const bf16x8*)(a.a_t + (int64_t)(16 * (t0 + t) + l15) * Kp + kw + 32 * sl + 8 * lq);
          }
      }
    };
    load_af(0);
    // quantize the slab: threads 0..127 take block (row t >> 4, segment t & 15); rows >= M and k >= K are zeros
    if (tid < MAXM * 16) {
      const int row = tid >> 4, seg = tid & 15;
      const int64_t k0 = (int64_t)p * SLAB_K + seg * 16;
      uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (row < M && k0 < a.K) {
        float v[16];
        load_block16<DT>(a.x, row * a.ldx + k0, v);
        quant_block16<DT>(v, a.qx, w);
      }
      if (tid == 0) D1_STAMP_AFTER(6, w[0]);
      *(uint4*)(pslab + img_off(row, 2 * seg, SLAB_K * 2)) = make_uint4(w[0], w[1], w[2], w[3]);
      *(uint4*)(pslab + img_off(row, 2 * seg + 1, SLAB_K * 2)) = make_uint4(w[4], w[5], w[6], w[7]);
    }
    __syncthreads();
    for (int t0 = 0; t0 < nt16_all; t0 += MAXNT) {
      if (t0 > 0) load_af(t0);
      const int nt16 = nt16_all - t0 < MAXNT ? nt16_all - t0 : MAXNT;  // live rank tiles of this chunk
      // A operand = tokens (row l15 & 7: rows 8..15 duplicate 0..7 and are never published), B operand = A^T (lane: rank
      // entry 16 t + l15, k + 8 lq)
      f32x4 acc[MAXNT];
#pragma unroll
      for (int t = 0; t < MAXNT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (wave < PW) {
#pragma unroll
        for (int sl = 0; sl < PS; ++sl)
          if (kw + 32 * sl < Kp) {
            const bf16x8 xf = *(const bf16x8*)(pslab + img_off(l15 & 7, 4 * (PS * wave + sl) + lq, SLAB_K * 2));
#pragma unroll
            for (int t = 0; t < MAXNT; ++t)
              if (t < nt16) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, af[sl][t], acc[t], 0, 0, 0);
          }
      }
      if (wave > 0 && wave < PW) {
#pragma unroll
        for (int t = 0; t < MAXNT; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (t < nt16) pred[(((wave - 1) * MAXNT + t) * 4 + j) * 64 + lane] = acc[t][j];
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int t = 0; t < MAXNT; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int row = 4 * lq + j;  // lane holds token rows 4 lq + j, rank entry 16 (t0 + t) + l15
            if (t < nt16 && row < M) {
              float sum = acc[t][j];
#pragma unroll
              for (int w2 = 0; w2 < PW - 1; ++w2) sum += pred[((w2 * MAXNT + t) * 4 + j) * 64 + lane];
              const u32x2_t gv = {__float_as_uint(sum), tag};
              __builtin_amdgcn_raw_buffer_store_b64(gv, gran_rsrc, (((p * MAXM + row) * rp_all) + 16 * (t0 + t) + l15) * 8, 0, 16);  // sc1
            }
          }
      }
      __syncthreads();  // pslab / pred may be reused (next chunk, fall-back loop)
    }
  };

  D1_STAMP(0);
  if ((int)blockIdx.x < a.np) {
    produce((int)blockIdx.x);
    D1_STAMP(5);
    return;
  }

  // ================================================== consumer =========================================================
  // which Linear of the launch this workgroup belongs to (consumer ids run member-major; scalar index into the argument table)
  // (what the weight stream's first request depends on - the member's packed image, its first workgroup and block count - is
  // selected from STATIC argument offsets: one batch of scalar loads with everything else; a table entry fetched by a computed
  // index would put a second, dependent scalar-memory round trip in front of the first weight panel.  The epilogue's fields come
  // from the indexed entry: their latency passes under the stream.)
  const int cb_all = (int)blockIdx.x - a.np;
  int mi = 0, m_cb0 = 0, m_nblk = a.mem[0].nblk;
  const uint8_t* m_wp = a.mem[0].wp;
  if constexpr (GROUP) {
#pragma unroll
    for (int i = 1; i < MAXMEM; ++i) {
      const bool in = cb_all >= a.mem[i].cb0;  // (unused entries: cb0 = INT_MAX)
      mi = in ? i : mi, m_cb0 = in ? a.mem[i].cb0 : m_cb0, m_nblk = in ? a.mem[i].nblk : m_nblk, m_wp = in ? a.mem[i].wp : m_wp;
    }
    mi = __builtin_amdgcn_readfirstlane(mi);
  }
  const Member& mb = a.mem[GROUP ? mi : 0];
  const int rp = mb.rp, r_off = mb.r_off;
  // this workgroup's column blocks of 16: cpw consecutive ones of its member (x is quantized into LDS once for all of them,
  // the member's x A is gathered once; the launch then fits the chip in ONE round whatever N and the group size are)
  const int cb_first = (cb_all - m_cb0) * a.cpw;
  const int nb_here = m_nblk - cb_first < a.cpw ? m_nblk - cb_first : a.cpw;
  const int row = l15, q = lq;  // weight row / token within the tile; k group (gemm_smallm.hip's names)
  const int nk = Kp / 64;
  const int codes_off = row * 32 + (q & 1) * 16;
  const int exps_off = 512 + row * 4;
  const bool hi = (q >> 1) != 0;
  const int sh0 = 8 * (q >> 1), sh1 = 16 + 8 * (q >> 1);
  const bool lowrank = mb.bt != nullptr && rp > 0;

  // weight panels through a buffer descriptor: a request past the end is dropped by the range check,
  // so no load sits under a branch and the compiler's vmcnt counts stay exact
  auto block_rsrc = [&](int cb) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(m_wp + (int64_t)cb * nk * LQER_PANEL_BYTES), 0, nk * LQER_PANEL_BYTES, 0x00020000);
  };
  auto load_panel = [&](const auto& w_rsrc, int kt, SmPanel& p) {  // kt >= nk: zeros
    // (the panel's base as the SCALAR offset - part of the range check on gfx9 -, the lane's place inside a panel as the one
    // vector offset: no per-panel address registers to keep alive across column blocks)
    const int base = __builtin_amdgcn_readfirstlane(kt < nk ? kt * LQER_PANEL_BYTES : 0x7ffffff0);
#ifndef LQER_D1_WAUX
#define LQER_D1_WAUX 2  // cache policy of the weight stream's loads: 0 default, 2 non-temporal (once-read bytes: a model walks 3.6 GB
                        // of packed weights per token - 9.4 -> 9.0 us per forward over 48 rotating weights, bench d1; a single resident
                        // weight re-read from the Infinity Cache is slower with it, 4.8 -> 5.5 us: not what a model does)
#endif
    p.cw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, codes_off, base, LQER_D1_WAUX));
    p.ex = __builtin_amdgcn_raw_buffer_load_b32(w_rsrc, exps_off, base, LQER_D1_WAUX);
  };
  // wave w takes panels w, w + 8, ...; two register buffers of 4 panels
  constexpr int UNR = 4;
  SmPanel pa[UNR], pb[UNR];
  const int per_wave = (nk - wave + NW - 1) / NW;
  auto kt_of = [&](int i) { return wave + NW * i; };
  // this thread's first XB blocks of x are requested BEFORE the weight panels (loads complete in order: the small L2-hot
  // reads must not queue behind the weight stream) and all together - one round trip for up to 4 x 512 blocks (M = 8, K = 4096)
  // instead of one per block.  Always loaded, from a clamped in-range offset: a load under a branch would make the compiler
  // wait for everything in flight at its first use; threads without such a block ignore the values.
  const int segs = Kp / 16;
  constexpr int XB = DT == LQER_F32 ? 2 : 4;  // (8 registers per 16-bit block, 16 per fp32 block)
  RawBlk<DT> rb[XB];
  auto request_x = [&](int b0) {
#pragma unroll
    for (int u = 0; u < XB; ++u) {
      const int b = b0 + 64 * NW * u;
      const int r = b / segs, seg = b - r * segs;
      load_raw16<DT>(a.x, b < M * segs && seg * 16 < a.K ? r * a.ldx + seg * 16 : 0, rb[u]);
    }
  };
  request_x(tid);
  asm volatile("" ::: "memory");  // (keeps the requests HERE: the compiler otherwise sinks them to their use, behind the panels)
  {
    const auto w_first = block_rsrc(cb_first);
#pragma unroll
    for (int u = 0; u < UNR; ++u) load_panel(w_first, kt_of(u), pa[u]);
  }
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  // ---- the activation image of ALL of x in LDS: block b = (row b / (Kp/16), segment b % (Kp/16))
  for (int b0 = tid; b0 < M * segs; b0 += 64 * NW * XB) {
    if (b0 != tid) request_x(b0);
#pragma unroll
    for (int u = 0; u < XB; ++u) {
      const int b = b0 + 64 * NW * u;
      if (b < M * segs) {
        const int r = b / segs, seg = b - r * segs;
        uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (seg * 16 < a.K) {
          float v[16];
          raw16_to_f32<DT>(rb[u], v);
          quant_block16<DT>(v, a.qx, w);
        }
        if (u == 0 && tid == 0) D1_STAMP_AFTER(6, w[0]);
        *(uint4*)(xs + img_off(r, 2 * seg, xpitch)) = make_uint4(w[0], w[1], w[2], w[3]);
        *(uint4*)(xs + img_off(r, 2 * seg + 1, xpitch)) = make_uint4(w[4], w[5], w[6], w[7]);
      }
    }
  }
  __syncthreads();
  D1_STAMP(1);

  // ---- x A from the producers' granules: one thread per 4 rank entries of a token, slabs summed in a fixed order.
  // The 512 threads read the granules as 4 groups x 128 items: group pg takes the slabs pg, pg + 4, pg + 8, ... (4 requested
  // together: one memory round trip per batch), the groups' sums are added in the order 0..3 through LDS.  The FIRST batch
  // is requested at the end of the first column block's weight stream: loads return in order, so the granules are there when
  // the cross-wave combine is done (their ~1.3 us round trip past this CU's caches is not paid after it).
  const int items = lowrank ? M * rp / 4 : 0;  // <= 128
  const int rq = lowrank ? rp / 4 : 1;
  const int pg = tid >> 7, it = tid & 127;
  const int g_r = it / rq, g_c4 = it - g_r * rq;
  struct GBatch {
    u32x4_t v0[4], v1[4];
  };
  auto gather_issue = [&](int p0, GBatch& b) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = p0 + 4 * u;
      const uint32_t off = (it < items && p < a.np) ? (uint32_t)((((p * MAXM + g_r) * rp_all) + r_off + 4 * g_c4) * 8) : 0x7ffffff0u;  // (past the range: zeros)
      b.v0[u] = __builtin_amdgcn_raw_buffer_load_b128(gran_rsrc, (int)off, 0, 16);       // sc1: past this CU's L1
      b.v1[u] = __builtin_amdgcn_raw_buffer_load_b128(gran_rsrc, (int)(off + 16u), 0, 16);
    }
  };
  // ---- one column block's weight stream (gemm_smallm.hip's main loop with the activation fragments from LDS; token rows >= M
  // read row 0: their output columns are never stored), then (prefetch_c) the request for the NEXT block's first panels: they
  // travel under the combine and wave 0's epilogue (unconditional - past the workgroup's blocks the descriptor's range drops
  // them: no load under a branch)
  auto request_first_panels = [&](int jb) {  // of column block jb (past the workgroup's blocks: an empty range, no traffic)
    const auto w_blk = __builtin_amdgcn_make_buffer_rsrc((void*)(m_wp + (int64_t)(cb_first + jb) * nk * LQER_PANEL_BYTES), 0,
                                                         jb < nb_here ? nk * LQER_PANEL_BYTES : 0, 0x00020000);
#pragma unroll
    for (int u = 0; u < UNR; ++u) load_panel(w_blk, kt_of(u), pa[u]);
  };
  auto stream_block = [&](int jb, auto prefetch_c) -> f32x4 {
    const int cb = cb_first + jb;
    const auto w_rsrc = block_rsrc(cb);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int lrow = l15;
    asm volatile("" : "+v"(lrow));  // (re-derived per block: kept alive from the top of the kernel it costs a spill)
    const int xrow = lrow < M ? lrow : 0;
    auto compute_panel = [&](int kt, const SmPanel& p) {
      const uint32_t w0 = hi ? p.cw[1] : p.cw[0], w1 = hi ? p.cw[3] : p.cw[2];
      const bf16x8 wb0 = expand_frag(w0, ((p.ex >> sh0) & 0xffu) << 23);
      const bf16x8 wb1 = expand_frag(w1, ((p.ex >> sh1) & 0xffu) << 23);
      const int kc = kt < nk ? kt : 0;  // (past the end: zero weights, any finite activation chunk)
      const bf16x8 x0 = *(const bf16x8*)(xs + img_off(xrow, kc * 8 + q, xpitch));
      const bf16x8 x1 = *(const bf16x8*)(xs + img_off(xrow, kc * 8 + 4 + q, xpitch));
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb0, x0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb1, x1, acc, 0, 0, 0);
    };
    // (no tests around loads or MFMAs: a panel past the end is zeros times the image's first chunk)
    for (int i = 0; i < per_wave; i += 2 * UNR) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) load_panel(w_rsrc, kt_of(i + UNR + u), pb[u]);
#pragma unroll
      for (int u = 0; u < UNR; ++u) compute_panel(kt_of(i + u), pa[u]);
#pragma unroll
      for (int u = 0; u < UNR; ++u) load_panel(w_rsrc, kt_of(i + 2 * UNR + u), pa[u]);
#pragma unroll
      for (int u = 0; u < UNR; ++u) compute_panel(kt_of(i + UNR + u), pb[u]);
    }
    if constexpr (decltype(prefetch_c)::value) request_first_panels(jb + 1);
    return acc;
  };
  // the waves' partial sums of a column block, parked in the block's own LDS buffer; its epilogue wave adds them in the fixed
  // order (((w0 + w1) + w2) + ...) + w7
  auto park = [&](int jb, const f32x4& acc) {
    float* const redj = red + jb * (RED1_BYTES / 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) redj[(wave * 4 + j) * 64 + lane] = acc[j];
  };
  auto combine_sum = [&](int jb) -> f32x4 {
    const float* const redj = red + jb * (RED1_BYTES / 4);
    f32x4 acc;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t = redj[j * 64 + lane];
#pragma unroll
      for (int w2 = 1; w2 < NW; ++w2) t += redj[(w2 * 4 + j) * 64 + lane];
      acc[j] = t;
    }
    return acc;
  };
  // wave 0's operands of the epilogue that do not depend on x A: requested right behind the combine barrier
  auto side_b_first = [&](int n0) -> bf16x8 {
    bf16x8 sp_b = zero8;
    if (lowrank && mb.b_limbs > 0 && 8 * q < rp) sp_b = *(const bf16x8*)(mb.bt + (int64_t)(n0 + row) * rp + 8 * q);
    return sp_b;
  };
  auto bias_of = [&](int n0) -> f32x4 {
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};  // (the bias image is padded to Np: columns n0 + 4 q .. + 3 exist)
    if (mb.bias) {
#pragma unroll
      for (int j = 0; j < 4; ++j) bias4[j] = mb.bias[n0 + 4 * q + j];
    }
    return bias4;
  };
  // ---- side path + bias + store (wave 0, gemm_smallm.hip's epilogue): lane = token `row`, output columns n0 + 4 q + j
  auto epilogue = [&](int n0, const f32x4& acc, const bf16x8& sp_b, const f32x4& bias4) {
    const int nq = n0 + 4 * q;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (lowrank) {
      bf16x8 sp_x = zero8;
      if (8 * q < rp && row < M) sp_x = *(const bf16x8*)(xaq_l + row * rp + 8 * q);
      if (mb.b_limbs > 0) s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sp_b, sp_x, s, 0, 0, 0);  // prefetched
      for (int l = 0; l < mb.b_limbs; ++l)
        for (int ks = (l == 0 ? 1 : 0); ks * 32 < rp; ++ks) {
          const int j0 = ks * 32 + 8 * q;
          bf16x8 bb = zero8, xv = zero8;
          if (j0 < rp) {
            bb = *(const bf16x8*)(mb.bt + ((int64_t)l * mb.Np + n0 + row) * rp + j0);
            if (row < M) xv = *(const bf16x8*)(xaq_l + row * rp + j0);
          }
          s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb, xv, s, 0, 0, 0);
        }
      D1_STAMP_AFTER(9, s[0]);
      if constexpr (BOUT == 1) {
        float amax = fmaxf(fmaxf(fabsf(s[0]), fabsf(s[1])), fmaxf(fabsf(s[2]), fabsf(s[3])));
        {  // max over lanes l, l ^ 16, l ^ 32 without the LDS crossbar: v_permlane16_swap / v_permlane32_swap
          auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(amax), __float_as_uint(amax), false, false);
          amax = fmaxf(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
          auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(amax), __float_as_uint(amax), false, false);
          amax = fmaxf(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
        }
        const int e = block_exponent(amax, g.bout);
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] = fabsf(s[j]) <= 1e-8f ? s[j] : mxint_value(s[j], e, g.bout);
      }
    }
    D1_STAMP_AFTER(10, s[0]);
    if (row < M) {
      float out[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) out[j] = (s[j] + bias4[j]) + acc[j];
      store_row4<DT>(mb.y, (int64_t)row * mb.ldy + nq, nq, mb.N, out);
    }
  };

  // ================= every column block's weight stream first (the next block's first panels are requested as the current
  // stream ends), the partial sums parked per block; then - ONE barrier - the member's x A (gathered and re-quantized once);
  // then the epilogues, block b by wave b, side by side
  for (int jb = 0; jb < nb_here; ++jb) {  // (behind the last block the request has an empty range: eight loads without traffic)
    const f32x4 acc = stream_block(jb, std::true_type{});
    park(jb, acc);
  }
  GBatch gb;
  gather_issue(lowrank ? pg : (1 << 20), gb);  // at the end of the weight streams: under the barrier and the epilogue waves' first loads
  D1_STAMP(2);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (not __syncthreads(): that would wait for the loads in flight)
  D1_STAMP(3);
  // the epilogue waves request their operands that do not depend on x A now: their latency passes under the granule round trip
  const bool ep_wave = wave < nb_here;
  const int n0 = (cb_first + wave) * 16;
  bf16x8 sp_b = zero8;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (ep_wave) {
    sp_b = side_b_first(n0);
    bias4 = bias_of(n0);
  }
  if (lowrank) {
    // prefetched: the first batch is already in flight (gb)
    auto gather = [&](int sweeps, bool& complete, bool prefetched) -> float4 {
      float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
      complete = true;
      if (it < items) {
        for (int p0 = pg; p0 < a.np; p0 += 16) {
          if (!(prefetched && p0 == pg)) gather_issue(p0, gb);
          auto tagged = [&](const u32x4_t& x0, const u32x4_t& x1) {
            return x0[1] == tag && x0[3] == tag && x1[1] == tag && x1[3] == tag;
          };
          bool ok[4], all_ok = true;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            ok[u] = p0 + 4 * u >= a.np || tagged(gb.v0[u], gb.v1[u]);
            all_ok = all_ok && ok[u];
          }
          for (int tries = 0; tries < sweeps && !all_ok; ++tries) {  // not all there yet: poll the missing slabs TOGETHER
            __builtin_amdgcn_s_sleep(8);
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (!ok[u]) {
                const int off = ((((p0 + 4 * u) * MAXM + g_r) * rp_all) + r_off + 4 * g_c4) * 8;
                gb.v0[u] = __builtin_amdgcn_raw_buffer_load_b128(gran_rsrc, off, 0, 16);
                gb.v1[u] = __builtin_amdgcn_raw_buffer_load_b128(gran_rsrc, off + 16, 0, 16);
              }
            all_ok = true;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              if (!ok[u]) ok[u] = tagged(gb.v0[u], gb.v1[u]);
              all_ok = all_ok && ok[u];
            }
          }
          if (!all_ok) complete = false;
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (p0 + 4 * u < a.np) {  // ascending slabs within the group
              sum.x += __uint_as_float(gb.v0[u][0]), sum.y += __uint_as_float(gb.v0[u][2]);
              sum.z += __uint_as_float(gb.v1[u][0]), sum.w += __uint_as_float(gb.v1[u][2]);
            }
        }
      }
      float4* const gs = (float4*)pred;  // [4][128]
      gs[tid] = sum;
      if (!complete) *miss_flag = 1u;  // the workgroup's vote rides on the barrier of the group sums
      __syncthreads();
      complete = *miss_flag == 0u;
      if (tid < 128) {
        const float4 s1 = gs[128 + tid], s2 = gs[256 + tid], s3 = gs[384 + tid];
        sum.x = ((sum.x + s1.x) + s2.x) + s3.x, sum.y = ((sum.y + s1.y) + s2.y) + s3.y;
        sum.z = ((sum.z + s1.z) + s2.z) + s3.z, sum.w = ((sum.w + s1.w) + s2.w) + s3.w;
      }
      return sum;  // (threads 0 .. items-1 hold the totals)
    };
    bool complete;
    float4 s = gather(a.spin, complete, true);
    D1_STAMP(4);
    if (!complete) {  // (workgroup-uniform: read back from LDS behind the barrier)
      // a producer has not been seen: compute every partial tile here (same routine, same bits), then read them back
      for (int p = 0; p < a.np; ++p) produce(p);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (tid == 0) *miss_flag = 0u;  // the vote of the second gather starts clean (its first use is behind the barrier)
      __syncthreads();
      s = gather(64, complete, false);  // (written by this very workgroup: there after its own drain + barrier)
      if (!complete) __builtin_trap();  // its own write-through stores not visible after vmcnt(0): never sum untagged bytes
    }
    D1_STAMP(7);
    // A_out in blocks of 16 = 4 consecutive threads
    float amax = fmaxf(fmaxf(fabsf(s.x), fabsf(s.y)), fmaxf(fabsf(s.z), fabsf(s.w)));
    amax = fmaxf(amax, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(amax), 0xB1, 0xf, 0xf, true)));  // quad_perm [1,0,3,2]
    amax = fmaxf(amax, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(amax), 0x4E, 0xf, 0xf, true)));  // quad_perm [2,3,0,1]
    if (tid < items) {
      const bool any = amax > 0.f;
      const int e = any ? block_exponent(amax, g.aout) : 0;
      const float v[4] = {s.x, s.y, s.z, s.w};
      uint32_t w[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float q0 = any && fabsf(v[2 * i]) > g.aout.tiny ? mxint_value(v[2 * i], e, g.aout) : 0.f;
        const float q1 = any && fabsf(v[2 * i + 1]) > g.aout.tiny ? mxint_value(v[2 * i + 1], e, g.aout) : 0.f;
        w[i] = exact_bf16_bits(q0) | (exact_bf16_bits(q1) << 16);
      }
      *(uint2*)(xaq_l + tid * 4) = make_uint2(w[0], w[1]);
    }
    __syncthreads();
  }
  D1_STAMP(8);
  if (ep_wave) {
    const f32x4 acc = combine_sum(wave);
    epilogue(n0, acc, sp_b, bias4);
  }
  D1_STAMP(5);
}

static std::atomic<uint32_t> g_nonce{1};  // the call's granule tag: any value the granule area does not hold yet

}  // namespace d1

#ifdef LQER_D1_STAMPS
extern "C" int lqer_debug_set_d1_stamps(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(d1::g_d1_stamps), &p, sizeof(p)); }
#endif

size_t decode1_lds_bytes(int M, int64_t Kp, int cpw) {  // cpw: column blocks per consumer workgroup (one 8-KiB sum buffer each)
  const size_t xs = (size_t)M * ((Kp * 2 + 255) / 256 * 256), prod = d1::PSLAB_BYTES + d1::PRED_BYTES;  // (aliased: see the kernel)
  return (xs > prod ? xs : prod) + (size_t)cpw * d1::RED1_BYTES + d1::XAQ_BYTES + d1::FLAG_BYTES;
}

size_t decode1_scratch_bytes(int64_t Kp, int rp) { return (size_t)((Kp + d1::SLAB_K - 1) / d1::SLAB_K) * d1::MAXM * rp * 8; }

// g: the shared part, filled like for the small-M kernel (M, Kp, aout, bout, tuning; xq / xaq / xa_part and the per-Linear fields
// unused) - mem[0 .. nmem-1]: the Linears of the launch (one, or a group that is handed the same tokens), a_t: the
// concatenation of their A^T images along the rank.  Returns LQER_E_UNSUPPORTED when the shape is outside.
int decode1_dispatch(GemmArgs g, int dtype, const void* x, int64_t ldx, int K, const QP& qx, const bf16_t* a_t, int bout,
                     const DecodeMember* mem, int nmem, void* scratch, size_t scratch_bytes, hipStream_t st) {
#ifdef LQER_HOST_TIMING
  timespec ht0; clock_gettime(CLOCK_MONOTONIC, &ht0);
#endif
  if (g.M < 1 || g.M > d1::MAXM || nmem < 1 || nmem > d1::MAXMEM || bout > 1 || K % 16 != 0) return LQER_E_UNSUPPORTED;
  d1::Args<d1::MAXMEM> a;
  int rp_all = 0, blocks = 0;
  for (int i = 0; i < nmem; ++i) {
    if (mem[i].rp <= 0 || mem[i].rp > 16 * d1::MAXNT || mem[i].rp % 16 || mem[i].Np % 16 || !mem[i].bt || mem[i].b_limbs < 1 ||
        mem[i].b_limbs > 3)
      return LQER_E_UNSUPPORTED;
    rp_all += mem[i].rp;
    blocks += mem[i].Np / 16;
  }
  if (rp_all > 128) return LQER_E_UNSUPPORTED;
  if (scratch_bytes < decode1_scratch_bytes(g.Kp, rp_all) || ((uintptr_t)scratch & 15)) return LQER_E_UNSUPPORTED;
  a.np = (int)((g.Kp + d1::SLAB_K - 1) / d1::SLAB_K);
  // ONE round: at most what the chip holds at once (two workgroups per CU by registers, one when a workgroup's LDS - the x image
  // above all - takes more than half of the CU's); a consumer then walks `cpw` consecutive column blocks of its member.  The
  // smallest cpw whose own LDS footprint still leaves that many workgroups resident:
  constexpr int CUS = 256;
  size_t lds = 0;
  a.cpw = 0;
  for (int c = 1; c <= d1::MAXCPW; ++c) {
    lds = decode1_lds_bytes(g.M, g.Kp, c);
    if (lds > 150 * 1024) break;
    const int slots = CUS * (2 * lds <= 160 * 1024 ? 2 : 1) - a.np;
    if (slots > 0 && (int64_t)slots * c >= blocks) {
      a.cpw = c;
      break;
    }
  }
  if (a.cpw == 0) return LQER_E_UNSUPPORTED;  // (K too long for the LDS image, or more than 4 x 16 columns per resident workgroup)
  int cbs = 0, roff = 0;
  for (int i = 0; i < nmem; ++i) {
    const int nblk = mem[i].Np / 16;
    a.mem[i] = d1::Member{mem[i].wp, mem[i].bt, mem[i].bias, mem[i].y, mem[i].ldy, mem[i].N, mem[i].Np, mem[i].rp, mem[i].b_limbs, roff, cbs, nblk};
    roff += mem[i].rp;
    cbs += (nblk + a.cpw - 1) / a.cpw;
  }
  for (int i = nmem; i < d1::MAXMEM; ++i) a.mem[i] = d1::Member{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0x7fffffff, 0};
  a.M = g.M, a.Kp = g.Kp, a.aout = g.aout, a.bout = g.bout;
  a.nmem = nmem, a.rp_all = rp_all;
  a.x = x, a.ldx = ldx, a.K = K, a.qx = qx, a.a_t = a_t;
  a.gran = (uint32_t*)scratch;
  // the call's tag: a counter spread over all 32 bits (odd multiplier: a bijection, so two calls never share a tag before
  // 2^32 calls) - small integers, zeros and the bit patterns of ordinary floats are what stale workspace bytes look like
  const uint32_t n = d1::g_nonce.fetch_add(1, std::memory_order_relaxed);
  a.nonce = n * 0x9E3779B1u ^ 0xA5C35A3Cu;
  a.spin = (g.tuning & LQER_TUNE_DECODE_NO_POLL) ? 0 : LQER_QD1_SPIN;  // (tests: every consumer computes the tiles itself)
  const unsigned grid = (unsigned)(a.np + cbs);
  d1::Args<1> a1;  // the single Linear's compact block
  a1.mem[0] = a.mem[0];
  a1.M = a.M, a1.Kp = a.Kp, a1.aout = a.aout, a1.bout = a.bout, a1.nmem = 1, a1.rp_all = a.rp_all, a1.x = a.x, a1.ldx = a.ldx, a1.K = a.K,
  a1.qx = a.qx, a1.a_t = a.a_t, a1.gran = a.gran, a1.nonce = a.nonce, a1.np = a.np, a1.spin = a.spin, a1.cpw = a.cpw;
#ifdef LQER_HOST_TIMING  // diagnostic build: host nanoseconds before / inside the launch call, printed at exit
  struct HostT {
    double pre = 0, launch = 0; long n = 0;
    ~HostT() { if (n) fprintf(stderr, "[decode1 host] calls %ld: dispatch before launch %.0f ns, launch call %.0f ns\n", n, pre / n, launch / n); }
  };
  static HostT host_t;
  timespec ht1; clock_gettime(CLOCK_MONOTONIC, &ht1);
#endif
#define D1_LAUNCH(DT, BO)                                                                     \
  do {                                                                                        \
    if (nmem > 1) {                                                                           \
      static LdsLimitOnce lds_once;                                                           \
      lds_once.set((const void*)d1::k_decode1<DT, BO, true>, 150 * 1024);                     \
      d1::k_decode1<DT, BO, true><<<grid, 64 * d1::NW, lds, st>>>(a);                         \
    } else {                                                                                  \
      static LdsLimitOnce lds_once;                                                           \
      lds_once.set((const void*)d1::k_decode1<DT, BO, false>, 150 * 1024);                    \
      d1::k_decode1<DT, BO, false><<<grid, 64 * d1::NW, lds, st>>>(a1);                       \
    }                                                                                         \
  } while (0)
  switch (dtype) {
    case LQER_F32: if (bout == 1) D1_LAUNCH(LQER_F32, 1); else D1_LAUNCH(LQER_F32, 0); break;
    case LQER_F16: if (bout == 1) D1_LAUNCH(LQER_F16, 1); else D1_LAUNCH(LQER_F16, 0); break;
    case LQER_BF16: if (bout == 1) D1_LAUNCH(LQER_BF16, 1); else D1_LAUNCH(LQER_BF16, 0); break;
    default: set_error("unknown dtype %d", dtype); return LQER_E_INVALID;
  }
#undef D1_LAUNCH
#ifdef LQER_HOST_TIMING
  {
    timespec ht2; clock_gettime(CLOCK_MONOTONIC, &ht2);
    host_t.pre += (ht1.tv_sec - ht0.tv_sec) * 1e9 + (ht1.tv_nsec - ht0.tv_nsec);
    host_t.launch += (ht2.tv_sec - ht1.tv_sec) * 1e9 + (ht2.tv_nsec - ht1.tv_nsec);
    host_t.n++;
  }
#endif
  return check_launch("lqer_decode1");
}

}  // namespace lqer
