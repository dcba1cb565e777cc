// Feed + compute mock of an e3m2 ("fp6 signed-digit") main loop at the C2 geometry (M 2048, K 4096, N 4096), the gate for
// building that route (DESIGN.md §4 "fp6 block-scaled MFMA", NOTEBOOK.md §8.4): does the operand feed - 25.6 KB of LDS-DMA per 64-k step of a
// 128 x 256 tile, twice the rate of the bf16 kernel - keep 4x-rate scaled MFMAs busy?  Everything of the real loop that
// costs time is here (global -> LDS ring by LDS-DMA with the real footprints and L2 sharing, fragment reads, scaled MFMAs,
// one barrier per step); the data are random bits, there is no prologue / epilogue and nothing is checked.
//
// Workgroup = one 128(m) x 256(n) tile, 12 waves: waves 0-7 consume (2 x 4, wave tile 64 x 64 = 2 token tiles x 2 digits x 2
// weight tiles: 8 v_mfma_scale_f32_32x32x64_f8f6f4 per step), waves 8-11 only issue LDS-DMA (an issuing wave stalls
// 60-180 cycles per 1-KiB piece: kept off the MFMA waves).  Ring of 4 slots; a step = 2 pairs of 32 k:
//   x chunk 12800 B = [pair][row group of 32][digit: 512 B plane of 16 B per row + 256 B plane of 8 B per row][64 B scales]
//   W chunk 12800 B = [pair][4096 B plane16 | 2048 B plane8 | 256 B scales]
// The two consumer waves of a SIMD run half a step apart (LOAD: fragment reads; COMPUTE: 8 MFMAs), the shipped kernels' ping-pong
// without the LDS-DMA in the LOAD section.
// Variants: MODE 0 full; 1 no MFMA (feed only); 2 no DMA (compute + LDS reads only).
// build: hipcc --offload-arch=gfx950 -O3 -o mx6_gemm_mock mx6_gemm_mock.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) int i8v;
typedef __attribute__((ext_vector_type(4))) uint32_t u4;
typedef __attribute__((ext_vector_type(2))) uint32_t u2;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((address_space(3))) void lds_void;

constexpr int CHUNK = 12800;           // bytes of x (and of W) per step
constexpr int XREG = 13 * 1024;        // LDS bytes reserved per operand and slot (13 pieces)
constexpr int SLOT = 2 * XREG;         // 26 KiB
#ifndef MOCK_NSLOT
#define MOCK_NSLOT 6
#endif
constexpr int NSLOT = MOCK_NSLOT;        // ring slots; the producers run NSLOT - 1 steps ahead
constexpr int AHEAD = NSLOT - 1;
constexpr int LDS_BYTES = NSLOT * SLOT;  // 104 KiB (4 slots) / 156 KiB (6)
constexpr int PIECES = 26;               // per step: 13 x + 13 W (the 13th of each is half used)

template <int MODE>
__global__ __launch_bounds__(768) void k_mock(const uint8_t* __restrict__ x6, const uint8_t* __restrict__ w6, int nk, int tiles_n,
                                              float* __restrict__ sink, int xcd_bm) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = gridDim.x;
  int tile;
  {
    const int b = blockIdx.x, xcd = b & 7, q8 = nt >> 3, r8 = nt & 7;
    tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
  }
  int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  if (xcd_bm > 0) {
    // XCD-local tile BLOCKS: the XCD's nt / 8 tiles form a block of xcd_bm token tiles x (nt / 8 / xcd_bm) weight tiles (instead of
    // whole rows of weight tiles), so that the bytes the XCD pulls through its L2 are W / (xcds along n) + x / (xcds along m)
    const int b = blockIdx.x, xcd = b & 7, l = b >> 3, per = nt >> 3;
    const int bn = per / xcd_bm;                 // weight tiles per block
    const int gm = (nt / tiles_n) / xcd_bm;      // XCD grid along m
    const int xm = xcd % gm, xn = xcd / gm;
    tm = xm * xcd_bm + (l % xcd_bm), tn = xn * bn + l / xcd_bm;
  }
  const uint8_t* const xb = x6 + (size_t)tm * nk * CHUNK;
  const uint8_t* const wb = w6 + (size_t)tn * nk * CHUNK;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem;
  for (int i = tid; i < LDS_BYTES / 4; i += 768) ((uint32_t*)smem)[i] = 0x7b7b7b7bu;
  __syncthreads();

  if (wave >= 8) {
    // ---- producers: piece p of a step = x piece p (p < 13) or W piece p - 13; producer q issues pieces q, q + 4, ...
    const int q = wave - 8;
    const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, nk * CHUNK, 0x00020000);
    const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, nk * CHUNK, 0x00020000);
    auto issue = [&](int kt) {
      if (MODE == 2) return;
      const int slot = kt % NSLOT;
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const int p = q + 4 * j;
        if (p >= PIECES) break;
        const bool isx = p < 13;
        const int pp = isx ? p : p - 13;
        // (the 13th piece covers bytes 12288 .. 13311 of a 12800-byte chunk: its upper half belongs to the next step - harmless
        // here, it lands in the slot's padding)
        if (isx)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void*)(smem + slot * SLOT + pp * 1024), 16, lane * 16 + pp * 1024,
                                                   kt * CHUNK, 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(smem + slot * SLOT + XREG + pp * 1024), 16,
                                                   lane * 16 + pp * 1024, kt * CHUNK, 0, 0);
      }
    };
    for (int d = 0; d < AHEAD; ++d) issue(d);
    // step 0 landed before the consumers' first reads: AHEAD - 1 batches (7 or 6 loads each) may stay in flight
    auto wait_landed = [&]() {
      if (q < 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(7 * (AHEAD - 1) > 63 ? 63 : 7 * (AHEAD - 1)) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * (AHEAD - 1) > 63 ? 63 : 6 * (AHEAD - 1)) : "memory");
    };
    wait_landed();
    asm volatile("s_barrier" ::: "memory");
    for (int kt = 0; kt < nk; ++kt) {
      issue(kt + AHEAD);  // (past the end of K: dropped by the buffer range) - its slot held step kt-1, last read before this step's start
      wait_landed();      // step kt+1 landed (read after this step's second barrier); the AHEAD - 1 younger batches stay in flight
      asm volatile("s_barrier" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
    }
    asm volatile("s_barrier" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // ---- consumers
  const int wm = wave >> 2, wn = wave & 3;  // 2 x 4
  const int r = lane & 31, h = lane >> 5;   // row of the 32-tile, pair of the step
  // fragment addresses inside a slot: x (row group g = 2 wm + i, digit d): pair h * 6400 + g * 1600 + d * 768 (+ 512: 8-byte plane),
  // scales at + 1536 + 2 r; W (32-row tile j of this wave's 64 columns): XREG + h * 6400 + row * 16, + 4096 + row * 8, + 6144 + row
  uint32_t xa16[2], xa8[2], xsc[2], wa16[2], wa8[2], wsc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int g = 2 * wm + i;
    xa16[i] = lds0 + h * 6400 + g * 1600 + r * 16;
    xa8[i] = lds0 + h * 6400 + g * 1600 + 512 + r * 8;
    xsc[i] = lds0 + h * 6400 + g * 1600 + 1536 + r * 2;
    const int row = wn * 64 + i * 32 + r;
    wa16[i] = lds0 + XREG + h * 6400 + row * 16;
    wa8[i] = lds0 + XREG + h * 6400 + 4096 + row * 8;
    wsc[i] = lds0 + XREG + h * 6400 + 6144 + row;
  }
  f16v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // LOAD / COMPUTE ping-pong of the two consumer waves of a SIMD (waves w and w + 4), half a step apart, two barriers per step
  // (every wave of the workgroup, producers included, passes 2 nk + 2 barriers)
  const bool late = wave >= 4;
  asm volatile("s_barrier" ::: "memory");  // step 0 has landed
  if (late) asm volatile("s_barrier" ::: "memory");
  for (int kt = 0; kt < nk; ++kt) {
    __builtin_amdgcn_s_setprio(1);
    // (plain locals: a struct handed to lambdas by reference ends up in scratch memory)
    u4 x16[2][2], w16[2];
    u2 x8[2][2], w8[2];
    uint32_t xs[2], ws[2];
    const uint32_t so = (uint32_t)((kt % NSLOT) * SLOT);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int d = 0; d < 2; ++d)
        asm volatile("ds_read_b128 %0, %2 offset:%4\n\tds_read_b64 %1, %3 offset:%4"
                     : "=&v"(x16[i][d]), "=&v"(x8[i][d])
                     : "v"(xa16[i] + so), "v"(xa8[i] + so), "i"(d * 768));
      asm volatile("ds_read_u16 %0, %1" : "=v"(xs[i]) : "v"(xsc[i] + so));
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b64 %1, %3" : "=&v"(w16[i]), "=&v"(w8[i]) : "v"(wa16[i] + so), "v"(wa8[i] + so));
      asm volatile("ds_read_u8 %0, %1" : "=v"(ws[i]) : "v"(wsc[i] + so));
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier"
                 : "+v"(x16[0][0]), "+v"(x16[0][1]), "+v"(x16[1][0]), "+v"(x16[1][1]), "+v"(x8[0][0]), "+v"(x8[0][1]), "+v"(x8[1][0]),
                   "+v"(x8[1][1]), "+v"(w16[0]), "+v"(w16[1]), "+v"(w8[0]), "+v"(w8[1]), "+v"(xs[0]), "+v"(xs[1]), "+v"(ws[0]), "+v"(ws[1])
                 :: "memory");
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 1) {
      asm volatile("" ::"v"(x16[0][0]), "v"(x16[1][1]), "v"(x8[0][0]), "v"(w16[0]), "v"(w8[1]), "v"(xs[0]), "v"(ws[1]));
    } else {
      i8v wf[2], xf[2][2];
#pragma unroll
      for (int j = 0; j < 2; ++j) wf[j] = (i8v){(int)w16[j][0], (int)w16[j][1], (int)w16[j][2], (int)w16[j][3], (int)w8[j][0], (int)w8[j][1], 0, 0};
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int d = 0; d < 2; ++d)
          xf[i][d] = (i8v){(int)x16[i][d][0], (int)x16[i][d][1], (int)x16[i][d][2], (int)x16[i][d][3], (int)x8[i][d][0], (int)x8[i][d][1], 0, 0};
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            if (d == 0) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[j], xf[i][0], acc[i][j], 3, 3, 0, (int)ws[j], 0, (int)xs[i]);
            else acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[j], xf[i][1], acc[i][j], 3, 3, 0, (int)ws[j], 1, (int)xs[i]);
          }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
  }
  if (!late) asm volatile("s_barrier" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  sink[(size_t)blockIdx.x * 512 + tid] = s;
}

template <int MODE>
static float run(const uint8_t* x6, const uint8_t* w6, int nk, int tiles_m, int tiles_n, float* sink, int reps, int xcd_bm) {
  hipFuncSetAttribute((const void*)k_mock<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) k_mock<MODE><<<tiles_m * tiles_n, 768, LDS_BYTES>>>(x6, w6, nk, tiles_n, sink, xcd_bm);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) k_mock<MODE><<<tiles_m * tiles_n, 768, LDS_BYTES>>>(x6, w6, nk, tiles_n, sink, xcd_bm);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
  const int M = 2048, K = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 4096;
  const int tiles_m = M / 128, tiles_n = N / 256, nk = K / 64;
  const size_t xbytes = (size_t)tiles_m * nk * CHUNK + 4096, wbytes = (size_t)tiles_n * nk * CHUNK + 4096;
  uint8_t *x6, *w6;
  float* sink;
  hipMalloc(&x6, xbytes), hipMalloc(&w6, wbytes), hipMalloc(&sink, (size_t)tiles_m * tiles_n * 512 * 4);
  uint8_t* h = (uint8_t*)malloc(xbytes > wbytes ? xbytes : wbytes);
  srand(3);
  for (size_t i = 0; i < (xbytes > wbytes ? xbytes : wbytes); ++i) h[i] = (uint8_t)rand();
  // scale bytes near 127 (x: 64 B per row group block of 1600 B at + 1536; W: 256 B per pair block of 6400 B at + 6144)
  for (size_t o = 0; o + 1600 <= xbytes; o += 1600)
    for (int i = 0; i < 64; ++i) h[o + 1536 + i] = 120 + (rand() & 7);
  hipMemcpy(x6, h, xbytes, hipMemcpyHostToDevice);
  for (size_t i = 0; i < wbytes; ++i) h[i] = (uint8_t)rand();
  for (size_t o = 0; o + 6400 <= wbytes; o += 6400)
    for (int i = 0; i < 256; ++i) h[o + 6144 + i] = 120 + (rand() & 7);
  hipMemcpy(w6, h, wbytes, hipMemcpyHostToDevice);
  const double flop = 2.0 * M * K * N;
  printf("M %d K %d N %d: %d workgroups x 12 waves, %d steps of 64 k, %.1f MB x + %.1f MB W images\n", M, K, N, tiles_m * tiles_n, nk,
         xbytes / 1e6, wbytes / 1e6);
  for (int bm : {0, 8, 4, 16}) {
    if (bm && (tiles_m % bm || (tiles_m * tiles_n / 8) % bm || 8 % (tiles_m / bm) || tiles_m * tiles_n != 256)) continue;
    const float t0 = run<0>(x6, w6, nk, tiles_m, tiles_n, sink, 200, bm);
    const float t1 = run<1>(x6, w6, nk, tiles_m, tiles_n, sink, 200, bm);
    const float t2 = run<2>(x6, w6, nk, tiles_m, tiles_n, sink, 200, bm);
    printf(" tile map: %s\n", bm ? "XCD-local blocks" : "whole rows of weight tiles per XCD (the shipped map)");
    if (bm) printf("   %d token tiles x %d weight tiles per XCD\n", bm, tiles_m * tiles_n / 8 / bm);
    printf("  full main loop (LDS-DMA ring + fragment reads + scaled MFMAs): %.1f us per launch = %.3f us per step, %.2f PFLOP/s-equiv\n", t0,
           t0 / nk, flop / (t0 * 1e-6) / 1e15);
    printf("  feed only (no MFMA):                                          %.1f us = %.3f us per step\n", t1, t1 / nk);
    printf("  compute + LDS reads only (no LDS-DMA):                        %.1f us = %.3f us per step\n", t2, t2 / nk);
  }
  printf("  (the shipped bf16 kernel at this shape: 0.65 us per step + 3.5 us ring fill = 45 us of its 54.5)\n");
  return 0;
}
