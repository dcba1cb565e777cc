// Microbenchmark: wave-side issue cost (s_memtime delta) of the ways to bring 1 KiB per wave from L2 to LDS.
// Build: hipcc --offload-arch=gfx950 -O3 -o issue_cost issue_cost.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")

template <int MODE, bool WITH_MFMA>
__global__ __launch_bounds__(512) void k(const unsigned char* src, unsigned long long* out, float* sink, int iters, int stride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)blockIdx.x * 8 * 4096 * 4 + wave * 4096 * 4), 0, 0x7fffffff, 0x00020000);
  const unsigned char* gp = src + (size_t)blockIdx.x * 8 * 4096 * 4 + wave * 4096 * 4 + lane * 16;
  unsigned long long sum = 0;
  f32x16 acc = {0};
  bf16x8 fa = {1, 2, 3, 4, 5, 6, 7, 8}, fb = {1, 1, 1, 1, 1, 1, 1, 1};
  const uint32_t lds_w = (uint32_t)(uintptr_t)(lds_void*)(smem + wave * 8192 + lane * 16);
  u32x4 v0, v1, v2, v3;
  for (int it = 0; it < iters; ++it) {
    unsigned long long t0, t1;
    const int soff = (it & 3) * 4096;
    if (WITH_MFMA) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
    }
    STAMP(t0);
    if (MODE == 0) {  // 4 x buffer_load ... lds
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(smem + wave * 8192 + i * 1024), 16, lane * 16 + i * 1024, soff, 0, 0);
    } else if (MODE == 1) {  // 4 x global_load_dwordx4 to VGPRs (asm, not waited here)
      asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:1024\n\t"
                   "global_load_dwordx4 %2, %4, off offset:2048\n\tglobal_load_dwordx4 %3, %4, off offset:3072"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(gp + soff) : "memory");
    } else if (MODE == 2) {  // 4 x ds_write_b128 (of registers already held)
      asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:1024\n\tds_write_b128 %0, %3 offset:2048\n\tds_write_b128 %0, %4 offset:3072"
                   ::"v"(lds_w), "v"(v0), "v"(v1), "v"(v2), "v"(v3) : "memory");
    } else if (MODE == 3) {  // 4 x ds_read_b128
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(lds_w) : "memory");
    }
    asm volatile("s_memtime %0" : "=s"(t1)::"memory");  // no lgkm wait: issue cost only
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    sum += t1 - t0;
    if (MODE == 1 || MODE == 3) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
    if ((it & 3) == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (lane == 0) out[blockIdx.x * 8 + wave] = sum;
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  if (MODE == 1 || MODE == 3) s += (float)(v0[0] + v1[1] + v2[2] + v3[3]);
  if (s == 123.456f) sink[0] = s;
}

template <int MODE, bool WITH_MFMA>
void run(const char* name, unsigned char* src, unsigned long long* out, float* sink, int waves) {
  const int iters = 256;
  hipFuncSetAttribute((const void*)k<MODE, WITH_MFMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  k<MODE, WITH_MFMA><<<256, 64 * waves, 65536>>>(src, out, sink, iters, 0);
  hipDeviceSynchronize();
  k<MODE, WITH_MFMA><<<256, 64 * waves, 65536>>>(src, out, sink, iters, 0);
  hipDeviceSynchronize();
  unsigned long long h[2048];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) s += h[b * 8 + w];
  printf("%-44s waves/CU %d mfma %d : %7.1f cycles per 4 instr (%.1f each)\n", name, waves, (int)WITH_MFMA, s / (256.0 * waves * iters), s / (256.0 * waves * iters) / 4);
}

int main() {
  unsigned char* src; unsigned long long* out; float* sink;
  hipMalloc(&src, (size_t)256 * 8 * 4096 * 4 + 65536); hipMemset(src, 1, (size_t)256 * 8 * 4096 * 4);
  hipMalloc(&out, 2048 * 8); hipMalloc(&sink, 4);
  for (int waves : {1, 4, 8}) {
    if (waves == 1) { run<0,false>("buffer_load_dwordx4 lds (LDS-DMA)", src, out, sink, 1); run<1,false>("global_load_dwordx4 -> VGPR", src, out, sink, 1); run<2,false>("ds_write_b128", src, out, sink, 1); run<3,false>("ds_read_b128", src, out, sink, 1); }
    if (waves == 4) { run<0,false>("buffer_load_dwordx4 lds (LDS-DMA)", src, out, sink, 4); run<1,false>("global_load_dwordx4 -> VGPR", src, out, sink, 4); run<2,false>("ds_write_b128", src, out, sink, 4); run<3,false>("ds_read_b128", src, out, sink, 4);
                      run<0,true>("buffer_load_dwordx4 lds (LDS-DMA)", src, out, sink, 4); run<1,true>("global_load_dwordx4 -> VGPR", src, out, sink, 4); run<2,true>("ds_write_b128", src, out, sink, 4); run<3,true>("ds_read_b128", src, out, sink, 4); }
    if (waves == 8) { run<0,false>("buffer_load_dwordx4 lds (LDS-DMA)", src, out, sink, 8); run<1,false>("global_load_dwordx4 -> VGPR", src, out, sink, 8); run<2,false>("ds_write_b128", src, out, sink, 8); run<3,false>("ds_read_b128", src, out, sink, 8);
                      run<0,true>("buffer_load_dwordx4 lds (LDS-DMA)", src, out, sink, 8); run<1,true>("global_load_dwordx4 -> VGPR", src, out, sink, 8); run<2,true>("ds_write_b128", src, out, sink, 8); run<3,true>("ds_read_b128", src, out, sink, 8); }
  }
  return 0;
}
