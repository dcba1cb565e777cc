// Microbenchmark: issue cost (cycles per instruction, one wave's stream) of the VALU ops used by the weight expand.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
template <int OP>
__global__ void k(unsigned long long* out, uint32_t* sink, int iters) {
  uint32_t a = threadIdx.x * 2654435761u, b = a ^ 0x12345678u, c = a + 77u, d = b + 99u, e = 5, f = 6;
  float sc = 0.25f;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (OP == 0) { asm volatile("v_cvt_scalef32_pk_bf16_fp8 %0, %1, %2" : "=v"(a) : "v"(b), "v"(sc)); asm volatile("v_cvt_scalef32_pk_bf16_fp8 %0, %1, %2" : "=v"(c) : "v"(d), "v"(sc)); }
      if (OP == 1) { asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a) : "v"(b), "v"(c), "v"(d)); asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(c) : "v"(d), "v"(b), "v"(d)); }
      if (OP == 2) { asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(a) : "v"(b), "v"(c), "v"(d)); asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(c) : "v"(d), "v"(b), "v"(d)); }
      if (OP == 3) { asm volatile("v_lshlrev_b32 %0, 4, %1" : "=v"(a) : "v"(b)); asm volatile("v_and_b32 %0, 0x7070707, %1" : "=v"(c) : "v"(d)); }
      if (OP == 4) { asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a) : "v"(b)); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(c) : "v"(d), "v"(sc)); }
      if (OP == 6) {  // independent: 4 destinations in rotation, sources never written
        asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(a) : "v"(b), "v"(d), "v"(sc));
        asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(c) : "v"(d), "v"(b), "v"(sc));
        asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(e) : "v"(b), "v"(d), "v"(sc));
        asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(f) : "v"(d), "v"(b), "v"(sc));
      }
      if (OP == 7) {  // dependent chain
        asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(d));
        asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(d));
        asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(d));
        asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(d));
      }
      if (OP == 8) {  // VOP2 e32 encodings, independent
        asm volatile("v_and_b32 %0, %1, %2" : "=v"(a) : "v"(b), "v"(d));
        asm volatile("v_and_b32 %0, %1, %2" : "=v"(c) : "v"(d), "v"(b));
        asm volatile("v_and_b32 %0, %1, %2" : "=v"(e) : "v"(b), "v"(d));
        asm volatile("v_and_b32 %0, %1, %2" : "=v"(f) : "v"(d), "v"(b));
      }
      if (OP == 9) {  // v_add_f32 independent
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(a) : "v"(b), "v"(d));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(c) : "v"(d), "v"(b));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(e) : "v"(b), "v"(d));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(f) : "v"(d), "v"(b));
      }
      if (OP == 5) { asm volatile("v_cvt_pk_f32_fp8 %0, %1" : "=v"(*(uint64_t*)&a) : "v"(b)); asm volatile("v_cvt_pk_f32_fp8 %0, %1" : "=v"(*(uint64_t*)&c) : "v"(d)); }
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
  if (a + c + e + f == 0x7fffffff) sink[0] = a;
}
template <int OP> void run(const char* name, int waves, unsigned long long* out, uint32_t* sink) {
  const int iters = 200;
  k<OP><<<256, 64 * waves>>>(out, sink, iters); hipDeviceSynchronize();
  k<OP><<<256, 64 * waves>>>(out, sink, iters); hipDeviceSynchronize();
  unsigned long long h[4096]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) s += h[b * 16 + w];
  printf("%-34s waves/CU %2d: %.2f cycles per instruction\n", name, waves, s / (256.0 * waves * iters * (OP >= 6 ? 64 : 32)));
}
int main() {
  unsigned long long* out; uint32_t* sink; hipMalloc(&out, 4096 * 8); hipMalloc(&sink, 4);
  for (int w : {4, 8, 16}) {
    run<6>("v_and_or_b32 independent x4", w, out, sink); run<7>("v_and_or_b32 dependent chain", w, out, sink);
    run<8>("v_and_b32 (VOP2) independent", w, out, sink); run<9>("v_add_f32 independent", w, out, sink);
    run<0>("v_cvt_scalef32_pk_bf16_fp8", w, out, sink); run<1>("v_perm_b32", w, out, sink); run<2>("v_and_or_b32", w, out, sink);
    run<3>("v_lshlrev/v_and (literal)", w, out, sink); run<4>("v_cvt_f32_i32 / v_mul_f32", w, out, sink); run<5>("v_cvt_pk_f32_fp8", w, out, sink);
  }
  return 0;
}
