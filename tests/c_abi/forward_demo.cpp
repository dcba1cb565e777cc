// A consumer of the C ABI with no Python and no torch in the process: plain HIP runtime calls for memory, then the
// library's pack + forward entry points (include/lqer_hip.h), as a reference maintainer binding the ABI from C would
// use them.  Inputs come from a small LCG that tests/test_c_abi_gpu.py repeats in numpy; the output tensor is written
// to a file and compared with the CPU oracle there.
//   build: hipcc -O2 -o forward_demo forward_demo.cpp -L<dir of liblqer_hip.so> -llqer_hip -Wl,-rpath,<dir>
//   run:   forward_demo M K N r out.bin [a16]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/lqer_hip.h"

#define HIP_OK(e)                                                                  \
  do {                                                                             \
    hipError_t err_ = (e);                                                         \
    if (err_ != hipSuccess) {                                                      \
      fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(err_)); \
      return 2;                                                                    \
    }                                                                              \
  } while (0)
#define LQER_OK_(e)                                                                     \
  do {                                                                                  \
    int rc_ = (e);                                                                      \
    if (rc_ != 0) {                                                                     \
      fprintf(stderr, "%s:%d: code %d: %s\n", __FILE__, __LINE__, rc_, lqer_last_error()); \
      return 3;                                                                         \
    }                                                                                   \
  } while (0)

static uint32_t lcg_state = 12345u;
static float lcg_uniform() {  // in [-1, 1), 24 bits
  lcg_state = lcg_state * 1664525u + 1013904223u;
  return (float)(lcg_state >> 8) / 8388608.0f - 1.0f;
}
static void fill_f16(std::vector<_Float16>& v, float scale) {
  for (auto& e : v) e = (_Float16)(scale * lcg_uniform());
}

int main(int argc, char** argv) {
  if (argc < 6) {
    fprintf(stderr, "usage: %s M K N r out.bin [a16]\n", argv[0]);
    return 1;
  }
  const int64_t M = atoll(argv[1]), K = atoll(argv[2]), N = atoll(argv[3]), r = atoll(argv[4]);
  const bool a16 = argc > 6 && strcmp(argv[6], "a16") == 0;
  if (lqer_version() != LQER_ABI_VERSION) {
    fprintf(stderr, "ABI mismatch\n");
    return 1;
  }
  std::vector<_Float16> x(M * K), W(N * K), A(K * r), B(r * N), bias(N);
  fill_f16(x, 2.0f), fill_f16(W, 0.05f), fill_f16(A, 0.02f), fill_f16(B, 0.02f), fill_f16(bias, 0.1f);

  lqer_linear_desc_t d;
  memset(&d, 0, sizeof(d));
  d.in_features = (int32_t)K, d.out_features = (int32_t)N, d.rank = (int32_t)r, d.has_bias = 1;
  const lqer_qfmt_t mx8 = {LQER_Q_MXINT, 8, 16, 8, 127}, mx4 = {LQER_Q_MXINT, 4, 16, 8, 127};
  const lqer_qfmt_t pass_x = {LQER_Q_PASSTHROUGH, 11, 0, 8, 127}, pass_xa = {LQER_Q_PASSTHROUGH, 16, 0, 8, 127};
  d.w_fmt = mx4, d.b_fmt = mx8;
  d.x_fmt = a16 ? pass_x : mx8;  // a16: fp16 activations as two bf16 limbs (the limb route needs no eligibility check)
  d.a_out_fmt = a16 ? pass_xa : mx8;
  d.b_out_fmt = a16 ? pass_xa : mx8;
  lqer_linear_desc_t single = d;  // sizes of ONE copy of every image
  single.x_fmt = single.a_out_fmt = mx8;
  lqer_linear_sizes_t sz, sz1;
  LQER_OK_(lqer_linear_sizes(&d, M, &sz));
  LQER_OK_(lqer_linear_sizes(&single, M, &sz1));
  int xl = 1, al = 1;
  LQER_OK_(lqer_desc_limbs(&d, &xl, &al));

  void *dx, *dW, *dA, *dB, *dbias, *dy, *w1, *a1, *b1, *wp, *at, *bt, *bq, *ws, *scr;
  int32_t* flags;
  HIP_OK(hipMalloc(&dx, x.size() * 2));
  HIP_OK(hipMalloc(&dW, W.size() * 2));
  HIP_OK(hipMalloc(&dA, A.size() * 2 + 16));
  HIP_OK(hipMalloc(&dB, B.size() * 2 + 16));
  HIP_OK(hipMalloc(&dbias, bias.size() * 2));
  HIP_OK(hipMalloc(&dy, M * N * 2));
  HIP_OK(hipMalloc(&w1, sz1.w_packed));
  HIP_OK(hipMalloc(&a1, sz1.a_t + 16));
  HIP_OK(hipMalloc(&b1, sz1.b_t + 16));
  HIP_OK(hipMalloc(&wp, sz.w_packed));
  HIP_OK(hipMalloc(&at, sz.a_t + 16));
  HIP_OK(hipMalloc(&bt, sz.b_t + 16));
  HIP_OK(hipMalloc(&bq, sz.bias_q));
  HIP_OK(hipMalloc(&ws, sz.workspace + 16));
  HIP_OK(hipMalloc(&scr, N * ((K + 15) / 16) + 16));
  HIP_OK(hipMalloc((void**)&flags, 8));
  HIP_OK(hipMemcpy(dx, x.data(), x.size() * 2, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dW, W.data(), W.size() * 2, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dbias, bias.data(), bias.size() * 2, hipMemcpyHostToDevice));
  if (r > 0) {
    HIP_OK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
  }
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));

  // one-time packing (the reference's first-forward quantization, linear.py:149-153)
  LQER_OK_(lqer_pack_weight_mxint(dW, LQER_F16, N, K, K, &d.w_fmt, w1, scr, st));
  LQER_OK_(lqer_pack_bias(dbias, LQER_F16, N, &d.b_fmt, (float*)bq, st));
  int32_t limbs[2] = {0, 0};
  if (r > 0) {
    LQER_OK_(lqer_pack_lowrank(dA, dB, LQER_F16, K, N, r, a1, b1, flags, st));
    HIP_OK(hipMemcpyAsync(limbs, flags, 8, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
  }
  const int64_t Kp = lqer_padded_k(K), Np = lqer_padded_n(N), rp = lqer_padded_r(r);
  LQER_OK_(lqer_replicate_rows(w1, wp, Np / LQER_PANEL_ROWS, (Kp / 64) * LQER_PANEL_BYTES, xl, st));
  if (r > 0) {
    LQER_OK_(lqer_replicate_rows(a1, at, 3 * rp, Kp * 2, xl, st));
    LQER_OK_(lqer_replicate_rows(b1, bt, 3 * Np, rp * 2, al, st));
  }
  // the forward (linear.py:145-157), twice: the second call must reproduce the first bit for bit
  std::vector<_Float16> y(M * N), y2(M * N);
  for (int pass = 0; pass < 2; ++pass) {
    LQER_OK_(lqer_linear_forward(&d, dx, LQER_F16, M, K, wp, r > 0 ? at : nullptr, r > 0 ? bt : nullptr, limbs[0], limbs[1],
                                 (const float*)bq, dy, N, ws, sz.workspace, st));
    HIP_OK(hipMemcpyAsync(pass ? y2.data() : y.data(), dy, M * N * 2, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
  }
  if (memcmp(y.data(), y2.data(), M * N * 2) != 0) {
    fprintf(stderr, "second forward differs from the first\n");
    return 4;
  }
  FILE* f = fopen(argv[5], "wb");
  if (!f || fwrite(y.data(), 2, M * N, f) != (size_t)(M * N)) return 5;
  fclose(f);
  double sum = 0;
  for (auto e : y) sum += (double)(float)e;
  printf("forward_demo M=%lld K=%lld N=%lld r=%lld %s limbs(A,B)=(%d,%d) image copies(x,xA)=(%d,%d) checksum %.6f\n", (long long)M,
         (long long)K, (long long)N, (long long)r, a16 ? "a16" : "mxint", limbs[0], limbs[1], xl, al, sum);
  return 0;
}
