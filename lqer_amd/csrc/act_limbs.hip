// "A16" activations: x_quantizer = passthrough (reference quantizers/passthrough.py:1; every *-int.toml template,
// e.g. experiments/configs/template/llama-7b-int.toml q_config.linear.x_quantizer).  The GEMM kernels multiply exact
// bf16 operands, so a pass-through activation is written as a sum of bf16 limbs, x = x0 + x1 (+ x2), each limb the
// round-to-nearest bf16 of what the previous ones left: 8 significand bits per limb, i.e. one limb for a bf16
// tensor, two for fp16 (11 bits), three for fp32 (24 bits).  The limbs are laid side by side along k,
//   xq [Mp][L * Kp],  limb l of x[m][k] at column l * Kp + k,
// and the weight image is repeated L times along k (lqer_replicate_rows), so the unchanged W4 kernels compute
// sum_l x_l W^T = x W^T with every product exact and fp32 accumulation - the arithmetic of the reference's
// F.linear on fp16 tensors (fp32 accumulate), without its fp16 rounding of intermediate results.
#include "common.h"

namespace lqer {

template <int DT, int L>
__global__ __launch_bounds__(256) void k_split_act(const void* __restrict__ x, int64_t M, int64_t K, int64_t ldx, bool vec,
                                                   bf16_t* __restrict__ xq, int64_t Kp) {
  const int64_t chunks = Kp / 8;
  const int64_t total = M * chunks;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t row = idx / chunks, k0 = (idx - row * chunks) * 8;
    float v[8];
    if (vec && k0 + 8 <= K) {
      if constexpr (DT == LQER_F32) {
        const float4 a = *(const float4*)((const float*)x + row * ldx + k0), b = *(const float4*)((const float*)x + row * ldx + k0 + 4);
        v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
      } else {
        const uint4 t = *(const uint4*)((const bf16_t*)x + row * ldx + k0);
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (DT == LQER_F16) {
            typedef __attribute__((ext_vector_type(2))) _Float16 h2;
            const h2 h = __builtin_bit_cast(h2, w[j]);
            v[2 * j] = (float)h[0], v[2 * j + 1] = (float)h[1];
          } else {
            v[2 * j] = __uint_as_float(w[j] << 16), v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = (k0 + i < K) ? load_elem<DT>(x, row * ldx + k0 + i) : 0.0f;
    }
#pragma unroll
    for (int l = 0; l < L; ++l) {
      uint32_t w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16_t b0 = f32_to_bf16_rne(v[2 * j]), b1 = f32_to_bf16_rne(v[2 * j + 1]);
        v[2 * j] -= __uint_as_float((uint32_t)b0 << 16);  // exact: the residual has fewer significant bits
        v[2 * j + 1] -= __uint_as_float((uint32_t)b1 << 16);
        w[j] = (uint32_t)b0 | ((uint32_t)b1 << 16);
      }
      *(uint4*)(xq + row * (L * Kp) + l * Kp + k0) = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
}

// LQER_Q_PASSTHROUGH_F16: the fp16 tensor itself, row- and k-padded, is the activation image (fp16 main loops).
__global__ __launch_bounds__(256) void k_copy_act_f16(const bf16_t* __restrict__ x, int64_t M, int64_t K, int64_t ldx, bool vec,
                                                      bf16_t* __restrict__ xq, int64_t Kp) {
  const int64_t chunks = Kp / 8;
  const int64_t total = M * chunks;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t row = idx / chunks, k0 = (idx - row * chunks) * 8;
    uint4 t;
    if (vec && k0 + 8 <= K) {
      t = *(const uint4*)(x + row * ldx + k0);
    } else {
      uint32_t w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t lo = k0 + 2 * j < K ? x[row * ldx + k0 + 2 * j] : 0u, hi = k0 + 2 * j + 1 < K ? x[row * ldx + k0 + 2 * j + 1] : 0u;
        w[j] = lo | (hi << 16);
      }
      t = make_uint4(w[0], w[1], w[2], w[3]);
    }
    *(uint4*)(xq + row * Kp + k0) = t;
  }
}

int copy_act_f16_dispatch(const void* x, int64_t M, int64_t K, int64_t ldx, bf16_t* xq, hipStream_t st) {
  if (M == 0) return LQER_OK;
  const int64_t Kp = lqer_padded_k(K);
  const bool vec = ((uintptr_t)x % 16 == 0) && ((ldx * 2) % 16 == 0);
  const int64_t total = M * (Kp / 8);
  const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  k_copy_act_f16<<<grid, 256, 0, st>>>((const bf16_t*)x, M, K, ldx, vec, xq, Kp);
  return check_launch("copy_act_f16");
}

int split_act_dispatch(const void* x, int dtype, int64_t M, int64_t K, int64_t ldx, int limbs, bf16_t* xq, hipStream_t st) {
  if (M == 0) return LQER_OK;
  const int64_t Kp = lqer_padded_k(K);
  const int esz = dtype == LQER_F32 ? 4 : 2;
  const bool vec = ((uintptr_t)x % 16 == 0) && ((ldx * esz) % 16 == 0);
  const int64_t total = M * (Kp / 8);
  const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
#define SPLIT_LAUNCH(DT, L) k_split_act<DT, L><<<grid, 256, 0, st>>>(x, M, K, ldx, vec, xq, Kp)
  if (dtype == LQER_BF16 && limbs == 1)
    SPLIT_LAUNCH(LQER_BF16, 1);
  else if (dtype == LQER_F16 && limbs == 2)
    SPLIT_LAUNCH(LQER_F16, 2);
  else if (dtype == LQER_F32 && limbs == 3)
    SPLIT_LAUNCH(LQER_F32, 3);
  else {
    set_error("x_quantizer passthrough: %d limb(s) do not hold element type %d exactly (bf16: width 8 = 1 limb, fp16: "
              "width 11 = 2, fp32: width 24 = 3)", limbs, dtype);
    return LQER_E_INVALID;
  }
#undef SPLIT_LAUNCH
  return check_launch("split_act");
}

}  // namespace lqer
