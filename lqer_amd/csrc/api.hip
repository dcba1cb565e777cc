// C ABI of liblqer_hip.so (include/lqer_hip.h): argument checking, buffer carving, kernel dispatch.
// No allocation, no synchronisation, no global mutable state besides the thread-local error text.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "common.h"

namespace lqer {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return LQER_E_LAUNCH;
  }
  return LQER_OK;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static bool fmt_ok(const lqer_qfmt_t* f, const char* name, int max_width) {
  if (!f) {
    set_error("%s: null format", name);
    return false;
  }
  if (f->kind == LQER_Q_PASSTHROUGH) return true;
  if (f->kind == LQER_Q_PASSTHROUGH_F16 && strcmp(name, "x_quantizer") == 0) return true;
  if (f->kind == LQER_Q_MXINT_I8 && strcmp(name, "x_quantizer") != 0) {
    set_error("%s: LQER_Q_MXINT_I8 is an x_quantizer kind", name);
    return false;
  }
  if (f->kind == LQER_Q_INT) {  // fixed point: exp_bias = frac_width, exp_width = is_signed
    const bool role_ok = !strcmp(name, "x_quantizer") || !strcmp(name, "b_quantizer") || !strcmp(name, "A_out_quantizer") ||
                         !strcmp(name, "B_out_quantizer") || !strcmp(name, "quantize_mxint") || !strcmp(name, "w_quantizer");
    if (!role_ok) {
      set_error("%s: the integer quantizer is not implemented in this role", name);
      return false;
    }
    if (!strcmp(name, "w_quantizer") && !f->exp_width) {  // codes travel as two's-complement nibbles: signed, 2..4 bits
      set_error("w_quantizer: an unsigned integer weight is not implemented (4-bit codes 0..15 do not fit the two's-complement nibble)");
      return false;
    }
    const int maxw = f->exp_width ? max_width : max_width - 1;  // unsigned: one magnitude bit more per width
    if (f->width < 1 || f->width > maxw || f->exp_bias < -100 || f->exp_bias > 100) {
      set_error("%s: integer width %d outside [1,%d] or frac_width %d outside [-100,100]", name, f->width, maxw, f->exp_bias);
      return false;
    }
    return true;
  }
  if (f->kind != LQER_Q_MXINT && f->kind != LQER_Q_MXINT_I8) {
    set_error("%s: quantizer kind %d is not implemented on the HIP path", name, f->kind);
    return false;
  }
  if (f->width < 2 || f->width > max_width) {
    set_error("%s: width %d outside [2,%d]", name, f->width, max_width);
    return false;
  }
  if (f->exp_width < 1 || f->exp_width > 8) {
    set_error("%s: exponent_width %d outside [1,8]", name, f->exp_width);
    return false;
  }
  return true;
}

// bf16 limbs of a pass-through tensor: fmt.width = significand bits to carry (8 per limb); block_fp images need one
static int limbs_of(const lqer_qfmt_t& f) {
  if (f.kind != LQER_Q_PASSTHROUGH) return 1;
  return f.width <= 8 ? 1 : (f.width <= 16 ? 2 : 3);
}
static int act_limbs(const lqer_linear_desc_t* d) { return limbs_of(d->x_fmt); }
// 4-bit limbs of the packed weight: block_fp weights of 5..8 bits travel as three signed base-8 digits side by side along k (pack.hip),
// the activation image is repeated to match (include/lqer_hip.h "weights of 5..8 bits")
static int w_panel_limbs(const lqer_linear_desc_t* d) { return (d->w_fmt.kind == LQER_Q_MXINT && d->w_fmt.width > 4) ? 3 : 1; }
// ... and the copies of the activation image that go with them: none on the int8 route (its kernel multiplies the int8 CODES of such
// a weight, one image of its own behind the limb panels)
static int w_limbs(const lqer_linear_desc_t* d) { return d->x_fmt.kind == LQER_Q_MXINT_I8 ? 1 : w_panel_limbs(d); }
// bytes of the activation image buffer `xq` of the split calls: [Mp][act limbs x weight limbs x Kp] bf16 and, when the weight has
// limbs, the single-copy image behind it (256-byte aligned) - the quantizer and the side GEMM work on that one, the GEMM on the wide one
static size_t act_image_bytes(const lqer_linear_desc_t* d, int64_t m_max) {
  const size_t one = align_up((size_t)lqer_padded_m(m_max) * lqer_padded_k(d->in_features) * 2 * act_limbs(d), 256);
  // (an LQER_Q_MXINT_I8 descriptor sizes for the bf16 kernels too: token counts the int8 kernel does not serve run them on the
  // same buffers - lqer_linear_forward switches the kind)
  const int wl = d->x_fmt.kind == LQER_Q_MXINT_I8 ? w_panel_limbs(d) : w_limbs(d);
  return wl > 1 ? align_up(one * wl, 256) + one : one;
}
static void* act_single_copy(const lqer_linear_desc_t* d, void* xq, int64_t M) {
  const int wl = w_limbs(d);
  if (wl == 1) return xq;
  const size_t one = align_up((size_t)lqer_padded_m(M) * lqer_padded_k(d->in_features) * 2 * act_limbs(d), 256);
  return (unsigned char*)xq + align_up(one * wl, 256);
}
static bool x_is_f16(const lqer_linear_desc_t* d) { return d->x_fmt.kind == LQER_Q_PASSTHROUGH_F16; }
static bool x_is_i8(const lqer_linear_desc_t* d) { return d->x_fmt.kind == LQER_Q_MXINT_I8; }
// byte offset of the int8 weight image inside w_packed (behind the sign-magnitude panels)
static size_t i8_image_offset(const lqer_linear_desc_t* d) {
  const size_t Kp = lqer_padded_k(d->in_features), Np = lqer_padded_n(d->out_features);
  return align_up((Np / LQER_PANEL_ROWS) * (Kp / 64) * LQER_PANEL_BYTES * w_panel_limbs(d), 256);
}
// static requirements of the int8 route (include/lqer_hip.h "int8 route")
static bool i8_formats_ok(const lqer_linear_desc_t* d) {
  const lqer_qfmt_t &x = d->x_fmt, &w = d->w_fmt;
  const int64_t K = d->in_features;
  if (x.width < 2 || x.width > 8 || !(x.block <= 0 || x.block >= K)) {
    set_error("LQER_Q_MXINT_I8: x_quantizer must be block_fp with width <= 8 and one block per row (got width %d block %d)", x.width, x.block);
    return false;
  }
  if (w.kind != LQER_Q_MXINT || w.width > 8 || !(w.block <= 0 || w.block >= K || w.block % I8_BK == 0)) {
    set_error("LQER_Q_MXINT_I8: w_quantizer blocks must span a multiple of 128 k or the whole row (got block %d)", w.block);
    return false;
  }
  if (K < I8_BK) {
    set_error("LQER_Q_MXINT_I8: in_features %lld < 128", (long long)K);
    return false;
  }
  return true;
}
// LQER_Q_PASSTHROUGH_F16: a dense fp16 tensor whose extents are already the padded ones IS the activation image
// (M a multiple of the row padding: the tile kernels read whole row tiles; M <= 64: the small-M kernel and the side
// GEMM never read past row M - 1 - unless B_out blocks other than 16 send a decode-size call to the tile kernel)
static bool f16_image_is_input(const lqer_linear_desc_t* d, const void* x, int64_t M, int64_t ldx) {
  // (asked of the dispatcher itself, so the two can not drift: an integer B_out, or B_out blocks other than 16, send a
  // decode-size call to the tile kernel, whose buffer range covers whole row tiles)
  const bool smallm = M <= 64 && x_is_f16(d) && lqer_gemm_route(d, M, LQER_F16) == LQER_ROUTE_SMALLM;
  return x_is_f16(d) && ldx == d->in_features && d->in_features % LQER_K_ALIGN == 0 && (M % LQER_M_ALIGN == 0 || smallm) &&
         ((uintptr_t)x & 15) == 0 && w_limbs(d) == 1;
}
// Decode sizes with the fused-quantizer formats: the small-M GEMM reduces the split-K partial tiles of x A itself
// (lqer_quantize_act_xa / lqer_linear_gemm with xaq == NULL), one launch less on a launch-bound path.
static bool decode_partials_ok(const lqer_linear_desc_t* d, int64_t M) {
  if (!d || d->rank <= 0 || M <= 0 || M > 64) return false;
  if (d->w_fmt.kind != LQER_Q_MXINT) return false;  // (integer weights: two's-complement nibbles - the tile kernel at every M)
  if (w_limbs(d) != 1) return false;  // (5..8-bit weights: the GEMM walks three limb images over a repeated activation image)
  if (d->x_fmt.kind != LQER_Q_MXINT || d->a_out_fmt.kind != LQER_Q_MXINT) return false;
  if (!xa_fused_partials_ok(make_qp(d->x_fmt), make_qp(d->a_out_fmt), d->rank)) return false;
  const lqer_qfmt_t& bo = d->b_out_fmt;  // the small-M kernel: B_out pass-through or blocks of 16
  return bo.kind == LQER_Q_PASSTHROUGH || (bo.kind == LQER_Q_MXINT && bo.block == 16);
}
static bool need_f16(const lqer_linear_desc_t* d, int dtype, const char* what) {
  if (x_is_f16(d) && dtype != LQER_F16) {
    set_error("%s: x_quantizer LQER_Q_PASSTHROUGH_F16 takes fp16 tensors (dtype %d given)", what, dtype);
    return false;
  }
  return true;
}
// the side product x A is an fp32 sum: two limbs at least
static int xa_limbs(const lqer_linear_desc_t* d) {
  if (d->a_out_fmt.kind != LQER_Q_PASSTHROUGH) return 1;
  return d->a_out_fmt.width <= 16 ? 2 : 3;
}
static bool passthrough_width_ok(const lqer_qfmt_t& f, const char* name) {
  if (f.kind == LQER_Q_PASSTHROUGH && (f.width < 1 || f.width > 24)) {
    set_error("%s: a passthrough format must say how many significand bits to carry in `width` (8 = bf16, 11 = fp16, "
              "16, 24 = fp32), got %d", name, f.width);
    return false;
  }
  return true;
}

}  // namespace lqer

using namespace lqer;
#ifdef LQER_HOST_TIMING  // diagnostic build: host nanoseconds between marks of lqer_linear_forward, printed at exit
#include <ctime>
namespace {
struct FwdT {
  double seg[8] = {0}; long n = 0; timespec last;
  void start() { clock_gettime(CLOCK_MONOTONIC, &last); }
  void mark(int i) { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); seg[i] += (t.tv_sec - last.tv_sec) * 1e9 + (t.tv_nsec - last.tv_nsec); last = t; }
  ~FwdT() { if (n) { fprintf(stderr, "[forward host] calls %ld:", n); for (int i = 0; i < 8; ++i) fprintf(stderr, " seg%d %.0f ns", i, seg[i] / n); fprintf(stderr, "\n"); } }
};
FwdT fwd_t;
}
#define HT_START() fwd_t.start()
#define HT_MARK(i) fwd_t.mark(i)
#define HT_DONE() (fwd_t.n++)
#else
#define HT_START()
#define HT_MARK(i)
#define HT_DONE()
#endif

extern "C" {
static int gemm_shape_args(const lqer_linear_desc_t* d, int64_t M, int dtype, GemmArgs& g);
// Token counts of the 128-row tile kernel, on request (LQER_TUNE_XA_REDUCE_IN_GEMM - measured slower, include/lqer_hip.h): its workgroups sum the partial tiles for their own rows on the way into the
// side product's LDS stage (k_lqer_gemm XAPART) - the reduce launch between the quantizer and the GEMM is skipped
static bool tile_partials_ok(const lqer_linear_desc_t* d, int64_t M, int dtype) {
  if (!d || d->rank <= 0 || M <= 64 || (dtype != LQER_F16 && dtype != LQER_BF16) || !(d->tuning & LQER_TUNE_XA_REDUCE_IN_GEMM)) return false;
  if (d->w_fmt.kind != LQER_Q_MXINT || d->x_fmt.kind != LQER_Q_MXINT || d->a_out_fmt.kind != LQER_Q_MXINT || w_limbs(d) != 1) return false;
  if (!xa_fused_partials_ok(make_qp(d->x_fmt), make_qp(d->a_out_fmt), d->rank)) return false;
  const lqer_qfmt_t& bo = d->b_out_fmt;  // (other B_out blocks: the pre-pass reads xAq)
  if (!(bo.kind == LQER_Q_PASSTHROUGH || (bo.kind == LQER_Q_MXINT && bo.block == 16))) return false;
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  if (gemm_shape_args(d, M, dtype, g)) return false;
  return gemm_route(g, true) == LQER_ROUTE_TILE128 && gemm_tile_rows(g) == 128;
}

int lqer_version(void) { return LQER_ABI_VERSION; }
size_t lqer_sizeof_qfmt(void) { return sizeof(lqer_qfmt_t); }
size_t lqer_sizeof_linear_desc(void) { return sizeof(lqer_linear_desc_t); }
size_t lqer_sizeof_linear_sizes(void) { return sizeof(lqer_linear_sizes_t); }
size_t lqer_sizeof_group_member(void) { return sizeof(lqer_group_member_t); }
const char* lqer_last_error(void) { return g_err; }

int64_t lqer_padded_k(int64_t K) { return (K + LQER_K_ALIGN - 1) / LQER_K_ALIGN * LQER_K_ALIGN; }
int64_t lqer_padded_n(int64_t N) { return (N + LQER_N_ALIGN - 1) / LQER_N_ALIGN * LQER_N_ALIGN; }
int64_t lqer_padded_m(int64_t M) { return (M + LQER_M_ALIGN - 1) / LQER_M_ALIGN * LQER_M_ALIGN; }
int64_t lqer_padded_r(int64_t r) { return (r + LQER_R_ALIGN - 1) / LQER_R_ALIGN * LQER_R_ALIGN; }

int lqer_quantize_mxint(const void* x, int dtype, int64_t rows, int64_t cols, int64_t ld, const lqer_qfmt_t* fmt,
                        float* deq_f32, int8_t* codes, int8_t* exps, void* stream) {
  if (!x && rows * cols > 0) {
    set_error("quantize_mxint: null input");
    return LQER_E_INVALID;
  }
  if (rows < 0 || cols < 0 || ld < cols) {
    set_error("quantize_mxint: bad shape rows=%lld cols=%lld ld=%lld", (long long)rows, (long long)cols, (long long)ld);
    return LQER_E_INVALID;
  }
  if (!fmt_ok(fmt, "quantize_mxint", 24)) return LQER_E_UNSUPPORTED;
  if (fmt->kind != LQER_Q_MXINT && fmt->kind != LQER_Q_INT) {
    set_error("quantize_mxint: format is neither block_fp nor integer");
    return LQER_E_INVALID;
  }
  if (codes && fmt->width > 8) {
    set_error("quantize_mxint: int8 codes need width <= 8");
    return LQER_E_UNSUPPORTED;
  }
  QP q = make_qp(*fmt);
  QuantOut o{deq_f32, codes, exps, nullptr, 0, 0};
  const int64_t L = (q.block <= 0 || q.block >= cols) ? cols : q.block;
  o.nblk = L > 0 ? (cols + L - 1) / L : 0;
  return quantize_dispatch(x, dtype, rows, cols, ld, q, o, (hipStream_t)stream);
}

int lqer_quantize_mxint_tiles(const void* x, int dtype, int64_t batches, int64_t rows, int64_t cols, const lqer_qfmt_t* fmt,
                              int64_t tile_rows, int64_t tile_cols, float* deq_f32, float* amax_scratch, void* stream) {
  if (batches < 0 || rows < 0 || cols < 0) {
    set_error("quantize_mxint_tiles: bad shape batches=%lld rows=%lld cols=%lld", (long long)batches, (long long)rows, (long long)cols);
    return LQER_E_INVALID;
  }
  if (batches * rows * cols == 0) return LQER_OK;
  if (!x || !deq_f32 || !amax_scratch) {
    set_error("quantize_mxint_tiles: null pointer");
    return LQER_E_INVALID;
  }
  if (!fmt_ok(fmt, "quantize_mxint_tiles", 24)) return LQER_E_UNSUPPORTED;
  if (fmt->kind != LQER_Q_MXINT) {
    set_error("quantize_mxint_tiles: the format is not block_fp");
    return LQER_E_INVALID;
  }
  const int64_t R = (tile_rows <= 0 || tile_rows > rows) ? rows : tile_rows, L = (tile_cols <= 0 || tile_cols > cols) ? cols : tile_cols;
  return quantize_tiles_dispatch(x, dtype, batches, rows, cols, R, L, make_qp(*fmt), deq_f32, amax_scratch, (hipStream_t)stream);
}

int lqer_quantize_act_mxint(const void* x, int dtype, int64_t M, int64_t K, int64_t ldx, const lqer_qfmt_t* fmt,
                            void* xq_bf16, void* stream) {
  if ((!x || !xq_bf16) && M * K > 0) {
    set_error("quantize_act: null pointer");
    return LQER_E_INVALID;
  }
  if (M < 0 || K < 0 || ldx < K) {
    set_error("quantize_act: bad shape M=%lld K=%lld ldx=%lld", (long long)M, (long long)K, (long long)ldx);
    return LQER_E_INVALID;
  }
  if (!fmt_ok(fmt, "x_quantizer", 9)) return LQER_E_UNSUPPORTED;
  if (fmt->kind != LQER_Q_MXINT && fmt->kind != LQER_Q_INT) {
    set_error("x_quantizer: only block_fp and integer activations have a bf16 image on the HIP path");
    return LQER_E_UNSUPPORTED;
  }
  QP q = make_qp(*fmt);
  QuantOut o{nullptr, nullptr, nullptr, (bf16_t*)xq_bf16, lqer_padded_k(K), 0};
  return quantize_dispatch(x, dtype, M, K, ldx, q, o, (hipStream_t)stream);
}

int lqer_quantize_act_i8(const void* x, int dtype, int64_t M, int64_t K, int64_t ldx, const lqer_qfmt_t* fmt, void* xq_i8,
                         void* stream) {
  if ((!x || !xq_i8) && M * K > 0) {
    set_error("quantize_act_i8: null pointer");
    return LQER_E_INVALID;
  }
  if (M < 0 || K < 0 || ldx < K) {
    set_error("quantize_act_i8: bad shape M=%lld K=%lld ldx=%lld", (long long)M, (long long)K, (long long)ldx);
    return LQER_E_INVALID;
  }
  if (!fmt || (fmt->kind != LQER_Q_MXINT && fmt->kind != LQER_Q_MXINT_I8) || fmt->width < 2 || fmt->width > 8 ||
      !(fmt->block <= 0 || fmt->block >= K)) {
    set_error("quantize_act_i8: needs block_fp with width <= 8 and one block per row");
    return LQER_E_UNSUPPORTED;
  }
  lqer_qfmt_t f = *fmt;
  f.kind = LQER_Q_MXINT;
  f.block = -1;
  QuantOut o{nullptr, nullptr, nullptr, nullptr, 0, 0};
  o.xq8 = (int8_t*)xq_i8;
  o.cols_p8 = padded_k8(K);
  o.xscale = const_cast<float*>(i8_row_scales(xq_i8, M, K));
  return quantize_dispatch(x, dtype, M, K, ldx, make_qp(f), o, (hipStream_t)stream);
}

int lqer_i8_prepare(void* w_packed, int64_t N, int64_t K, const lqer_qfmt_t* w_fmt, int32_t* flags, void* stream) {
  if (!w_packed || !flags || !w_fmt || N <= 0 || K <= 0) {
    set_error("i8_prepare: bad argument");
    return LQER_E_INVALID;
  }
  lqer_linear_desc_t d;
  memset(&d, 0, sizeof(d));
  d.in_features = (int32_t)K, d.out_features = (int32_t)N;
  d.w_fmt = *w_fmt;
  d.x_fmt.kind = LQER_Q_MXINT_I8, d.x_fmt.width = 8, d.x_fmt.block = -1;
  if (!i8_formats_ok(&d)) return LQER_E_UNSUPPORTED;
  return i8_prepare_dispatch(w_packed, N, K, w_fmt->width - 1, (unsigned char*)w_packed + i8_image_offset(&d), flags,
                             (hipStream_t)stream);
}

int lqer_unpack_weight_i8(const void* w_packed, int64_t N, int64_t K, float* w_f32, void* stream) {
  if (!w_packed || !w_f32 || N <= 0 || K <= 0) {
    set_error("unpack_weight_i8: bad argument");
    return LQER_E_INVALID;
  }
  lqer_linear_desc_t d;
  memset(&d, 0, sizeof(d));
  d.in_features = (int32_t)K, d.out_features = (int32_t)N;
  return i8_unpack_dispatch((const unsigned char*)w_packed + i8_image_offset(&d), N, K, w_f32, (hipStream_t)stream);
}

int lqer_unpack_weight_i8_fmt(const void* w_packed, int64_t N, int64_t K, const lqer_qfmt_t* w_fmt, float* w_f32, void* stream) {
  if (!w_packed || !w_f32 || !w_fmt || N <= 0 || K <= 0) {
    set_error("unpack_weight_i8: bad argument");
    return LQER_E_INVALID;
  }
  lqer_linear_desc_t d;
  memset(&d, 0, sizeof(d));
  d.in_features = (int32_t)K, d.out_features = (int32_t)N;
  d.w_fmt = *w_fmt;
  return i8_unpack_dispatch((const unsigned char*)w_packed + i8_image_offset(&d), N, K, w_f32, (hipStream_t)stream, w_panel_limbs(&d) > 1);
}

int lqer_linear_sizes(const lqer_linear_desc_t* d, int64_t m_max, lqer_linear_sizes_t* out) {
  if (!d || !out || d->in_features <= 0 || d->out_features <= 0 || d->rank < 0 || m_max < 0) {
    set_error("linear_sizes: bad descriptor");
    return LQER_E_INVALID;
  }
  const size_t Kp = lqer_padded_k(d->in_features), Np = lqer_padded_n(d->out_features);
  const size_t rp = lqer_padded_r(d->rank), Mp = lqer_padded_m(m_max);
  if (!passthrough_width_ok(d->x_fmt, "x_quantizer") || (d->rank > 0 && !passthrough_width_ok(d->a_out_fmt, "A_out_quantizer")))
    return LQER_E_INVALID;
  const size_t xl = act_limbs(d), al = xa_limbs(d);  // pass-through activations: images repeated per limb
  const size_t wl = w_panel_limbs(d);
  if (!fmt_ok(&d->w_fmt, "w_quantizer", 8)) return LQER_E_UNSUPPORTED;
  out->w_packed = (Np / LQER_PANEL_ROWS) * (Kp / 64) * LQER_PANEL_BYTES * xl * wl;
  if (x_is_i8(d)) {
    if (!i8_formats_ok(d)) return LQER_E_UNSUPPORTED;
    out->w_packed = i8_image_offset(d) + (wl > 1 ? i8_weight8_image_bytes(d->out_features, d->in_features)
                                                 : i8_weight_image_bytes(d->out_features, d->in_features));
  }
  out->a_t = 3 * rp * Kp * 2 * xl;
  out->b_t = 3 * Np * rp * 2 * al;
  out->bias_q = Np * 4;
  size_t side = 0;
  if (rp) {
    const size_t a = xa_scratch_bytes(m_max, d->in_features, rp);
    const size_t b = d->b_out_fmt.kind == LQER_Q_MXINT ? gemm_scratch_bytes(m_max, d->out_features, make_qp(d->b_out_fmt)) : 0;
    side = align_up(a > b ? a : b, 256);  // the two scratch uses never overlap in time
  }
  // (the int8 activation image + row scales of LQER_Q_MXINT_I8 fit the bf16 image's slot: K >= 128)
  const size_t act = act_image_bytes(d, m_max);
  (void)wl;
  if (x_is_i8(d) && i8_act_image_bytes(m_max, d->in_features) + Mp * sizeof(float) > act) {
    set_error("linear_sizes: int8 activation image larger than the bf16 one (K %d)", d->in_features);
    return LQER_E_UNSUPPORTED;
  }
  out->workspace = act + align_up(Mp * rp * 2 * al, 256) + side;
  return LQER_OK;
}

int lqer_pack_weight_mxint(const void* W, int dtype, int64_t N, int64_t K, int64_t ldw, const lqer_qfmt_t* fmt,
                           void* w_packed, void* scratch, void* stream) {
  if (!W || !w_packed || !scratch || N <= 0 || K <= 0 || ldw < K) {
    set_error("pack_weight: bad argument");
    return LQER_E_INVALID;
  }
  if (!fmt_ok(fmt, "w_quantizer", 8)) return LQER_E_UNSUPPORTED;
  if (fmt->kind != LQER_Q_MXINT && fmt->kind != LQER_Q_INT) {
    set_error("w_quantizer: only block_fp and integer weights can be packed");
    return LQER_E_UNSUPPORTED;
  }
  return pack_weight_dispatch(W, dtype, N, K, ldw, make_qp(*fmt), 1, w_packed, scratch, (hipStream_t)stream);
}

int lqer_pack_weight_mxint_2d(const void* W, int dtype, int64_t N, int64_t K, int64_t ldw, const lqer_qfmt_t* fmt, int64_t block_rows,
                              void* w_packed, void* scratch, void* stream) {
  if (!W || !w_packed || !scratch || N <= 0 || K <= 0 || ldw < K) {
    set_error("pack_weight: bad argument");
    return LQER_E_INVALID;
  }
  if (!fmt_ok(fmt, "w_quantizer", 8)) return LQER_E_UNSUPPORTED;
  if (fmt->kind != LQER_Q_MXINT) {
    set_error("w_quantizer: only block_fp weights can be packed");
    return LQER_E_UNSUPPORTED;
  }
  return pack_weight_dispatch(W, dtype, N, K, ldw, make_qp(*fmt), block_rows == 0 ? -1 : block_rows, w_packed, scratch, (hipStream_t)stream);
}

int lqer_unpack_weight_mxint(const void* w_packed, int64_t N, int64_t K, const lqer_qfmt_t* fmt, float* w_f32,
                             void* stream) {
  if (!w_packed || !w_f32 || !fmt || N <= 0 || K <= 0) {
    set_error("unpack_weight: bad argument");
    return LQER_E_INVALID;
  }
  return unpack_weight_dispatch(w_packed, N, K, fmt->width - 1, fmt->kind == LQER_Q_INT, w_f32, (hipStream_t)stream);
}

int lqer_pack_lowrank(const void* A, const void* B, int dtype, int64_t K, int64_t N, int64_t r, void* a_t, void* b_t,
                      int32_t* limb_flags, void* stream) {
  if (!A || !B || !a_t || !b_t || !limb_flags || K <= 0 || N <= 0 || r <= 0) {
    set_error("pack_lowrank: bad argument");
    return LQER_E_INVALID;
  }
  return pack_lowrank_dispatch(A, B, dtype, K, N, r, a_t, b_t, limb_flags, (hipStream_t)stream);
}

int lqer_pack_bias(const void* bias, int dtype, int64_t N, const lqer_qfmt_t* fmt, float* bias_q, void* stream) {
  if (!bias || !bias_q || N <= 0) {
    set_error("pack_bias: bad argument");
    return LQER_E_INVALID;
  }
  if (!fmt_ok(fmt, "b_quantizer", 24)) return LQER_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (fmt->kind == LQER_Q_PASSTHROUGH) return bias_passthrough_dispatch(bias, dtype, N, bias_q, st);
  (void)hipMemsetAsync(bias_q, 0, lqer_padded_n(N) * sizeof(float), st);
  QP q = make_qp(*fmt);
  QuantOut o{bias_q, nullptr, nullptr, nullptr, 0, 0};
  return quantize_dispatch(bias, dtype, 1, N, N, q, o, st);
}

size_t lqer_lowrank_xa_scratch_bytes(const lqer_linear_desc_t* d, int64_t m_max) {
  return (d && d->rank > 0) ? xa_scratch_bytes(m_max, d->in_features, lqer_padded_r(d->rank)) : 0;
}

int lqer_lowrank_xa(const lqer_linear_desc_t* d, const void* xq, int64_t M, const void* a_t, int a_limbs, void* xaq,
                    void* scratch, size_t scratch_bytes, void* stream) {
  if (!d || !xq || !a_t || !xaq || M < 0 || d->rank <= 0) {
    set_error("lowrank_xa: bad argument");
    return LQER_E_INVALID;
  }
  if (!fmt_ok(&d->a_out_fmt, "A_out_quantizer", 9)) return LQER_E_UNSUPPORTED;
  if (!passthrough_width_ok(d->x_fmt, "x_quantizer") || !passthrough_width_ok(d->a_out_fmt, "A_out_quantizer")) return LQER_E_INVALID;
  if ((a_limbs < 0 || a_limbs > 3) && !(a_limbs == -1 && x_is_i8(d))) {
    set_error("lowrank_xa: a_limbs %d outside [0,3] (-1 = one fp16 image of A^T: the int8 route only)", a_limbs);
    return LQER_E_INVALID;
  }
  return lowrank_xa_dispatch((const bf16_t*)xq, M, d->in_features, x_is_f16(d) ? 0 : (x_is_i8(d) ? -1 : act_limbs(d)), (const bf16_t*)a_t, a_limbs, d->rank,
                             make_qp(d->a_out_fmt), d->a_out_fmt.kind == LQER_Q_PASSTHROUGH ? xa_limbs(d) : 0,
                             (bf16_t*)xaq, (float*)scratch, scratch_bytes, (hipStream_t)stream);
}

// lqer_linear_forward only: the GEMM call behind this one wants `bytes` at the head of the shared scratch zeroed (gemm_amax_zero_bytes);
// `done` says whether a kernel of this call wrote the zeros (the one-launch int8 activation kernel does, the other routes do not)
struct AmaxZeroReq {
  void* p;       // head of the scratch the GEMM call will be handed
  size_t bytes;
  bool done;
};
static int gemm_shape_args(const lqer_linear_desc_t* d, int64_t M, int dtype, GemmArgs& g);
// the request for M tokens of this descriptor: bytes > 0 when the GEMM launches a pre-pass on atomicMax cells
static AmaxZeroReq amax_zero_request(const lqer_linear_desc_t* d, int64_t M, int dtype, int a_limbs, void* gemm_scratch) {
  AmaxZeroReq zr{gemm_scratch, 0, false};
  if (d && gemm_scratch && d->rank > 0 && x_is_i8(d) && a_limbs == -1 && M > 0) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    if (gemm_shape_args(d, M, dtype, g) == LQER_OK) zr.bytes = gemm_amax_zero_bytes(g, true);
  }
  return zr;
}
static int quantize_act_xa_single(const lqer_linear_desc_t* d, const void* x, int dtype, int64_t M, int64_t ldx, const void* a_t,
                                  int a_limbs, void* xq, void* xaq, void* scratch, size_t scratch_bytes, void* stream, AmaxZeroReq* zr);
static int quantize_act_xa_impl(const lqer_linear_desc_t* d, const void* x, int dtype, int64_t M, int64_t ldx, const void* a_t,
                                int a_limbs, void* xq, void* xaq, void* scratch, size_t scratch_bytes, void* stream, AmaxZeroReq* zr);

int lqer_quantize_act_xa(const lqer_linear_desc_t* d, const void* x, int dtype, int64_t M, int64_t ldx, const void* a_t,
                         int a_limbs, void* xq, void* xaq, void* scratch, size_t scratch_bytes, void* stream) {
  return quantize_act_xa_impl(d, x, dtype, M, ldx, a_t, a_limbs, xq, xaq, scratch, scratch_bytes, stream, nullptr);
}

int lqer_quantize_act_xa_prep(const lqer_linear_desc_t* d, const void* x, int dtype, int64_t M, int64_t ldx, const void* a_t,
                              int a_limbs, void* xq, void* xaq, void* scratch, size_t scratch_bytes, void* gemm_scratch,
                              size_t* ready_bytes, void* stream) {
  if (ready_bytes) *ready_bytes = 0;
  AmaxZeroReq zr = amax_zero_request(d, M, dtype, a_limbs, ready_bytes ? gemm_scratch : nullptr);
  const int rc = quantize_act_xa_impl(d, x, dtype, M, ldx, a_t, a_limbs, xq, xaq, scratch, scratch_bytes, stream, &zr);
  if (rc == LQER_OK && ready_bytes && zr.done) *ready_bytes = zr.bytes;
  return rc;
}

static int quantize_act_xa_impl(const lqer_linear_desc_t* d, const void* x, int dtype, int64_t M, int64_t ldx, const void* a_t,
                                int a_limbs, void* xq, void* xaq, void* scratch, size_t scratch_bytes, void* stream, AmaxZeroReq* zr) {
  if (!d || (!x && M > 0) || !xq || M < 0 || ldx < d->in_features) {
    set_error("quantize_act_xa: bad argument");
    return LQER_E_INVALID;
  }
  if (!fmt_ok(&d->w_fmt, "w_quantizer", 8)) return LQER_E_UNSUPPORTED;
  const int wl = w_limbs(d);
  if (wl == 1) return quantize_act_xa_single(d, x, dtype, M, ldx, a_t, a_limbs, xq, xaq, scratch, scratch_bytes, stream, zr);
  // a weight of three 4-bit limbs: the quantizer and the side GEMM work on the single-copy image behind the wide one, which is
  // then written as three copies side by side along k (one per weight limb) for the GEMM
  if (!passthrough_width_ok(d->x_fmt, "x_quantizer")) return LQER_E_INVALID;
  void* const one = act_single_copy(d, xq, M);
  const int rc = quantize_act_xa_single(d, x, dtype, M, ldx, a_t, a_limbs, one, xaq, scratch, scratch_bytes, stream, zr);
  if (rc || M == 0) return rc;
  return lqer_replicate_rows(one, xq, M, lqer_padded_k(d->in_features) * 2 * act_limbs(d), wl, stream);
}

static int quantize_act_xa_single(const lqer_linear_desc_t* d, const void* x, int dtype, int64_t M, int64_t ldx, const void* a_t,
                                  int a_limbs, void* xq, void* xaq, void* scratch, size_t scratch_bytes, void* stream, AmaxZeroReq* zr) {
  if (a_limbs == -2) {
    // the bf16 image of A^T with its fragment-major copy (lqer_a_b16_prepare): block-16 MXINT activations run quantizer + x A + A_out as
    // ONE launch where act16_fused.hip applies; everything else reads the image's first part as the one-limb image it is
    if (d->rank > 0 && a_t && xaq && xq && d->x_fmt.kind == LQER_Q_MXINT && fmt_ok(&d->x_fmt, "x_quantizer", 9) &&
        fmt_ok(&d->a_out_fmt, "A_out_quantizer", 9)) {
      const int rc = act16_fused_dispatch(x, dtype, M, d->in_features, ldx, make_qp(d->x_fmt), (bf16_t*)xq, a_t, d->rank, make_qp(d->a_out_fmt),
                                          (bf16_t*)xaq, d->tuning, (hipStream_t)stream);
      if (rc != LQER_E_UNSUPPORTED) return rc;
    }
    a_limbs = 1;
  }
  if (d->rank > 0 && a_t && !xaq) {  // partial tiles only: the GEMM reduces them (decode sizes)
    if (!decode_partials_ok(d, M) && !tile_partials_ok(d, M, dtype)) {
      set_error("quantize_act_xa: xaq == NULL needs M <= 64 or the 128-row tile kernel's token counts with fp16 / bf16 tensors, "
                "x / A_out block_fp in blocks of 16 (width <= 9), padded rank <= 64 and B_out pass-through or in blocks of 16");
      return LQER_E_INVALID;
    }
    const int rc = quant_xa_fused_dispatch(x, dtype, M, d->in_features, ldx, make_qp(d->x_fmt), (bf16_t*)xq, (const bf16_t*)a_t,
                                           a_limbs, d->rank, make_qp(d->a_out_fmt), nullptr, (float*)scratch, scratch_bytes,
                                           (hipStream_t)stream);
    if (rc == LQER_E_UNSUPPORTED) set_error("quantize_act_xa: scratch too small for the partial tiles");
    return rc == LQER_E_UNSUPPORTED ? LQER_E_WORKSPACE : rc;
  }
  if (d->rank > 0 && a_t && xaq) {
    if (!fmt_ok(&d->x_fmt, "x_quantizer", 9) || !fmt_ok(&d->a_out_fmt, "A_out_quantizer", 9)) return LQER_E_UNSUPPORTED;
    const int rc = quant_xa_fused_dispatch(x, dtype, M, d->in_features, ldx, make_qp(d->x_fmt), (bf16_t*)xq,
                                           (const bf16_t*)a_t, a_limbs, d->rank, make_qp(d->a_out_fmt), (bf16_t*)xaq,
                                           (float*)scratch, scratch_bytes, (hipStream_t)stream);
    if (rc != LQER_E_UNSUPPORTED) return rc;
  }
  int rc;
  if (x_is_f16(d)) {
    if (!need_f16(d, dtype, "quantize_act_xa")) return LQER_E_INVALID;
    if (xq == x) {  // the tensor itself is the image: nothing to write
      if (!f16_image_is_input(d, x, M, ldx)) {
        set_error("quantize_act_xa: xq == x needs a dense fp16 tensor with K %% %d == 0, M %% %d == 0 or M <= 64, 16-byte aligned",
                  LQER_K_ALIGN, LQER_M_ALIGN);
        return LQER_E_INVALID;
      }
      rc = LQER_OK;
    } else {
      rc = copy_act_f16_dispatch(x, M, d->in_features, ldx, (bf16_t*)xq, (hipStream_t)stream);
    }
  } else if (d->x_fmt.kind == LQER_Q_PASSTHROUGH) {
    if (!passthrough_width_ok(d->x_fmt, "x_quantizer")) return LQER_E_INVALID;
    rc = split_act_dispatch(x, dtype, M, d->in_features, ldx, act_limbs(d), (bf16_t*)xq, (hipStream_t)stream);
  } else if (x_is_i8(d)) {
    // one launch for quantizer + x A + A_out where the fused kernel applies (act8_fused.hip: 16-bit tensor, fp16 image of A^T with
    // its fragment-major copy, small token counts); else quantizer, split-K side GEMM and reduce as three launches
    if (d->rank > 0 && a_t && xaq && a_limbs == -1) {
      lqer_qfmt_t fx = d->x_fmt;
      fx.kind = LQER_Q_MXINT, fx.block = -1;
      const bool zreq = zr && zr->bytes > 0 && zr->p;
      bool zeroed = false;
      rc = act8_fused_dispatch(x, dtype, M, d->in_features, ldx, make_qp(fx), xq, a_t, d->rank, make_qp(d->a_out_fmt), (bf16_t*)xaq,
                               d->tuning, (hipStream_t)stream, zreq ? (float*)zr->p : nullptr, zreq ? zr->bytes : 0, &zeroed);
      if (rc == LQER_OK && zr) zr->done = zeroed;
      if (rc != LQER_E_UNSUPPORTED) return rc;
    }
    rc = lqer_quantize_act_i8(x, dtype, M, d->in_features, ldx, &d->x_fmt, xq, stream);
  } else {
    rc = lqer_quantize_act_mxint(x, dtype, M, d->in_features, ldx, &d->x_fmt, xq, stream);
  }
  if (rc || d->rank <= 0) return rc;
  return lqer_lowrank_xa(d, xq, M, a_t, a_limbs, xaq, scratch, scratch_bytes, stream);
}

size_t lqer_act_image_bytes(const lqer_linear_desc_t* d, int64_t m_max) {
  return (d && m_max >= 0 && d->in_features > 0) ? act_image_bytes(d, m_max) : 0;
}

size_t lqer_linear_gemm_scratch_bytes(const lqer_linear_desc_t* d, int64_t m_max) {
  if (!d || d->rank <= 0 || d->b_out_fmt.kind != LQER_Q_MXINT) return 0;
  return gemm_scratch_bytes(m_max, d->out_features, make_qp(d->b_out_fmt));
}

int lqer_linear_gemm(const lqer_linear_desc_t* d, const void* xq, int64_t M, const void* w_packed, const void* xaq,
                     const void* b_t, int b_limbs, const float* bias_q, void* y, int dtype, int64_t ldy, void* scratch,
                     size_t scratch_bytes, void* stream) {
  return lqer_linear_gemm_ld(d, xq, M, w_packed, xaq, d ? lqer_padded_r(d->rank) * xa_limbs(d) : 0, b_t, b_limbs, bias_q, y,
                             dtype, ldy, scratch, scratch_bytes, stream);
}

// fills the kernel arguments that depend on the descriptor and the token count alone (shapes, formats, limb counts)
static int gemm_shape_args(const lqer_linear_desc_t* d, int64_t M, int dtype, GemmArgs& g) {
  const bool lowrank = d->rank > 0;
  if (!fmt_ok(&d->w_fmt, "w_quantizer", 8)) return LQER_E_UNSUPPORTED;
  if (lowrank && !fmt_ok(&d->b_out_fmt, "B_out_quantizer", 24)) return LQER_E_UNSUPPORTED;
  if (!passthrough_width_ok(d->x_fmt, "x_quantizer") || (lowrank && !passthrough_width_ok(d->a_out_fmt, "A_out_quantizer")))
    return LQER_E_INVALID;
  if (!need_f16(d, dtype, "linear_gemm")) return LQER_E_INVALID;
  const int xl = act_limbs(d), al = lowrank ? xa_limbs(d) : 1;
  g.M = (int)M;
  g.N = d->out_features;
  g.Np = (int)lqer_padded_n(d->out_features);
  g.x_f16 = x_is_f16(d) ? 1 : 0;
  // activation limbs side by side along k with the weight image repeated to match - and the other way round for the three
  // 4-bit limbs of a 5..8-bit weight
  g.Kp = (int)lqer_padded_k(d->in_features) * xl * w_limbs(d);
  g.rp = (int)lqer_padded_r(d->rank) * al;
  g.w_mbits = w_limbs(d) > 1 ? 3 : d->w_fmt.width - 1;
  g.tuning = d->tuning;
  g.w_twos = d->w_fmt.kind == LQER_Q_INT ? 1 : 0;
  if (lowrank) g.bout = make_qp(d->b_out_fmt);
  if (x_is_i8(d)) {
    if (!i8_formats_ok(d)) return LQER_E_UNSUPPORTED;
    g.Kp = (int)padded_k8(d->in_features);  // row stride of the int8 activation image
    g.i8_shift = !(d->w_fmt.block <= 0 || d->w_fmt.block >= d->in_features);  // one weight block per row: no shifts at all
    g.w_i8codes = w_panel_limbs(d) > 1 ? 1 : 0;  // weights of 5..8 bits: the image holds the codes, one exponent per row (lqer_i8_prepare checked)
    if (g.w_i8codes) g.i8_shift = 0;
    g.w8 = (const uint8_t*)(uintptr_t)1;  // (route query: "the image exists"; lqer_linear_gemm sets the real pointer)
  }
  return LQER_OK;
}

int lqer_gemm_route(const lqer_linear_desc_t* d, int64_t M, int dtype) {
  if (!d || M < 0) {
    set_error("gemm_route: bad argument");
    return LQER_E_INVALID;
  }
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  const int rc = gemm_shape_args(d, M, dtype, g);
  if (rc) return rc;
  return gemm_route(g, d->rank > 0);
}

int lqer_gemm_tile_rows(const lqer_linear_desc_t* d, int64_t M, int dtype) {
  if (!d || M < 0) {
    set_error("gemm_tile_rows: bad argument");
    return LQER_E_INVALID;
  }
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  const int rc = gemm_shape_args(d, M, dtype, g);
  if (rc) return rc;
  const int route = gemm_route(g, d->rank > 0);
  if (route < 0) return route;
  switch (route) {
    case LQER_ROUTE_SMALLM: return 0;
    case LQER_ROUTE_TILE256: return 256;
    case LQER_ROUTE_I8: return i8_tile_rows(g);
    default: return gemm_tile_rows(g);
  }
}

static int linear_gemm_impl(const lqer_linear_desc_t* d, const void* xq, int64_t M, const void* w_packed, const void* xaq,
                            int64_t xaq_ld, const void* b_t, int b_limbs, const float* bias_q, void* y, int dtype, int64_t ldy,
                            void* scratch, size_t scratch_bytes, void* stream, bool amax_zeroed);

int lqer_linear_gemm_ld(const lqer_linear_desc_t* d, const void* xq, int64_t M, const void* w_packed, const void* xaq,
                        int64_t xaq_ld, const void* b_t, int b_limbs, const float* bias_q, void* y, int dtype, int64_t ldy,
                        void* scratch, size_t scratch_bytes, void* stream) {
  return linear_gemm_impl(d, xq, M, w_packed, xaq, xaq_ld, b_t, b_limbs, bias_q, y, dtype, ldy, scratch, scratch_bytes, stream, false);
}

int lqer_linear_gemm_prepared(const lqer_linear_desc_t* d, const void* xq, int64_t M, const void* w_packed, const void* xaq,
                              const void* b_t, int b_limbs, const float* bias_q, void* y, int dtype, int64_t ldy, void* scratch,
                              size_t scratch_bytes, size_t ready_bytes, void* stream) {
  bool zeroed = false;
  if (d && ready_bytes > 0) {  // (covers this launch's cells?  a_limbs = -1: the request's own condition, met by whoever prepared them)
    const AmaxZeroReq zr = amax_zero_request(d, M, dtype, -1, scratch);
    zeroed = zr.bytes > 0 && ready_bytes >= zr.bytes;
  }
  return linear_gemm_impl(d, xq, M, w_packed, xaq, d ? lqer_padded_r(d->rank) * xa_limbs(d) : 0, b_t, b_limbs, bias_q, y, dtype, ldy,
                          scratch, scratch_bytes, stream, zeroed);
}

static int linear_gemm_impl(const lqer_linear_desc_t* d, const void* xq, int64_t M, const void* w_packed, const void* xaq,
                            int64_t xaq_ld, const void* b_t, int b_limbs, const float* bias_q, void* y, int dtype, int64_t ldy,
                            void* scratch, size_t scratch_bytes, void* stream, bool amax_zeroed) {
  if (!d || !xq || !w_packed || !y || M < 0 || ldy < d->out_features) {
    set_error("linear_gemm: bad argument");
    return LQER_E_INVALID;
  }
  const bool lowrank = d->rank > 0;
  const bool from_partials = lowrank && !xaq && b_t && scratch && (decode_partials_ok(d, M) || tile_partials_ok(d, M, dtype));
  if (lowrank && ((!xaq && !from_partials) || !b_t)) {
    set_error("linear_gemm: rank %d but no side-path operands (xaq == NULL: only the decode route, see lqer_decode_partials)", d->rank);
    return LQER_E_INVALID;
  }
  if (b_limbs < 0 || b_limbs > 3) {
    set_error("linear_gemm: b_limbs %d outside [0,3]", b_limbs);
    return LQER_E_INVALID;
  }
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  const int rc = gemm_shape_args(d, M, dtype, g);
  if (rc) return rc;
  g.xq = (const bf16_t*)xq;
  g.wp = (const uint8_t*)w_packed;
  g.xaq = (const bf16_t*)xaq;
  if (x_is_i8(d)) {
    g.w8 = (const uint8_t*)w_packed + i8_image_offset(d);
    g.xscale = i8_row_scales(xq, M, d->in_features);
  }
  const int al = lowrank ? xa_limbs(d) : 1;
  if (lowrank && !from_partials && (xaq_ld < lqer_padded_r(d->rank) * al || xaq_ld % 8 != 0 || ((uintptr_t)xaq & 15) != 0)) {
    set_error("linear_gemm: xaq row stride %lld (elements) must be a multiple of 8 and at least the padded rank %lld, "
              "xaq 16-byte aligned", (long long)xaq_ld, (long long)lqer_padded_r(d->rank));
    return LQER_E_INVALID;
  }
  g.xaq_ld = (int)xaq_ld;
  g.bt = (const bf16_t*)b_t;
  g.bias = d->has_bias ? bias_q : nullptr;
  g.y = y;
  g.ldy = ldy;
  g.b_limbs = b_limbs;
  if (from_partials) {
    xa_fused_plan(M, d->in_features, d->rank, &g.xa_nchunk, &g.xa_cstride);
    if (scratch_bytes < (size_t)g.xa_nchunk * g.xa_cstride * sizeof(float)) {
      set_error("linear_gemm: scratch %zu B does not hold the partial tiles of x A", scratch_bytes);
      return LQER_E_WORKSPACE;
    }
    g.xa_part = (const float*)scratch;
    g.aout = make_qp(d->a_out_fmt);
  }
  g.amax_zeroed = amax_zeroed ? 1 : 0;
  return gemm_dispatch(g, dtype, lowrank, scratch, scratch_bytes, (hipStream_t)stream);
}

int lqer_linear_forward(const lqer_linear_desc_t* d, const void* x, int dtype, int64_t M, int64_t ldx,
                        const void* w_packed, const void* a_t, const void* b_t, int a_limbs, int b_limbs,
                        const float* bias_q, void* y, int64_t ldy, void* workspace, size_t workspace_bytes,
                        void* stream) {
  if (!d) {
    set_error("linear_forward: null descriptor");
    return LQER_E_INVALID;
  }
  HT_START();
  lqer_linear_sizes_t sz;
  int rc = lqer_linear_sizes(d, M, &sz);
  if (rc) return rc;
  HT_MARK(0);
  lqer_linear_desc_t plain;
  if (x_is_i8(d) && lqer_gemm_route(d, M, dtype) != LQER_ROUTE_I8) {
    // token counts the int8 tile kernel does not serve run the bf16 kernels on the sign-magnitude image (same buffers)
    plain = *d;
    plain.x_fmt.kind = LQER_Q_MXINT;
    d = &plain;
  }
  if (workspace_bytes < sz.workspace || (!workspace && sz.workspace)) {
    set_error("linear_forward: workspace %zu B < %zu B needed for M=%lld", workspace_bytes, sz.workspace, (long long)M);
    return LQER_E_WORKSPACE;
  }
  if (M == 0) return LQER_OK;
  const size_t Kp = lqer_padded_k(d->in_features), Mp = lqer_padded_m(M);
  unsigned char* ws = (unsigned char*)workspace;
  void* xq = ws;
  const size_t rp = lqer_padded_r(d->rank);
  const size_t xl = act_limbs(d), al = xa_limbs(d);
  HT_MARK(1);
  if (dtype == LQER_F16 && f16_image_is_input(d, x, M, ldx)) xq = const_cast<void*>(x);  // no copy (never written)
  HT_MARK(2);
  const size_t act = act_image_bytes(d, M);
  void* xaq = ws + act;
  void* xa_scratch = ws + act + align_up(Mp * rp * 2 * al, 256);
  (void)Kp, (void)xl;
  if (decode_partials_ok(d, M) && a_t && b_t) {
    HT_MARK(3);
    const size_t nscr = lqer_lowrank_xa_scratch_bytes(d, M);
    HT_MARK(4);
#ifndef LQER_NO_DECODE1
    // up to 8 tokens: ONE launch (decode1.hip) - producer workgroups publish the partial tiles of x A, the weight-streaming
    // workgroups quantize x themselves and pick the tiles up at their very end
    const int esz = dtype == LQER_F32 ? 4 : 2;
    // (capturable: the kernel mixes its dispatch id into the granule tag, so a replayed graph node never accepts the
    // previous replay's tiles although its arguments - the host's per-call counter among them - are frozen)
    if (M <= 8 && a_limbs == 1 && !x_is_f16(d) && ((uintptr_t)x & 15) == 0 &&
        (ldx * esz) % 16 == 0 && b_limbs >= 1 && b_limbs <= 3) {
      GemmArgs g;
      memset(&g, 0, sizeof(g));
      rc = gemm_shape_args(d, M, dtype, g);
      if (rc) return rc;
      HT_MARK(5);
      g.wp = (const uint8_t*)w_packed;
      g.bt = (const bf16_t*)b_t;
      g.bias = d->has_bias ? bias_q : nullptr;
      g.y = y;
      g.ldy = ldy;
      g.b_limbs = b_limbs;
      g.aout = make_qp(d->a_out_fmt);
      const int bout = d->b_out_fmt.kind == LQER_Q_PASSTHROUGH ? 0 : 1;  // (decode_partials_ok: pass-through or blocks of 16)
      const DecodeMember one{g.wp, g.bt, g.bias, g.y, g.ldy, g.N, g.Np, g.rp, g.b_limbs};
      rc = decode1_dispatch(g, dtype, x, ldx, d->in_features, make_qp(d->x_fmt), (const bf16_t*)a_t, bout, &one, 1, xa_scratch, nscr,
                            (hipStream_t)stream);
      HT_MARK(6);
      HT_DONE();
      if (rc != LQER_E_UNSUPPORTED) return rc;
    }
#endif
    // two launches: the GEMM sums the partial tiles of x A itself
    rc = lqer_quantize_act_xa(d, x, dtype, M, ldx, a_t, a_limbs, xq, nullptr, xa_scratch, nscr, stream);
    if (rc) return rc;
    return lqer_linear_gemm(d, xq, M, w_packed, nullptr, b_t, b_limbs, bias_q, y, dtype, ldy, xa_scratch, nscr, stream);
  }
#ifndef LQER_NO_TILE_PARTIALS
  if (a_t && b_t && a_limbs == 1 && tile_partials_ok(d, M, dtype)) {  // two launches: no reduce kernel in between
    const size_t nscr = lqer_lowrank_xa_scratch_bytes(d, M);
    rc = lqer_quantize_act_xa(d, x, dtype, M, ldx, a_t, a_limbs, xq, nullptr, xa_scratch, nscr, stream);
    if (rc) return rc;
    return lqer_linear_gemm(d, xq, M, w_packed, nullptr, b_t, b_limbs, bias_q, y, dtype, ldy, xa_scratch, nscr, stream);
  }
#endif
  // (a pre-pass on atomicMax cells behind the one-launch int8 activation kernel: that kernel zeroes the cells - the two calls share the
  // scratch, whose size is the larger of their needs - instead of a memset launch between them)
  AmaxZeroReq zr = amax_zero_request(d, M, dtype, a_limbs, b_t ? xa_scratch : nullptr);
  const size_t side_bytes = lqer_lowrank_xa_scratch_bytes(d, M), gemm_bytes = lqer_linear_gemm_scratch_bytes(d, M);
  rc = quantize_act_xa_impl(d, x, dtype, M, ldx, a_t, a_limbs, xq, xaq, xa_scratch, side_bytes, stream, &zr);
  if (rc) return rc;
  return linear_gemm_impl(d, xq, M, w_packed, d->rank > 0 ? xaq : nullptr, lqer_padded_r(d->rank) * xa_limbs(d), b_t, b_limbs, bias_q, y, dtype,
                          ldy, xa_scratch, gemm_bytes, stream, zr.done);
}

size_t lqer_group_workspace_bytes(int64_t K, int64_t rank_padded_sum) {
  return (K > 0 && rank_padded_sum > 0) ? (decode1_scratch_bytes(lqer_padded_k(K), (int)rank_padded_sum) + 255) / 256 * 256 : 0;
}

int lqer_linear_forward_group(const lqer_group_member_t* members, int n_members, const void* x, int dtype, int64_t M, int64_t ldx,
                              const void* a_t_cat, int a_limbs, void* workspace, size_t workspace_bytes, void* stream) {
  if (!members || n_members < 1 || !x || !a_t_cat || !workspace || M < 0) {
    set_error("linear_forward_group: bad argument");
    return LQER_E_INVALID;
  }
  if (M == 0) return LQER_OK;
  const lqer_linear_desc_t* d0 = members[0].desc;
  const int esz = dtype == LQER_F32 ? 4 : 2;
  auto same = [](const lqer_qfmt_t& a, const lqer_qfmt_t& b) {
    return a.kind == b.kind && a.width == b.width && a.block == b.block && a.exp_width == b.exp_width && a.exp_bias == b.exp_bias;
  };
  // the one-launch route's conditions (lqer_linear_forward), for every member; anything else: the caller's member-by-member loop
  bool ok = n_members >= 2 && n_members <= 4 && M <= 8 && a_limbs == 1 && d0 && !x_is_f16(d0) && ((uintptr_t)x & 15) == 0 &&
            (ldx * esz) % 16 == 0 && ((uintptr_t)workspace & 15) == 0;
  DecodeMember mem[4];
  int64_t rp_all = 0;
  // malformed members are errors, not "outside the route" (what lqer_linear_forward gets from lqer_linear_sizes / gemm_shape_args)
  for (int i = 0; i < n_members && i < 4; ++i) {
    const lqer_linear_desc_t* d = members[i].desc;
    if (!d || d->in_features <= 0 || d->out_features <= 0 || d->rank < 0) {
      set_error("linear_forward_group: member %d: bad descriptor", i);
      return LQER_E_INVALID;
    }
    if (!fmt_ok(&d->w_fmt, "w_quantizer", 4) || !fmt_ok(&d->x_fmt, "x_quantizer", 9) || !fmt_ok(&d->a_out_fmt, "A_out_quantizer", 9) ||
        !fmt_ok(&d->b_out_fmt, "B_out_quantizer", 24))
      return LQER_E_UNSUPPORTED;
    if (d->has_bias && !members[i].bias_q) {
      set_error("linear_forward_group: member %d: has_bias = 1 but bias_q == NULL", i);
      return LQER_E_INVALID;
    }
  }
  for (int i = 0; ok && i < n_members; ++i) {
    const lqer_group_member_t& m = members[i];
    const lqer_linear_desc_t* d = m.desc;
    ok = d && m.w_packed && m.b_t && m.y && decode_partials_ok(d, M) && d->in_features == d0->in_features && same(d->x_fmt, d0->x_fmt) &&
         same(d->a_out_fmt, d0->a_out_fmt) && same(d->b_out_fmt, d0->b_out_fmt) && m.b_limbs >= 1 && m.b_limbs <= 3 &&
         m.ldy >= d->out_features && ldx >= d->in_features;
    if (!ok) break;
    mem[i] = DecodeMember{(const uint8_t*)m.w_packed, (const bf16_t*)m.b_t, d->has_bias ? m.bias_q : nullptr, m.y, m.ldy, d->out_features,
                          (int)lqer_padded_n(d->out_features), (int)lqer_padded_r(d->rank), m.b_limbs};
    rp_all += lqer_padded_r(d->rank);
  }
  if (!ok || rp_all > 128) {
    set_error("linear_forward_group: outside the one-launch decode route (2..4 members with equal K and x / A_out / B_out formats, "
              "lqer_decode_partials, M <= 8, one limb of A, aligned x): run the members one by one");
    return LQER_E_UNSUPPORTED;
  }
  if (workspace_bytes < lqer_group_workspace_bytes(d0->in_features, rp_all)) {
    set_error("linear_forward_group: workspace %zu B < %zu B", workspace_bytes, lqer_group_workspace_bytes(d0->in_features, rp_all));
    return LQER_E_WORKSPACE;
  }
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  int rc = gemm_shape_args(d0, M, dtype, g);  // the shared part: M, Kp, formats
  if (rc) return rc;
  g.aout = make_qp(d0->a_out_fmt);
  const int bout = d0->b_out_fmt.kind == LQER_Q_PASSTHROUGH ? 0 : 1;  // (decode_partials_ok: pass-through or blocks of 16)
  rc = decode1_dispatch(g, dtype, x, ldx, d0->in_features, make_qp(d0->x_fmt), (const bf16_t*)a_t_cat, bout, mem, n_members, workspace,
                        workspace_bytes, (hipStream_t)stream);
  if (rc == LQER_E_UNSUPPORTED) set_error("linear_forward_group: shape outside the one-launch decode kernel (K too long for its LDS image)");
  return rc;
}

int lqer_desc_limbs(const lqer_linear_desc_t* d, int* act, int* xa) {
  if (!d || !passthrough_width_ok(d->x_fmt, "x_quantizer") || (d->rank > 0 && !passthrough_width_ok(d->a_out_fmt, "A_out_quantizer"))) {
    if (!d) set_error("desc_limbs: null descriptor");
    return LQER_E_INVALID;
  }
  if (act) *act = act_limbs(d);
  if (xa) *xa = d->rank > 0 ? xa_limbs(d) : 1;
  return LQER_OK;
}

int lqer_decode_partials(const lqer_linear_desc_t* d, int64_t M) { return decode_partials_ok(d, M) ? 1 : 0; }

int lqer_tile_partials(const lqer_linear_desc_t* d, int64_t M, int dtype) { return tile_partials_ok(d, M, dtype) ? 1 : 0; }

int lqer_f16_prepare(const void* w_packed, int64_t N, int64_t K, const void* a_t_limbs, int a_limbs, int64_t r, void* a_t_f16,
                     int32_t* flags, void* stream) {
  if (!w_packed || !flags || N <= 0 || K <= 0 || r < 0 || (r > 0 && (!a_t_limbs || !a_t_f16 || a_limbs < 0 || a_limbs > 3))) {
    set_error("f16_prepare: bad argument");
    return LQER_E_INVALID;
  }
  const int rc = f16_prepare_dispatch(w_packed, N, K, a_t_limbs, a_limbs, r, a_t_f16, flags, (hipStream_t)stream);
  if (rc || r <= 0) return rc;
  return a_frag_dispatch(a_t_f16, K, r, (hipStream_t)stream);  // the fragment-major copy behind [rp][Kp] (lqer_a_f16_image_bytes)
}

size_t lqer_a_b16_image_bytes(int64_t K, int64_t r) { return (K > 0 && r > 0) ? a_b16_image_bytes(K, r) : 0; }

int lqer_a_b16_prepare(const void* a_t_limbs, int64_t K, int64_t r, void* out, void* stream) {
  if (!a_t_limbs || !out || K <= 0 || r <= 0) {
    set_error("a_b16_prepare: bad argument");
    return LQER_E_INVALID;
  }
  return a_b16_prepare_dispatch(a_t_limbs, K, r, out, (hipStream_t)stream);
}

size_t lqer_a_f16_image_bytes(int64_t K, int64_t r) {
  return (K > 0 && r > 0) ? a_f16_image_bytes(K, r) : 0;
}

size_t lqer_matmul_q_workspace_bytes(int64_t batch, int64_t K, int64_t S2) {
  return (batch > 0 && K > 0 && S2 > 0) ? qmatmul_workspace_bytes(batch, K, S2) : 0;
}

size_t lqer_matmul_q_workspace_bytes_fmt(int64_t batch, int64_t S1, int64_t K, int64_t S2, const lqer_qfmt_t* x_fmt, const lqer_qfmt_t* y_fmt) {
  if (batch <= 0 || K <= 0 || S2 <= 0 || S1 < 0 || !x_fmt || !y_fmt) return 0;
  return qmatmul_workspace_bytes_ex(batch, S1, K, S2, x_fmt->block != 16, y_fmt->block != 16);
}

int lqer_matmul_q(const void* x, const void* y, void* out, int dtype, int64_t batch, int64_t S1, int64_t K, int64_t S2, int64_t x_bs,
                  int64_t x_rs, int64_t y_bs, int64_t y_ks, int64_t y_js, const lqer_qfmt_t* x_fmt, const lqer_qfmt_t* y_fmt,
                  void* workspace, size_t workspace_bytes, void* stream) {
  if (batch < 0 || S1 < 0 || K <= 0 || S2 < 0) {
    set_error("matmul_q: bad shape batch=%lld S1=%lld K=%lld S2=%lld", (long long)batch, (long long)S1, (long long)K, (long long)S2);
    return LQER_E_INVALID;
  }
  if (batch == 0 || S1 == 0 || S2 == 0) return LQER_OK;
  if (!x || !y || !out || !workspace) {
    set_error("matmul_q: null pointer");
    return LQER_E_INVALID;
  }
  if (!fmt_ok(x_fmt, "matmul x_quantizer", 8) || !fmt_ok(y_fmt, "matmul w_quantizer", 8)) return LQER_E_UNSUPPORTED;
  auto blk_ok = [](const lqer_qfmt_t* f, int64_t cols) { return f->block <= 0 || f->block >= cols || f->block % 16 == 0; };
  if (x_fmt->kind != LQER_Q_MXINT || y_fmt->kind != LQER_Q_MXINT || !blk_ok(x_fmt, K) || !blk_ok(y_fmt, S2)) {
    set_error("matmul_q: both quantizers must be block_fp with blocks of 16 n elements, or whole rows, along the last dim (got kinds %d / %d, "
              "blocks %d / %d)", x_fmt->kind, y_fmt->kind, x_fmt->block, y_fmt->block);
    return LQER_E_UNSUPPORTED;
  }
  if (x_rs < K || (y_ks != 1 && y_js != 1)) {
    set_error("matmul_q: x rows must be dense along k (row stride %lld < K) and y dense along k or along j (strides %lld, %lld)",
              (long long)x_rs, (long long)y_ks, (long long)y_js);
    return LQER_E_INVALID;
  }
  // blocks other than 16: the standalone quantizer writes the operand's bf16 image first - rows evenly spaced over the batch,
  // y dense along j (its blocks' dim)
  const bool x_pre = x_fmt->block != 16, y_pre = y_fmt->block != 16;
  if ((x_pre && batch > 1 && x_bs != S1 * x_rs) || (y_pre && (y_js != 1 || (batch > 1 && y_bs != K * y_ks)))) {
    set_error("matmul_q: operands with blocks other than 16 must have evenly spaced rows over the batch (x_bs == S1 x_rs; y dense along j with "
              "y_bs == K y_ks)");
    return LQER_E_INVALID;
  }
  if (workspace_bytes < qmatmul_workspace_bytes_ex(batch, S1, K, S2, x_pre, y_pre)) {
    set_error("matmul_q: workspace %zu B < %zu B (lqer_matmul_q_workspace_bytes_fmt)", workspace_bytes,
              qmatmul_workspace_bytes_ex(batch, S1, K, S2, x_pre, y_pre));
    return LQER_E_WORKSPACE;
  }
  return qmatmul_dispatch(x, y, out, dtype, batch, S1, K, S2, x_bs, x_rs, y_bs, y_ks, y_js, make_qp(*x_fmt), make_qp(*y_fmt), workspace,
                          (hipStream_t)stream);
}

int lqer_replicate_rows(const void* src, void* dst, int64_t rows, int64_t row_bytes, int copies, void* stream) {
  if (!src || !dst || rows < 0 || row_bytes < 0 || copies < 1) {
    set_error("replicate_rows: bad argument");
    return LQER_E_INVALID;
  }
  if (rows == 0 || row_bytes == 0) return LQER_OK;
  for (int c = 0; c < copies; ++c) {
    const hipError_t e = hipMemcpy2DAsync((char*)dst + (size_t)c * row_bytes, (size_t)copies * row_bytes, src, (size_t)row_bytes,
                                          (size_t)row_bytes, (size_t)rows, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) {
      set_error("replicate_rows: %s", hipGetErrorString(e));
      return LQER_E_LAUNCH;
    }
  }
  return LQER_OK;
}

}  // extern "C"
