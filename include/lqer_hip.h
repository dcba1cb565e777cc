/*
 * lqer_hip.h - C ABI of liblqer_hip.so: the MI355X (gfx950) implementation of the LQER quantized
 * Linear forward      y = Q_x(x) W_q^T + b_q + Q_Bout( Q_Aout( Q_x(x) A ) B )
 *
 * The reference (ChengZhang-98/lqer) is pure Python and has no FFI: its boundary for this path is
 * the nn.Module `LinearFlexibleLqer` (src/lqer/quantize/quantized_layers/linear.py:112-166).  Each
 * entry point below replaces one step of that module's forward; the reference line it stands in
 * for is cited on the declaration.  The Python mirror of the module (lqer_amd/linear.py) binds
 * these with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless named host_*;
 *  - kernels are launched on the calling thread's CURRENT HIP device: the caller makes the device that owns the buffers
 *    current (hipSetDevice) before the call - the Python mirror does (lqer_amd/linear.py, ops.py);
 *  - the caller owns every buffer (incl. workspace); the library never allocates, frees or
 *    retains pointers, and is re-entrant (its only process-wide state are std::once_flag-guarded, idempotent kernel
 *    attribute settings, one atomic call counter - the tag of the one-launch decode route, see lqer_linear_forward - and
 *    the thread-local error text; kernel-selection knobs for tests travel in the descriptor: lqer_linear_desc_t.tuning);
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default
 *    stream) and performs no host synchronisation, so calls may be captured in a hipGraph;
 *  - return value 0 = success, <0 = error (LQER_E_*); lqer_last_error() gives the text of the
 *    last error raised on the calling thread.
 *  - "MXINT(w, L)" = the reference's block_fp format (quantizers/block_fp.py:7-82): blocks of L
 *    consecutive elements along the last dim share an exponent e = ceil(log2(max|block|)); each
 *    element is a sign and a (w-1)-bit magnitude m; value = +-m * 2^(e-(w-1)).
 */
#ifndef LQER_HIP_H
#define LQER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LQER_ABI_VERSION 13

/* error codes */
#define LQER_OK 0
#define LQER_E_INVALID (-1)     /* bad argument (null pointer, negative size, misalignment) */
#define LQER_E_UNSUPPORTED (-2) /* format / shape outside what the kernels implement       */
#define LQER_E_LAUNCH (-3)      /* HIP reported an error at launch                          */
#define LQER_E_WORKSPACE (-4)   /* workspace too small                                     */

/* element types of caller tensors */
#define LQER_F32 0
#define LQER_F16 1
#define LQER_BF16 2

/* quantizer kinds (reference quantizers/__init__.py:7-18) */
#define LQER_Q_PASSTHROUGH 0
#define LQER_Q_MXINT 1 /* "block_fp" */
#define LQER_Q_PASSTHROUGH_F16 2 /* x_fmt only: pass-through fp16 activations multiplied natively (v_mfma_*_f16) - needs
                                    lqer_f16_prepare to report both operands exact, see "pass-through activations" */

#define LQER_Q_MXINT_I8 3 /* x_fmt only: block_fp activations with ONE block per row (width <= 8) carried as int8 mantissas and
                            multiplied on the int8 MFMA - needs lqer_i8_prepare to report the weight eligible, see "int8 route" */

#define LQER_Q_INT 4 /* "integer" (reference quantizers/integer.py:10-43): fixed point, clamp(rne(x 2^frac), lo, hi) / 2^frac with
                        lo, hi = -2^(width-1), 2^(width-1)-1 (signed) or 0, 2^width-1.  Fields: width; exp_bias = frac_width;
                        exp_width = is_signed (1 / 0); block is ignored.  Implemented for the x, b and A_out quantizers (width
                        <= 9 signed / 8 unsigned, so that every value is a bf16 number), for B_out (any width <= 24: applied to
                        the fp32 side product inside the tile kernels' prologues - the reference's fall-back when x_quantizer is
                        integer, linear.py:115-119; decode sizes then take the tile kernel) and in lqer_quantize_mxint; as
                        w_quantizer: signed, width 2..4 - the codes -8..7 travel as two's-complement nibbles in the same panels
                        (lqer_pack_weight_mxint; the 128-row tile kernel at every token count); unsigned or wider integer
                        weights: LQER_E_UNSUPPORTED */

/* Geometry of the packed operands (fixed by the kernels; exported so callers can size buffers). */
#define LQER_K_ALIGN 64     /* K is zero-padded to a multiple of this                        */
#define LQER_M_ALIGN 256    /* activation workspaces are row-padded to a multiple of this    */
#define LQER_N_ALIGN 256    /* packed weights are row-padded to a multiple of this           */
#define LQER_R_ALIGN 16     /* rank is zero-padded to a multiple of this                     */
#define LQER_PANEL_ROWS 16  /* packed W: panels of 16 rows x 64 k                            */
#define LQER_PANEL_BYTES 576 /* 16*32 B of 4-bit codes + 16*4 B of block exponents           */

/* One MXINT quantizer: `width` bits per element incl. sign; `block` elements share an exponent
 * (block <= 0: one block per row); exponent clamped to [-exp_bias, 2^exp_width-1-exp_bias]
 * (block_fp.py:46-51; exp_width=8, exp_bias=127 in every template config). */
typedef struct lqer_qfmt {
  int32_t kind;      /* LQER_Q_* */
  int32_t width;     /* passthrough (x / A_out): significand bits to carry - 8 bf16, 11 fp16, 24 fp32 - see
                        "pass-through activations" below; ignored for passthrough b / B_out               */
  int32_t block;
  int32_t exp_width;
  int32_t exp_bias;
} lqer_qfmt_t;

/* Static description of one Linear.  M (tokens) is a per-call argument. */
typedef struct lqer_linear_desc {
  int32_t in_features;   /* K */
  int32_t out_features;  /* N */
  int32_t rank;          /* r; 0 = no side path (LinearFlexible, linear.py:88-109) */
  int32_t has_bias;
  lqer_qfmt_t x_fmt;     /* linear.py:148   x_quantizer                        */
  lqer_qfmt_t w_fmt;     /* linear.py:150   w_quantizer (block_fp width <= 8: 5..8 bits see "weights of 5..8 bits"; integer <= 4) */
  lqer_qfmt_t b_fmt;     /* linear.py:152   b_quantizer                        */
  lqer_qfmt_t a_out_fmt; /* linear.py:154   A_out_quantizer                    */
  lqer_qfmt_t b_out_fmt; /* linear.py:155   B_out_quantizer                    */
  int32_t tuning;        /* 0 = the library decides everything (what every caller wants).  LQER_TUNE_* bits: per-CALL kernel
                            selection knobs for tests and measurements - they change which kernel variant runs, never a result
                            bit.  Carried by the descriptor, not by process-wide state: two threads never see each other's. */
} lqer_linear_desc_t;

/* lqer_linear_desc_t.tuning (all variants give the same bits) */
#define LQER_TUNE_TILE_ROWS_128 0x1   /* 128-row kernel family (LQER_ROUTE_TILE128): always 128-row tiles                     */
#define LQER_TUNE_TILE_ROWS_64 0x2    /* ... always 64-row tiles (two workgroups per CU); default: 64 rows when the 128-row grid
                                         covers at most half of the CUs                                                        */
#define LQER_TUNE_XCD_BLOCK(t) (((t) & 0x3f) << 4) /* XCD-local tile BLOCKS of `t` token tiles x (tiles / 8 / t) weight tiles in the
                                         128-row kernel instead of rows of weight tiles; applied only where the tile grid divides
                                         (16 x 16 tiles: 8, 4 or 16); measured +-0 (the weight stream through every XCD's L2 is
                                         served by the Infinity Cache)                                                        */
#define LQER_TUNE_I8_ROWS_128 0x4      /* int8 kernel (LQER_ROUTE_I8): always 128-row tiles                                      */
#define LQER_TUNE_I8_ROWS_256 0x8      /* ... always 256-row tiles; default: whichever takes fewer weighted rounds of one tile per
                                         CU (lqer_gemm_tile_rows says which)                                                    */
#define LQER_TUNE_AMAX_ATOMIC 0x40000  /* int8 route, B_out with one block per row: the pre-pass of row maxima with one atomicMax cell per
                                         row behind a zero-fill launch (the round-4 form) instead of per-segment partial maxima in plain
                                         stores that the GEMM folds (no zero-fill launch); max is order-independent: same bits         */
#define LQER_TUNE_AMAX_PARTS 0x80000   /* ... the segment partials at every N (default: up to N = 4096, where they are faster)     */
#define LQER_TUNE_AMAX_XCH_MISS 0x100000 /* int8 route, ONE round of 128-row tiles (at most one tile per CU) and one B_out block per row:
                                         by default there is NO pre-pass launch - every workgroup publishes the row maxima of its own
                                         tile's side product as tagged granules and gathers its row band's at the epilogue (either
                                         AMAX pin above restores the pre-pass).  This bit makes every workgroup treat the others'
                                         granules as missing, i.e. take the fall-back that computes the whole band's maxima itself (a
                                         workgroup that times out on a neighbour does the same): same bits                          */
#define LQER_TUNE_XA_REDUCE_IN_GEMM 0x20000 /* lqer_linear_forward on 128-row tiles: no reduce launch between the quantizer and the
                                         GEMM - its workgroups sum the partial tiles of x A for their own rows (lqer_tile_partials).
                                         Off by default: measured slower (C2: the GEMM grows by 5.3 us, the launch it saves took
                                         4.9 us - 16 partial tiles at K = 4096, summed in front of the main loop, whose accumulators it opens)           */
#define LQER_TUNE_ACT8_SPLIT 0x200000 /* int8 route (LQER_Q_MXINT_I8), lqer_quantize_act_xa / lqer_linear_forward: quantizer, split-K side GEMM
                                         and reduce as three launches even where the one-launch kernel applies (16-bit tensors, a_limbs =
                                         -1, padded rank 16 / 32 / 64, M <= 4096): bit-identical int8 image and row scales, x A summed in
                                         another order (A_out of it inside the summation-order envelope)                            */
#define LQER_TUNE_ACT8_FUSED 0x400000 /* ... the one-launch kernel at every token count (default: M <= 4096 - every workgroup of 8 token
                                         rows streams the whole A^T image)                                                          */
#define LQER_TUNE_ACT16_SPLIT 0x800000 /* block-16 MXINT activations with a_limbs = -2: the quantizer + split-K kernel and the reduce as two
                                         launches even where the one-launch kernel applies (act16_fused.hip; same image, x A in another
                                         summation order)                                                                            */
#define LQER_TUNE_ACT16_FUSED 0x1000000 /* ... the one-launch kernel at every token count (default: 1024 <= M <= 4096; the same window
                                         applies to LQER_TUNE_ACT8_FUSED's kernel)                                                    */
#define LQER_TUNE_AMAX_NO_MRX 0x4000000 /* int8 route, SEVERAL rounds of 128-row tiles on a resident grid (up to 2048 tokens), one B_out block per
                                           row, 4-bit weights: keep the k_bout_amax pre-pass launch (rounds 3-5) instead of the default - the
                                           GEMM's workgroups compute the pre-pass themselves, one item each at their start, and exchange
                                           tagged granules (csrc/gemm_w4a8_i8.hip, MRX)                                                   */
#define LQER_TUNE_BOUT_IN_PROLOGUE 0x2000000 /* 128-row tile kernel, B_out in blocks of 16: re-quantize the side product in front of the main
                                              loop (rounds 1-5) instead of under its first 16 k-steps with the product added behind the
                                              last one (default from K = 1024; csrc/gemm_w4a8.hip, DEFER)                               */
#define LQER_TUNE_DECODE_NO_POLL 0x10000 /* one-launch decode route: no wait for the producers' tiles - every weight-streaming
                                         workgroup computes the partial tiles of x A itself (the bounded wait's fall-back)     */

/* Sizes (bytes) of the derived, caller-allocated device buffers of one Linear. */
typedef struct lqer_linear_sizes {
  size_t w_packed;   /* 4-bit codes + block exponents, panel layout            */
  size_t a_t;        /* A^T as bf16 limbs [3][rp][Kp]                          */
  size_t b_t;        /* B^T as bf16 limbs [3][Np][rp]                          */
  size_t bias_q;     /* quantized bias, fp32 [Np]                              */
  size_t workspace;  /* per-call scratch for `m_max` tokens                    */
} lqer_linear_sizes_t;

int lqer_version(void);
const char* lqer_last_error(void);
/* sizeof of the structs above as THIS library was compiled: a binding in another language asserts its own layouts against
 * these once (the library reads sizeof(lqer_linear_desc_t) bytes behind a descriptor pointer - a shorter struct is undefined
 * behaviour).  INTEGRATION.md section B's ctypes stub does; tests/test_abi_cpu.py executes that stub against the library. */
size_t lqer_sizeof_qfmt(void);
size_t lqer_sizeof_linear_desc(void);
size_t lqer_sizeof_linear_sizes(void);
size_t lqer_sizeof_group_member(void);

/* Padded extents used by the packed buffers. */
int64_t lqer_padded_k(int64_t K);
int64_t lqer_padded_n(int64_t N);
int64_t lqer_padded_m(int64_t M);
int64_t lqer_padded_r(int64_t r);

/* ---- quantizers -------------------------------------------------------------------------- */

/* MXINT quantizer over the rows of a [rows, cols] matrix (row stride `ld` elements), blocks along
 * the last dim.  Replaces x_quantizer / A_out_quantizer / B_out_quantizer / b_quantizer calls of
 * linear.py:148,152,154,155 -> block_fp.py:111.  Any of the outputs may be NULL:
 *   deq_f32 [rows, cols]   dequantized values (|x| <= 1e-8 kept as is, block_fp.py:79-80)
 *   codes   [rows, cols]   signed mantissas, int8 (needs width <= 8); |x| <= 1e-8 -> 0
 *   exps    [rows, ceil(cols/L)]  shared exponents, int8 (0 for an all-zero block)          */
int lqer_quantize_mxint(const void* x, int dtype, int64_t rows, int64_t cols, int64_t ld,
                        const lqer_qfmt_t* fmt, float* deq_f32, int8_t* codes, int8_t* exps,
                        void* stream);

/* The same quantizer over 2-D TILES of an activation - blocks that span token rows: x [batches][rows][cols] dense, tiles of
 * tile_rows x tile_cols (<= 0 or larger than the extent: the whole extent) anchored at (0, 0) of every batch element, ragged
 * edges clipped (the reference pads them with zeros, which cannot raise a block maximum).  Replaces block_fp.py:111 ->
 * quantizers/utils.py:211-237 (`_block_3d_activation`: x / A_out / B_out of a [batch, tokens, features] tensor with
 * block_size [R, L], skip_first_dim = true) and utils.py:161-183 as utils.py:261-270 applies it to a 2-D activation with
 * skip_first_dim = false (batches = 1).  deq_f32 [batches][rows][cols]; amax_scratch: one float per tile
 * (batches * ceil(rows / R) * ceil(cols / L)).  A cold path - no template configuration has such blocks; the fused kernels
 * keep per-row blocks. */
int lqer_quantize_mxint_tiles(const void* x, int dtype, int64_t batches, int64_t rows, int64_t cols, const lqer_qfmt_t* fmt,
                              int64_t tile_rows, int64_t tile_cols, float* deq_f32, float* amax_scratch, void* stream);

/* Activation quantizer of the fast path (linear.py:148): x [M,K] -> exact bf16 image
 * xq [lqer_padded_m(M), lqer_padded_k(K)] (K padding zeroed; |x| <= 1e-8 flushed to 0).
 * Needs fmt->width <= 9 so that every value m*2^(e-w+1) is exactly representable in bf16.   */
int lqer_quantize_act_mxint(const void* x, int dtype, int64_t M, int64_t K, int64_t ldx,
                            const lqer_qfmt_t* fmt, void* xq_bf16, void* stream);

/* ---- one-time packing (replaces the in-place first-forward quantization, linear.py:149-153) */

int lqer_linear_sizes(const lqer_linear_desc_t* desc, int64_t m_max, lqer_linear_sizes_t* out);

/* W [N,K] (row stride ldw) -> packed panels: w_quantizer(W) as 4-bit sign-magnitude mantissas
 * (bit 3 = sign) + block exponents.  Panel (n/16, k/64) = 16 rows x 32 B codes followed by
 * 16 x 4 exponent bytes (one per 16 k; a coarser block repeats its exponent), stored biased:
 * byte = clamp(e - (width-1) + 127, 1, 254).  Within each 32-bit word of codes (8 consecutive k)
 * nibble p holds k = p/2 (p even) or 4 + p/2 (p odd); a row's 8 words are stored in the order
 * {0,2,4,6,1,3,5,7}.  |w| <= 1e-8 is flushed to code 0.  `scratch` needs N*ceil(K/16) bytes.   */
int lqer_pack_weight_mxint(const void* W, int dtype, int64_t N, int64_t K, int64_t ldw,
                           const lqer_qfmt_t* fmt, void* w_packed, void* scratch, void* stream);
/* The same for 2-D tiles (w_quantizer block_size [R, L] with skip_first_dim = false, reference quantizers/utils.py:161-183):
 * one exponent per tile of `block_rows` weight rows x fmt->block k (block_rows <= 0 or >= N: all rows; 1 = the call above).  The
 * image layout does not change - a tile's exponent is repeated for each of its rows - so every GEMM route reads it as it is. */
int lqer_pack_weight_mxint_2d(const void* W, int dtype, int64_t N, int64_t K, int64_t ldw,
                              const lqer_qfmt_t* fmt, int64_t block_rows, void* w_packed, void* scratch,
                              void* stream);

/* Test hook: packed panels -> dequantized fp32 [N,K]. */
int lqer_unpack_weight_mxint(const void* w_packed, int64_t N, int64_t K, const lqer_qfmt_t* fmt,
                             float* w_f32, void* stream);

/* A [K,r] and B [r,N] (linear.py:142-143; values as stored in the state dict) -> transposed bf16
 * limb images a_t [3][rp][Kp], b_t [3][Np][rp]; v = limb0 + limb1 + limb2 exactly (an 8-bit
 * MXINT value needs one limb, fp16 two, fp32 three).  limb_flags[0..1] (device int32[2]) receive
 * the number of non-zero limbs of A and of B.                                                */
int lqer_pack_lowrank(const void* A, const void* B, int dtype, int64_t K, int64_t N, int64_t r,
                      void* a_t, void* b_t, int32_t* limb_flags, void* stream);

/* bias [N] -> b_quantizer(bias) as fp32 [Np] (linear.py:151-152). fmt->kind may be passthrough. */
int lqer_pack_bias(const void* bias, int dtype, int64_t N, const lqer_qfmt_t* fmt, float* bias_q,
                   void* stream);

/* ---- the forward (linear.py:145-157, PTQ branch) ------------------------------------------- */

/* y[M,N] (row stride ldy, same dtype as x) from x[M,K] (row stride ldx).  a_limbs / b_limbs =
 * the limb counts reported by lqer_pack_lowrank.  workspace >= lqer_linear_sizes(...).workspace
 * for m_max >= M.  Two to four stream-ordered launches, no host synchronisation: the activation quantizer (fused
 * with the split-K partials of the rank-r side GEMM when x blocks are 16 and the padded rank <= 64; nothing at all for
 * a dense fp16 tensor on the LQER_Q_PASSTHROUGH_F16 route), a fixed-order reduce of the partials with the A_out
 * re-quantization (taken over by the GEMM at decode sizes, lqer_decode_partials), a pre-pass for B_out blocks other
 * than 16 columns, and the fused W4 GEMM with the B side GEMM, B_out re-quantization, bias and add in its prologue.
 * a_limbs = -1 (LQER_Q_MXINT_I8 descriptors at token counts of LQER_ROUTE_I8 only; also lqer_quantize_act_xa and
 * lqer_lowrank_xa): a_t is ONE fp16 image of A^T [padded rank][padded K] as lqer_f16_prepare writes it (its flag for A
 * clear: every element exact in fp16) - the side GEMM then multiplies the int8 mantissas with it on the fp16 MFMA, half
 * the A^T bytes and MFMAs of the two-limb image. */
int lqer_linear_forward(const lqer_linear_desc_t* desc, const void* x, int dtype, int64_t M,
                        int64_t ldx, const void* w_packed, const void* a_t, const void* b_t,
                        int a_limbs, int b_limbs, const float* bias_q, void* y, int64_t ldy,
                        void* workspace, size_t workspace_bytes, void* stream);

/* Up to 8 tokens (block_fp activations in blocks of 16, A_out in blocks of 16, padded rank <= 64, one limb of A, B_out
 * pass-through or in blocks of 16 - lqer_decode_partials - and 16-byte aligned rows of x) lqer_linear_forward issues ONE
 * launch (csrc/decode1.hip): producer workgroups publish the split-K partial tiles of x A as {value, tag} granules in the
 * workspace, the weight-streaming workgroups quantize x themselves and read the tiles at their very end.  The tag is a
 * per-LAUNCH value: a process-wide atomic call counter (the library's only mutable state besides the thread-local error
 * text) mixed, inside the kernel, with the launch's AQL dispatch id and queue address - so the route is taken under stream
 * capture as well: a replayed graph node carries the captured counter but gets a fresh dispatch id, and never accepts the
 * previous replay's tiles.  The wait for the tiles is bounded; a workgroup that does not see them computes them itself
 * (same bits). */

/* The same, split for callers that share one quantized activation between several Linears
 * (q/k/v, gate/up) and for per-stage timing.  xq = output of lqer_quantize_act_mxint.         */
/* x_quantizer + side path in one call: xq and (rank > 0) xaq.  Uses a fused kernel (the activation
 * image is quantized and multiplied by A from LDS, never re-read from HBM) when x blocks are 16 and the
 * padded rank <= 64 - or <= 128 from 512 tokens on, with 16-byte aligned 16-bit rows -, the two separate
 * steps otherwise (same bits).  scratch as for lqer_lowrank_xa.                                      */
int lqer_quantize_act_xa(const lqer_linear_desc_t* desc, const void* x, int dtype, int64_t M,
                         int64_t ldx, const void* a_t, int a_limbs, void* xq_bf16, void* xaq_bf16,
                         void* scratch, size_t scratch_bytes, void* stream);
size_t lqer_lowrank_xa_scratch_bytes(const lqer_linear_desc_t* desc, int64_t m_max);
/* Bytes of the activation image buffer `xq_bf16` of the split calls for up to m_max tokens (what lqer_linear_sizes counts inside
 * `workspace`): [padded M][padded K x activation limbs x weight limbs] bf16, plus - for weights of 5..8 bits - the single-copy
 * image behind it.  xaq and the scratch follow at 256-byte aligned offsets in lqer_linear_forward's own carving. */
size_t lqer_act_image_bytes(const lqer_linear_desc_t* desc, int64_t m_max);
int lqer_lowrank_xa(const lqer_linear_desc_t* desc, const void* xq_bf16, int64_t M,
                    const void* a_t, int a_limbs, void* xaq_bf16, void* scratch,
                    size_t scratch_bytes, void* stream);
/* scratch of lqer_linear_gemm: 0 unless B_out blocks differ from 16 columns (then the kernel needs
 * the per-row-block maxima of xAq @ B: a pre-pass it launches itself - or, on the int8 route with one block per row and one round
 * of 128-row tiles, tagged granules that the GEMM's own workgroups exchange through this scratch, LQER_TUNE_AMAX_XCH_MISS above).
 * The contents of the scratch need no initialisation and may be anything left by earlier calls. */
size_t lqer_linear_gemm_scratch_bytes(const lqer_linear_desc_t* desc, int64_t m_max);
int lqer_linear_gemm(const lqer_linear_desc_t* desc, const void* xq_bf16, int64_t M,
                     const void* w_packed, const void* xaq_bf16, const void* b_t, int b_limbs,
                     const float* bias_q, void* y, int dtype, int64_t ldy, void* scratch,
                     size_t scratch_bytes, void* stream);
/* The zero fill of lqer_linear_gemm's pre-pass, handed to the activation call in front of it (ABI 13).  Launches whose B_out blocks
 * differ from 16 columns and whose grid is more than one round of tiles start with a pre-pass that folds the row-block maxima of
 * xAq @ B with atomicMax into cells at the head of `scratch`, which lqer_linear_gemm zero-fills first - a memset launch between two
 * kernels (4.8 us of a 116-us forward at 2048 x 4096 -> 11008, rank 32).  lqer_quantize_act_xa_prep is lqer_quantize_act_xa which,
 * where its work runs as the one-launch int8 kernel (csrc/act8_fused.hip: LQER_Q_MXINT_I8, a_limbs = -1, 1024..4096 tokens), lets that
 * kernel write those zeros to `gemm_scratch` (one store per thread) and reports in *ready_bytes how many bytes at its head are ready -
 * 0 when nothing was prepared (another route, or a GEMM that needs none).  lqer_linear_gemm_prepared is lqer_linear_gemm which skips
 * its own zero fill when ready_bytes covers its cells.  The caller's side of the contract: the same gemm_scratch / scratch pointer,
 * the same stream, and nothing else writing that scratch between the two calls (it may alias the first call's own `scratch`: the
 * one-launch kernel does not use it).  lqer_linear_forward does this hand-over on its own; same results either way. */
int lqer_quantize_act_xa_prep(const lqer_linear_desc_t* desc, const void* x, int dtype, int64_t M,
                              int64_t ldx, const void* a_t, int a_limbs, void* xq_bf16, void* xaq_bf16,
                              void* scratch, size_t scratch_bytes, void* gemm_scratch, size_t* ready_bytes, void* stream);
int lqer_linear_gemm_prepared(const lqer_linear_desc_t* desc, const void* xq_bf16, int64_t M,
                              const void* w_packed, const void* xaq_bf16, const void* b_t, int b_limbs,
                              const float* bias_q, void* y, int dtype, int64_t ldy, void* scratch,
                              size_t scratch_bytes, size_t ready_bytes, void* stream);

/* Which GEMM kernel lqer_linear_gemm launches for `M` tokens of this descriptor and element type (the choice depends on
 * nothing else): LQER_ROUTE_SMALLM (M <= 64: one workgroup per 16 output columns streams its packed weight rows),
 * LQER_ROUTE_TILE128 (128 x 256 tiles; 64 x 256 tiles of the same kernel when the 128-row grid would cover at most half of
 * the CUs and the side product is at most two 16-deep slices - M = 65..1024 at N = 4096), LQER_ROUTE_TILE256 (256 x 256 tiles, M >= 512 when that needs fewer rounds of one
 * tile per CU).  < 0: the error lqer_linear_gemm would return.  For benchmarks and tests that must know which kernel
 * they are looking at. */
#define LQER_ROUTE_SMALLM 0
#define LQER_ROUTE_TILE128 1
#define LQER_ROUTE_TILE256 2
#define LQER_ROUTE_I8 3 /* int8 MFMA main loop (x_fmt.kind = LQER_Q_MXINT_I8 only): 256 x 256 tiles, or 128 x 256 tiles where those
                           take fewer weighted rounds of one tile per CU (Llama-7B projections at M = 2048) - lqer_gemm_tile_rows */
int lqer_gemm_route(const lqer_linear_desc_t* desc, int64_t M, int dtype);
/* Token rows of a tile of the kernel lqer_gemm_route names: 64 / 128 (LQER_ROUTE_TILE128 family), 256 (LQER_ROUTE_TILE256),
 * 128 / 256 (LQER_ROUTE_I8); 0 for LQER_ROUTE_SMALLM; < 0: the error.  For tests and benchmarks that assert the variant. */
int lqer_gemm_tile_rows(const lqer_linear_desc_t* desc, int64_t M, int dtype);

/* Decode sizes (launch-bound: each kernel runs ~3 us).  Returns 1 when, for this descriptor and token count, the
 * re-quantized side product never has to be materialised: M <= 64, x and A_out block_fp in blocks of 16 (width <= 9),
 * padded rank <= 64, B_out pass-through or in blocks of 16.  Then lqer_quantize_act_xa may be called with
 * xaq_bf16 == NULL - it leaves the split-K partial tiles of x A in `scratch` and skips the reduce launch - and
 * lqer_linear_gemm with xaq_bf16 == NULL and the SAME scratch buffer (scratch_bytes = lqer_lowrank_xa_scratch_bytes):
 * the GEMM sums the tiles in the same fixed order and applies A_out itself.  lqer_linear_forward does this on its own. */
int lqer_decode_partials(const lqer_linear_desc_t* desc, int64_t M);
/* The same hand-over at the token counts of the 128-row tile kernel, for callers that ask for it (descriptor tuning bit
 * LQER_TUNE_XA_REDUCE_IN_GEMM; 1 when lqer_linear_forward then takes it for M tokens of `dtype`: LQER_ROUTE_TILE128 with
 * 128-row tiles, fp16 / bf16 tensors, x / A_out in blocks of 16, padded rank <= 64, one limb of A, B_out pass-through or in
 * blocks of 16): every GEMM workgroup sums the partial tiles for its own 128 rows in ascending chunk order and applies A_out
 * on the way into the side product's LDS stage - k_xa_reduce4's arithmetic item by item, same bits (csrc/gemm_w4a8.hip,
 * XAPART).  A measured dead end kept selectable: the sum delays the main loop by more than the launch it replaces.  Same
 * calling convention: lqer_quantize_act_xa and lqer_linear_gemm with xaq_bf16 == NULL and the same scratch. */
int lqer_tile_partials(const lqer_linear_desc_t* desc, int64_t M, int dtype);

/* lqer_linear_gemm with an explicit row stride of xaq (elements; a multiple of 8, >= the padded rank): Linears that
 * share one input (q/k/v, gate/up; llama_decoder.py:246-248, :176) can run ONE lqer_quantize_act_xa over the
 * concatenation of their A matrices (descriptor rank = sum of the padded ranks, a_t = the concatenated limb image)
 * and hand each GEMM its columns of the result: xaq = xaq_cat + column offset, xaq_ld = total padded rank. */
int lqer_linear_gemm_ld(const lqer_linear_desc_t* desc, const void* xq_bf16, int64_t M, const void* w_packed,
                        const void* xaq_bf16, int64_t xaq_ld, const void* b_t, int b_limbs, const float* bias_q,
                        void* y, int dtype, int64_t ldy, void* scratch, size_t scratch_bytes, void* stream);

/* ---- decode: Linears that are handed the SAME tokens in ONE launch ---------------------------------------------------
 * q/k/v and gate/up receive one tensor (reference models/llama_decoder.py:222-224, :104; opt_decoder.py q/k/v).  At decode
 * sizes a forward is a chain of latencies around a 9 MB weight stream; a group's members run as ONE launch of the one-launch
 * decode route (see lqer_linear_forward above): the producers multiply x with the concatenation of the members' A, every
 * member's weight-streaming workgroups pick their rank columns out of the shared tiles.  Per member the arithmetic is that of
 * its own lqer_linear_forward: the same bits.
 *   members[i]: the member's descriptor, packed operands (as for lqer_linear_forward) and output y_i [M, N_i] (row stride ldy);
 *   a_t_cat:    A^T of the members concatenated along the rank: bf16 [sum of padded ranks][padded K] - lqer_pack_lowrank of
 *               [A_0 | A_1 | ...] (each A_i zero-padded to its padded rank); a_limbs must be 1 (8-bit MXINT values);
 *   workspace:  >= lqer_group_workspace_bytes(K, sum of padded ranks), 16-byte aligned.
 * Served: 2..4 members with equal in_features and equal x / A_out / B_out formats, each with lqer_decode_partials(desc, M) == 1,
 * M <= 8, 16-byte aligned rows of x, sum of padded ranks <= 128.  Anything else returns LQER_E_UNSUPPORTED WITHOUT launching:
 * the caller then runs the members one by one. */
typedef struct lqer_group_member {
  const lqer_linear_desc_t* desc;
  const void* w_packed;
  const void* b_t;
  int32_t b_limbs;
  const float* bias_q;
  void* y;
  int64_t ldy;
} lqer_group_member_t;
size_t lqer_group_workspace_bytes(int64_t K, int64_t rank_padded_sum);
int lqer_linear_forward_group(const lqer_group_member_t* members, int n_members, const void* x, int dtype, int64_t M,
                              int64_t ldx, const void* a_t_cat, int a_limbs, void* workspace, size_t workspace_bytes,
                              void* stream);

/* ---- pass-through activations ("A16": x_quantizer = passthrough, reference quantizers/passthrough.py:1, every
 * experiments/configs/template/ *-int.toml) ------------------------------------------------------------------
 * The kernels multiply exact bf16 operands.  A pass-through activation is therefore carried as the sum of
 * L = ceil(x_fmt.width / 8) bf16 limbs (1 for a bf16 tensor, 2 for fp16, 3 for fp32) laid side by side along k:
 * xq is [Mp][L*Kp], and w_packed / a_t must hold L copies of the packed image along k.  Likewise a pass-through
 * A_out hands x A on as LA limbs (2 when a_out_fmt.width <= 16, else 3): xaq is [Mp][LA*rp] and b_t holds LA
 * copies along r.  lqer_linear_sizes reports the enlarged sizes; pack into a buffer of the single-copy size (the
 * sizes of a descriptor whose x / A_out formats are block_fp) and expand with lqer_replicate_rows:
 *   w_packed: rows = Np/16,  row_bytes = (Kp/64)*LQER_PANEL_BYTES, copies = L
 *   a_t:      rows = 3*rp,   row_bytes = Kp*2,                      copies = L
 *   b_t:      rows = 3*Np,   row_bytes = rp*2,                      copies = LA
 * Every product stays exact and accumulation is fp32, as in the reference's F.linear on 16-bit tensors. */
int lqer_desc_limbs(const lqer_linear_desc_t* desc, int* act_limbs, int* xa_limbs);
/* fp16 tensors have a faster exact route: x_fmt.kind = LQER_Q_PASSTHROUGH_F16.  The activation image is then the
 * fp16 tensor itself ([Mp][Kp], one copy of w_packed), the main loops expand the weights to fp16 and run the fp16 MFMA
 * (products exact, fp32 accumulation - the reference's F.linear on fp16 tensors), and the side GEMM reads A as ONE
 * fp16 image a_t = [rp][Kp] fp16 written by this call from the limb image of lqer_pack_lowrank - followed (ABI 12) by its
 * FRAGMENT-MAJOR copy for the int8 route's one-launch activation kernel (per 32-k step and 16-rank tile 64 lanes x 8 halves, the B
 * operand of v_mfma_f32_16x16x32_f16, over ceil(K / 128) * 128 columns): a_t_f16 must hold lqer_a_f16_image_bytes(K, r) bytes.
 * Allowed only when
 * this call leaves flags[0] (a weight block scale outside the fp16 range 2^-24 .. 2^13) and flags[1] (an element of
 * A that is not an fp16 number) at zero (device int32[2]); otherwise use LQER_Q_PASSTHROUGH with width 11.
 * A_out / xaq / b_t are as for LQER_Q_PASSTHROUGH.  Calls take dtype = LQER_F16.  A dense tensor (ldx == K) with
 * K % LQER_K_ALIGN == 0, M % LQER_M_ALIGN == 0 (or M <= 64) and 16-byte alignment already is its image: lqer_linear_forward
 * then skips the copy, and the split API accepts xq == x in lqer_quantize_act_xa / lqer_linear_gemm. */
int lqer_f16_prepare(const void* w_packed, int64_t N, int64_t K, const void* a_t_limbs, int a_limbs, int64_t r,
                     void* a_t_f16, int32_t* flags, void* stream);
/* bytes of the fp16 image of A^T that lqer_f16_prepare writes (the [rp][Kp] image every a_limbs = -1 consumer reads + its
 * fragment-major copy); the reference has nothing to replace here - A is an fp16 nn.Parameter (quantized_layers/linear.py:142) */
size_t lqer_a_f16_image_bytes(int64_t K, int64_t r);
/* The same for block-16 MXINT activations (x_fmt block_fp [1, 16], the llama-7b.toml / opt-6.7b.toml templates) and an A of ONE bf16 limb
 * (8-bit block_fp A, llama-7b.toml:60-73): lqer_a_b16_prepare copies limb 0 of lqer_pack_lowrank's a_t ([rp][Kp] bf16) into `out` and
 * writes its fragment-major copy (the B operand of v_mfma_f32_16x16x32_bf16) behind it; out holds lqer_a_b16_image_bytes(K, r) bytes.
 * Passed as a_t with a_limbs = -2 to lqer_quantize_act_xa / lqer_linear_forward, it lets quantizer + x A + A_out run as ONE launch
 * (csrc/act16_fused.hip) for 16-bit, 16-byte aligned tensors with K % 16 == 0, padded rank 16 / 32 / 64 and 1024 <= M <= 4096; every
 * other call reads the image's first part exactly like a_limbs = 1.  Reference: linear.py:154-156 (x_quantizer, matmul, A_out_quantizer). */
size_t lqer_a_b16_image_bytes(int64_t K, int64_t r);
int lqer_a_b16_prepare(const void* a_t_limbs, int64_t K, int64_t r, void* out, void* stream);

/* ---- weights of 5..8 bits (the reference's no-LQER baseline: W8A8 block_fp with one block per row and per token,
 * experiments/pipeline/sweep_baseline_no_lqer.sh:73-76, through LinearFlexible, quantized_layers/linear.py:50-64) ---------------
 * The packed image holds 4-bit codes, so a mantissa m of up to 8 bits (|m| <= 127) travels as three signed base-8 digits,
 * m = 64 a + 8 b + c (a in [-2, 2]; b, c in [-4, 3]): three 4-bit sign-magnitude LIMB images with block exponents e - mbits + 6,
 * + 3, + 0, side by side along k - w_packed is [Np / 16][3][Kp / 64] panels (lqer_linear_sizes: three times the 4-bit size) - and
 * the activation image is repeated three times along k to match (the dual of the pass-through activation limbs below).  Every
 * kernel of the 4-bit path then multiplies the 8-bit weight EXACTLY (digits x powers of two; fp32 accumulation of exact products)
 * at three times the MFMA work and 13.5 bits per weight: the universal route, at every token count and for every x format.
 * lqer_pack_weight_mxint writes the three limbs for w_fmt.width in 5..8; lqer_quantize_act_xa writes the single-copy image behind
 * the wide one (lqer_act_image_bytes) and repeats it; the side path (x A, A_out) works on the single copy.  The one-launch decode
 * route and the shared-input groups do not take such weights (lqer_decode_partials = 0): decode sizes run the two-launch route.
 * The FAST route for the reference's own W8A8 configuration - per-token 8-bit activations, one weight block per row - is the
 * int8 MFMA kernel on an int8 image of the codes themselves (no expand at all): see "int8 route". */

/* ---- int8 route (the "W4A8 INT" configurations: x_quantizer block_fp with block_size [1,-1], i.e. one exponent per token,
 * reference experiments/pipeline/sweep_lqer_act_int.sh:83; w_quantizer blocks of 128 k or one block per row,
 * llama-7b-int.toml:87) ---------------------------------------------------------------------------------------------
 * With one activation exponent per token and one weight exponent per 128 k (or more) the products of a 128-k group share
 * one scale per output element, so their integer mantissas can be summed exactly on v_mfma_i32_32x32x32_i8 at twice the
 * bf16 rate.  x_fmt.kind = LQER_Q_MXINT_I8 selects this:
 *  - lqer_linear_sizes reports w_packed = the 4-bit sign-magnitude image of lqer_pack_weight_mxint followed (256-byte
 *    aligned) by a second image for the int8 main loop: two's-complement nibbles + one left shift per (row, 128-k group)
 *    + one scale per row (4.06 bit per weight).  Pack the first image as usual, then call lqer_i8_prepare on the same
 *    buffer.  flags[0] != 0 (device int32[2]): some row's integer sums could leave the i32 range, or the image holds two
 *    exponents inside one 128-k group - do not use LQER_Q_MXINT_I8 for this weight (plain LQER_Q_MXINT is always exact).
 *  - the activation image xq is then int8: [Mp][K padded to 128] mantissas followed (256-byte aligned) by Mp fp32 row
 *    scales; never larger than the bf16 image (K >= 128 required), so the workspace carving does not change.
 *  - the split calls lqer_quantize_act_xa / lqer_lowrank_xa / lqer_linear_gemm[_ld] with this kind ALWAYS produce /
 *    consume the int8 images; lqer_gemm_route says whether that is possible for M tokens (LQER_ROUTE_I8: M >= 128,
 *    B_out pass-through or one block per row, padded rank x limbs of x A <= 128);
 *    otherwise call them with kind LQER_Q_MXINT on the same buffers.  lqer_linear_forward chooses by itself.
 * Requires x_fmt.width <= 8, x_fmt.block <= 0 or >= K, w_fmt.block <= 0, >= K or a multiple of 128.
 *  - weights of 5..8 bits (the reference's W8A8 baseline, sweep_baseline_no_lqer.sh:73-76): the second image holds the int8 CODES
 *    themselves - per (256-row tile, 64-k half-step) 256 rows x 64 B, then one scale per row - and the main loop is LDS-DMA, one
 *    16-byte LDS read per weight fragment and the MFMA: no expand.  It needs ONE exponent per weight row (block_size [1,-1], or
 *    coarser blocks whose exponents happen to agree): lqer_i8_prepare sets flags[0] otherwise, and such a weight keeps the limb
 *    route (exact, three times the work).  128- or 256-row tiles (lqer_gemm_tile_rows says which): on 128-row tiles a wave's codes go
 *    straight from this image into registers (four coalesced 16-byte loads per lane and step, no LDS), on 256-row tiles through a
 *    half-step LDS ring. */
int lqer_i8_prepare(void* w_packed, int64_t N, int64_t K, const lqer_qfmt_t* w_fmt, int32_t* flags, void* stream);
/* Test hooks: the int8 weight image (inside w_packed) -> dequantized fp32 [N,K]; x [M,K] -> the int8 activation image. */
int lqer_unpack_weight_i8(const void* w_packed, int64_t N, int64_t K, float* w_f32, void* stream);
int lqer_unpack_weight_i8_fmt(const void* w_packed, int64_t N, int64_t K, const lqer_qfmt_t* w_fmt, float* w_f32, void* stream); /* ... of
   a weight of any width (5..8 bits: the image of codes behind the three limb images) */
int lqer_quantize_act_i8(const void* x, int dtype, int64_t M, int64_t K, int64_t ldx, const lqer_qfmt_t* fmt, void* xq_i8,
                         void* stream);
/* dst[row] = src[row] repeated `copies` times (device to device, stream-ordered; dst != src). */
int lqer_replicate_rows(const void* src, void* dst, int64_t rows, int64_t row_bytes, int copies, void* stream);

/* ---- quantized attention products (reference quantized_functions/matmul.py:12-37: matmul_flexible / bmm_flexible; call
 * sites models/llama_decoder.py:263,294, opt_decoder.py:125,190) -------------------------------------------------------
 *     out[b] = x_quantizer(x[b]) @ w_quantizer(y[b]),   b < batch;  x[b]: [S1, K], y[b]: [K, S2], out[b]: [S1, S2] (dense)
 * Both quantizers block_fp with blocks along the LAST dim of their operand (width <= 8; blocks of 16 - the templates' - run fused,
 * other lengths see lqer_matmul_q_workspace_bytes_fmt) - for y that is the output dim j, not the contraction dim (llama-7b.toml:110-126).  x is quantized in the GEMM's load path (read from HBM
 * once, no quantized copy); y goes through a bf16 image [batch][S2 padded to 128][K padded to 64] in `workspace`
 * (lqer_matmul_q_workspace_bytes).  Element strides: x[b][i][k] at b x_bs + i x_rs + k (k contiguous);
 * y[b][k][j] at b y_bs + k y_ks + j y_js with y_ks == 1 or y_js == 1 (Q K^T hands over the transposed VIEW of K: y_ks == 1).
 * fp32 accumulation of exact products (every 8-bit MXINT value is a bf16 number); out has the element type `dtype` of x, y. */
size_t lqer_matmul_q_workspace_bytes(int64_t batch, int64_t K, int64_t S2);
/* Blocks other than 16 (16 n elements, or whole rows: quantized_functions/matmul.py:12-29 takes any block_size): the library's
 * standalone quantizer writes that operand's bf16 image into the workspace first and the image / product kernels take it as
 * it is - same bits, one more pass over the operand.  Such an operand needs evenly spaced rows over the batch (x_bs == S1
 * x_rs; y dense along j with y_bs == K y_ks); the workspace then also holds those images: */
size_t lqer_matmul_q_workspace_bytes_fmt(int64_t batch, int64_t S1, int64_t K, int64_t S2, const lqer_qfmt_t* x_fmt,
                                         const lqer_qfmt_t* y_fmt);
int lqer_matmul_q(const void* x, const void* y, void* out, int dtype, int64_t batch, int64_t S1, int64_t K, int64_t S2,
                  int64_t x_bs, int64_t x_rs, int64_t y_bs, int64_t y_ks, int64_t y_js, const lqer_qfmt_t* x_fmt,
                  const lqer_qfmt_t* y_fmt, void* workspace, size_t workspace_bytes, void* stream);

/* ---- measurement aid ---------------------------------------------------------------------------------------------------
 * The shader clock the chip holds while other work runs: `nblocks` (1..64) one-wave workgroups on `stream` (a stream of its
 * own, beside the kernels under study) each write {shader cycles, 100 MHz ticks} elapsed over the last three quarters of about `duration_us` of real time to
 * out_pairs[2 b], out_pairs[2 b + 1] (device uint64).  MHz = cycles / ticks x 100.  bench.py: roofline.sustained_mhz. */
int lqer_clock_probe(unsigned long long* out_pairs, int nblocks, int64_t duration_us, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LQER_HIP_H */
