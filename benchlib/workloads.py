"""BASELINE.json's configurations as bench workloads, and the synthetic operands of SURVEY.md §8d."""
from __future__ import annotations

import torch


def _bfp(width, block, skip):
    return dict(name="block_fp", width=width, exponent_width=8, exponent_bias=None, block_size=block, skip_first_dim=skip)


# reference experiments/configs/template/llama-7b.toml:78-105 (W4A8 MXINT, blocks of 16)
MXINT_Q = dict(name="flexible_lqer", is_ptq=True, default=False, x_quantizer=_bfp(8, [1, 16], True),
               w_quantizer=_bfp(4, [1, 16], False), b_quantizer=_bfp(8, [-1], False))
# opt-6.7b.toml:98-102: bias in blocks of 16
OPT_Q = dict(MXINT_Q, b_quantizer=_bfp(8, [1, 16], False))

# reference sweep_lqer_act_int.sh:83 / llama-7b-int.toml (W block 128, A/B unquantized fp16) with the 8-bit
# per-token activation format BASELINE.json's "W4A8 L2QER-INT" pins: block_fp(8, [1,-1])
INT_Q = dict(name="flexible_lqer", is_ptq=True, default=False, x_quantizer=_bfp(8, [1, -1], True),
             w_quantizer=_bfp(4, [1, 128], False), b_quantizer=dict(name="passthrough"))
# the same with one weight block per row (llama-7b-int.toml:87, block_size [1, -1])
INTROW_Q = dict(INT_Q, w_quantizer=_bfp(4, [1, -1], False))

# 8-bit weights, one block per row, 8-bit per-token activations: the reference's W8A8 format (experiments/pipeline/
# sweep_baseline_no_lqer.sh:73-76 runs it through LinearFlexible; here with the rank-32 side path of the INT sweeps, unquantized A / B)
W8A8_Q = dict(INT_Q, w_quantizer=_bfp(8, [1, -1], False))

# the INT templates as shipped (llama-7b-int.toml q_config.linear): pass-through fp16 activations ("W4A16"), A_out and
# B_out falling back to the same pass-through (linear.py:115-124), A/B unquantized
A16_Q = dict(INT_Q, x_quantizer=dict(name="passthrough", width=16, frac_width=12))
# the reference's weight-only sweep (experiments/pipeline/sweep_lqer_act_w-only.sh:74-77, the paper's "W3A16" row): 3-bit weights in
# blocks of [1, 32], pass-through activations / bias, A / B unquantized, rank 64
W3A16_Q = dict(A16_Q, w_quantizer=_bfp(3, [1, 32], False))
UNQUANTIZED_AB = (INT_Q, INTROW_Q, A16_Q, W8A8_Q, W3A16_Q)

BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
INT8_MFMA_PEAK_TOPS = 5000.0    # 2x bf16 per clock (same guide, "Matrix cores", I8 row)
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md, HBM3E

LLAMA13B = [(5120, 5120, 4), (5120, 13824, 2), (13824, 5120, 1)]
LLAMA7B = [(4096, 4096, 4), (4096, 11008, 2), (11008, 4096, 1)]
WORKLOADS = {
    # name: (description, M, rank, bias, q_config, [(K, N, count per layer)], decoder layers of the model)
    "c2": ("LqerLinear 4096x4096 rank32 W4A8-MXINT16 M=2048 (BASELINE configs[1])", 2048, 32, False, MXINT_Q, [(4096, 4096, 1)], 1),
    "c3": ("Llama-7B 7 projections x 32 layers rank32 W4A8-MXINT16 M=2048 (BASELINE configs[2])", 2048, 32, False, MXINT_Q,
           [(4096, 4096, 4), (4096, 11008, 2), (11008, 4096, 1)], 32),
    # north_star's own shape on the instruction it names: the reference's Llama-7B INT configuration
    # (sweep_lqer_act_int.sh:81-83: W4 blocks of 128, rank 32; llama-7b-int.toml:87: one block per row) with 8-bit per-token
    # activations, on the int8 MFMA kernel's 128-row tiles
    "c2int": ("LqerLinear 4096x4096 rank32 W4(block128)A8(per-token) M=2048 (north_star shape, int8 MFMA)", 2048, 32, False, INT_Q,
              [(4096, 4096, 1)], 1),
    "c2introw": ("LqerLinear 4096x4096 rank32 W4(one block per row, llama-7b-int.toml:87)A8(per-token) M=2048", 2048, 32, False,
                 INTROW_Q, [(4096, 4096, 1)], 1),
    "c3int": ("Llama-7B 7 projections x 32 layers rank32 W4(block128)A8(per-token) M=2048 (llama-7b-int.toml / "
              "sweep_lqer_act_int.sh:81-83)", 2048, 32, False, INT_Q, LLAMA7B, 32),
    # 8-bit weights on the int8 MFMA kernel's code image (no expand in the main loop); 256-row tiles: M = 2048 fills half of the CUs
    # at 4096 x 4096, M = 8192 two full rounds
    "c2w8a8": ("LqerLinear 4096x4096 rank32 W8(one block per row)A8(per-token) M=2048 (sweep_baseline_no_lqer.sh:73-76 format, int8 MFMA)",
               2048, 32, False, W8A8_Q, [(4096, 4096, 1)], 1),
    "c2w8a8m8k": ("LqerLinear 4096x4096 rank32 W8(one block per row)A8(per-token) M=8192 (512 tiles of 256 x 256: two full rounds)",
                  8192, 32, False, W8A8_Q, [(4096, 4096, 1)], 1),
    "c4": ("Llama-13B 7 projections x 40 layers rank64 W4(block128)A8(per-token) M=16384 (BASELINE configs[3])", 16384, 64, False,
           INT_Q, LLAMA13B, 40),
    "c4row": ("Llama-13B 7 projections x 40 layers rank64 W4(one block per row, llama-7b-int.toml:87)A8(per-token) M=16384", 16384, 64,
              False, INTROW_Q, LLAMA13B, 40),
    "c5": ("OPT-6.7B 6 projections x 32 layers rank128 W4A8-MXINT16 M=2048 (BASELINE configs[4])", 2048, 128, True, OPT_Q,
           [(4096, 4096, 4), (4096, 16384, 1), (16384, 4096, 1)], 32),
    "c4a16": ("Llama-13B 7 projections x 40 layers rank64 W4(block128)A16 (the reference's INT template as shipped) M=16384",
              16384, 64, False, A16_Q, LLAMA13B, 40),
    # (round 6, VERDICT r5 missing 3) the W3A16 weight-only sweep at the Llama-7B shapes: fp16 MFMA main loop over 3-bit codes
    "c3w3a16": ("Llama-7B 7 projections x 32 layers rank64 W3(block32)A16 M=2048 (sweep_lqer_act_w-only.sh:74-77)", 2048, 64, False,
                W3A16_Q, LLAMA7B, 32),
    "d1a16": ("LqerLinear 4096x4096 rank32 W4(block128)A16 M=1 (decode)", 1, 32, False, A16_Q, [(4096, 4096, 1)], 1),
    # decode sizes (HBM-bound on the packed weight; roofline quoted in GB/s): the small-M kernel
    "d1": ("LqerLinear 4096x4096 rank32 W4A8-MXINT16 M=1 (decode)", 1, 32, False, MXINT_Q, [(4096, 4096, 1)], 1),
    "d16": ("LqerLinear 4096x4096 rank32 W4A8-MXINT16 M=16 (decode)", 16, 32, False, MXINT_Q, [(4096, 4096, 1)], 1),
    # a decode step the way a model runs it: the 7 projections of Llama-7B decoder layers at M = 1 - q/k/v and gate/up as
    # one-launch groups (lqer_linear_forward_group), o and down as single launches - walking 4 distinct layers (454 MB of packed weights)
    "d1layer": ("Llama-7B decoder layer, 7 projections rank32 W4A8-MXINT16 M=1: q/k/v + gate/up as group launches, o + down single (decode)",
                1, 32, False, MXINT_Q, LLAMA7B, 4),
}


def flops(M, K, N, r):
    """Reference multiply model (experiments/hw_performance/README.md:81-106) x 2."""
    return 2 * M * K * N + 2 * M * K * r + 2 * M * r * N


def _snap_mxint8_dim0(t):
    """t -> the 8-bit MXINT grid with blocks of 16 along dim 0 (the reference approximator's A / B format,
    llama-7b.toml:60-73).  On a GPU box this is the library's own HIP quantizer (the GPU's inputs do not depend on the
    checker); without a GPU (CPU-only tools and tests) the CPU oracle's.  Setup of synthetic inputs only - any values would do."""
    if torch.cuda.is_available():
        from lqer_amd import ops

        fmt = ops.make_qfmt(_bfp(8, [1, 16], True), "x")
        return ops.quantize_mxint(t.t().contiguous().cuda(), fmt, want=("deq",))["deq"].t().contiguous().cpu()
    from oracle import lqer_oracle as O

    return O.mxint_quantize(t, width=8, block_size=[16, 1], skip_first_dim=False)


def make_x(M, K, seed=0):
    """Synthetic token batch of SURVEY.md §8d: x ~ N(0,1) with three x30 outlier channels."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K, generator=g)
    for c in (7, 1033, 2900):
        if c < K:
            x[:, c] *= 30.0
    return x, g


def make_weights(g, K, N, r, bias=False, quantize_ab=True):
    """W ~ N(0, 0.02^2); A, B ~ 0.01 N(0,1), snapped to the 8-bit MXINT grid for the MXINT configurations, left
    unquantized for the INT ones (llama-7b-int.toml:61-68); optional bias ~ 0.01 N(0,1)."""
    W = 0.02 * torch.randn(N, K, generator=g)
    A = B = None
    if r > 0:
        A = 0.01 * torch.randn(K, r, generator=g)
        B = 0.01 * torch.randn(r, N, generator=g)
        if quantize_ab:
            A, B = _snap_mxint8_dim0(A), _snap_mxint8_dim0(B)
    return (W, A, B, 0.01 * torch.randn(N, generator=g)) if bias else (W, A, B)


def make_case(M, K, N, r, seed=0, bias=False, quantize_ab=True):
    """(x, W, A, B[, bias]) from one seed (tests, tools, smoke)."""
    x, g = make_x(M, K, seed)
    return (x,) + make_weights(g, K, N, r, bias, quantize_ab)


def check_rows(M, every=False):
    """Rows compared with the oracle after the timed region: the first and the last 96 (first / last row tile) - every row of a
    single-Linear workload of up to 2048 tokens (the headline: 0.3 s more of the oracle)."""
    n = M if every and M <= 2048 else min(96, M)
    return torch.tensor(sorted(set(range(n)) | set(range(M - n, M))))
