"""64-row tiles of the fused GEMM (csrc/gemm_w4a8.hip, k_lqer_gemm<..., MT = 2>): taken when the 128-row grid would cover at
most half of the CUs (token counts between the decode kernel and the full tile grid).  Same kernel, same per-element
accumulation order: the outputs must equal the 128-row tiles' bit for bit, for every output type, ragged K / N, bias, both
B_out modes and no side path at all; and they must be the oracle's (reference quantized_layers/linear.py:145-157).
Run on the GPU box:  python -m pytest tests -m gpu -x -q"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import lqer_oracle as O  # the checker

DEV = "cuda:0"


@pytest.fixture(scope="module")
def lq():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import lqer_amd

    return lqer_amd


CASES = [  # M, K, N, rank, bias, B_out, dtype
    (65, 512, 384, 32, False, "mx", torch.float16),      # one live row in the second tile
    (96, 4096, 768, 32, False, "mx", torch.float16),
    (130, 1100, 520, 20, True, "mx", torch.bfloat16),    # ragged K and N (edge column tile), padded rank, bias
    (300, 512, 256, 16, True, "pass", torch.float32),    # B_out pass-through, fp32 outputs
    (1000, 2048, 1280, 32, False, "mx", torch.float16),
    (1024, 1024, 4096, 32, False, "mx", torch.bfloat16),  # the largest token count the rule takes at N = 4096
    (200, 768, 512, 0, True, "mx", torch.float16),       # LinearFlexible: no side path
    (200, 1024, 512, 48, False, "mx", torch.float16),    # padded rank 48: the staged side path
    (330, 512, 768, 128, True, "mx", torch.bfloat16),    # rank 128: two staging passes
    (260, 1024, 512, 32, False, "row", torch.float16),   # B_out with one block per token row: maxima from the pre-pass
    (300, 640, 384, 64, False, "int", torch.float16),    # the INT template: per-token x, unquantized A / B (bf16 limbs), B_out per row
]


@pytest.mark.parametrize("M,K,N,r,bias,bout,dtype", CASES)
def test_64_row_tiles_equal_128_row_tiles_and_the_oracle(lq, M, K, N, r, bias, bout, dtype):
    from bench import INT_Q, MXINT_Q, _bfp, make_case
    from lqer_amd import _lib

    qc = {"mx": MXINT_Q, "pass": dict(MXINT_Q, B_out_quantizer={"name": "passthrough"}),
          "row": dict(MXINT_Q, B_out_quantizer=_bfp(8, [1, -1], True)), "int": INT_Q}[bout]
    case = make_case(M, K, N, max(r, 16), seed=31, bias=bias, quantize_ab=bout != "int")
    x, W, A, B = case[:4]
    bvec = case[4] if bias else None
    if r > 0:
        A, B = A[:, :r].contiguous(), B[:r].contiguous()
        mod = lq.LinearFlexibleLqer(K, N, bias=bias, q_config=qc, l_config={"rank": r})
        sd = {"weight": W, "A": A, "B": B}
    else:
        mod = lq.LinearFlexible(K, N, bias=bias, q_config=dict(qc, name="flexible"))
        sd = {"weight": W}
    if bias:
        sd["bias"] = bvec
    mod.load_state_dict(sd)
    mod = mod.to(DEV).to(dtype)
    xd = x.to(dtype).to(DEV)
    mod.tuning = _lib.TUNE_TILE_ROWS_128  # (per call, in the descriptor: no process-wide switch)
    y128 = mod(xd).clone()
    mod.tuning = 0
    y64 = mod(xd).clone()
    assert torch.equal(y64, y128)
    h = lambda t: None if t is None else t.to(dtype).float()
    ref = O.lqer_linear_forward(h(x), h(W), h(bvec), h(A) if r > 0 else None, h(B) if r > 0 else None, qc)
    err = float((y64.float().cpu() - ref).norm() / ref.norm())
    assert err <= (4e-3 if dtype == torch.bfloat16 else 1e-3), err


def test_forced_64_row_tiles_and_xcd_blocks_give_the_same_bits(lq):
    """Round-3 experiment hooks of the 128-row kernel at the C2 shape (16 x 16 tiles, one per CU): 64-row tiles forced at
    M = 2048 (two workgroups per CU, 69.6 KB of LDS each) and XCD-local tile blocks (8 / 4 / 16 token tiles per XCD) are pure
    re-mappings of the same arithmetic - bit-identical outputs."""
    from bench import MXINT_Q, make_case
    from lqer_amd import _lib

    M, K, N, r = 2048, 1024, 4096, 32  # (K short: the mapping, not the loop length, is under test)
    x, W, A, B = make_case(M, K, N, r, seed=77)
    mod = lq.LinearFlexibleLqer(K, N, bias=False, q_config=MXINT_Q, l_config={"rank": r})
    mod.load_state_dict({"weight": W, "A": A, "B": B})
    mod = mod.to(DEV).half()
    xd = x.half().to(DEV)
    y0 = mod(xd).clone()
    mod.tuning = _lib.TUNE_TILE_ROWS_64
    assert torch.equal(mod(xd), y0)
    for bm in (8, 4, 16, 3):  # (3 does not divide the grid: ignored)
        mod.tuning = _lib.tune_xcd_block(bm)
        assert torch.equal(mod(xd), y0), bm
    mod.tuning = 0


PARTIAL_CASES = [  # M, K, N, rank, bias, B_out, dtype
    (2048, 4096, 4096, 32, False, "mx", torch.float16),   # C2: sixteen chunks (one per 256 k), one item per thread
    (300, 1100, 520, 20, True, "mx", torch.bfloat16),     # ragged M / K / N, padded rank, bias (128-row tiles pinned)
    (1500, 11008, 512, 32, False, "mx", torch.float16),   # 43 chunks
    (640, 2048, 768, 64, True, "mx", torch.bfloat16),     # rank 64: two items per thread, two chunks each up front
    (513, 1024, 1024, 48, False, "pass", torch.float16),  # padded rank 48, B_out pass-through
]


@pytest.mark.parametrize("M,K,N,r,bias,bout,dtype", PARTIAL_CASES)
def test_gemm_summing_the_partial_tiles_equals_the_reduce_launch(lq, M, K, N, r, bias, bout, dtype):
    """LQER_TUNE_XA_REDUCE_IN_GEMM: lqer_linear_forward on 128-row tiles skips k_xa_reduce4 - the GEMM's workgroups sum the
    split-K partial tiles of x A in ascending chunk order and apply A_out on the way into the side product's LDS stage
    (k_lqer_gemm XAPART) - the same arithmetic item by item, so y must carry the bits of the default three-launch route.
    (Selectable, not the default: measured slower, include/lqer_hip.h.)"""
    from bench import MXINT_Q, make_case
    from lqer_amd import _lib

    qc = {"mx": MXINT_Q, "pass": dict(MXINT_Q, B_out_quantizer={"name": "passthrough"})}[bout]
    case = make_case(M, K, N, max(r, 16), seed=37, bias=bias)
    x, W, A, B = case[:4]
    A, B = A[:, :r].contiguous(), B[:r].contiguous()
    mod = lq.LinearFlexibleLqer(K, N, bias=bias, q_config=qc, l_config={"rank": r})
    sd = {"weight": W, "A": A, "B": B}
    if bias:
        sd["bias"] = case[4]
    mod.load_state_dict(sd)
    mod = mod.to(DEV).to(dtype)
    xd = x.to(dtype).to(DEV)
    mod.tuning = _lib.TUNE_TILE_ROWS_128
    y3 = mod(xd).clone()
    mod.tuning = _lib.TUNE_TILE_ROWS_128 | _lib.TUNE_XA_REDUCE_IN_GEMM
    assert _lib.lib().lqer_tile_partials(mod._desc(), M, _lib.F16 if dtype == torch.float16 else _lib.BF16) == 1
    y2 = mod(xd).clone()
    assert torch.equal(y2, y3)
    h = lambda t: None if t is None else t.to(dtype).float()
    ref = O.lqer_linear_forward(h(x), h(W), h(case[4]) if bias else None, h(A), h(B), qc)
    err = float((y2.float().cpu() - ref).norm() / ref.norm())
    assert err <= (4e-3 if dtype == torch.bfloat16 else 1e-3), err


DEFER_CASES = [  # M, K, N, rank, bias, dtype
    (2048, 4096, 4096, 32, False, torch.float16),   # BASELINE configs[1]: the direct side path (two 16-deep slices)
    (300, 1024, 520, 20, True, torch.bfloat16),     # the shortest K that defers (16 k-steps), ragged M / N, padded rank, bias
    (1500, 2048, 768, 64, True, torch.float16),     # rank 64: the staged side path
    (640, 1100, 512, 32, False, torch.float32),     # ragged K (18 k-steps), fp32 outputs: the accumulator's own bits
]


@pytest.mark.parametrize("M,K,N,r,bias,dtype", DEFER_CASES)
def test_b_out_requantized_under_the_main_loop_equals_in_front_of_it(lq, M, K, N, r, bias, dtype):
    """Round 6 (k_lqer_gemm DEFER): B_out in blocks of 16 is re-quantized piece by piece under the first 16 k-steps and the side product
    added behind the last one; LQER_TUNE_BOUT_IN_PROLOGUE pins the rounds-1-5 order (re-quantized in front of the main loop, the
    accumulators opened on it).  Same formula: on MXINT data (every partial sum exact in fp32) the same bits; with a side product whose
    blocks are zero, below the 1e-8 pass-through, and spread over 28 binades both stay the oracle's (linear.py:145-157) - there the
    two summation orders may differ in the last fp32 bit, like gemm_smallm.hip / decode1.hip (which always added last) and the tile kernels did."""
    from bench import MXINT_Q, make_case
    from lqer_amd import _lib

    case = make_case(M, K, N, max(r, 16), seed=41, bias=bias)
    x, W, A, B = case[:4]
    A, B = A[:, :r].contiguous(), B[:r].contiguous()
    h = lambda t: None if t is None else t.to(dtype).float()

    def both(Bm):
        mod = lq.LinearFlexibleLqer(K, N, bias=bias, q_config=MXINT_Q, l_config={"rank": r})
        sd = {"weight": W, "A": A, "B": Bm}
        if bias:
            sd["bias"] = case[4]
        mod.load_state_dict(sd)
        mod = mod.to(DEV).to(dtype)
        xd = x.to(dtype).to(DEV)
        mod.tuning = _lib.TUNE_TILE_ROWS_128
        y_new = mod(xd).clone()
        mod.tuning = _lib.TUNE_TILE_ROWS_128 | _lib.TUNE_BOUT_IN_PROLOGUE
        y_old = mod(xd).clone()
        ref = O.lqer_linear_forward(h(x), h(W), h(case[4]) if bias else None, h(A), h(Bm), MXINT_Q)
        tol = 4e-3 if dtype == torch.bfloat16 else 1e-3
        for y in (y_new, y_old):
            assert float((y.float().cpu() - ref).norm() / ref.norm()) <= tol
        return y_new, y_old, ref

    y_new, y_old, _ = both(B)
    assert torch.equal(y_new, y_old)
    # columns of B: zero (blocks with maximum 0), 1e-10 (every element of the side product below 1e-8: passed through unquantized),
    # and powers of two over 28 binades (block exponents far apart; sums no longer exact; fp16 outputs stay finite)
    g = torch.Generator().manual_seed(5)
    scale = torch.pow(2.0, torch.randint(-20, 8, (N,), generator=g).float())
    scale[: N // 8] = 0.0
    scale[N // 8: N // 4] = 1e-10
    y_new, y_old, ref = both((B * scale[None, :]).contiguous())
    assert float((y_new.float() - y_old.float()).abs().max()) <= 2e-3 * float(ref.abs().max())
