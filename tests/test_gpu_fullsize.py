"""Parity at the bench's OWN shapes (BASELINE.json configs[1..4]): the HIP module runs the full-size forward, and row
slices of the result are compared with the CPU oracle.  Rows are independent in every configuration of the path (the
activation, A_out and B_out blocks never span two tokens - `test_size_independent_properties_full_size` checks that
on the GPU), so a few hundred rows of the oracle pin the kernel route the bench times: tile kernel, side-path
route (direct / staged, one or two 64-rank passes), B_out route (in-register blocks of 16 / pre-pass row maxima), bias.
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


def _rows(M):
    """192 rows: the first 64, 64 straddling the middle tile boundary, the last 64 (first / interior / last row tile)."""
    mid = (M // 2 // 256) * 256
    idx = list(range(0, 64)) + list(range(mid - 32, mid + 32)) + list(range(M - 64, M))
    return torch.tensor(sorted(set(i for i in idx if 0 <= i < M)))


def _run(lq, M, K, N, r, qc, bias, quantize_ab, seed, tol=1e-3, route=None, tile_rows=None):
    import ctypes as C

    from bench import make_case
    from lqer_amd import _lib

    case = make_case(M, K, N, r, seed=seed, bias=bias, quantize_ab=quantize_ab)
    x, W, A, B = case[:4]
    b = case[4] if bias else None
    mod = lq.LinearFlexibleLqer(K, N, bias=bias, q_config=qc, l_config={"rank": r})
    sd = {"weight": W, "A": A, "B": B}
    if bias:
        sd["bias"] = b
    mod.load_state_dict(sd)
    mod = mod.to(DEV).half()
    xin = x.half()
    y = mod(xin.to(DEV))
    assert y.shape == (M, N) and torch.isfinite(y).all()
    if route is not None:  # the kernel the bench times at this shape
        assert _lib.lib().lqer_gemm_route(C.byref(mod._desc()), M, _lib.F16) == getattr(_lib, "ROUTE_" + route)
        if tile_rows is not None:
            assert _lib.lib().lqer_gemm_tile_rows(C.byref(mod._desc()), M, _lib.F16) == tile_rows
    idx = _rows(M)
    got = y[idx.to(DEV)].float().cpu()
    ref = O.lqer_linear_forward(xin[idx].float(), W.half().float(), b.half().float() if bias else None, A.half().float(),
                                B.half().float(), qc)
    err = float((got - ref).norm() / ref.norm())
    assert err <= tol, err
    # fp16 output of an fp32 sum: besides the rounding of y, no element may be further than a few fp16 ulps of the
    # largest value in its row from the oracle (a wrong tile, block exponent or side-path slice would be)
    row_scale = ref.abs().amax(dim=1, keepdim=True)
    assert float(((got - ref).abs() / row_scale).max()) <= 2.0 ** -8
    return mod, y


# C5: OPT-6.7B, rank 128, bias in blocks of 16 (opt-6.7b.toml:98-102; sweep_lqer_svd.sh:80-84).  M = 2048:
# 4096 -> 16384 takes the 256-row kernel (staged side path, two 64-rank passes), 16384 -> 4096 and 4096 -> 4096 the
# 128-row kernel (staged, rank 128).
@pytest.mark.parametrize("K,N", [(4096, 16384), (16384, 4096), (4096, 4096)])
def test_c5_opt_rank128_bias_full_size(lq, K, N):
    from bench import OPT_Q

    _run(lq, 2048, K, N, 128, OPT_Q, True, True, seed=51, route="TILE256" if N == 16384 else "TILE128")


# C4: Llama-13B shapes, rank 64, W blocks of 128 (sweep_lqer_act_int.sh:83) or one block per row
# (llama-7b-int.toml:87), 8-bit per-token activations, unquantized fp16 A / B, per-row B_out (pre-pass).  M = 16384 is
# the bench's token count (seq 2048 x batch 8).
@pytest.mark.parametrize("K,N,wblock", [(5120, 13824, 128), (13824, 5120, 128), (5120, 5120, -1)])
def test_c4_int_rank64_full_size(lq, K, N, wblock):
    from bench import INT_Q, _bfp

    qc = dict(INT_Q, w_quantizer=_bfp(4, [1, wblock], False))
    mod, _ = _run(lq, 16384, K, N, 64, qc, False, False, seed=41, route="I8", tile_rows=256)
    assert mod._x_i8  # the int8 main loop


# c2int / c3int: the shapes north_star names - Llama-7B projections at M = 2048, rank 32 - with the
# reference's Llama-7B INT template (experiments/configs/template/llama-7b-int.toml:70-93: W4 one block per row; the sweep
# experiments/pipeline/sweep_lqer_act_int.sh:81-83: W4 blocks of 128, rank 32), 8-bit per-token activations, unquantized
# fp16 A / B.  The int8 MFMA kernel on 128-row tiles: 256 x 256 tiles would leave half of the CUs idle at 4096 x 4096.
@pytest.mark.parametrize("K,N,wblock", [(4096, 4096, 128), (4096, 11008, 128), (11008, 4096, 128), (4096, 4096, -1)])
def test_c2int_c3int_llama7b_int_rank32_full_size(lq, K, N, wblock):
    from bench import INT_Q, _bfp

    qc = dict(INT_Q, w_quantizer=_bfp(4, [1, wblock], False))
    mod, y = _run(lq, 2048, K, N, 32, qc, False, False, seed=61, route="I8", tile_rows=128)
    assert mod._x_i8
    # the 256-row tiles of the same kernel: the same bits
    from lqer_amd import _lib

    mod.tuning = _lib.TUNE_I8_ROWS_256
    x = __import__("bench").make_case(2048, K, N, 32, seed=61, quantize_ab=False)[0].half().to(DEV)
    assert torch.equal(mod(x), y)


# C2 / C3: Llama-7B shapes at the bench's M = 2048, rank 32, MXINT blocks of 16.
@pytest.mark.parametrize("K,N", [(4096, 4096), (4096, 11008), (11008, 4096)])
def test_c2_c3_mxint_rank32_full_size(lq, K, N):
    from bench import MXINT_Q

    _run(lq, 2048, K, N, 32, MXINT_Q, False, True, seed=21, route="TILE128")


def test_c4_a16_template_full_size(lq):
    """The INT template as shipped (pass-through fp16 activations, fp16 MFMA main loop) at the bench's c4a16 shape."""
    from bench import A16_Q

    mod, _ = _run(lq, 16384, 5120, 5120, 64, A16_Q, False, False, seed=43)
    assert mod._x_f16


@pytest.mark.parametrize("K,N", [(4096, 4096), (4096, 11008), (11008, 4096)])
def test_w3a16_weight_only_sweep_full_size(lq, K, N):
    """The reference's weight-only sweep (experiments/pipeline/sweep_lqer_act_w-only.sh:74-77, the paper's "W3A16" row): 3-bit weights in
    blocks of [1, 32], pass-through fp16 activations, rank 64, at the Llama-7B shapes of the bench's c3w3a16 workload - the fp16 MFMA main
    loop over 3-bit codes (round 6; pinned at small sizes by the reference-generated `w3b32_a16_r64` vectors)."""
    from bench import W3A16_Q

    mod, _ = _run(lq, 2048, K, N, 64, W3A16_Q, False, False, seed=71)
    assert mod._x_f16
