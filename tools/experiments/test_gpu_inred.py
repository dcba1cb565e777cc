"""The in-launch reduce of the 128-row tile kernel (csrc/gemm_w4a8.hip INRED; reference linear.py:154-156): with the fused
quantizer's formats the GEMM launch sums the split-K partial tiles of x A itself - a share per workgroup, published as tagged
granules, gathered under the ring fill - instead of a separate reduce launch.  Same arithmetic, so the same bits as the
three-launch route; also with the gather's fall-back forced, under graph capture, and through the module.
Run on the GPU box:  python -m pytest tests -m gpu -x -q"""
import ctypes as C

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


def _two_routes(mod, xd, spin=None):
    """y through lqer_quantize_act_xa + lqer_linear_gemm with xaq materialised (three launches) and with xaq == NULL (two)."""
    from lqer_amd import _lib, ops

    L = _lib.lib()
    M, K = xd.shape
    N = mod.out_features
    desc = mod._desc()
    dref = C.byref(desc)
    assert L.lqer_decode_partials(dref, M) == 1, "shape not eligible for the in-launch reduce"
    p = mod._packed
    st = torch.cuda.current_stream().cuda_stream
    Mp, Kp, rp = L.lqer_padded_m(M), L.lqer_padded_k(K), L.lqer_padded_r(mod.rank)
    xq = torch.empty(Mp * Kp, dtype=torch.bfloat16, device=DEV)
    xaq = torch.empty(Mp * rp, dtype=torch.bfloat16, device=DEV)
    nscr = L.lqer_lowrank_xa_scratch_bytes(dref, M)
    scr = torch.full((nscr,), 0x5A, dtype=torch.uint8, device=DEV)  # stale bytes: never a valid tag
    dt = ops.dtype_code(xd)
    outs = []
    for use_xaq in (True, False):
        y = torch.empty(M, N, dtype=xd.dtype, device=DEV)
        xa = xaq.data_ptr() if use_xaq else None
        _lib.check(L.lqer_quantize_act_xa(dref, xd.data_ptr(), dt, M, K, p["a_t"].data_ptr(), p["a_limbs"], xq.data_ptr(), xa,
                                          scr.data_ptr(), nscr, st), "qxa")
        if spin is not None and not use_xaq:
            L.lqer_debug_set_decode_spin(spin)
        try:
            gs = nscr if not use_xaq else L.lqer_linear_gemm_scratch_bytes(dref, M)
            _lib.check(L.lqer_linear_gemm(dref, xq.data_ptr(), M, p["w"].data_ptr(), xa, p["b_t"].data_ptr(), p["b_limbs"],
                                          ops._ptr(p.get("bias")), y.data_ptr(), dt, N, scr.data_ptr(), gs, st), "gemm")
            torch.cuda.synchronize()
        finally:
            L.lqer_debug_set_decode_spin(-1)
        outs.append(y)
    return outs


def _module(lq, M, K, N, r, bias, dtype, seed=3):
    from bench import MXINT_Q, OPT_Q, make_case

    qc = OPT_Q if bias else MXINT_Q
    case = make_case(M, K, N, r, seed=seed, bias=bias)
    x, W, A, B = case[:4]
    mod = lq.LinearFlexibleLqer(K, N, bias=bias, q_config=qc, l_config={"rank": r})
    sd = {"weight": W, "A": A, "B": B}
    if bias:
        sd["bias"] = case[4]
    mod.load_state_dict(sd)
    mod = mod.to(DEV).to(dtype)
    xd = x.to(dtype).to(DEV)
    mod(xd[:8])  # packs the operands
    return mod, xd, (x, W, A, B, case[4] if bias else None, qc)


CASES = [
    # M, K, N, r, bias, dtype
    (2048, 1024, 4096, 32, False, torch.float16),   # the C2 tile grid (16 x 16), 4 chunks
    (1100, 4096, 4096, 16, False, torch.float16),   # ragged M (rows of the last tile beyond M), rank 16, 16 chunks
    (2048, 768, 4096, 64, True, torch.bfloat16),    # rank 64 (two 16-byte chunks per thread), bias, bf16
    (1536, 11008, 4096, 48, False, torch.float32),  # 43 chunks, rank 48 (three A_out blocks per row), fp32 in / out
]


@pytest.mark.parametrize("M,K,N,r,bias,dtype", CASES)
def test_in_launch_reduce_equals_the_three_launch_route(lq, M, K, N, r, bias, dtype):
    mod, xd, (x, W, A, B, b, qc) = _module(lq, M, K, N, r, bias, dtype)
    y3, y2 = _two_routes(mod, xd)
    assert torch.equal(y2, y3)
    h = lambda t: None if t is None else t.to(dtype).float()
    idx = torch.cat([torch.arange(0, 64), torch.arange(M - 64, M)])
    ref = O.lqer_linear_forward(h(x)[idx], h(W), h(b), h(A), h(B), qc)
    err = float((y2[idx.to(DEV)].float().cpu() - ref).norm() / ref.norm())
    assert err <= {torch.float32: 1e-5, torch.float16: 1e-3, torch.bfloat16: 8e-3}[dtype], err
    # the module takes the route by itself (lqer_linear_forward)
    assert torch.equal(mod(xd), y3)


def test_gather_fallback_computes_the_same_bits(lq):
    """Poll bound 0: every workgroup whose rows are not there at its first look sums them itself (the producers' routine)."""
    mod, xd, _ = _module(lq, 2048, 2048, 4096, 32, False, torch.float16, seed=5)
    y3, y2 = _two_routes(mod, xd, spin=0)
    assert torch.equal(y2, y3)


def test_captured_forward_with_the_in_launch_reduce_follows_its_inputs(lq):
    """The granule tag carries the launch's dispatch id: replays of one captured forward with new inputs (no host
    synchronisation in between) each equal their eager result."""
    mod, xd, _ = _module(lq, 2048, 512, 4096, 32, False, torch.float16, seed=7)
    static_x = xd.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        mod(static_x)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = mod(static_x)
    scales = [1.0, -0.5, 2.0, 0.25, -3.0, 0.75]
    want = [mod((xd * sc).half()).clone() for sc in scales]
    got = []
    for sc in scales:
        static_x.copy_((xd * sc).half())
        g.replay()
        got.append(y.clone())
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), i


def test_shapes_outside_the_route_keep_three_launches(lq):
    from lqer_amd import _lib

    L = _lib.lib()
    mod, xd, _ = _module(lq, 300, 512, 512, 32, False, torch.float16)  # thin grid: 64-row tiles
    assert L.lqer_decode_partials(C.byref(mod._desc()), 300) == 0
    assert L.lqer_decode_partials(C.byref(mod._desc()), 8) == 1       # (decode sizes: the small-M / one-launch routes)
    mod128, _, _ = _module(lq, 2048, 512, 4096, 128, False, torch.float16)  # rank 128: not the fused quantizer's formats
    assert L.lqer_decode_partials(C.byref(mod128._desc()), 2048) == 0
