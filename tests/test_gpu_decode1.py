"""The one-launch forward of decode sizes (csrc/decode1.hip, behind lqer_linear_forward for M <= 8; reference
quantized_layers/linear.py:145-157): against the oracle, against the two-launch route it replaces, with the consumers'
fall-back forced (every workgroup computes the partial tiles of x A itself), across calls that reuse the granule scratch,
and under graph capture (the one-launch route is capturable: its granule tag carries the launch's dispatch id).
Run on the GPU box:  python -m pytest tests -m gpu -x -q"""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import lqer_oracle as O  # the checker

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from lqer_amd import ops as _ops

    return _ops


def _module(K, N, r, bias, cfg, dtype, seed=5, M=8):
    import lqer_amd
    from bench import make_case

    case = make_case(M, K, N, r, seed=seed, bias=bias)
    x, W, A, B = case[:4]
    mod = lqer_amd.LinearFlexibleLqer(K, N, bias=bias, q_config=cfg, l_config={"rank": r})
    sd = {"weight": W, "A": A, "B": B}
    if bias:
        sd["bias"] = case[4]
    mod.load_state_dict(sd)
    return mod.to(DEV).to(dtype), x, W, A, B, (case[4] if bias else None)


def _ref(x, W, b, A, B, cfg, dtype):
    h = (lambda t: None if t is None else t.to(dtype).float())
    return O.lqer_linear_forward(h(x), h(W), h(b), h(A), h(B), cfg)


CASES = [
    # M, K, N, r, bias, dtype, tol
    (1, 4096, 4096, 32, False, torch.float16, 1e-3),
    (8, 4096, 1024, 32, False, torch.float16, 1e-3),
    (3, 1088, 272, 16, True, torch.float32, 1e-5),    # ragged K (5 slabs, the last one short), N not a multiple of 16 x 17
    (4, 11008, 512, 64, False, torch.bfloat16, 8e-3),  # Llama down-projection width: 43 producers, rank 64
    (5, 512, 4096, 48, True, torch.float16, 1e-3),     # three rank tiles, OPT-style bias
]


@pytest.mark.parametrize("M,K,N,r,bias,dtype,tol", CASES)
def test_one_launch_decode_vs_oracle_and_two_launch_route(ops, M, K, N, r, bias, dtype, tol):
    from bench import MXINT_Q, OPT_Q
    from lqer_amd import _lib

    cfg = OPT_Q if bias else MXINT_Q
    mod, x, W, A, B, b = _module(K, N, r, bias, cfg, dtype, M=M)
    xd = x[:M].to(dtype).to(DEV)
    y = mod(xd)
    ref = _ref(x[:M], W, b, A, B, cfg, dtype)
    err = float((y.float().cpu() - ref).norm() / ref.norm())
    assert err <= tol, err
    # the two-launch route on the same operands (quantizer + partial tiles, then the small-M kernel): same quantizers, another
    # summation order of x A -> equal within the output rounding
    L = _lib.lib()
    desc, p = mod._desc(), mod._packed
    assert L.lqer_decode_partials(C.byref(desc), M) == 1
    sz = ops.linear_sizes(desc, M).workspace
    ws = torch.zeros(sz, dtype=torch.uint8, device=DEV)
    Kp, Mp = L.lqer_padded_k(K), L.lqer_padded_m(M)
    xq = ws.data_ptr()
    rp = L.lqer_padded_r(r)
    scr = xq + ((Mp * Kp * 2 + 255) // 256) * 256 + ((Mp * rp * 2 + 255) // 256) * 256
    nscr = L.lqer_lowrank_xa_scratch_bytes(C.byref(desc), M)
    y2 = torch.empty_like(y)
    dt = ops.dtype_code(xd)
    _lib.check(L.lqer_quantize_act_xa(C.byref(desc), xd.data_ptr(), dt, M, K, p["a_t"].data_ptr(), p["a_limbs"], xq, None, scr, nscr, None), "q")
    _lib.check(L.lqer_linear_gemm(C.byref(desc), xq, M, p["w"].data_ptr(), None, p["b_t"].data_ptr(), p["b_limbs"], ops._ptr(p.get("bias")),
                                  y2.data_ptr(), dt, N, scr, nscr, None), "g")
    torch.cuda.synchronize()
    d = float((y.float() - y2.float()).norm() / y2.float().norm())
    assert d <= tol, d
    # run-to-run bit stability (fixed-order sums on both sides of the hand-off), fresh output buffer
    for _ in range(3):
        assert torch.equal(mod(xd), y)


def test_consumer_fallback_computes_the_same_bits(ops):
    """With the poll bound at 0 every consumer workgroup computes the partial tiles itself (the producers' routine): the
    result must be bit-identical - the path that guarantees termination under any dispatch order is also exact."""
    from bench import MXINT_Q
    from lqer_amd import _lib

    mod, x, W, A, B, _ = _module(2048, 1024, 32, False, MXINT_Q, torch.float16)
    xd = x[:6].half().to(DEV)
    y = mod(xd)
    mod.tuning = _lib.TUNE_DECODE_NO_POLL  # (per call, in the descriptor)
    y_fb = mod(xd)
    torch.cuda.synchronize()
    mod.tuning = 0
    assert torch.equal(y_fb, y)


def test_granule_scratch_reused_across_calls_and_shapes(ops):
    """The granule area lives in the shared workspace: other token counts, other modules and prefill forwards (which write
    fp32 partial tiles into the same bytes) in between must never leak into a decode forward."""
    from bench import MXINT_Q

    mod, x, W, A, B, _ = _module(1024, 512, 32, False, MXINT_Q, torch.float16, M=300)
    mod2, x2, *_ = _module(1024, 256, 16, False, MXINT_Q, torch.float16, seed=9, M=300)
    want = {}
    for M in (1, 8, 3):
        want[M] = _ref(x[:M], W, None, A, B, MXINT_Q, torch.float16)
    for rnd in range(3):
        for M in (8, 1, 3):
            xs = (x[:M] * (1.0 if rnd == 0 else 1.0 + 0.25 * rnd)).half()  # new values through the same scratch
            y = mod(xs.to(DEV)).float().cpu()
            ref = _ref(xs.float(), W, None, A, B, MXINT_Q, torch.float16)
            assert float((y - ref).norm() / ref.norm()) <= 1e-3, (rnd, M)
            mod2(x2[: 2 + rnd].half().to(DEV))       # another Linear, other rank, same workspace
            mod(x[:300].half().to(DEV))              # a prefill forward: fp32 partial tiles over the granule area
    assert want


def test_captured_forward_is_one_launch_and_replays_follow_their_inputs(ops):
    """The granule tag mixes the launch's AQL dispatch id into the host's per-call counter, so the one-launch route is taken
    under stream capture too: a replayed graph node carries frozen arguments but a fresh dispatch id.  Replays issued back to
    back with NEW inputs (copied in on the same stream, no host synchronisation in between) must each follow their own
    input bit for bit - a replay that accepted the previous replay's tiles of x A would show the previous input's side
    product."""
    from bench import MXINT_Q

    mod, x, W, A, B, _ = _module(1024, 512, 32, False, MXINT_Q, torch.float16)
    xd = x[:4].half().to(DEV).clone()
    mod(xd)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        mod(xd)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        y = mod(xd)
        y2 = mod(y.repeat(1, 2).contiguous())  # a second Linear forward in the same graph, through the same workspace
    scales = [1.0, -0.5, 3.0] + [0.1 * (i + 1) * (-1) ** i for i in range(40)]
    inputs = [(x[:4] * sc).half().to(DEV) for sc in scales]
    direct = []
    for xi in inputs:  # the same two forwards outside any graph
        yi = mod(xi)
        direct.append((yi.clone(), mod(yi.repeat(1, 2).contiguous()).clone()))
    got = []
    for xi in inputs:
        xd.copy_(xi)
        g.replay()
        got.append((y.clone(), y2.clone()))
    torch.cuda.synchronize()
    for i, ((a1, a2), (b1, b2)) in enumerate(zip(got, direct)):
        assert torch.equal(a1, b1) and torch.equal(a2, b2), f"replay {i} (scale {scales[i]}) does not follow its input"
    ref = _ref((x[:4] * scales[2]).half().float(), W, None, A, B, MXINT_Q, torch.float16)
    assert float((got[2][0].float().cpu() - ref).norm() / ref.norm()) <= 1e-3


def test_graphed_callable_runs_a_chain_of_module_forwards(ops):
    """lqer_amd.graph.GraphedCallable: a chain of decode-size module forwards captured once and replayed with new inputs
    equals the same chain run eagerly, bit for bit (one-launch route at M <= 8, two-launch route at M = 16)."""
    from bench import MXINT_Q
    from lqer_amd.graph import GraphedCallable

    mod, x, W, A, B, _ = _module(1024, 1024, 32, False, MXINT_Q, torch.float16, M=16)
    for M in (2, 16):
        xs = x[:M].half().to(DEV).clone()
        chain = lambda t: mod(mod(mod(t)))
        gc = GraphedCallable(chain, xs, warmup=1)
        for scale in (1.0, 0.37, -2.0):
            xn = (x[:M] * scale).half().to(DEV)
            want = chain(xn)
            got = gc(xn).clone()
            torch.cuda.synchronize()
            assert torch.equal(got, want), (M, scale)
    with pytest.raises(ValueError):
        gc(xs[:1])
