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


def _group(Ns, K, r, bias, cfg, dtype, seed=70):
    """Linears that will be handed the same tokens (q/k/v, gate/up): own weights, A, B and bias each."""
    import lqer_amd
    from bench import make_case

    mods, x = [], None
    for i, N in enumerate(Ns):
        case = make_case(8, K, N, r, seed=seed + i, bias=bias)
        x = case[0] if x is None else x
        m = lqer_amd.LinearFlexibleLqer(K, N, bias=bias, q_config=cfg, l_config={"rank": r})
        sd = {"weight": case[1], "A": case[2], "B": case[3]}
        if bias:
            sd["bias"] = case[4]
        m.load_state_dict(sd)
        mods.append(m.to(DEV).to(dtype))
    return mods, x


@pytest.mark.parametrize("Ns,K,r,bias,dtype", [((512, 256, 384), 1024, 32, False, torch.float16),   # q/k/v with GQA-style widths
                                               ((1376, 1376), 4096, 32, False, torch.float16),        # gate/up
                                               ((272, 512), 1088, 16, True, torch.float32),           # ragged K and N, bias
                                               ((256, 256, 256, 256), 512, 32, False, torch.bfloat16),  # four members: rank columns 0..127
                                               ((512, 512), 2048, 64, False, torch.float16)])         # 2 x rank 64 = 128 columns
def test_group_decode_is_one_launch_with_the_members_own_bits(ops, Ns, K, r, bias, dtype):
    """lqer_linear_forward_group through SharedActivation: the members of a group (handed the very same tensor) at M <= 8 run
    as ONE launch; every member's output carries the bits of its own one-launch forward, with the producers' wait bounded at
    0 (every workgroup computes the tiles itself) as well; a member that comes with ANOTHER tensor starts a new round."""
    from bench import MXINT_Q, OPT_Q
    from lqer_amd import _lib
    from lqer_amd.linear import SharedActivation

    cfg = OPT_Q if bias else MXINT_Q
    mods, x = _group(Ns, K, r, bias, cfg, dtype)
    for M in (1, 5, 8):
        xd = x[:M].to(dtype).to(DEV)
        alone = [m(xd).clone() for m in mods]  # (no group yet: each member's own launch)
        grp = SharedActivation(mods)
        assert grp.enabled
        got = [m(xd) for m in mods]
        assert grp._dplans[(M, ops.dtype_code(xd))] is not None  # the group launch was taken ...
        assert grp._dx is None and grp._dys is None              # ... and every member has been served
        for a, g_ in zip(alone, got):
            assert torch.equal(a, g_)
        # a second round with new values in a NEW tensor; only two of the members come (the third output is dropped)
        x2 = (xd * 0.5).contiguous()
        assert torch.equal(mods[1](x2), (lambda t: (setattr(mods[1], "_group", None), mods[1](t), setattr(mods[1], "_group", grp))[1])(x2))
        assert torch.equal(mods[0](x2), (lambda t: (setattr(mods[0], "_group", None), mods[0](t), setattr(mods[0], "_group", grp))[1])(x2))
        # an in-place write to the tensor between two members' calls: the second member must not be handed stale outputs
        x3 = xd.clone()
        y0 = mods[0](x3)
        x3.mul_(2.0)
        y1 = mods[1](x3)
        mods[1]._group = None
        assert torch.equal(y1, mods[1](x3))
        mods[1]._group = grp
        for m in mods:  # the consumers' fall-back inside the group launch
            m.tuning = _lib.TUNE_DECODE_NO_POLL
        grp.invalidate()
        got_fb = [m(xd) for m in mods]
        for a, g_ in zip(alone, got_fb):
            assert torch.equal(a, g_)
        for m in mods:
            m.tuning = 0
            m._group = None


def test_group_decode_under_graph_capture(ops):
    """A decoder layer's q/k/v (one group launch) captured in a graph: replays with new inputs follow them bit for bit."""
    from bench import MXINT_Q
    from lqer_amd.graph import GraphedCallable
    from lqer_amd.linear import SharedActivation

    mods, x = _group((512, 512, 512), 1024, 32, False, MXINT_Q, torch.float16)
    grp = SharedActivation(mods)
    assert grp.enabled
    fn = lambda t: torch.cat([m(t) for m in mods], dim=-1)
    xs = x[:4].half().to(DEV).clone()
    gc = GraphedCallable(fn, xs, warmup=1)
    for scale in (1.0, -0.25, 3.0, 0.5):
        xn = (x[:4] * scale).half().to(DEV)
        want = fn(xn)
        got = gc(xn).clone()
        torch.cuda.synchronize()
        assert torch.equal(got, want), scale


def test_group_forward_c_abi_refuses_what_it_does_not_serve(ops):
    """lqer_linear_forward_group returns LQER_E_UNSUPPORTED without launching for member sets outside the one-launch route:
    M > 8, one member, members of different K; the Python group then runs member by member (same results)."""
    from bench import MXINT_Q
    from lqer_amd import _lib
    from lqer_amd.linear import SharedActivation

    L = _lib.lib()
    mods, x = _group((256, 256), 512, 32, False, MXINT_Q, torch.float16)
    xd = x.half().to(DEV)
    for m in mods:
        m(xd)
    descs = [m._desc() for m in mods]
    tab = (_lib.GroupMember * 2)()
    ys = [torch.empty(8, 256, dtype=torch.float16, device=DEV) for _ in mods]
    for i, (m, d) in enumerate(zip(mods, descs)):
        p = m._packed
        tab[i].desc, tab[i].w_packed, tab[i].b_t, tab[i].b_limbs = C.pointer(d), p["w"].data_ptr(), p["b_t"].data_ptr(), p["b_limbs"]
        tab[i].bias_q, tab[i].y, tab[i].ldy = None, ys[i].data_ptr(), 256
    grp = SharedActivation(mods)
    grp._pack_cat(torch.device(DEV))
    ws = torch.empty(L.lqer_group_workspace_bytes(512, 64), dtype=torch.uint8, device=DEV)
    call = lambda n, M: L.lqer_linear_forward_group(tab, n, xd.data_ptr(), _lib.F16, M, 512, grp._cat["a_t"].data_ptr(), 1, ws.data_ptr(),
                                                    ws.numel(), None)
    assert call(2, 8) == 0
    torch.cuda.synchronize()
    for m, y in zip(mods, ys):
        m._group = None
        assert torch.equal(m(xd), y)
        m._group = grp
    assert call(2, 9) == -2 and call(1, 8) == -2
    assert L.lqer_linear_forward_group(tab, 2, xd.data_ptr(), _lib.F16, 8, 512, grp._cat["a_t"].data_ptr(), 1, ws.data_ptr(), 64, None) == -4
    # M = 16: the group keeps the shared-image route (one quantizer + side GEMM launch, then each member's GEMM)
    x16 = torch.cat([xd, xd * 0.5]).contiguous()
    outs = [m(x16) for m in mods]
    for m, o in zip(mods, outs):
        m._group = None
        assert float((m(x16).float() - o.float()).norm() / o.float().norm()) <= 2e-3


def test_group_decode_sees_an_in_place_write_to_any_member(ops):
    """the group launch reads every member's images, so a write to a NON-leading member's parameters
    (`k_proj.weight.copy_(W2)`, a bias write) must be noticed by the next decode step although only the leading member's
    forward runs its own version check - the outputs equal those of FRESH ungrouped modules built from the new values."""
    import lqer_amd
    from bench import OPT_Q, make_case
    from lqer_amd.linear import SharedActivation

    K, r, Ns = 512, 32, (256, 384, 256)
    cases = [make_case(8, K, N, r, seed=40 + i, bias=True) for i, N in enumerate(Ns)]

    def build(i, W=None, b=None):
        m = lqer_amd.LinearFlexibleLqer(K, Ns[i], bias=True, q_config=OPT_Q, l_config={"rank": r})
        m.load_state_dict({"weight": cases[i][1] if W is None else W, "A": cases[i][2], "B": cases[i][3],
                           "bias": cases[i][4] if b is None else b})
        return m.to(DEV).half()

    mods = [build(i) for i in range(3)]
    xd = cases[0][0][:4].half().to(DEV)
    grp = SharedActivation(mods)
    assert grp.enabled
    first = [m(xd).clone() for m in mods]
    assert grp._dplans[(4, ops.dtype_code(xd))] is not None
    g2 = torch.Generator().manual_seed(99)
    W2 = 0.02 * torch.randn(Ns[1], K, generator=g2)
    b2 = 0.01 * torch.randn(Ns[2], generator=g2)
    with torch.no_grad():
        mods[1].weight.copy_(W2.half().to(DEV))   # k: new unquantized weight
        mods[2].bias.copy_(b2.half().to(DEV))     # v: new bias (its weight already holds w_quantizer(W): not quantized again)
    x2 = (xd * 0.5).contiguous()                  # a new round, started by q
    got = [m(x2) for m in mods]
    assert grp._dplans[(4, ops.dtype_code(xd))] is not None  # (still one launch)
    want = [build(0)(x2), build(1, W=W2)(x2), build(2, b=b2)(x2)]
    for i in range(3):
        assert torch.equal(got[i], want[i]), i
    assert not torch.equal(got[1], (first[1].float() * 0.5).half())
    for m in mods:
        m._group = None


def test_group_member_on_another_stream_takes_its_own_route(ops):
    """A member that comes with the round's tensor on ANOTHER stream is not handed the output the group launch is still
    writing on the first stream: it runs its own forward there (same bits)."""
    from bench import MXINT_Q
    from lqer_amd.linear import SharedActivation

    mods, x = _group((256, 256), 512, 32, False, MXINT_Q, torch.float16)
    xd = x[:2].half().to(DEV)
    alone = [m(xd).clone() for m in mods]
    grp = SharedActivation(mods)
    y0 = mods[0](xd)
    side = torch.cuda.Stream(DEV)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        assert grp.take(mods[1], xd) is None
        y1 = mods[1](xd)
    side.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(y0, alone[0]) and torch.equal(y1, alone[1])
    for m in mods:
        m._group = None


def test_group_forward_c_abi_rejects_malformed_members(ops):
    """Every member's descriptor goes through the format checks of lqer_linear_forward (not members[0] alone): an 8-bit
    w_quantizer in member 1 is LQER_E_UNSUPPORTED, has_bias without bias_q LQER_E_INVALID - nothing is launched."""
    from bench import MXINT_Q
    from lqer_amd import _lib
    from lqer_amd.linear import SharedActivation

    L = _lib.lib()
    mods, x = _group((256, 256), 512, 32, False, MXINT_Q, torch.float16)
    xd = x.half().to(DEV)
    for m in mods:
        m(xd)
    descs = [m._desc() for m in mods]
    tab = (_lib.GroupMember * 2)()
    ys = [torch.full((8, 256), 7.0, dtype=torch.float16, device=DEV) for _ in mods]
    for i, (m, d) in enumerate(zip(mods, descs)):
        p = m._packed
        tab[i].desc, tab[i].w_packed, tab[i].b_t, tab[i].b_limbs = C.pointer(d), p["w"].data_ptr(), p["b_t"].data_ptr(), p["b_limbs"]
        tab[i].bias_q, tab[i].y, tab[i].ldy = None, ys[i].data_ptr(), 256
    grp = SharedActivation(mods)
    grp._pack_cat(torch.device(DEV))
    ws = torch.empty(L.lqer_group_workspace_bytes(512, 64), dtype=torch.uint8, device=DEV)
    call = lambda: L.lqer_linear_forward_group(tab, 2, xd.data_ptr(), _lib.F16, 8, 512, grp._cat["a_t"].data_ptr(), 1, ws.data_ptr(),
                                               ws.numel(), None)
    descs[1].w_fmt.width = 12
    assert call() == -2 and b"w_quantizer" in L.lqer_last_error()
    descs[1].w_fmt.width = 4
    descs[1].has_bias = 1
    assert call() == -1 and b"bias_q" in L.lqer_last_error()
    descs[1].has_bias = 0
    torch.cuda.synchronize()
    assert all(bool((y == 7.0).all()) for y in ys)  # nothing was launched
    assert call() == 0
    torch.cuda.synchronize()
    for m in mods:
        m._group = None
