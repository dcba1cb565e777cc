"""Parity of the HIP path (through the C ABI) against the oracle and the golden vectors.
Run on the GPU box:  python -m pytest tests -m gpu -x -q"""
import numpy as np
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import lqer_oracle as O  # the checker (tests may import it; the product never does)

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from lqer_amd import ops as _ops

    return _ops


def _fmt(ops, width, block):
    return ops.make_qfmt(dict(name="block_fp", width=width, exponent_width=8, exponent_bias=None, block_size=[1, block]))


def _q_cases(golden_q):
    """(name, x, y_ref, width, block along last dim) for every golden case the HIP path covers."""
    out = []
    for name in sorted({k.split("/")[1] for k in golden_q.files if k.startswith("q/")}):
        x = torch.from_numpy(golden_q[f"q/{name}/x"])
        y = torch.from_numpy(golden_q[f"q/{name}/y"])
        meta = golden_q[f"q/{name}/meta"].tolist()
        width, block = meta[0], meta[2:]
        if width > 8:
            continue
        if x.ndim == 2 and len(block) == 2 and block[0] == 16 and block[1] == 1:
            out.append((name + "^T", x.t().contiguous(), y.t().contiguous(), width, 16))  # column blocks = row blocks of x^T
            continue
        if any(b != 1 for b in block[:-1]) or (x.ndim == 2 and len(block) == 1 and not meta[1]):
            continue  # 2-D tiles (a lone [L] without skip_first_dim = all rows x L): the weight packer's, test_pack_weight_2d_tiles_bit_exact
        out.append((name, x, y, width, block[-1]))
    return out


def test_quantizer_bit_exact_vs_reference_vectors(ops, golden_q):
    cases = _q_cases(golden_q)
    assert len(cases) >= 18
    for name, x, y, width, block in cases:
        got = ops.quantize_mxint(x.to(DEV), _fmt(ops, width, block), want=("deq",))["deq"].cpu()
        assert torch.equal(got.view(torch.int32), y.view(torch.int32)) or torch.equal(got, y), name


def test_quantizer_codes_and_exponents_bit_exact(ops, golden_q):
    for name, x, y, width, block in _q_cases(golden_q):
        x2 = x.reshape(-1, x.shape[-1])
        _, codes, exps = O.mxint_quantize(x2, width=width, block_size=[1, block], skip_first_dim=True, decompose=True)
        got = ops.quantize_mxint(x2.to(DEV), _fmt(ops, width, block))
        assert torch.equal(got["codes"].cpu().to(torch.int32), codes), name
        assert torch.equal(got["exps"].cpu().to(torch.int32), exps.reshape(x2.shape[0], -1).clamp(-128, 127)), name


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_quantizer_half_inputs(ops, dtype):
    torch.manual_seed(3)
    x = (torch.randn(33, 200) * 3).to(dtype)
    x[:, 7] *= 30
    for block in (16, 32, -1):
        ref, codes, exps = O.mxint_quantize(x.float(), width=8, block_size=[1, block], skip_first_dim=True, decompose=True)
        got = ops.quantize_mxint(x.to(DEV), _fmt(ops, 8, block))
        assert torch.equal(got["deq"].cpu(), ref)
        assert torch.equal(got["codes"].cpu().to(torch.int32), codes)


def test_quantize_act_bf16_image_exact(ops):
    torch.manual_seed(4)
    for (M, K) in ((5, 176), (300, 4096), (1, 50)):
        x = torch.randn(M, K, dtype=torch.float16)
        x[:, 3] *= 30
        ref = O.mxint_quantize(x.float(), width=8, block_size=[1, 16], skip_first_dim=True)
        xq = ops.quantize_act(x.to(DEV), _fmt(ops, 8, 16)).cpu()
        assert xq.shape[0] % 256 == 0 and xq.shape[1] % 64 == 0
        assert torch.equal(xq[:M, :K].float(), ref)
        assert not xq[:M, K:].any()


def _panels_from_rowmajor(codes, exps, L, Kp, K):
    """oracle layout ([Np,Kp/2] codes, [Np,nblk] exps) -> the library's panel layout (bytes).
    Segments that lie entirely in the K padding carry exponent 0."""
    Np = codes.shape[0]
    e16 = exps.repeat_interleave(max(L // 16, 1), dim=1)[:, : Kp // 16] if L < Kp else exps[:, :1].expand(Np, Kp // 16)
    e16 = e16.clone().to(torch.int32)
    e16[:, -(-K // 16):] = 0
    e16 = (e16 - 3 + 127).clamp(1, 254).to(torch.uint8)  # stored biased for mbits = 3
    # a row's 8 words (4 B = 8 k each) of a 64-k group are stored in the order {0,2,4,6,1,3,5,7}
    words = codes.reshape(Np, Kp // 64, 8, 4)[:, :, [0, 2, 4, 6, 1, 3, 5, 7], :].reshape(Np, Kp // 2)
    c = words.reshape(Np // 16, 16, Kp // 64, 32).permute(0, 2, 1, 3)  # [pn, pk, 16, 32]
    e = e16.reshape(Np // 16, 16, Kp // 64, 4).permute(0, 2, 1, 3).contiguous()
    return torch.cat([c.reshape(Np // 16, Kp // 64, 512), e.reshape(Np // 16, Kp // 64, 64)], dim=2).reshape(-1)


def test_pack_unpack_weight_bit_exact(ops, golden_q):
    for name, block in (("w4_b16", 16), ("w4_b128", 128), ("w4_row", -1), ("w4_ragged", 16), ("w2_b32", 32)):
        x = torch.from_numpy(golden_q[f"q/{name}/x"])
        y = torch.from_numpy(golden_q[f"q/{name}/y"])
        width = int(golden_q[f"q/{name}/meta"][0])
        fmt = _fmt(ops, width, block)
        packed = ops.pack_weight(x.to(DEV), fmt)
        w = ops.unpack_weight(packed, x.shape[0], x.shape[1], fmt).cpu()
        ref = torch.where(x.abs() <= 1e-8, torch.zeros_like(y), y)  # declared flush (block_fp.py:79-80)
        assert torch.equal(w, ref), name
        if width == 4:
            codes, exps = O.pack_weight_mxint4(x, block, n_pad=256, k_pad=64)
            Kp = codes.shape[1] * 2
            L = Kp if block <= 0 or block >= x.shape[1] else block
            want = _panels_from_rowmajor(codes, exps, L, Kp, x.shape[1])
            assert torch.equal(packed.cpu(), want), name


def test_pack_unpack_8bit_weight_with_tiny_rows(ops):
    """An 8-bit weight travels as three 4-bit limbs whose exponent bytes stand 2^3 apart (pack.hip k_w_pack8).  Rows of |w| < 2^-120 sit
    where the lowest limb's byte would clamp: their blocks pack as zeros - the declared |w| <= 1e-8 flush of packed images - and the
    other rows read back as the oracle's quantizer, bit for bit (round 6)."""
    from oracle import lqer_oracle as O

    g = torch.Generator().manual_seed(5)
    W = 0.02 * torch.randn(48, 256, generator=g)
    W[3] = torch.randn(256, generator=g) * 2.0 ** -124
    W[7, :64] = torch.randn(64, generator=g) * 2.0 ** -121
    W[9] = 0.0
    for block in (16, 64, -1):
        cfg = dict(name="block_fp", width=8, exponent_width=8, exponent_bias=None, block_size=[1, block], skip_first_dim=False)
        fmt = ops.make_qfmt(cfg, "w")
        w = ops.unpack_weight(ops.pack_weight(W.to(DEV), fmt), 48, 256, fmt).cpu()
        ref = O.get_quantizer(cfg)(W.clone())
        assert torch.equal(w, torch.where(W.abs() <= 1e-8, torch.zeros_like(ref), ref)), block
        assert float(w[3].abs().max()) == 0.0 and float(w[9].abs().max()) == 0.0


def test_pack_weight_2d_tiles_bit_exact(ops, golden_q):
    """Weight tiles of R rows x L k (block_size [R, L], skip_first_dim = false; a lone [L] = all rows x L): the packed image,
    read back, against the reference's vectors - ragged tiles in both dims included (quantizers/utils.py:161-183)."""
    for name in ("w4_tile_8x16", "w4_tile_16x32_ragged", "w4_tile_allrows_16", "w4_tile_4xrow"):
        x = torch.from_numpy(golden_q[f"q/{name}/x"])
        y = torch.from_numpy(golden_q[f"q/{name}/y"])
        meta = golden_q[f"q/{name}/meta"].tolist()
        fmt = ops.make_qfmt(dict(name="block_fp", width=int(meta[0]), exponent_width=8, exponent_bias=None, block_size=meta[2:],
                                 skip_first_dim=bool(meta[1])), "w")
        assert getattr(fmt, "block_rows", 1) != 1
        packed = ops.pack_weight(x.to(DEV), fmt)
        w = ops.unpack_weight(packed, x.shape[0], x.shape[1], fmt).cpu()
        ref = torch.where(x.abs() <= 1e-8, torch.zeros_like(y), y)
        assert torch.equal(w, ref), name
        # the standalone quantizer honours the tile rows as well (it used to quantize per row silently)
        assert torch.equal(ops.quantize_mxint(x.to(DEV), fmt, want=("deq",))["deq"].cpu(), y), name
        with pytest.raises(NotImplementedError):
            ops.quantize_mxint(x.to(DEV), fmt, want=("deq", "codes"))


def test_pack_lowrank_limbs(ops):
    torch.manual_seed(5)
    K, N, r = 100, 70, 24
    A8 = O.mxint_quantize(0.05 * torch.randn(K, r), width=8, block_size=[16, 1], skip_first_dim=False)
    B8 = O.mxint_quantize(0.05 * torch.randn(r, N), width=8, block_size=[16, 1], skip_first_dim=False)
    for A, B, want in ((A8, B8, 1), (A8.half(), B8.half(), 1), (torch.randn(K, r).half(), torch.randn(r, N).half(), 2),
                       (torch.randn(K, r), torch.randn(r, N), 3)):
        a_t, b_t, la, lb = ops.pack_lowrank(A.to(DEV), B.to(DEV))
        assert (la, lb) == (want, want)
        Kp, Np, rp = 128, 256, 32
        a3 = a_t.cpu().float().reshape(3, rp, Kp)
        b3 = b_t.cpu().float().reshape(3, Np, rp)
        assert torch.equal(a3.sum(0)[:r, :K].t().double(), A.double()) or torch.equal((a3[0].double() + a3[1].double() + a3[2].double())[:r, :K].t(), A.double())
        assert torch.equal((b3[0].double() + b3[1].double() + b3[2].double())[:N, :r].t(), B.double())
        assert not a3[:, r:].any() and not a3[:, :, K:].any() and not b3[:, N:].any() and not b3[:, :, r:].any()


def _module_from_case(g, cfgs, name, dtype=torch.float32):
    import lqer_amd

    t = lambda k: torch.from_numpy(g[f"{name}/{k}"])
    qc = cfgs[name]
    W = t("W")
    N, K = W.shape
    has_b = f"{name}/bias" in g.files
    r = int(g[f"{name}/rank"][0])
    mod = lqer_amd.get_quantized_layer_cls("linear", qc)(K, N, bias=has_b, q_config=qc, l_config={"rank": r})
    sd = {"weight": W, "A": t("A"), "B": t("B")}
    if has_b:
        sd["bias"] = t("bias")
    mod.load_state_dict(sd)  # same keys as the reference module
    return mod.to(DEV).to(dtype), t


FWD_CASES = ["m1", "m7", "m64", "b2s5", "r128", "ragged", "int128", "introw", "a16", "a16row", "a16mix", "tile8", "tileall",
             "w8row", "w8g128", "w8b16", "w6b32",  # (w8*: weights of 5..8 bits as three 4-bit limbs, round 5)
             "w3b32_a16_r64", "w3b32_a16_r64_3d"]  # (round 6: the W3A16 weight-only sweep, sweep_lqer_act_w-only.sh:74-77)


@pytest.mark.parametrize("name", FWD_CASES)
def test_forward_vs_reference_vectors(ops, golden_fwd, name):
    g, cfgs = golden_fwd
    mod, t = _module_from_case(g, cfgs, name)
    y = mod(t("x").to(DEV)).cpu()
    ref = t("y")
    assert y.shape == ref.shape and y.dtype == torch.float32
    err = (y - ref).norm() / ref.norm()
    assert err <= 1e-5, float(err)  # fp32 in/out: only accumulation order differs
    # like the reference after its first forward, the parameters now hold the quantized values
    assert torch.equal(mod.weight.detach().cpu(), t("wq"))
    if mod.bias is not None:
        assert torch.equal(mod.bias.detach().cpu(), t("bq"))


ACT_TILE_CASES = ["acttile3d", "acttile_ragged", "acttile2d", "acttile2d_all", "acttile_r_on_2d", "acttile_whole", "acttile_bout",
                  "acttile_lone3d", "acttile_lone2d"]  # (round 6: the default block_size [16] - [1, T, 16] on a 3-D tensor, per row on a 2-D one)


@pytest.mark.parametrize("name", ACT_TILE_CASES)
def test_forward_activation_tiles_vs_reference_vectors(ops, golden_fwd, name):
    """Activation quantizers whose blocks span token rows (round 5; reference quantizers/utils.py:211-237 for 3-D tensors,
    :161-183 / :261-270 for 2-D tensors with skip_first_dim = false): the HIP tile quantizer is bit-exact on x and on the
    reference's own intermediates, the module's tile route matches the reference's y, quantizes weight / bias once, and its
    results do not depend on the call count, the dtype's route (fp16 within the fp16 bar) or a copy of the module."""
    import copy

    g, cfgs = golden_fwd
    mod, t = _module_from_case(g, cfgs, name)
    assert mod._tiles
    x = t("x").to(DEV)
    assert torch.equal(ops.quantize_act_tiles(x, mod._fmt["x"]).cpu(), t("xq"))
    assert torch.equal(ops.quantize_act_tiles(t("xA").to(DEV), mod._fmt["A_out"]).cpu(), t("xAq"))
    assert torch.equal(ops.quantize_act_tiles(t("xAB").to(DEV), mod._fmt["B_out"]).cpu(), t("xABq"))
    y = mod(x).cpu()
    ref = t("y")
    assert y.shape == ref.shape and y.dtype == torch.float32
    err = (y - ref).norm() / ref.norm()
    assert err <= 1e-5, float(err)
    assert torch.equal(mod.weight.detach().cpu(), t("wq"))  # quantized once, in place (linear.py:149-153)
    if mod.bias is not None:
        assert torch.equal(mod.bias.detach().cpu(), t("bq"))
    assert torch.equal(mod(x).cpu(), y)                      # ... and not a second time
    assert torch.equal(copy.deepcopy(mod)(x).cpu(), y)
    assert list(mod.state_dict()) == (["weight", "bias", "A", "B"] if mod.bias is not None else ["weight", "A", "B"])  # (the twin is no submodule)
    # fp16 module and tokens: against the oracle on the same fp16-rounded tokens (the weight / bias already hold their quantized
    # values, A and B are 8-bit MXINT: all exact in fp16), at the fp16-output bar
    from oracle import lqer_oracle as O

    mh = mod.half()
    xh = x.half()
    yh = mh(xh).float().cpu()
    bq = t("bq") if mod.bias is not None else None
    refh = O.lqer_linear_forward(xh.float().cpu(), t("wq"), bq, t("A"), t("B"), cfgs[name], weight_is_quantized=True)
    assert (yh - refh).norm() / refh.norm() <= 1e-3
    # new values in place: the twin's images follow
    with torch.no_grad():
        mod.weight.copy_(t("W").to(DEV).to(mod.weight.dtype) * 0.5)
    y2 = mod(x.half()).float().cpu()
    assert not torch.equal(y2, yh)


def test_activation_tiles_refusals_mirror_the_reference(ops):
    import lqer_amd

    bfp = lambda w, bs, skip: dict(name="block_fp", width=w, exponent_width=8, exponent_bias=None, block_size=bs, skip_first_dim=skip)
    qc = dict(name="flexible_lqer", is_ptq=True, default=False, x_quantizer=bfp(8, [4, 16], False), w_quantizer=bfp(4, [1, 16], False))
    mod = lqer_amd.LinearFlexibleLqer(64, 32, bias=False, q_config=qc, l_config={"rank": 16}).to(DEV)
    assert mod(torch.randn(8, 64, device=DEV)).shape == (8, 32)
    with pytest.raises(NotImplementedError, match="block 3d weight"):   # utils.py:279
        mod(torch.randn(2, 8, 64, device=DEV))
    with pytest.raises(RuntimeError, match="Unsupported x.ndim"):        # utils.py:284
        mod(torch.randn(2, 2, 8, 64, device=DEV))
    # the standalone op on empty / tiny inputs
    f = mod._fmt["x"]
    assert ops.quantize_act_tiles(torch.zeros(0, 64, device=DEV), f).shape == (0, 64)
    z = ops.quantize_act_tiles(torch.zeros(5, 64, device=DEV), f)
    assert torch.equal(z, torch.zeros_like(z))


def test_forward_stages_vs_reference_vectors(ops, golden_fwd):
    """xq and xAq (the two quantized intermediates that are materialised) against the vectors."""
    import ctypes as C

    from lqer_amd import _lib

    g, cfgs = golden_fwd
    for name in ("m7", "m64", "r128"):
        mod, t = _module_from_case(g, cfgs, name)
        x = t("x").to(DEV)
        mod(x)
        M, K = x.reshape(-1, x.shape[-1]).shape
        xq = ops.quantize_act(x.reshape(M, K), mod._fmt["x"])
        assert torch.equal(xq[:M, :K].float().cpu(), t("xq").reshape(M, K))
        rp = _lib.lib().lqer_padded_r(mod.rank)
        xaq = torch.empty(xq.shape[0], rp, dtype=torch.bfloat16, device=DEV)
        desc = mod._desc()
        p = mod._packed
        nscr = _lib.lib().lqer_lowrank_xa_scratch_bytes(C.byref(desc), M)
        scr = torch.empty(nscr, dtype=torch.uint8, device=DEV)
        _lib.check(_lib.lib().lqer_lowrank_xa(C.byref(desc), xq.data_ptr(), M, p["a_t"].data_ptr(), p["a_limbs"], xaq.data_ptr(),
                                              scr.data_ptr(), nscr, None), "xa")
        torch.cuda.synchronize()
        got = xaq[:M, : mod.rank].float().cpu()
        ref = t("xAq").reshape(M, -1)
        # a re-quantizer turns fp32 summation-order noise into (rare) one-step differences.  The bound that follows from
        # it (tests/_envelope.py): every product is exact, any summation order lands within D ulps of the exact sum, so each
        # block of 16 must be the quantizer's image of SOME point of that interval - however few entries differ
        from _envelope import envelope_check

        s64 = t("xq").reshape(M, K).double().numpy() @ t("A").double().numpy()
        bad = envelope_check(s64, got.numpy(), 16, 7, max(16.0, K ** 0.5))
        assert bad == 0, (name, bad)
        assert (got != ref).float().mean() <= 0.02, name  # and stays close to torch's own order


def test_forward_no_side_path(ops, golden_fwd):
    import lqer_amd

    g, cfgs = golden_fwd
    t = lambda k: torch.from_numpy(g[f"flex/{k}"])
    qc = cfgs["flex"]
    mod = lqer_amd.get_quantized_layer_cls("linear", qc)(96, 48, bias=True, q_config=qc, l_config=None)
    mod.load_state_dict({"weight": t("W"), "bias": t("bias")})
    y = mod.to(DEV)(t("x").to(DEV)).cpu()
    assert (y - t("y")).norm() / t("y").norm() <= 1e-5


@pytest.mark.parametrize("name", ["flex_w8a8_row", "flex_w8a8_g128"])
def test_forward_no_side_path_w8a8(ops, golden_fwd, name):
    """SURVEY row a11 in the configuration the reference runs LinearFlexible with (sweep_baseline_no_lqer.sh:73-76: W8A8, one block
    per weight row and per token) and with weight blocks of 128, against the reference's own outputs; the parameters hold the
    quantized values afterwards; the packed image gives the quantized weight back bit for bit."""
    import lqer_amd
    from lqer_amd import ops as O_

    g, cfgs = golden_fwd
    t = lambda k: torch.from_numpy(g[f"{name}/{k}"])
    qc = cfgs[name]
    N, K = t("W").shape
    mod = lqer_amd.get_quantized_layer_cls("linear", qc)(K, N, bias=True, q_config=qc, l_config=None)
    assert type(mod).__name__ == "LinearFlexible"
    mod.load_state_dict({"weight": t("W"), "bias": t("bias")})
    mod = mod.to(DEV)
    y = mod(t("x").to(DEV)).cpu()
    assert y.shape == t("y").shape
    assert (y - t("y")).norm() / t("y").norm() <= 1e-5
    assert torch.equal(mod.weight.detach().cpu(), t("wq")) and torch.equal(mod.bias.detach().cpu(), t("bq"))
    w_img = mod._single_copy("w")
    assert w_img.numel() == 3 * (-(-N // 256) * 256 // 16) * (-(-K // 64)) * 576  # three 4-bit limb images
    assert torch.equal(O_.unpack_weight(w_img, N, K, mod._fmt["w"]).cpu(), t("wq"))
    # decode sizes and a few rows: the same module at other token counts (other kernels of the 4-bit path), same values
    x2 = t("x").reshape(-1, K)
    for rows in (1, 5, x2.shape[0]):
        yr = mod(x2[:rows].to(DEV)).cpu()
        ref = t("y").reshape(-1, N)[:rows]
        assert (yr - ref).norm() / ref.norm() <= 1e-5, rows
    # 16-bit tensors
    yh = mod.half()(x2.half().to(DEV)).float().cpu()
    ref_h = O.lqer_linear_forward(x2.half().float(), t("wq").half().float(), t("bq").half().float(), None, None,
                                  dict(qc, w_quantizer=dict(name="passthrough"), b_quantizer=dict(name="passthrough")))
    assert (yh - ref_h).norm() / ref_h.norm() <= 1e-3


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 1e-3), (torch.bfloat16, 5e-3), (torch.float32, 1e-5)])
def test_forward_llama_shape_vs_oracle(ops, dtype, tol):
    """4096 -> 4096, rank 32, W4A8 MXINT (BASELINE config 2) against the CPU oracle; tolerance is the
    north star's 1e-3 relative L2 for fp16 output (bf16 output rounding alone is 2^-9)."""
    import lqer_amd
    from bench import make_case, MXINT_Q

    K = N = 4096
    M, r = 384, 32
    x, W, A, B = make_case(M, K, N, r, seed=0)
    mod = lqer_amd.LinearFlexibleLqer(K, N, bias=False, q_config=MXINT_Q, l_config={"rank": r})
    mod.load_state_dict({"weight": W, "A": A, "B": B})
    mod = mod.to(DEV).to(dtype)
    xin = x.to(dtype)
    y = mod(xin.to(DEV)).float().cpu()
    ref = O.lqer_linear_forward(xin.float(), W.to(dtype).float(), None, A.to(dtype).float(), B.to(dtype).float(), MXINT_Q)
    err = (y - ref).norm() / ref.norm()
    assert err <= tol, float(err)


@pytest.mark.parametrize("K,N,r,bias,cfg", [(4096, 11008, 32, False, "mxint"), (11008, 4096, 32, False, "mxint"),
                                            (1000, 1500, 48, True, "opt"), (5120, 1280, 64, False, "int")])
def test_forward_model_shapes_vs_oracle(ops, K, N, r, bias, cfg):
    """Llama-7B MLP shapes (N and K = 11008 = 43 * 256), a ragged shape (K, N, rank not multiples of the tile
    constants, bias in blocks of 16) and the INT configuration (per-token activation blocks, weight blocks of
    128, unquantized A/B) against the oracle at M = 200."""
    import lqer_amd
    from bench import INT_Q, MXINT_Q, OPT_Q, make_case

    qc = {"mxint": MXINT_Q, "opt": OPT_Q, "int": INT_Q}[cfg]
    M = 200
    case = make_case(M, K, N, r, seed=11, bias=bias, quantize_ab=cfg != "int")
    x, W, A, B = case[:4]
    b = case[4] if bias else None
    mod = lqer_amd.LinearFlexibleLqer(K, N, bias=bias, q_config=qc, l_config={"rank": r})
    sd = {"weight": W, "A": A, "B": B}
    if bias:
        sd["bias"] = b
    mod.load_state_dict(sd)
    y = mod.to(DEV)(x.to(DEV)).cpu()
    ref = O.lqer_linear_forward(x, W, b, A, B, qc)
    err = (y - ref).norm() / ref.norm()
    assert err <= 2e-5, float(err)


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 5e-3), (torch.float32, 2e-5)])
def test_small_m_kernel_output_dtypes(ops, dtype, tol):
    """bf16 and fp32 in/out through the small-M kernel (M = 20: two token tiles), B_out pass-through variant included."""
    import lqer_amd
    from bench import MXINT_Q, make_case

    M, K, N, r = 20, 768, 1024, 32
    x, W, A, B = make_case(M, K, N, r, seed=21)
    for qc in (MXINT_Q, dict(MXINT_Q, B_out_quantizer={"name": "passthrough"})):
        mod = lqer_amd.LinearFlexibleLqer(K, N, bias=False, q_config=qc, l_config={"rank": r})
        mod.load_state_dict({"weight": W, "A": A, "B": B})
        mod = mod.to(DEV).to(dtype)
        xin = x.to(dtype)
        y = mod(xin.to(DEV))
        assert y.dtype == dtype
        ref = O.lqer_linear_forward(xin.float(), W.to(dtype).float(), None, A.to(dtype).float(), B.to(dtype).float(), qc)
        err = (y.float().cpu() - ref).norm() / ref.norm()
        assert err <= tol, (float(err), qc.get("B_out_quantizer"))


@pytest.mark.parametrize("M", [1, 16, 17, 33, 48, 64, 65])
@pytest.mark.parametrize("cfg,K,N,r,bias", [("mxint", 4096, 4096, 32, False), ("opt", 1000, 1500, 48, True)])
def test_small_m_kernel_vs_oracle(ops, M, cfg, K, N, r, bias):
    """Decode sizes: M <= 64 runs the HBM-bound small-M kernel (one workgroup per 16 output columns, 1..4 token
    tiles of 16), M = 65 the tile kernel; fp16 in/out against the oracle at the north star's tolerance."""
    import lqer_amd
    from bench import MXINT_Q, OPT_Q, make_case

    qc = {"mxint": MXINT_Q, "opt": OPT_Q}[cfg]
    case = make_case(M, K, N, r, seed=5, bias=bias)
    x, W, A, B = case[:4]
    b = case[4] if bias else None
    mod = lqer_amd.LinearFlexibleLqer(K, N, bias=bias, q_config=qc, l_config={"rank": r})
    sd = {"weight": W, "A": A, "B": B}
    if bias:
        sd["bias"] = b
    mod.load_state_dict(sd)
    mod = mod.to(DEV).half()
    xin = x.half()
    y = mod(xin.to(DEV)).float().cpu()
    ref = O.lqer_linear_forward(xin.float(), W.half().float(), b.half().float() if bias else None, A.half().float(),
                                B.half().float(), qc)
    assert y.shape == ref.shape
    err = (y - ref).norm() / ref.norm()
    assert err <= 1e-3, float(err)


def test_quantized_attention_matmuls_vs_reference_vectors(ops):
    """lqer_amd.matmul_flexible / bmm_flexible (the fused HIP GEMM for the templates' blocks of 16) against the reference's
    outputs; the second operand of Q K^T is passed as the transposed view the model code uses."""
    import json
    import os

    import numpy as np

    import lqer_amd

    here = os.path.join(os.path.dirname(__file__), "golden")
    g = np.load(os.path.join(here, "matmul.npz"))
    qc = json.load(open(os.path.join(here, "matmul_config.json")))
    t = lambda k: torch.from_numpy(g[k]).to(DEV)
    out = lqer_amd.get_quantized_func("matmul", qc)(t("qk/x"), t("qk/y").transpose(1, 2).contiguous().transpose(1, 2), q_config=qc)
    assert (out.cpu() - torch.from_numpy(g["qk/out"])).norm() / torch.from_numpy(g["qk/out"]).norm() <= 1e-6
    out = lqer_amd.matmul_flexible(t("pv/x"), t("pv/y"), qc)
    assert (out.cpu() - torch.from_numpy(g["pv/out"])).norm() / torch.from_numpy(g["pv/out"]).norm() <= 1e-6
    out = lqer_amd.bmm_flexible(t("bmm/x"), t("bmm/y"), qc)
    assert (out.cpu() - torch.from_numpy(g["bmm/out"])).norm() / torch.from_numpy(g["bmm/out"]).norm() <= 1e-6
    with pytest.raises(KeyError):
        lqer_amd.matmul_flexible(t("pv/x"), t("pv/y"), {"name": "flexible", "x_quantizer": qc["x_quantizer"]})
    with pytest.raises(RuntimeError):  # no software fallback
        lqer_amd.matmul_flexible(t("pv/x").cpu(), t("pv/y").cpu(), qc)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.float16, 1e-3), (torch.bfloat16, 5e-3)])
@pytest.mark.parametrize("bh,s1,s2,d", [(3, 300, 200, 128), (2, 129, 70, 64), (1, 16, 16, 40), (4, 2048, 2048, 128),
                                        (2, 333, 1100, 72), (1, 130, 1537, 40)])  # (the last two: k_qmatmul_xr with ragged tiles / K)
def test_fused_quantized_matmul_vs_oracle(ops, dtype, tol, bh, s1, s2, d):
    """lqer_matmul_q (x quantized in the GEMM's load path, y through a bf16 image) against the oracle's
    matmul(x_quantizer(x), w_quantizer(y)) for both attention products: Q K^T with y the transposed VIEW of K (blocks of 16
    consecutive tokens per head feature) and P V with y = V (blocks along the head dim); ragged s / d, three dtypes."""
    import json

    import lqer_amd

    here = os.path.join(os.path.dirname(__file__), "golden")
    qc = json.load(open(os.path.join(here, "matmul_config.json")))
    g = torch.Generator().manual_seed(bh * 1000 + s1 + d)
    q = torch.randn(bh, s1, d, generator=g).to(dtype)
    k = (torch.randn(bh, s2, d, generator=g) * torch.logspace(-2, 1, d)).to(dtype)
    v = torch.randn(bh, s2, d, generator=g).to(dtype)
    if bh * s1 * s2 > 4e6:  # the BASELINE-size case: compare two heads' worth of rows with the oracle
        sl = slice(0, 1)
    else:
        sl = slice(0, bh)
    kt = k.to(DEV).transpose(1, 2)  # a view: dense along the contraction dim
    out = lqer_amd.matmul_flexible(q.to(DEV), kt, qc)
    assert out.shape == (bh, s1, s2) and out.dtype == dtype
    ref = O.matmul_flexible(q[sl].float(), k[sl].float().transpose(1, 2), qc)
    err = (out[sl].float().cpu() - ref).norm() / ref.norm()
    assert err <= tol, float(err)
    p = torch.softmax(out.float() / d ** 0.5, dim=-1).to(dtype)
    o2 = lqer_amd.matmul_flexible(p, v.to(DEV), qc)
    assert o2.shape == (bh, s1, d)
    ref2 = O.matmul_flexible(p[sl].float().cpu(), v[sl].float(), qc)
    err2 = (o2[sl].float().cpu() - ref2).norm() / ref2.norm()
    assert err2 <= tol, float(err2)
    # the same bits whatever the layout of y: a dense copy of the transposed view takes the j-contiguous image kernel
    out_c = lqer_amd.matmul_flexible(q.to(DEV), kt.contiguous(), qc)
    assert torch.equal(out_c, out)
    # 2-D operands
    assert torch.equal(lqer_amd.matmul_flexible(q[0].to(DEV), kt[0], qc), out[0])


def test_quantized_matmul_blocks_other_than_16(ops, monkeypatch):
    """matmul_flexible with block lengths other than the templates' 16 (quantized_functions/matmul.py:12-29 takes any
    block_size): the library's standalone quantizer writes the operand's bf16 image, the library's own image / product kernels
    take it as it is - torch.matmul is never reached.  Against the reference's vectors (blocks of 32 on both operands; x in
    blocks of 32 with one block per row of y) and, for mixed settings at larger shapes, against the oracle."""
    import json

    import numpy as np

    import lqer_amd
    from lqer_amd import functional

    here = os.path.join(os.path.dirname(__file__), "golden")
    g = np.load(os.path.join(here, "matmul.npz"))
    cfgs = json.load(open(os.path.join(here, "matmul_config_blocks.json")))
    monkeypatch.setitem(functional.MATMUL_MAP, "matmul", lambda *a, **k: (_ for _ in ()).throw(AssertionError("torch.matmul reached")))
    t = lambda k: torch.from_numpy(g[k]).to(DEV)
    for name in ("b32", "brow"):
        out = lqer_amd.matmul_flexible(t(f"{name}/x"), t(f"{name}/y"), cfgs[name]).cpu()
        ref = torch.from_numpy(g[f"{name}/out"])
        assert float((out - ref).norm() / ref.norm()) <= 1e-6, name
    # the transposed VIEW of K as y (dense along k): made dense along j first, same result
    out_v = lqer_amd.matmul_flexible(t("b32/x"), t("b32/y").transpose(1, 2).contiguous().transpose(1, 2), cfgs["b32"]).cpu()
    assert float((out_v - torch.from_numpy(g["b32/out"])).norm() / torch.from_numpy(g["b32/out"]).norm()) <= 1e-6
    bfp = lambda blk: dict(name="block_fp", width=8, exponent_width=8, exponent_bias=None, block_size=[1, blk], skip_first_dim=True)
    gen = torch.Generator().manual_seed(9)
    for dtype, tol in ((torch.float32, 2e-6), (torch.float16, 1e-3)):
        for bx, by, (bh, s1, s2, d) in ((64, 16, (3, 300, 200, 128)), (16, 32, (2, 129, 70, 64)), (-1, 64, (2, 260, 520, 128)), (32, -1, (1, 16, 16, 40))):
            qc = dict(name="flexible", default=False, x_quantizer=bfp(bx), w_quantizer=bfp(by))
            x = torch.randn(bh, s1, d, generator=gen).to(dtype)
            y = (torch.randn(bh, d, s2, generator=gen) * torch.logspace(-2, 1, s2)).to(dtype)
            out = lqer_amd.matmul_flexible(x.to(DEV), y.to(DEV), qc)
            assert out.dtype == dtype and out.shape == (bh, s1, s2)
            ref = O.matmul_flexible(x.float(), y.float(), qc)
            assert float((out.float().cpu() - ref).norm() / ref.norm()) <= tol, (bx, by, dtype)
    with pytest.raises(NotImplementedError):  # a block length the quantizer kernels do not have (not 16 n): refused, never approximated
        lqer_amd.matmul_flexible(t("b32/x"), t("b32/y"), dict(cfgs["b32"], x_quantizer=bfp(24)))
    # (round 6) the quantizer's default lone [16] on 3-D operands = [1, S, 16] tiles over all rows of a batch element: the HIP tile
    # quantizer + the library product (rounds 1-5 read it per row), against the reference's vector
    monkeypatch.undo()
    out = lqer_amd.matmul_flexible(t("lone/x"), t("lone/y"), cfgs["lone"]).cpu()
    ref = torch.from_numpy(g["lone/out"])
    assert float((out - ref).norm() / ref.norm()) <= 1e-6


def test_fused_quantized_matmul_takes_4d_operands_and_large_batches(ops, monkeypatch):
    """The llama call sites hand matmul_flexible 4-D [bsz, heads, ..] operands (reference llama_decoder.py:263,294): they are
    folded into one batch dim and run the fused kernel - same bits as head by head; a batch beyond the grid.z limit goes in
    chunks (limit lowered here); broadcasting leading dims keep the two-step route."""
    import json

    import lqer_amd
    from lqer_amd import functional

    qc = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "matmul_config.json")))
    g = torch.Generator().manual_seed(4)
    bsz, h, s, d = 2, 3, 70, 64
    q = torch.randn(bsz, h, s, d, generator=g).half().to(DEV)
    k = torch.randn(bsz, h, s, d, generator=g).half().to(DEV)
    v = torch.randn(bsz, h, s, d, generator=g).half().to(DEV)
    calls = []
    real = functional._matmul_fused
    monkeypatch.setattr(functional, "_matmul_fused", lambda *a: (calls.append(a[0].shape), real(*a))[1])
    s4 = lqer_amd.matmul_flexible(q, k.transpose(2, 3), qc)
    o4 = lqer_amd.matmul_flexible(torch.softmax(s4.float(), -1).half(), v, qc)
    assert len(calls) == 2 and s4.shape == (bsz, h, s, s) and o4.shape == (bsz, h, s, d)
    s3 = lqer_amd.matmul_flexible(q.reshape(bsz * h, s, d), k.reshape(bsz * h, s, d).transpose(1, 2), qc)
    assert torch.equal(s4.reshape(bsz * h, s, s), s3)
    ref = O.matmul_flexible(q[1, 2].float().cpu(), k[1, 2].float().cpu().t(), qc)
    assert float((s4[1, 2].float().cpu() - ref).norm() / ref.norm()) <= 1e-3
    monkeypatch.setattr(functional, "_MAX_GRID_Z", 4)  # 6 heads -> chunks of 4 + 2
    assert torch.equal(lqer_amd.matmul_flexible(q, k.transpose(2, 3), qc), s4)
    # leading dims that broadcast ([1, h, ..] x [bsz, h, ..]): not the fused kernel, the same bits
    n_before = len(calls)
    sb = lqer_amd.matmul_flexible(q[:1], k.transpose(2, 3), qc)
    assert len(calls) == n_before and sb.shape == s4.shape
    assert float((sb[0].float() - s4[0].float()).norm() / s4[0].float().norm()) <= 1e-3  # (another GEMM's summation order)
    with pytest.raises(RuntimeError):  # torch.bmm's own rule
        lqer_amd.bmm_flexible(q, k.transpose(2, 3), qc)


@pytest.mark.parametrize("M", [4, 300])
def test_forward_captured_in_a_graph(ops, M):
    """include/lqer_hip.h: calls are stream-ordered, perform no host synchronisation and may be captured in a
    hipGraph.  Capture one module forward (small-M kernel at M = 4, tile kernel at M = 300), replay it on new input
    and compare bit for bit with the eager call."""
    import lqer_amd
    from bench import MXINT_Q, make_case

    K, N, r = 512, 768, 32
    x, W, A, B = make_case(M, K, N, r, seed=9)
    mod = lqer_amd.LinearFlexibleLqer(K, N, bias=False, q_config=MXINT_Q, l_config={"rank": r})
    mod.load_state_dict({"weight": W, "A": A, "B": B})
    mod = mod.to(DEV).half()
    xs = x.half().to(DEV)
    x2 = (x.flip(0) * 1.5).half().to(DEV)
    eager1, eager2 = mod(xs).clone(), mod(x2).clone()  # also packs and sizes the workspace before the capture
    static_x = xs.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        mod(static_x)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static_y = mod(static_x)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_y, eager1)
    static_x.copy_(x2)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_y, eager2)


@pytest.mark.parametrize("with_scale,w_block", [(False, [1, 16]), (True, [1, 16]), (False, [8, 16]), (True, [16])])
def test_gpu_approximator_reconstruction_error(ops, with_scale, w_block):
    """lqer_amd.approximate.lqer_factors (HIP quantizers + rocSOLVER SVD) against the oracle's restatement of the
    reference approximators: the factors are not unique, the error they leave is - |E^T - A B| must agree, and it
    must beat the rank-0 error by the margin the oracle sees."""
    from lqer_amd.approximate import lqer_factors

    torch.manual_seed(3)
    N, K, r = 192, 160, 16
    W = 0.02 * torch.randn(N, K)
    W[:, 5] *= 20
    # (w_block [8, 16] / [16] with skip_first_dim = false: 2-D weight tiles - the factors must correct the error of THAT quantizer)
    w_cfg = dict(name="block_fp", width=4, exponent_width=8, exponent_bias=None, block_size=w_block, skip_first_dim=False)
    ab_cfg = dict(name="block_fp", width=8, exponent_width=8, exponent_bias=None, block_size=[16, 1], skip_first_dim=False)
    scale = (0.5 + torch.rand(K)) if with_scale else None
    A, B = lqer_factors(W.to(DEV), w_cfg, r, ab_cfg, ab_cfg, scale.to(DEV) if with_scale else None)
    Ao, Bo = O.lqer_factors(W, w_cfg, r, ab_cfg, ab_cfg, scale)
    assert A.shape == (K, r) and B.shape == (r, N)
    err_t = (W - O.get_quantizer(w_cfg)(W)).t()
    e_gpu = (err_t - A.cpu() @ B.cpu()).norm() / err_t.norm()
    e_ora = (err_t - Ao @ Bo).norm() / err_t.norm()
    assert abs(float(e_gpu) - float(e_ora)) <= 2e-3 * float(e_ora), (float(e_gpu), float(e_ora))
    assert float(e_gpu) < 0.97  # a rank-16 correction of a 160-dim error removes a visible part of it
    # the factors are 8-bit block-floating-point numbers: in every block of 16 along dim 0 all values are integer
    # multiples of (block maximum rounded up to a power of two) / 128
    for t in (A.cpu(), B.cpu()):
        blk = t.t().reshape(t.shape[1], -1, 16)
        step = 2.0 ** torch.ceil(torch.log2(blk.abs().amax(-1, keepdim=True).clamp_min(1e-30))) / 128
        assert torch.all((blk / step - torch.round(blk / step)).abs() < 1e-4)


@pytest.mark.parametrize("cfg,K,N", [("mxint", 320, 8192), ("int", 384, 8192), ("opt", 64, 8192), ("opt", 200, 8000)])
def test_large_m_tile_kernel_vs_oracle(ops, cfg, K, N):
    """M = 4096, N = 8192 (16 x 32 tiles of 256 x 256: the large-M kernel of gemm_w4a8_m256.hip) against the oracle:
    MXINT blocks of 16, the INT configuration (per-token B_out blocks: row-block maxima from the pre-pass) and a bias;
    K = 320 / 384 / 64 give 5, 6 and 1 k-steps (ring wrap-around and the shortest pipeline)."""
    import lqer_amd
    from bench import INT_Q, MXINT_Q, OPT_Q, make_case

    qc = {"mxint": MXINT_Q, "int": INT_Q, "opt": OPT_Q}[cfg]
    M, r = 4096, 32  # (K = 200, N = 8000: ragged K and N - padded k-steps, guarded output columns)
    bias = cfg == "opt"
    case = make_case(M, K, N, r, seed=31, bias=bias, quantize_ab=cfg != "int")
    x, W, A, B = case[:4]
    b = case[4] if bias else None
    mod = lqer_amd.LinearFlexibleLqer(K, N, bias=bias, q_config=qc, l_config={"rank": r})
    sd = {"weight": W, "A": A, "B": B}
    if bias:
        sd["bias"] = b
    mod.load_state_dict(sd)
    mod = mod.to(DEV).half()
    xin = x.half()
    y = mod(xin.to(DEV)).float().cpu()
    ref = O.lqer_linear_forward(xin.float(), W.half().float(), b.half().float() if bias else None, A.half().float(),
                                B.half().float(), qc)
    err = (y - ref).norm() / ref.norm()
    assert err <= 1e-3, float(err)
    # rows are independent: the 128-row kernel (M = 300: one round of either tile size, the small tiles are chosen) must
    # give the same bits for the same rows
    y2 = mod(xin[:300].to(DEV)).float().cpu()
    assert torch.equal(y2, y[:300])


@pytest.mark.parametrize("dtype,tol,native", [(torch.float16, 1e-3, True), (torch.float16, 1e-3, False), (torch.bfloat16, 5e-3, True),
                                              (torch.float32, 2e-5, True)])
@pytest.mark.parametrize("M,K,N", [(1, 1024, 768), (40, 1024, 768), (300, 1000, 700), (4096, 256, 8192)])
def test_passthrough_activations_vs_oracle(ops, dtype, tol, native, M, K, N):
    """The reference's INT templates as shipped (llama-7b-int.toml: x_quantizer = passthrough, i.e. W4A16; A_out and
    B_out fall back to it; A, B unquantized) through the three GEMM kernels (M = 1 / 40: small-M, 300: 128-row tiles,
    4096 x 8192: 256-row tiles).  fp16 tensors run the fp16 MFMA main loops (native) or, like bf16 (1) and fp32 (3),
    travel as exact bf16 limbs (2)."""
    import lqer_amd
    from bench import A16_Q, make_case

    r = 64
    x, W, A, B = make_case(M, K, N, r, seed=5, quantize_ab=False)
    mod = lqer_amd.LinearFlexibleLqer(K, N, bias=False, q_config=A16_Q, l_config={"rank": r})
    mod.load_state_dict({"weight": W, "A": A, "B": B})
    mod = mod.to(DEV).to(dtype)
    mod.a16_native = native
    xin = x.to(dtype)
    y = mod(xin.to(DEV))
    assert y.dtype == dtype
    assert mod._x_f16 == (native and dtype == torch.float16)
    assert mod._limbs() == {torch.bfloat16: (1, 2), torch.float16: (1, 2) if native else (2, 2), torch.float32: (3, 3)}[dtype]
    ref = O.lqer_linear_forward(xin.float(), W.to(dtype).float(), None, A.to(dtype).float(), B.to(dtype).float(), A16_Q)
    err = (y.float().cpu() - ref).norm() / ref.norm()
    assert err <= tol, float(err)
    if dtype != torch.float32:
        # exact products + fp32 accumulation: the output is the correctly rounded oracle value except where summation
        # order moves a sum across a rounding boundary - a 1-limb (8-bit) activation would miss most elements
        want = ref.to(dtype)
        miss = (y.cpu() != want)
        assert miss.float().mean() <= 0.02, float(miss.float().mean())
        assert (y.cpu().float() - want.float()).abs().max() <= 2.0 ** (-9 if dtype == torch.float16 else -6) * ref.abs().max()
    with pytest.raises(RuntimeError, match="pass-through x_quantizer"):
        mod(xin.to(DEV).to(torch.float32 if dtype != torch.float32 else torch.float16))
    # a strided view takes the copying route (the dense, aligned fp16 tensor above was its own activation image)
    wide = torch.zeros(M, K + 64, dtype=dtype, device=DEV)
    wide[:, :K] = xin.to(DEV)
    assert torch.equal(mod(wide[:, :K]), y)


def test_passthrough_fp16_route_falls_back_when_a_weight_scale_leaves_fp16(ops):
    """A weight block whose scale is below 2^-24 cannot be expanded to fp16: the module must notice at pack time and
    take the bf16-limb route (same results), never approximate."""
    import lqer_amd
    from bench import A16_Q, make_case

    M, K, N, r = 70, 512, 300, 32
    x, W, A, B = make_case(M, K, N, r, seed=12, quantize_ab=False)
    W[3, 128:256] = 0.0
    W[3, 130], W[3, 200] = 2.0 ** -23, -(2.0 ** -24)  # block of 128 with max 2^-23: exponent -23, code scale 2^-26 < 2^-24
    outs = []
    for tiny in (True, False):
        Wt = W.clone()
        if not tiny:
            Wt[3, 128:256] = 0.0
        mod = lqer_amd.LinearFlexibleLqer(K, N, bias=False, q_config=A16_Q, l_config={"rank": r})
        mod.load_state_dict({"weight": Wt, "A": A, "B": B})
        mod = mod.to(DEV).half()
        y = mod(x.half().to(DEV)).float().cpu()
        ref = O.lqer_linear_forward(x.half().float(), Wt.half().float(), None, A.half().float(), B.half().float(), A16_Q)
        assert (y - ref).norm() / ref.norm() <= 1e-3
        outs.append(mod._x_f16)
    assert outs == [False, True]


def test_passthrough_packed_checkpoint_single_copy(ops, tmp_path):
    """A packed checkpoint of a W4A16 Linear stores ONE copy of every image (the per-limb copies are rebuilt at load)
    and reproduces the forward bit for bit, also when loaded into a module of another dtype (other limb counts)."""
    import lqer_amd
    from bench import A16_Q, make_case

    M, K, N, r = 33, 320, 300, 32
    x, W, A, B = make_case(M, K, N, r, seed=9, quantize_ab=False)
    mod = lqer_amd.LinearFlexibleLqer(K, N, bias=False, q_config=A16_Q, l_config={"rank": r})
    mod.load_state_dict({"weight": W.half().float(), "A": A.half().float(), "B": B.half().float()})
    mod = mod.to(DEV).half()
    Kp, Np = 320, 512
    for native in (False, True):
        mod.invalidate_packed(weight_changed=False)
        mod.a16_native = native
        y = mod(x.half().to(DEV))
        st = mod.packed_state()
        assert st["w"].numel() == (Np // 16) * (Kp // 64) * 576
        assert mod._packed["w"].numel() == (1 if native else 2) * st["w"].numel() and mod._x_f16 == native
        for dt in (torch.float16, torch.float32):
            m2 = lqer_amd.LinearFlexibleLqer(K, N, bias=False, q_config=A16_Q, l_config={"rank": r}).to(dt)
            m2.a16_native = native
            m2.load_packed_state({k: v.cpu() for k, v in st.items()}, DEV)
            m2 = m2.to(DEV)
            y2 = m2(x.half().to(DEV).to(dt))
            if dt == torch.float16:
                assert torch.equal(y2, y)
            else:
                assert (y2 - y.float()).norm() / y.float().norm() <= 5e-4  # (fp32 output against its fp16 rounding)
        # a dtype cast of the packed-only module rebuilds the images for the new element type
        m3 = m2.half()
        assert m3._x_f16 == native and torch.equal(m3(x.half().to(DEV)), y)


def test_randomised_parity_sweep(ops):
    """tools/fuzz_parity.py with a fixed seed: 24 random (shape, rank, dtype, configuration) cases across the small-M,
    128-row and 256-row kernels against the oracle."""
    import random
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from tools import fuzz_parity as F

    rng = random.Random(2024)
    for i in range(24):
        c = F.one_case(rng)
        err, tol = F.run_case(*c, DEV)
        assert err <= tol, (i, c, err, tol)


def test_size_independent_properties_full_size(ops):
    """At BASELINE's full size (M=2048, 4096x4096, r=32): rows and output columns are independent,
    so a row permutation, a row split and a column split must reproduce the same bits."""
    import lqer_amd
    from bench import make_case, MXINT_Q

    K = N = 4096
    M, r = 2048, 32
    x, W, A, B = make_case(M, K, N, r, seed=1)
    x = x.half().to(DEV)
    mod = lqer_amd.LinearFlexibleLqer(K, N, bias=False, q_config=MXINT_Q, l_config={"rank": r})
    mod.load_state_dict({"weight": W, "A": A, "B": B})
    mod = mod.to(DEV).half()
    y = mod(x)
    assert torch.isfinite(y).all()
    perm = torch.randperm(M, device=DEV)
    assert torch.equal(mod(x[perm]), y[perm])
    assert torch.equal(mod(x[:1000]), y[:1000])
    assert torch.equal(mod(x[1000:1001]), y[1000:1001])
    half = lqer_amd.LinearFlexibleLqer(K, N // 2, bias=False, q_config=MXINT_Q, l_config={"rank": r})
    half.load_state_dict({"weight": W[: N // 2], "A": A, "B": B[:, : N // 2]})
    assert torch.equal(half.to(DEV).half()(x), y[:, : N // 2])
    # 3-D input = flattened 2-D input (reference blocks both the same way)
    assert torch.equal(mod(x.reshape(2, M // 2, K)).reshape(M, N), y)
    # zero A,B: the side path contributes nothing -> LinearFlexible
    z = lqer_amd.LinearFlexibleLqer(K, N, bias=False, q_config=MXINT_Q, l_config={"rank": r})
    z.load_state_dict({"weight": W}, strict=False)
    f = lqer_amd.LinearFlexible(K, N, bias=False, q_config=dict(MXINT_Q, name="flexible"), l_config=None)
    f.load_state_dict({"weight": W})
    assert torch.equal(z.to(DEV).half()(x), f.to(DEV).half()(x))


def test_run_to_run_bit_stability(ops):
    """Race screen: the kernel keeps loads in flight across barriers and hand-counts its waits, so the
    same launch must give the same bits every time (a too-early LDS read shows up as rare differences).
    Several shapes, 20 launches each, every output element compared."""
    import lqer_amd
    from bench import make_case, MXINT_Q

    for (M, K, N, r) in ((2048, 4096, 4096, 32), (300, 1024, 768, 32), (2048, 11008, 4096, 32)):
        x, W, A, B = make_case(M, K, N, r, seed=3)
        mod = lqer_amd.LinearFlexibleLqer(K, N, bias=False, q_config=MXINT_Q, l_config={"rank": r})
        mod.load_state_dict({"weight": W, "A": A, "B": B})
        mod = mod.to(DEV).half()
        xd = x.half().to(DEV)
        y0 = mod(xd).clone()
        for _ in range(20):
            assert torch.equal(mod(xd), y0), (M, K, N)


def test_edge_shapes_and_errors(ops):
    import lqer_amd
    from bench import MXINT_Q

    mod = lqer_amd.LinearFlexibleLqer(64, 48, bias=True, q_config=MXINT_Q, l_config={"rank": 16}).to(DEV)
    assert mod(torch.zeros(0, 64, device=DEV)).shape == (0, 48)
    assert mod(torch.zeros(2, 0, 64, device=DEV)).shape == (2, 0, 48)
    y0 = mod(torch.zeros(3, 64, device=DEV))
    assert torch.equal(y0, mod.bias.detach().expand(3, 48))  # all-zero activations: y = b_q
    with pytest.raises(RuntimeError):
        mod(torch.zeros(3, 65, device=DEV))
    with pytest.raises(RuntimeError):
        mod(torch.zeros(3, 64))  # CPU tensor: no fallback
    bad = dict(MXINT_Q, x_quantizer=dict(name="minifloat", width=8))
    with pytest.raises(NotImplementedError):
        lqer_amd.LinearFlexibleLqer(64, 48, q_config=bad, l_config={"rank": 16})
    # non-contiguous rows
    xb = torch.randn(5, 128, device=DEV)
    assert torch.equal(mod(xb[:, :64]), mod(xb[:, :64].contiguous()))


def test_passthrough_abi_errors(ops):
    """The C ABI refuses what it cannot do exactly: a pass-through format without its significand width, the fp16
    route with another element type, xq == x for a tensor that is not its own padded image."""
    import ctypes as C

    from lqer_amd import _lib
    from lqer_amd._lib import LinearDesc, LinearSizes, QFmt

    L = _lib.lib()
    mx = QFmt(_lib.Q_MXINT, 8, 16, 8, 127)
    w4 = QFmt(_lib.Q_MXINT, 4, 128, 8, 127)
    none = QFmt(_lib.Q_PASSTHROUGH, 0, 0, 8, 127)
    sz = LinearSizes()
    d = LinearDesc(256, 256, 16, 0, QFmt(_lib.Q_PASSTHROUGH, 0, 0, 8, 127), w4, none, mx, mx)
    assert L.lqer_linear_sizes(C.byref(d), 8, C.byref(sz)) != 0 and b"significand" in L.lqer_last_error()
    one = LinearSizes()
    assert L.lqer_linear_sizes(C.byref(LinearDesc(256, 256, 16, 0, mx, w4, none, mx, mx)), 8, C.byref(one)) == 0
    for width, copies in ((8, 1), (11, 2), (24, 3)):
        d = LinearDesc(256, 256, 16, 0, QFmt(_lib.Q_PASSTHROUGH, width, 0, 8, 127), w4, none, mx, mx)
        assert L.lqer_linear_sizes(C.byref(d), 8, C.byref(sz)) == 0
        assert (sz.w_packed, sz.a_t, sz.b_t) == (copies * one.w_packed, copies * one.a_t, one.b_t)
    d = LinearDesc(256, 256, 16, 0, mx, w4, none, QFmt(_lib.Q_PASSTHROUGH, 16, 0, 8, 127), mx)
    assert L.lqer_linear_sizes(C.byref(d), 8, C.byref(sz)) == 0 and sz.b_t == 2 * one.b_t
    # fp16 route: fp16 tensors only; xq == x only for a dense, aligned tensor with padded extents
    d = LinearDesc(256, 256, 0, 0, QFmt(_lib.Q_PASSTHROUGH_F16, 11, 0, 8, 127), w4, none, none, none)
    x = torch.zeros(300, 256, dtype=torch.float16, device=DEV)
    xq = torch.zeros(512 * 256, dtype=torch.float16, device=DEV)
    args = (x.data_ptr(), None, 0, xq.data_ptr(), None, None, 0, None)
    assert L.lqer_quantize_act_xa(C.byref(d), x.data_ptr(), _lib.BF16, 300, 256, None, 0, xq.data_ptr(), None, None, 0, None) != 0
    assert b"fp16 tensors" in L.lqer_last_error()
    assert L.lqer_quantize_act_xa(C.byref(d), x.data_ptr(), _lib.F16, 300, 256, None, 0, x.data_ptr(), None, None, 0, None) != 0
    assert b"xq == x" in L.lqer_last_error()  # 300 rows: neither a multiple of 256 nor a decode size
    assert L.lqer_quantize_act_xa(C.byref(d), x.data_ptr(), _lib.F16, 33, 256, None, 0, x.data_ptr(), None, None, 0, None) == 0
    assert L.lqer_quantize_act_xa(C.byref(d), x.data_ptr(), _lib.F16, 256, 256, None, 0, x.data_ptr(), None, None, 0, None) == 0
    assert L.lqer_quantize_act_xa(C.byref(d), x.data_ptr(), _lib.F16, 300, 256, None, 0, xq.data_ptr(), None, None, 0, None) == 0
    torch.cuda.synchronize()
    bad = QFmt(_lib.Q_PASSTHROUGH_F16, 11, 0, 8, 127)
    d = LinearDesc(256, 256, 16, 0, mx, w4, none, bad, mx)  # the fp16 kind is an x format only
    assert L.lqer_lowrank_xa(C.byref(d), xq.data_ptr(), 8, xq.data_ptr(), 1, xq.data_ptr(), xq.data_ptr(), 1 << 16, None) != 0


@pytest.mark.parametrize("M,r,bout", [(1, 32, "mx"), (16, 32, "mx"), (17, 64, "mx"), (64, 64, "pass"), (33, 16, "mx")])
def test_decode_route_reduces_partials_in_the_gemm(ops, M, r, bout):
    """Decode sizes: lqer_linear_forward runs two launches (the small-M GEMM sums the split-K partial tiles of x A and
    applies A_out itself).  Same bits as the three-launch split API with a materialised xaq; and not offered for
    formats or sizes it does not cover."""
    import ctypes as C

    import lqer_amd
    from bench import INT_Q, MXINT_Q, make_case
    from lqer_amd import _lib

    K, N = 1024, 768
    qc = MXINT_Q if bout == "mx" else dict(MXINT_Q, B_out_quantizer={"name": "passthrough"})
    x, W, A, B = make_case(M, K, N, r, seed=21)
    mod = lqer_amd.LinearFlexibleLqer(K, N, bias=False, q_config=qc, l_config={"rank": r})
    mod.load_state_dict({"weight": W, "A": A, "B": B})
    mod = mod.to(DEV).half()
    xd = x.half().to(DEV)
    y = mod(xd)  # lqer_linear_forward: the decode route
    L = _lib.lib()
    desc = mod._desc()
    assert L.lqer_decode_partials(C.byref(desc), M) == 1 and L.lqer_decode_partials(C.byref(desc), 65) == 0
    p = mod._packed
    Mp, Kp, rp = L.lqer_padded_m(M), L.lqer_padded_k(K), L.lqer_padded_r(r)
    xq = torch.empty(Mp * Kp, dtype=torch.bfloat16, device=DEV)
    xaq = torch.empty(Mp * rp, dtype=torch.bfloat16, device=DEV)
    nscr = L.lqer_lowrank_xa_scratch_bytes(C.byref(desc), M)
    scr = torch.empty(nscr, dtype=torch.uint8, device=DEV)
    y3 = torch.empty_like(y)
    _lib.check(L.lqer_quantize_act_xa(C.byref(desc), xd.data_ptr(), _lib.F16, M, K, p["a_t"].data_ptr(), p["a_limbs"], xq.data_ptr(),
                                      xaq.data_ptr(), scr.data_ptr(), nscr, None), "xa")
    _lib.check(L.lqer_linear_gemm(C.byref(desc), xq.data_ptr(), M, p["w"].data_ptr(), xaq.data_ptr(), p["b_t"].data_ptr(), p["b_limbs"],
                                  None, y3.data_ptr(), _lib.F16, N, scr.data_ptr(), nscr, None), "gemm")
    torch.cuda.synchronize()
    assert torch.equal(y3, y)
    # the split API with xaq == NULL is the same two-launch route
    y2 = torch.empty_like(y)
    _lib.check(L.lqer_quantize_act_xa(C.byref(desc), xd.data_ptr(), _lib.F16, M, K, p["a_t"].data_ptr(), p["a_limbs"], xq.data_ptr(),
                                      None, scr.data_ptr(), nscr, None), "xa partials")
    _lib.check(L.lqer_linear_gemm(C.byref(desc), xq.data_ptr(), M, p["w"].data_ptr(), None, p["b_t"].data_ptr(), p["b_limbs"],
                                  None, y2.data_ptr(), _lib.F16, N, scr.data_ptr(), nscr, None), "gemm from partials")
    torch.cuda.synchronize()
    assert torch.equal(y2, y)
    # not offered: per-token blocks (INT configuration), more than 64 tokens
    di = lqer_amd.LinearFlexibleLqer(K, N, bias=False, q_config=INT_Q, l_config={"rank": r})._desc()
    assert L.lqer_decode_partials(C.byref(di), M) == 0
    assert L.lqer_quantize_act_xa(C.byref(di), xd.data_ptr(), _lib.F16, M, K, p["a_t"].data_ptr(), p["a_limbs"], xq.data_ptr(),
                                  None, scr.data_ptr(), nscr, None) != 0
    assert L.lqer_linear_gemm(C.byref(desc), xq.data_ptr(), 65, p["w"].data_ptr(), None, p["b_t"].data_ptr(), p["b_limbs"],
                              None, y2.data_ptr(), _lib.F16, N, scr.data_ptr(), nscr, None) != 0
