"""CPU-side checks: the shared library loads and exports every symbol include/lqer_hip.h declares,
struct layouts agree, and the host-side mirror of the reference interface behaves like it
(constructor, parameter names, config fall-backs, errors).  No kernel is launched here."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from lqer_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "lqer_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lqer_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from lqer_amd import _lib

    names = _declared_symbols()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/lqer_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names


def test_abi_constants_and_padding(lib):
    from lqer_amd import _lib

    hdr = open(os.path.join(ROOT, "include", "lqer_hip.h")).read()
    consts = dict(re.findall(r"#define\s+(LQER_[A-Z0-9_]+)\s+(-?\d+)", hdr))
    assert int(consts["LQER_ABI_VERSION"]) == _lib.ABI_VERSION == lib.lqer_version()
    assert (int(consts["LQER_K_ALIGN"]), int(consts["LQER_M_ALIGN"]), int(consts["LQER_N_ALIGN"]), int(consts["LQER_R_ALIGN"])) == (
        _lib.K_ALIGN, _lib.M_ALIGN, _lib.N_ALIGN, _lib.R_ALIGN)
    assert lib.lqer_padded_k(4096) == 4096 and lib.lqer_padded_k(4097) == 4160 and lib.lqer_padded_k(1) == 64
    assert lib.lqer_padded_n(11008) == 11008 and lib.lqer_padded_n(50) == 256
    assert lib.lqer_padded_m(1) == 256 and lib.lqer_padded_r(32) == 32 and lib.lqer_padded_r(33) == 48
    assert C.sizeof(_lib.QFmt) == 20 and C.sizeof(_lib.LinearDesc) == 16 + 5 * 20 + 4  # (+ tuning, ABI 9)
    assert _lib.LinearDesc(1, 1, 0, 0).tuning == 0  # positional construction without the knob field: defaults


def test_struct_sizes_match_the_library(lib):
    """ABI 11: the library says how large it compiled its structs; the package's binding agrees."""
    from lqer_amd import _lib

    assert lib.lqer_sizeof_qfmt() == C.sizeof(_lib.QFmt)
    assert lib.lqer_sizeof_linear_desc() == C.sizeof(_lib.LinearDesc)
    assert lib.lqer_sizeof_linear_sizes() == C.sizeof(_lib.LinearSizes)
    assert lib.lqer_sizeof_group_member() == C.sizeof(_lib.GroupMember)


def test_integration_stub_layouts_match_the_library(lib):
    """INTEGRATION.md section B is a ctypes stub a reference maintainer would copy: EXECUTE its struct definitions (and the
    group stub's) and hold them to the library's own sizeof - the document cannot drift from the header again."""
    from lqer_amd import _lib

    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", doc, flags=re.S)
    stub = next(b for b in blocks if "class LinearDesc(C.Structure)" in b)
    defs = stub[stub.index("class QFmt"): stub.index("# the library reads sizeof")]
    ns = {"C": C}
    exec(defs, ns)  # noqa: S102 - our own document
    assert C.sizeof(ns["QFmt"]) == lib.lqer_sizeof_qfmt()
    assert C.sizeof(ns["LinearDesc"]) == lib.lqer_sizeof_linear_desc()
    assert C.sizeof(ns["LinearSizes"]) == lib.lqer_sizeof_linear_sizes()
    assert [n for n, _ in ns["LinearDesc"]._fields_] == [n for n, _ in _lib.LinearDesc._fields_]
    # the stub's own import-time assertion, run against the built library
    check = stub[stub.index("for _cls, _fn in"): stub.index("def _check")]
    exec(check, dict(ns, _lib=lib))  # noqa: S102
    grp = next(b for b in blocks if "class GroupMember(C.Structure)" in b)
    gdefs = grp[grp.index("class GroupMember"): grp.index("tab = ")]
    exec(gdefs, ns)  # noqa: S102
    assert C.sizeof(ns["GroupMember"]) == lib.lqer_sizeof_group_member()


def test_sizes_and_argument_errors_without_gpu(lib):
    from lqer_amd import _lib, ops

    f8 = ops.make_qfmt(dict(name="block_fp", width=8, block_size=[1, 16]))
    f4 = ops.make_qfmt(dict(name="block_fp", width=4, block_size=[1, 16]))
    d = _lib.LinearDesc(4096, 4096, 32, 0, f8, f4, f8, f8, f8)
    sz = _lib.LinearSizes()
    assert lib.lqer_linear_sizes(C.byref(d), 2048, C.byref(sz)) == 0
    assert sz.w_packed == (4096 // 16) * (4096 // 64) * 576  # 4.5 bits per weight
    assert sz.workspace >= 2048 * 4096 * 2 + 2048 * 32 * 2
    bad = _lib.LinearDesc(0, 4096, 32, 0, f8, f4, f8, f8, f8)
    assert lib.lqer_linear_sizes(C.byref(bad), 1, C.byref(sz)) == -1
    assert b"bad descriptor" in lib.lqer_last_error()
    # argument validation happens before any HIP call
    assert lib.lqer_quantize_mxint(None, 0, 4, 4, 4, C.byref(f8), None, None, None, None) == -1
    assert lib.lqer_pack_weight_mxint(None, 0, 4, 4, 4, C.byref(f4), None, None, None) == -1
    assert lib.lqer_linear_forward(None, None, 0, 1, 1, None, None, None, 0, 0, None, None, 1, None, 0, None) == -1
    assert lib.lqer_linear_forward(C.byref(d), None, 1, 8, 4096, None, None, None, 1, 1, None, None, 4096, None, 0, None) == -4


def test_passthrough_descriptors_host_logic():
    """Sizes, limb counts and the decode-route predicate are host arithmetic: checked without a GPU.  Pass-through
    activations travel as 1 / 2 / 3 bf16 limbs (width 8 / 11 / 24) with the packed images repeated per limb; the fp16
    route (kind 2) keeps one copy; a pass-through A_out doubles or triples b_t."""
    from lqer_amd import _lib

    L = _lib.lib()
    mx, w4 = _lib.QFmt(_lib.Q_MXINT, 8, 16, 8, 127), _lib.QFmt(_lib.Q_MXINT, 4, 128, 8, 127)
    none = _lib.QFmt(_lib.Q_PASSTHROUGH, 0, 0, 8, 127)
    one, sz = _lib.LinearSizes(), _lib.LinearSizes()
    assert L.lqer_linear_sizes(C.byref(_lib.LinearDesc(1000, 700, 48, 0, mx, w4, none, mx, mx)), 100, C.byref(one)) == 0
    a, b = C.c_int(0), C.c_int(0)
    for kind, width, xa_width, want in ((0, 8, 16, (1, 2)), (0, 11, 16, (2, 2)), (0, 24, 24, (3, 3)), (2, 11, 16, (1, 2))):
        d = _lib.LinearDesc(1000, 700, 48, 0, _lib.QFmt(kind, width, 0, 8, 127), w4, none, _lib.QFmt(0, xa_width, 0, 8, 127), none)
        assert L.lqer_desc_limbs(C.byref(d), C.byref(a), C.byref(b)) == 0 and (a.value, b.value) == want
        assert L.lqer_linear_sizes(C.byref(d), 100, C.byref(sz)) == 0
        assert (sz.w_packed, sz.a_t, sz.b_t) == (want[0] * one.w_packed, want[0] * one.a_t, want[1] * one.b_t)
        assert sz.workspace > one.workspace or want == (1, 2)
    d = _lib.LinearDesc(1000, 700, 48, 0, _lib.QFmt(0, 0, 0, 8, 127), w4, none, mx, mx)
    assert L.lqer_linear_sizes(C.byref(d), 100, C.byref(sz)) != 0 and b"significand" in L.lqer_last_error()
    # decode route: M <= 64, blocks of 16 for x and A_out, padded rank <= 64, B_out pass-through or blocks of 16
    ok = _lib.LinearDesc(4096, 4096, 32, 0, mx, w4, none, mx, mx)
    assert [L.lqer_decode_partials(C.byref(ok), m) for m in (0, 1, 64, 65)] == [0, 1, 1, 0]
    row = _lib.QFmt(_lib.Q_MXINT, 8, -1, 8, 127)
    for bad in (_lib.LinearDesc(4096, 4096, 80, 0, mx, w4, none, mx, mx), _lib.LinearDesc(4096, 4096, 32, 0, row, w4, none, mx, mx),
                _lib.LinearDesc(4096, 4096, 32, 0, mx, w4, none, row, mx), _lib.LinearDesc(4096, 4096, 32, 0, mx, w4, none, mx, row),
                _lib.LinearDesc(4096, 4096, 0, 0, mx, w4, none, mx, mx)):
        assert L.lqer_decode_partials(C.byref(bad), 8) == 0
    assert L.lqer_decode_partials(C.byref(_lib.LinearDesc(4096, 4096, 32, 0, mx, w4, none, mx, none)), 8) == 1


def test_make_qfmt_schema():
    from lqer_amd import _lib, ops

    f = ops.make_qfmt(dict(name="block_fp", width=8, exponent_width=8, exponent_bias="NA", block_size=[1, 16], skip_first_dim=True))
    assert (f.kind, f.width, f.block, f.exp_width, f.exp_bias) == (_lib.Q_MXINT, 8, 16, 8, 127)
    assert ops.make_qfmt(dict(name="block_fp", width=8, block_size=[-1])).block == -1
    assert ops.make_qfmt(dict(name="block_fp", width=4, block_size=[1, -1])).block == -1
    assert ops.make_qfmt(dict(name="block_fp", width=4, block_size=128)).block == 128
    assert ops.make_qfmt(dict(name="passthrough", width=16, frac_width=9)).kind == _lib.Q_PASSTHROUGH
    # (round 5) an activation's blocks may span token rows: the format carries (R, L, skip_first_dim) for the module's tile route
    assert ops.make_qfmt(dict(name="block_fp", width=8, block_size=[16, 1])).act_tiles == (16, 1, True, False)
    assert not hasattr(f, "act_tiles")
    # (round 6) the quantizer's DEFAULT block_size, a lone [L] with skip_first_dim = true: per-row blocks on a 2-D tensor (the fused
    # kernels), but [1, T, L] - all token rows x L columns - on a 3-D one (utils.py:56-66, :211-237): recorded, decided per call
    fl = ops.make_qfmt(dict(name="block_fp", width=8, block_size=[16]))
    assert fl.act_tiles == (-1, 16, True, True) and fl.block == 16
    assert ops.act_tile_shape(fl, 2) == (1, 16) and ops.act_tile_shape(fl, 3) == (-1, 16)
    assert ops.act_rows_per_block(fl, (5, 64)) == 1 and ops.act_rows_per_block(fl, (2, 7, 64)) == 7 and ops.act_rows_per_block(fl, (2, 1, 64)) == 1
    assert ops.act_rows_per_block(f, (2, 7, 64)) == 1
    with pytest.raises(NotImplementedError):
        ops.make_qfmt(dict(name="minifloat", width=8))
    # "integer" (quantizers/integer.py:10-43): fixed point for x / b / A_out; frac_width rides in exp_bias, is_signed in exp_width
    fi = ops.make_qfmt(dict(name="integer", width=8, frac_width=4), "x")
    assert (fi.kind, fi.width, fi.exp_width, fi.exp_bias) == (_lib.Q_INT, 8, 1, 4)
    assert ops.make_qfmt(dict(name="integer", width=8, frac_width=6, is_signed=False), "b").exp_width == 0
    # (round 4) an integer WEIGHT: signed, 2..4 bits - two's-complement nibbles of the packed image (the code -8 included)
    fw4 = ops.make_qfmt(dict(name="integer", width=4, frac_width=2), "w")
    assert (fw4.kind, fw4.width, fw4.exp_width, fw4.exp_bias) == (_lib.Q_INT, 4, 1, 2)
    for bad in (dict(name="integer", width=4, frac_width=2, is_signed=False), dict(name="integer", width=5, frac_width=2)):
        with pytest.raises(NotImplementedError):  # 0..15 or 5-bit codes do not fit the nibble: refused, never approximated
            ops.make_qfmt(bad, "w")
    assert ops.make_qfmt(dict(name="integer", width=8, frac_width=4), "B_out").kind == _lib.Q_INT  # (round 3: inside the tile kernels)
    # the role decides how a one-entry block_size is right-aligned (quantizers/utils.py:42-67, :261-284): per row for
    # activations / weights with skip_first_dim = true (the quantizer's default) and for the 1-D bias; a 2-D tensor with
    # skip_first_dim = false reads [L] as tiles of ALL rows x L - for the weight one exponent per tile at pack time (round 3:
    # `block_rows` rides beside the C fields), for activations not on the HIP path; never silently per row
    w = dict(name="block_fp", width=4, block_size=[128], skip_first_dim=False)
    fw = ops.make_qfmt(w, "w")
    assert fw.block == 128 and fw.block_rows == -1
    ft = ops.make_qfmt(dict(w, block_size=[8, 32]), "w")
    assert ft.block == 32 and ft.block_rows == 8
    assert not hasattr(ops.make_qfmt(dict(w, block_size=[1, 16]), "w"), "block_rows")  # the templates' per-row blocks
    fx = ops.make_qfmt(dict(w, width=8), "x")  # (round 5: the tile route - a 2-D activation is blocked like a weight, utils.py:261-270)
    assert fx.act_tiles == (-1, 128, False, False) and ops.act_tile_shape(fx, 2) == (-1, 128)
    with pytest.raises(NotImplementedError, match="block 3d weight"):  # (the reference's own refusal, utils.py:279)
        ops.act_tile_shape(fx, 3)
    fx = ops.make_qfmt(dict(w, width=8, block_size=[8, 16], skip_first_dim=True), "B_out")
    assert fx.act_tiles == (8, 16, True, False) and ops.act_tile_shape(fx, 3) == (8, 16) and ops.act_tile_shape(fx, 2) == (1, 16)
    with pytest.raises(RuntimeError, match="Unsupported x.ndim"):
        ops.act_tile_shape(fx, 4)
    assert ops.make_qfmt(dict(w, skip_first_dim=True), "w").block == 128
    assert ops.make_qfmt(dict(w, block_size=[1, 128]), "w").block == 128
    assert ops.make_qfmt(dict(name="block_fp", width=8, block_size=[16], skip_first_dim=False), "b").block == 16
    with pytest.raises(AssertionError):  # the reference asserts the same for a bias (utils.py:268-271)
        ops.make_qfmt(dict(name="block_fp", width=8, block_size=[16], skip_first_dim=True), "b")


def test_module_mirrors_reference_interface():
    import lqer_amd
    from bench import MXINT_Q

    cls = lqer_amd.get_quantized_layer_cls("linear", MXINT_Q)
    assert cls is lqer_amd.LinearFlexibleLqer and issubclass(cls, torch.nn.Linear)
    m = cls(64, 48, bias=True, q_config=MXINT_Q, l_config={"rank": 16})
    sd = m.state_dict()
    assert list(sd) == ["weight", "bias", "A", "B"]
    assert sd["A"].shape == (64, 16) and sd["B"].shape == (16, 48) and not sd["A"].any()
    assert m.is_ptq and m.w_is_quantized is False
    # A_out / B_out fall back to the x quantizer's config (reference linear.py:115-124)
    assert (m._fmt["A_out"].width, m._fmt["A_out"].block) == (m._fmt["x"].width, m._fmt["x"].block)
    assert lqer_amd.get_quantized_layer_cls("linear", dict(MXINT_Q, name="flexible")) is lqer_amd.LinearFlexible
    with pytest.raises(AssertionError):
        lqer_amd.get_quantized_layer_cls("conv", MXINT_Q)
    # the reference evaluates q_config["default"] eagerly (linear.py:90): same KeyError here
    no_default = {k: v for k, v in MXINT_Q.items() if k != "default"}
    with pytest.raises(KeyError):
        cls(64, 48, q_config=no_default, l_config={"rank": 16})
    with pytest.raises(TypeError):  # "did the weight change?" has no default
        m.invalidate_packed()
    m.invalidate_packed(weight_changed=True)
    assert m.w_is_quantized is False
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(2, 64))
    with pytest.raises(NotImplementedError):
        cls(64, 48, q_config=dict(MXINT_Q, is_ptq=False), l_config={"rank": 16})(torch.zeros(2, 64))
    # strict=False partial loads as the reference harness does (llama_decoder.py:507, runners.py:220-222)
    m.load_state_dict({"weight": torch.ones(48, 64)}, strict=False)
    m.load_state_dict({"A": torch.ones(64, 16), "B": torch.ones(16, 48)}, strict=False)
    assert m._packed is None and m.w_is_quantized is False


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under lqer_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "lqer_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".sh")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), os.path.join(dirpath, f)
                assert "/root/reference" not in txt


def test_activation_tiles_take_the_tile_route():
    """Activation quantizers whose blocks span token rows (a first-dim block != 1: quantizers/utils.py:211-237 [R, L] tiles of a
    3-D tensor; :161-183 / :261-270 a 2-D tensor with skip_first_dim = false) are never approximated per row: round 5 gave them a
    route of their own (linear.py `_forward_tiles`: HIP tile quantizers around the fused GEMM) - the module says so at construction,
    the templates' per-row form does not take it, and there is still no CPU path."""
    import lqer_amd

    bfp = lambda w, bs, skip: dict(name="block_fp", width=w, exponent_width=8, exponent_bias=None, block_size=bs, skip_first_dim=skip)
    base = dict(name="flexible_lqer", is_ptq=True, default=False, w_quantizer=bfp(4, [1, 16], False), b_quantizer=bfp(8, [-1], False))
    for role, cfg in (("x_quantizer", bfp(8, [4, 16], False)), ("x_quantizer", bfp(8, [16], False)), ("A_out_quantizer", bfp(8, [2, 16], False)),
                      ("B_out_quantizer", bfp(8, [8, 1], True))):
        qc = dict(base, x_quantizer=bfp(8, [1, 16], True))
        qc[role] = cfg
        mod = lqer_amd.LinearFlexibleLqer(64, 64, bias=False, q_config=qc, l_config={"rank": 16})
        assert mod._tiles, role
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            mod(torch.zeros(2, 3, 64))
        with pytest.raises(NotImplementedError, match="tile route"):  # packed checkpoints are images of the fused path
            mod.packed_state()
    mod = lqer_amd.LinearFlexibleLqer(64, 64, bias=False, q_config=dict(base, x_quantizer=bfp(8, [1, 16], True)), l_config={"rank": 16})  # the templates' form
    assert not mod._tiles
    # (round 6) the quantizer's default, a lone [16] with skip_first_dim = true: [1, T, 16] tiles on a 3-D tensor, per-row blocks on a
    # 2-D one - the route is chosen per call (A_out / B_out fall back to the same format, linear.py:115-124)
    mod = lqer_amd.LinearFlexibleLqer(64, 64, bias=False, q_config=dict(base, x_quantizer=bfp(8, [16], True)), l_config={"rank": 16})
    assert mod._tiles and not mod._tiles_only
    assert mod._needs_tiles(torch.zeros(2, 3, 64)) and not mod._needs_tiles(torch.zeros(6, 64)) and not mod._needs_tiles(torch.zeros(4, 1, 64))
