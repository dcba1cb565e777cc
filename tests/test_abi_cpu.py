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
    assert C.sizeof(_lib.QFmt) == 20 and C.sizeof(_lib.LinearDesc) == 16 + 5 * 20


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


def test_make_qfmt_schema():
    from lqer_amd import _lib, ops

    f = ops.make_qfmt(dict(name="block_fp", width=8, exponent_width=8, exponent_bias="NA", block_size=[1, 16], skip_first_dim=True))
    assert (f.kind, f.width, f.block, f.exp_width, f.exp_bias) == (_lib.Q_MXINT, 8, 16, 8, 127)
    assert ops.make_qfmt(dict(name="block_fp", width=8, block_size=[-1])).block == -1
    assert ops.make_qfmt(dict(name="block_fp", width=4, block_size=[1, -1])).block == -1
    assert ops.make_qfmt(dict(name="block_fp", width=4, block_size=128)).block == 128
    assert ops.make_qfmt(dict(name="passthrough", width=16, frac_width=9)).kind == _lib.Q_PASSTHROUGH
    with pytest.raises(NotImplementedError):
        ops.make_qfmt(dict(name="block_fp", width=8, block_size=[16, 1]))
    with pytest.raises(NotImplementedError):
        ops.make_qfmt(dict(name="minifloat", width=8))


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
