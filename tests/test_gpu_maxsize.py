"""Maximum sizes: operands whose ELEMENT counts pass 2^31 (activations, images and outputs of 4 GiB and more), where any
32-bit row * stride product in a kernel or in the C ABI's workspace carving would wrap.  Rows (and batch entries of the
attention products) are independent, so the first, a middle and the LAST rows - the ones a wrapped offset would hit or
leave unwritten - are compared with the CPU oracle; everything else is checked to be finite and written.  One case per
main-loop route: bf16 tile kernel (MXINT blocks of 16), int8 route (per-token activations), pass-through fp16
activations, the fused attention product.  The activations are drawn on the GPU (a host randn of 2^31 values would take
longer than the test).   Run on the GPU box:  python -m pytest tests -m gpu -x -q"""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import lqer_oracle as O  # the checker

DEV = "cuda:0"
M_BIG = (1 << 18) + 192  # 262336 rows: not a multiple of 128 or 256 (ragged last row tile)


@pytest.fixture(scope="module")
def lq():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if torch.cuda.get_device_properties(0).total_memory < 64 << 30:
        pytest.skip("needs 64 GB of device memory")
    import lqer_amd

    return lqer_amd


def _big_x(M, K, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.empty(M, K, dtype=torch.float16, device=DEV)
    step = 1 << 15
    for m0 in range(0, M, step):  # (in slabs: no 8-GiB fp32 temporary)
        x[m0:m0 + step] = torch.randn(min(step, M - m0), K, generator=g, device=DEV, dtype=torch.float32).half()
    for c in (7, 1033, 2900):
        x[:, c] *= 30.0
    return x


def _sample_rows(M):
    mid = (M // 2 // 256) * 256
    idx = list(range(0, 48)) + list(range(mid - 24, mid + 24)) + list(range(M - 200, M))  # the last 200: past row 2^18
    return torch.tensor(idx)


@pytest.mark.parametrize("cfg,K,N,r,bias", [("mxint", 8192, 8320, 32, False), ("int", 8192, 8320, 64, False),
                                            ("a16", 8192, 8320, 64, False), ("opt", 8192, 8320, 128, True)])
def test_linear_forward_past_2_31_elements(lq, cfg, K, N, r, bias):
    from bench import A16_Q, INT_Q, MXINT_Q, OPT_Q, make_weights

    qc = {"mxint": MXINT_Q, "int": INT_Q, "a16": A16_Q, "opt": OPT_Q}[cfg]
    M = M_BIG
    assert M * K > 2 ** 31 and M * N > 2 ** 31
    g = torch.Generator().manual_seed(77)
    wts = make_weights(g, K, N, r, bias=bias, quantize_ab=cfg in ("mxint", "opt"))
    W, A, B = wts[:3]
    mod = lq.LinearFlexibleLqer(K, N, bias=bias, q_config=qc, l_config={"rank": r})
    sd = {"weight": W, "A": A, "B": B}
    if bias:
        sd["bias"] = wts[3]
    mod.load_state_dict(sd)
    mod = mod.to(DEV).half()
    x = _big_x(M, K, seed=5)
    y = torch.full((M, N), float("nan"), dtype=torch.float16, device=DEV)  # (every element must be WRITTEN)
    y.copy_(mod(x))
    torch.cuda.synchronize()
    assert y.shape == (M, N)
    for m0 in range(0, M, 1 << 16):
        assert torch.isfinite(y[m0:m0 + (1 << 16)]).all(), m0
    idx = _sample_rows(M)
    got = y[idx.to(DEV)].float().cpu()
    xs = x[idx.to(DEV)].float().cpu()
    ref = O.lqer_linear_forward(xs, W.half().float(), wts[3].half().float() if bias else None, A.half().float(),
                                B.half().float(), qc)
    err = float((got - ref).norm() / ref.norm())
    assert err <= 1e-3, err
    row_scale = ref.abs().amax(dim=1, keepdim=True)
    assert float(((got - ref).abs() / row_scale).max()) <= 2.0 ** -8
    # the same rows run alone (another kernel geometry for the int8 / fused-quantizer routes): equal up to the fp16 rounding
    # of a differently ordered fp32 sum
    small = mod(x[idx[-200:].to(DEV)]).float().cpu()
    assert float((small - got[-200:]).norm() / got[-200:].norm()) <= 1e-3
    del x, y
    torch.cuda.empty_cache()


def test_attention_products_past_2_31_elements(lq):
    """Q K^T with 40 heads x 8192 x 8192 scores (2.7e9 elements) and P V over the same probabilities."""
    here = os.path.join(os.path.dirname(__file__), "golden")
    qc = json.load(open(os.path.join(here, "matmul_config.json")))
    bh, s, d = 40, 8192, 128
    g = torch.Generator(device=DEV).manual_seed(9)
    q = torch.randn(bh, s, d, generator=g, device=DEV, dtype=torch.float32).half()
    k = torch.randn(bh, s, d, generator=g, device=DEV, dtype=torch.float32).half()
    v = torch.randn(bh, s, d, generator=g, device=DEV, dtype=torch.float32).half()
    sc = lq.matmul_flexible(q, k.transpose(1, 2), qc)
    assert sc.shape == (bh, s, s) and sc.numel() > 2 ** 31
    for h in (0, bh // 2, bh - 1):  # (the last head sits past element 2^31)
        ref = O.matmul_flexible(q[h].float().cpu(), k[h].float().cpu().t(), qc)
        got = sc[h].float().cpu()
        assert float((got - ref).norm() / ref.norm()) <= 1e-3, h
    p = sc  # scores as the left operand of the second product (values do not matter to the indexing)
    p.mul_(1.0 / 64)
    o = lq.matmul_flexible(p, v, qc)
    assert o.shape == (bh, s, d)
    for h in (0, bh - 1):
        ref = O.matmul_flexible(p[h].float().cpu(), v[h].float().cpu(), qc)
        got = o[h].float().cpu()
        assert float((got - ref).norm() / ref.norm()) <= 1e-3, h
    del sc, p, o
    torch.cuda.empty_cache()
