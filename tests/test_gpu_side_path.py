"""Activation quantizer + first half of the side path (lqer_quantize_act_xa: k_quant_xa16 + k_xa_reduce4, or the separate
quantizer, k_xa_partial and reduce pass; reference quantized_layers/linear.py:148,154) at prefill sizes, against the
standalone quantizer (bit for bit) and the oracle.

xAq = A_out(xq @ A) re-quantizes an fp32 sum whose accumulation order differs from torch.matmul's.  The criterion is the
bound that follows from that: every product xq[m,k] A[k,j] is exact, so any fp32 summation order lands within D ulps of the
exact sum s (D = max(16, sqrt(K)), generous for blocked accumulation); the output must therefore equal the quantizer
applied to SOME value in [s - D ulp, s + D ulp] - element by element, with the block exponent taken from the same
interval of the block maximum.  Elements outside that envelope fail, however few they are.
Run on the GPU box:  python -m pytest tests -m gpu -x -q"""
import ctypes as C
import math

import numpy as np
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


def _bfp(width, block, skip=True):
    return dict(name="block_fp", width=width, exponent_width=8, exponent_bias=None, block_size=block, skip_first_dim=skip)


from _envelope import envelope_check as _envelope_check  # noqa: E402


CASES = [
    # M, K, r, dtype, A_out block (-1 = the whole row)
    (2048, 4096, 32, torch.float16, 16),   # C2 / C3
    (300, 1088, 32, torch.float16, 16),    # K not a multiple of the 1024-k stage, ragged last row group
    (129, 11008, 64, torch.bfloat16, 16),  # Llama down-projection width, rank 64
    (160, 4096, 128, torch.float16, 16),   # rank 128 (C5): separate quantizer, k_xa_partial, reduce pass
    (72, 512, 16, torch.float32, 16),      # one rank tile, fp32 input
    (200, 2048, 48, torch.float16, 16),    # three rank tiles
    (4096 + 8, 1024, 32, torch.float16, -1),  # A_out over the whole row
    (8192, 512, 32, torch.float16, 16),
    # rank 65..128 at M >= 512: the 128-row fused kernel (xal::k_quant_xa128: quantizer + both MFMA operands through LDS)
    (2048, 4096, 128, torch.float16, 16),   # C5
    (700, 1096, 96, torch.bfloat16, 16),    # ragged last tile, K a multiple of 8 but not of 64 (padded step), three rank tiles
    (1500, 1088, 128, torch.float16, -1),   # 17 steps over 15 chunks, A_out over the whole row
    (640, 64, 128, torch.float16, 16),      # a single step
]


@pytest.mark.parametrize("M,K,r,dtype,ablock", CASES)
def test_side_path_vs_quantizer_and_oracle(ops, M, K, r, dtype, ablock):
    from lqer_amd import _lib

    L = _lib.lib()
    g = torch.Generator().manual_seed(M * 7 + K)
    x = torch.randn(M, K, generator=g)
    for c in (7, 1033, 2900):
        if c < K:
            x[:, c] *= 30.0
    x[min(5, M - 1)] = 0.0            # a zero row: every block takes the zero path
    x[min(9, M - 1), 16:32] = 3e-9    # values inside the reference's pass-through range (fp32 inputs keep them non-zero)
    x[min(11, M - 1), 32:48] = 1e-37  # a block exponent outside the power-of-two fast path (bf16 / fp32 inputs; 0 in fp16)
    x = x.to(dtype)
    A = O.mxint_quantize(0.01 * torch.randn(K, r, generator=g), width=8, block_size=[16, 1], skip_first_dim=False)
    A = torch.where(A.abs() <= 1e-8, torch.zeros_like(A), A)  # (the reference's pass-through of |a| <= 1e-8 is not on the 8-bit grid)
    xf = ops.make_qfmt(_bfp(8, [1, 16]), "x")
    af = ops.make_qfmt(_bfp(8, [1, ablock]), "x")
    wf = ops.make_qfmt(_bfp(4, [1, 16], False), "w")
    desc = _lib.LinearDesc(K, 256, r, 0, xf, wf, xf, af, xf)
    a_t, _, a_limbs, _ = ops.pack_lowrank(A.to(DEV), torch.zeros(r, 256, device=DEV))
    assert a_limbs == 1
    xd = x.to(DEV)
    Mp, Kp, rp = L.lqer_padded_m(M), L.lqer_padded_k(K), L.lqer_padded_r(r)
    xq = torch.full((Mp, Kp), 7.0, dtype=torch.bfloat16, device=DEV)
    xaq = torch.full((Mp, rp), 7.0, dtype=torch.bfloat16, device=DEV)
    nscr = L.lqer_lowrank_xa_scratch_bytes(C.byref(desc), M)
    scr = torch.empty(max(nscr, 16), dtype=torch.uint8, device=DEV)

    def run(xq_, xaq_):
        _lib.check(L.lqer_quantize_act_xa(C.byref(desc), xd.data_ptr(), ops.dtype_code(xd), M, K, a_t.data_ptr(), a_limbs,
                                          xq_.data_ptr(), xaq_.data_ptr(), scr.data_ptr(), nscr, None), "quantize_act_xa")
        torch.cuda.synchronize()

    run(xq, xaq)
    # (1) the activation image: bit for bit the standalone quantizer's (itself pinned to the reference vectors), K padding zeroed
    ref_img = ops.quantize_act(xd, xf)
    assert torch.equal(xq[:M].view(torch.int16), ref_img[:M].view(torch.int16))
    # (2) xAq inside the envelope of the exact sum
    xq64 = xq[:M, :K].double().cpu().numpy()
    s64 = xq64 @ A.double().numpy()
    got = xaq[:M, :r].float().cpu().numpy()
    Lb = r if ablock < 0 else ablock
    bad = _envelope_check(s64, got, Lb, 7, max(16.0, math.sqrt(K)))
    assert bad == 0, f"{bad} row-blocks of xAq outside the fp32-summation envelope"
    # informational: how many entries differ from torch's own summation order (the blanket figure of round 1)
    ref = O.mxint_quantize(torch.from_numpy(xq64).float() @ A, width=8, block_size=[1, ablock], skip_first_dim=True).numpy()
    print(f"xAq entries differing from the oracle's summation order: {(got != ref).mean():.4%}")
    assert (got != ref).mean() <= 0.03
    # (2b) the two separate steps (standalone quantizer, then lqer_lowrank_xa on its image - the route of unaligned inputs) walk
    # other K chunks than the fused 64-row kernel: another fp32 summation order, the same envelope
    if rp > 64 and M >= 512:
        img = torch.zeros(Mp, Kp, dtype=torch.bfloat16, device=DEV)
        img[:M] = ref_img[:M]
        xaq3 = torch.full((Mp, rp), 7.0, dtype=torch.bfloat16, device=DEV)
        _lib.check(L.lqer_lowrank_xa(C.byref(desc), img.data_ptr(), M, a_t.data_ptr(), a_limbs, xaq3.data_ptr(), scr.data_ptr(), nscr,
                                     None), "lowrank_xa")
        torch.cuda.synchronize()
        assert _envelope_check(s64, xaq3[:M, :r].float().cpu().numpy(), Lb, 7, max(16.0, math.sqrt(K))) == 0
        assert (xaq3[:M, :r] != xaq[:M, :r]).float().mean() <= 0.03
        assert float(xq[:M, K:].float().abs().max()) == 0.0 if Kp > K else True  # (the padded k of the image are zeros)
    # (3) run-to-run bit stability (fixed-order combine), fresh output buffers
    xq2, xaq2 = torch.zeros_like(xq), torch.zeros_like(xaq)
    run(xq2, xaq2)
    assert torch.equal(xq2[:M].view(torch.int16), xq[:M].view(torch.int16))
    assert torch.equal(xaq2[:M, :r].view(torch.int16), xaq[:M, :r].view(torch.int16))


def test_module_forward_at_a_prefill_size_captures_in_a_graph(ops):
    """The module's forward at a prefill size: parity, and the same bits when replayed from a hipGraph (the kernels keep no
    state between launches)."""
    import lqer_amd
    from bench import MXINT_Q, make_case

    M, K, N, r = 512, 1024, 512, 32
    x, W, A, B = make_case(M, K, N, r, seed=3)
    mod = lqer_amd.LinearFlexibleLqer(K, N, bias=False, q_config=MXINT_Q, l_config={"rank": r})
    mod.load_state_dict({"weight": W, "A": A, "B": B})
    mod = mod.to(DEV).half()
    xd = x.half().to(DEV)
    y0 = mod(xd)
    ref = O.lqer_linear_forward(x.half().float(), W.half().float(), None, A.half().float(), B.half().float(), MXINT_Q)
    err = float((y0.float().cpu() - ref).norm() / ref.norm())
    assert err <= 1e-3, err
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        mod(xd)
        torch.cuda.synchronize()
        with torch.cuda.graph(gr):
            y1 = mod(xd)
    gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(y1, y0)


def test_fused_rank128_quantizer_takes_strided_rows_and_unaligned_views(ops):
    """The 128-row fused quantizer reads its 16-bit source through the row stride it is given: a view into a wider tensor (rows
    16-byte aligned) equals the dense tensor; a view whose rows are NOT 16-byte aligned takes the two separate kernels - same bits."""
    import lqer_amd
    from bench import OPT_Q, make_case

    M, K, N, r = 640, 512, 256, 128
    x, W, A, B, b = make_case(M, K, N, r, seed=11, bias=True)
    mod = lqer_amd.LinearFlexibleLqer(K, N, bias=True, q_config=OPT_Q, l_config={"rank": r})
    mod.load_state_dict({"weight": W, "A": A, "B": B, "bias": b})
    mod = mod.to(DEV).half()
    xd = x.half().to(DEV)
    y = mod(xd)
    wide = torch.zeros(M, K + 64, dtype=torch.float16, device=DEV)
    wide[:, 8:8 + K] = xd
    assert torch.equal(mod(wide[:, 8:8 + K]), y)      # rows start 16 bytes into a 1152-byte pitch: aligned, strided
    wide2 = torch.zeros(M, K + 68, dtype=torch.float16, device=DEV)
    wide2[:, 3:3 + K] = xd
    assert torch.equal(mod(wide2[:, 3:3 + K]), y)     # 6 bytes in: not 16-byte aligned -> the separate kernels
    ref = O.lqer_linear_forward(x.half().float(), W.half().float(), b.half().float(), A.half().float(), B.half().float(), OPT_Q)
    assert float((y.float().cpu() - ref).norm() / ref.norm()) <= 1e-3
