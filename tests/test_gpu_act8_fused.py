"""The int8 route's activation side in ONE launch (lqer_amd/csrc/act8_fused.hip, round 6): per-token x_quantizer + x_q A +
A_out_quantizer (reference quantized_layers/linear.py:154-156 with block_size [1, -1] activations and pass-through fp16 A,
experiments/configs/template/llama-7b-int.toml:61-93) against the three launches it replaces (LQER_TUNE_ACT8_SPLIT):
the int8 image and the row scales bit for bit, x A re-quantized inside the summation-order envelope of tests/_envelope.py
(the exact sum is computed here from the image itself), run-to-run bit-stable, ragged token counts / K / ranks, both 16-bit dtypes,
and the whole forward against the CPU oracle.   Run on the GPU box:  python -m pytest tests -m gpu -x -q"""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from _envelope import envelope_check  # noqa: E402
from oracle import lqer_oracle as O  # the checker  # noqa: E402

DEV = "cuda:0"


@pytest.fixture(scope="module")
def lq():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import lqer_amd

    return lqer_amd


def _module(lq, K, N, r, dtype, seed, M, aout=None):
    from bench import INT_Q, make_case

    qc = dict(INT_Q)
    if aout is not None:
        qc["A_out_quantizer"] = aout
    x, W, A, B = make_case(M, K, N, r, seed=seed, quantize_ab=False)
    if dtype == torch.bfloat16:
        # multiples of 2^-12 below 2^-4: exact in bf16 AND in fp16, so that a bf16 module still gets the single fp16 image of A^T
        # (lqer_f16_prepare refuses an A that fp16 cannot hold; the module then keeps two bf16 limbs and the three launches)
        A, B = (A * 4096).round().clamp(-200, 200) / 4096, (B * 4096).round().clamp(-200, 200) / 4096
    mod = lq.LinearFlexibleLqer(K, N, bias=False, q_config=qc, l_config={"rank": r})
    mod.load_state_dict({"weight": W, "A": A, "B": B})
    mod = mod.to(DEV).to(dtype)
    return mod, x, W, A, B, qc


def _act_side(mod, xd, tuning):
    """lqer_quantize_act_xa through the C ABI with the module's images -> (int8 image [M, Kp8], row scales [M], xAq [M, rp] fp32)."""
    from lqer_amd import _lib, ops

    L = _lib.lib()
    M, K = xd.shape
    desc = mod._desc()
    desc.tuning = tuning
    p = mod._packed
    assert mod._x_i8 and "a_t_f16" in p
    Kp, Mp, rp = L.lqer_padded_k(K), L.lqer_padded_m(M), L.lqer_padded_r(mod.rank)
    Kp8 = -(-K // 128) * 128
    ws = torch.full((ops.linear_sizes(desc, M).workspace,), 0x5A, dtype=torch.uint8, device=DEV)
    xq = ws.data_ptr()
    xaq = xq + ((Mp * Kp * 2 + 255) // 256) * 256
    scr = xaq + ((Mp * rp * 2 + 255) // 256) * 256
    nscr = L.lqer_lowrank_xa_scratch_bytes(C.byref(desc), M)
    _lib.check(L.lqer_quantize_act_xa(C.byref(desc), xd.data_ptr(), ops.dtype_code(xd), M, K,
                                      p["a_t_f16"].data_ptr(), -1, xq, xaq, scr, nscr, torch.cuda.current_stream().cuda_stream), "quantize_act_xa")
    torch.cuda.synchronize()
    img_bytes = (Mp * Kp8 + 255) // 256 * 256
    img = ws[: Mp * Kp8].view(torch.int8).view(Mp, Kp8)[:M].clone()
    sc = ws[img_bytes: img_bytes + Mp * 4].view(torch.float32)[:M].clone()
    off = xaq - xq
    xa = ws[off: off + Mp * rp * 2].view(torch.bfloat16).view(Mp, rp)[:M].float().clone()
    return img, sc, xa


CASES = [
    # M, K, N, r, dtype
    (2048, 4096, 512, 32, torch.float16),    # the Llama-7B INT shape of the activation side (llama-7b-int.toml, rank 32)
    (300, 1088, 256, 32, torch.float16),     # ragged token count (300 = 37 workgroups + 4 rows), K not a multiple of 128
    (1000, 11008, 256, 32, torch.float16),   # the down projection's K: 22 chunks per lane
    (2048, 5120, 256, 64, torch.float16),    # Llama-13B INT: rank 64, four rank tiles
    (130, 512, 256, 16, torch.float16),      # one rank tile
    (777, 4096, 256, 32, torch.bfloat16),
    (64, 5120, 256, 64, torch.bfloat16),
]


@pytest.mark.parametrize("M,K,N,r,dtype", CASES)
def test_one_launch_against_three(lq, M, K, N, r, dtype):
    from lqer_amd import _lib

    mod, x, W, A, B, qc = _module(lq, K, N, r, dtype, seed=M + K + r, M=M)
    xd = x.to(dtype).to(DEV)
    xd[5] = 0  # an all-zero row (scale 1, zero image, zero x A)
    mod(xd[:128])  # builds the images
    img3, sc3, xa3 = _act_side(mod, xd, _lib.TUNE_ACT8_SPLIT)
    img1, sc1, xa1 = _act_side(mod, xd, _lib.TUNE_ACT8_FUSED)  # the one launch, at every token count
    # the default takes it from 1024 to 4096 tokens (below, a grid of M / 8 workgroups leaves most CUs idle; above, every workgroup's
    # pass over A^T costs more than the launches): the default's bits are those of the route it names
    imgd, scd, xad = _act_side(mod, xd, 0)
    want = (img1, sc1, xa1) if 1024 <= M <= 4096 else (img3, sc3, xa3)
    assert torch.equal(imgd, want[0]) and torch.equal(scd, want[1]) and torch.equal(xad, want[2])
    # (1) image and scales: bit for bit
    assert torch.equal(img1, img3)
    assert torch.equal(sc1, sc3)
    Kp8 = img1.shape[1]
    if Kp8 > K:
        assert int(img1[:, K:].abs().max()) == 0  # the padded k of the image are zeros (the GEMM multiplies them)
    # ... and the oracle's quantizer on the same tensor
    xf = xd.float().cpu()
    ref = O.get_quantizer(qc["x_quantizer"])(xf)
    deq = (img1[:, :K].float() * sc1[:, None]).cpu()
    assert torch.equal(deq, torch.where(xf.abs() <= 1e-8, torch.zeros_like(ref), ref))
    # (2) xAq of both routes inside the envelope of the exact sum of exact products
    Ah = A.half().double().numpy()  # (what the module's fp16 image of A^T holds: fp16 modules round A to it, the bf16 cases are exact)
    s64 = (img1[:, :K].double().cpu().numpy() @ Ah) * sc1.double().cpu().numpy()[:, None]
    for xa in (xa1, xa3):
        got = xa[:, :r].cpu().numpy()
        assert envelope_check(s64, got, r, 7, max(16.0, math.sqrt(K))) == 0
    assert float((xa1 != xa3).float().mean()) <= 0.03
    if xa1.shape[1] > r:
        assert float(xa1[:, r:].abs().max()) == 0.0  # padded rank columns
    # (3) run-to-run bit stability (fixed summation order)
    img2, sc2, xa2 = _act_side(mod, xd, _lib.TUNE_ACT8_FUSED)
    assert torch.equal(img2, img1) and torch.equal(sc2, sc1) and torch.equal(xa2, xa1)


def test_forward_with_the_one_launch_activation_side_vs_oracle(lq):
    """The module's forward at the Llama-7B INT shape takes the one-launch activation kernel by default: against the CPU oracle, and
    against the same forward with the three launches pinned."""
    from lqer_amd import _lib

    M, K, N, r = 2048, 4096, 4096, 32
    mod, x, W, A, B, qc = _module(lq, K, N, r, torch.float16, seed=3, M=M)
    xd = x.half().to(DEV)
    y1 = mod(xd).float().cpu()
    mod.tuning = _lib.TUNE_ACT8_SPLIT
    mod._fw_cache.clear()
    y3 = mod(xd).float().cpu()
    h = lambda t: t.half().float()
    ref = O.lqer_linear_forward(h(x), h(W), None, h(A), h(B), qc)
    for y in (y1, y3):
        assert float((y - ref).norm() / ref.norm()) <= 1e-3
    assert float((y1 - y3).norm() / ref.norm()) <= 3e-4


def test_a_out_in_blocks_of_16(lq):
    """A_out blocks shorter than the rank (block_fp [1, 16]) ride through the same epilogue."""
    from lqer_amd import _lib

    M, K, N, r = 520, 1024, 256, 32
    aout = dict(name="block_fp", width=8, exponent_width=8, exponent_bias=None, block_size=[1, 16], skip_first_dim=True)
    mod, x, W, A, B, qc = _module(lq, K, N, r, torch.float16, seed=11, M=M, aout=aout)
    xd = x.half().to(DEV)
    mod(xd[:128])
    img1, sc1, xa1 = _act_side(mod, xd, _lib.TUNE_ACT8_FUSED)
    img3, sc3, xa3 = _act_side(mod, xd, _lib.TUNE_ACT8_SPLIT)
    assert torch.equal(img1, img3) and torch.equal(sc1, sc3)
    s64 = (img1[:, :K].double().cpu().numpy() @ A.half().double().numpy()) * sc1.double().cpu().numpy()[:, None]
    for xa in (xa1, xa3):
        assert envelope_check(s64, xa[:, :r].cpu().numpy(), 16, 7, max(16.0, math.sqrt(K))) == 0


def test_ties_at_every_exponent_through_the_one_launch_kernel(lq):
    """The packed-half quantizer of fp16 rows (common.h row8_chunk_h16) inside the one-launch kernel: rows whose maximum is 2^n for n from
    -12 to 15, filled with multiples of max / 256 (ties k + 1/2, the clamp), image and scales against the oracle bit for bit."""
    M, K, N, r = 56, 1024, 256, 32
    mod, x, W, A, B, qc = _module(lq, K, N, r, torch.float16, seed=5, M=M)
    g = torch.Generator().manual_seed(7)
    x = torch.zeros(M, K)
    for i in range(M):
        amax = 2.0 ** (i // 2 - 12)
        j = torch.randint(0, 257, (K,), generator=g).float()
        x[i] = torch.where(torch.rand(K, generator=g) < 0.5, -1.0, 1.0) * amax * j / 256.0
        x[i, i % K] = amax
    xd = x.half().to(DEV)
    from lqer_amd import _lib

    mod(xd)
    img, sc, xa = _act_side(mod, xd, _lib.TUNE_ACT8_FUSED)
    ref = O.get_quantizer(qc["x_quantizer"])(x)
    assert torch.equal((img[:, :K].float() * sc[:, None]).cpu(), torch.where(x.abs() <= 1e-8, torch.zeros_like(ref), ref))
    img3, sc3, _ = _act_side(mod, xd, _lib.TUNE_ACT8_SPLIT)
    assert torch.equal(img, img3) and torch.equal(sc, sc3)


def test_gemm_prepass_zero_fill_handed_to_the_activation_kernel(lq):
    """ABI 13: a GEMM whose grid is several rounds of tiles folds the B_out row maxima in a pre-pass with atomicMax on cells it zero-fills
    first (a memset launch); lqer_quantize_act_xa_prep lets the one-launch activation kernel in front write those zeros and
    lqer_linear_gemm_prepared skips the memset.  Same bits as the plain pair - on a scratch full of garbage -, 0 ready bytes where the
    activation side takes the three launches or the GEMM needs no cells, and lqer_linear_forward (which hands over internally) agrees."""
    from lqer_amd import _lib, ops

    L = _lib.lib()
    M, K, N, r = 2048, 512, 11008, 32  # 43 column tiles x 16 row tiles of 128: 2.7 rounds - the pre-pass on atomic cells
    mod, x, W, A, B, qc = _module(lq, K, N, r, torch.float16, seed=11, M=M)
    xd = x.half().to(DEV)
    y_fwd = mod(xd).clone()  # lqer_linear_forward
    p = mod._packed
    st = torch.cuda.current_stream().cuda_stream

    def run(tuning, prep, fill):
        desc = mod._desc()
        desc.tuning = tuning
        Kp, Mp, rp = L.lqer_padded_k(K), L.lqer_padded_m(M), L.lqer_padded_r(r)
        ws = torch.full((ops.linear_sizes(desc, M).workspace,), fill, dtype=torch.uint8, device=DEV)
        xq = ws.data_ptr()
        xaq = xq + ((Mp * Kp * 2 + 255) // 256) * 256
        scr = xaq + ((Mp * rp * 2 + 255) // 256) * 256
        nscr, gscr = L.lqer_lowrank_xa_scratch_bytes(C.byref(desc), M), L.lqer_linear_gemm_scratch_bytes(C.byref(desc), M)
        y = torch.empty(M, N, dtype=torch.float16, device=DEV)
        ready = C.c_size_t(12345)
        qa = (C.byref(desc), xd.data_ptr(), _lib.F16, M, K, p["a_t_f16"].data_ptr(), -1, xq, xaq, scr, nscr)
        ga = (C.byref(desc), xq, M, p["w"].data_ptr(), xaq, p["b_t"].data_ptr(), p["b_limbs"], None, y.data_ptr(), _lib.F16, N, scr, gscr)
        if prep:
            _lib.check(L.lqer_quantize_act_xa_prep(*qa, scr, C.byref(ready), st), "quantize_act_xa_prep")
            torch.cuda.synchronize()
            head = ws[scr - xq: scr - xq + ready.value].clone()
            _lib.check(L.lqer_linear_gemm_prepared(*ga, ready.value, st), "linear_gemm_prepared")
        else:
            head = None
            _lib.check(L.lqer_quantize_act_xa(*qa, st), "quantize_act_xa")
            _lib.check(L.lqer_linear_gemm(*ga, st), "linear_gemm")
        torch.cuda.synchronize()
        return y, ready.value, head

    NM = _lib.TUNE_AMAX_NO_MRX  # (the pre-pass launch pinned: by default this shape computes the pre-pass inside the GEMM - no cells at all)
    y0, _, _ = run(NM, False, 0x5A)
    y1, ready, head = run(NM, True, 0xFF)
    assert ready == L.lqer_padded_m(M) * 4 and int(head.max()) == 0  # one fp32 cell per row (one B_out block per row), zeroed
    assert torch.equal(y0.view(torch.int16), y1.view(torch.int16)) and torch.equal(y0.view(torch.int16), y_fwd.view(torch.int16))
    y2, ready2, _ = run(NM | _lib.TUNE_ACT8_SPLIT, True, 0xFF)  # three launches: nobody prepared anything, the GEMM fills its cells itself
    assert ready2 == 0 and torch.equal(y0.view(torch.int16), y2.view(torch.int16))
    ref = O.lqer_linear_forward(x.half().float(), W.half().float(), None, A.half().float(), B.half().float(), qc)
    assert float((y1.float().cpu() - ref).norm() / ref.norm()) <= 1e-3
    # a launch that needs no cells (one round of tiles: the in-GEMM exchange): 0 ready bytes
    mod4, x4, *_ = _module(lq, K, 4096, r, torch.float16, seed=12, M=M)
    mod4(x4.half().to(DEV)[:128])
    desc4 = mod4._desc()
    ws4 = torch.empty(ops.linear_sizes(desc4, M).workspace, dtype=torch.uint8, device=DEV)
    Kp, Mp, rp = L.lqer_padded_k(K), L.lqer_padded_m(M), L.lqer_padded_r(r)
    xq = ws4.data_ptr()
    xaq = xq + ((Mp * Kp * 2 + 255) // 256) * 256
    scr = xaq + ((Mp * rp * 2 + 255) // 256) * 256
    rdy = C.c_size_t(7)
    _lib.check(L.lqer_quantize_act_xa_prep(C.byref(desc4), xd.data_ptr(), _lib.F16, M, K, mod4._packed["a_t_f16"].data_ptr(), -1, xq, xaq, scr,
                                           L.lqer_lowrank_xa_scratch_bytes(C.byref(desc4), M), scr, C.byref(rdy), st), "quantize_act_xa_prep")
    torch.cuda.synchronize()
    assert rdy.value == 0
    y3, ready3, _ = run(0, True, 0xFF)  # ... nor does the default route of the 11008-column launch (the in-GEMM pre-pass): same bits
    assert ready3 == 0 and torch.equal(y0.view(torch.int16), y3.view(torch.int16))
