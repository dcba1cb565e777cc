"""The block-16 MXINT activation side in ONE launch (lqer_amd/csrc/act16_fused.hip, round 6): x_quantizer in blocks of [1, 16] + x_q A +
A_out_quantizer (reference quantized_layers/linear.py:154-156 with the llama-7b.toml:78-105 formats) against the two launches it
replaces (k_quant_xa16 + k_xa_reduce4, pinned by LQER_TUNE_ACT16_SPLIT): the bf16 activation image bit for bit (and against the oracle's
quantizer), x A re-quantized inside the summation-order envelope of tests/_envelope.py, run-to-run bit-stable; ragged token counts, the
down projection's K, ranks 16 / 32 / 64, both 16-bit dtypes; and the whole forward against the CPU oracle.
Run on the GPU box:  python -m pytest tests -m gpu -x -q"""
import ctypes as C
import math

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


def _module(lq, K, N, r, dtype, seed, M):
    from bench import MXINT_Q, make_case

    x, W, A, B = make_case(M, K, N, r, seed=seed, quantize_ab=True)  # (A, B on the 8-bit MXINT grid: one bf16 limb each)
    mod = lq.LinearFlexibleLqer(K, N, bias=False, q_config=MXINT_Q, l_config={"rank": r})
    mod.load_state_dict({"weight": W, "A": A, "B": B})
    return mod.to(DEV).to(dtype), x, W, A, B, MXINT_Q


def _act_side(mod, xd, tuning):
    """lqer_quantize_act_xa through the C ABI with the module's a_limbs = -2 image -> (bf16 image [M, Kp], xAq [M, rp] fp32)."""
    from lqer_amd import _lib, ops

    L = _lib.lib()
    M, K = xd.shape
    desc = mod._desc()
    desc.tuning = tuning
    p = mod._packed
    assert "a_t_b16" in p and not mod._x_i8
    Kp, Mp, rp = L.lqer_padded_k(K), L.lqer_padded_m(M), L.lqer_padded_r(mod.rank)
    ws = torch.full((ops.linear_sizes(desc, M).workspace,), 0x5A, dtype=torch.uint8, device=DEV)
    xq = ws.data_ptr()
    xaq = xq + ((Mp * Kp * 2 + 255) // 256) * 256
    scr = xaq + ((Mp * rp * 2 + 255) // 256) * 256
    nscr = L.lqer_lowrank_xa_scratch_bytes(C.byref(desc), M)
    _lib.check(L.lqer_quantize_act_xa(C.byref(desc), xd.data_ptr(), ops.dtype_code(xd), M, K, p["a_t_b16"].data_ptr(), -2, xq, xaq, scr, nscr,
                                      torch.cuda.current_stream().cuda_stream), "quantize_act_xa")
    torch.cuda.synchronize()
    img = ws[: Mp * Kp * 2].view(torch.bfloat16).view(Mp, Kp)[:M].clone()
    off = xaq - xq
    xa = ws[off: off + Mp * rp * 2].view(torch.bfloat16).view(Mp, rp)[:M].float().clone()
    return img, xa


CASES = [
    # M, K, N, r, dtype
    (2048, 4096, 512, 32, torch.float16),    # BASELINE configs[1] / the Llama-7B attention projections
    (1100, 1088, 256, 32, torch.float16),    # ragged token count (137 workgroups + 4 rows), K = 2 slabs + 64 (a part-filled slab)
    (1032, 11008, 256, 32, torch.float16),   # the down projection's K: 21.5 slabs over 8 waves
    (1024, 4096, 256, 64, torch.float16),    # rank 64: four rank tiles, quarter-slab fragment sets
    (1024, 512, 256, 16, torch.float16),     # one rank tile, one slab (seven of the eight waves idle)
    (1500, 4096, 256, 32, torch.bfloat16),
]


@pytest.mark.parametrize("M,K,N,r,dtype", CASES)
def test_one_launch_against_two(lq, M, K, N, r, dtype):
    from lqer_amd import _lib

    mod, x, W, A, B, qc = _module(lq, K, N, r, dtype, seed=M + K + r, M=M)
    xd = x.to(dtype).to(DEV)
    xd[5] = 0          # an all-zero row
    xd[7, 32:64] = 0   # all-zero blocks inside a row
    mod(xd[:128])      # builds the images
    img2, xa2 = _act_side(mod, xd, _lib.TUNE_ACT16_SPLIT)
    img1, xa1 = _act_side(mod, xd, 0)
    imgf, xaf = _act_side(mod, xd, _lib.TUNE_ACT16_FUSED)
    assert torch.equal(img1.view(torch.int16), imgf.view(torch.int16)) and torch.equal(xa1, xaf)  # (M >= 1024: the default IS the one launch)
    # (1) the image: bit for bit the two-launch route's and the oracle's quantizer (|x| <= 1e-8 flushed in images)
    assert torch.equal(img1.view(torch.int16), img2.view(torch.int16))
    xf = xd.float().cpu()
    ref = O.get_quantizer(qc["x_quantizer"])(xf)
    assert torch.equal(img1[:, :K].float().cpu(), torch.where(xf.abs() <= 1e-8, torch.zeros_like(ref), ref))
    if img1.shape[1] > K:
        assert float(img1[:, K:].float().abs().max()) == 0.0  # the padded k of the image are zeros
    # (2) xAq of both routes inside the envelope of the exact sum of exact products (A_out in blocks of 16)
    s64 = img1[:, :K].double().cpu().numpy() @ A.double().numpy()
    for xa in (xa1, xa2):
        assert envelope_check(s64, xa[:, :r].cpu().numpy(), 16, 7, max(16.0, math.sqrt(K))) == 0
    assert float((xa1 != xa2).float().mean()) <= 0.03
    # (3) run-to-run bit stability
    img3, xa3 = _act_side(mod, xd, 0)
    assert torch.equal(img3.view(torch.int16), img1.view(torch.int16)) and torch.equal(xa3, xa1)


def test_small_token_counts_keep_the_two_launches(lq):
    """Below 1024 tokens the default stays on the split-K kernels (a grid of M / 8 workgroups would leave most CUs idle); the image
    passed with a_limbs = -2 is then read as the one-limb image it starts with - same bits as a_limbs = 1."""
    from lqer_amd import _lib

    M, K, N, r = 300, 1024, 256, 32
    mod, x, W, A, B, qc = _module(lq, K, N, r, torch.float16, seed=4, M=M)
    xd = x.half().to(DEV)
    mod(xd[:128])
    img_d, xa_d = _act_side(mod, xd, 0)
    img_s, xa_s = _act_side(mod, xd, _lib.TUNE_ACT16_SPLIT)
    assert torch.equal(img_d.view(torch.int16), img_s.view(torch.int16)) and torch.equal(xa_d, xa_s)
    img_f, xa_f = _act_side(mod, xd, _lib.TUNE_ACT16_FUSED)  # ... and the one launch, forced: same image, x A inside the envelope
    assert torch.equal(img_f.view(torch.int16), img_s.view(torch.int16))
    s64 = img_f[:, :K].double().cpu().numpy() @ A.double().numpy()
    assert envelope_check(s64, xa_f[:, :r].cpu().numpy(), 16, 7, max(16.0, math.sqrt(K))) == 0


def test_forward_with_the_one_launch_activation_side_vs_oracle(lq):
    """The module's forward at BASELINE configs[1] takes the one-launch activation kernel by default: against the CPU oracle, and
    against the same forward with the two launches pinned; decode sizes still take the one-launch DECODE kernel (a_limbs = 1)."""
    from lqer_amd import _lib

    M, K, N, r = 2048, 4096, 4096, 32
    mod, x, W, A, B, qc = _module(lq, K, N, r, torch.float16, seed=3, M=M)
    xd = x.half().to(DEV)
    y1 = mod(xd).float().cpu()
    assert mod._side_image(M, mod._desc(), _lib.F16)[1] == -2 and mod._side_image(4, mod._desc(), _lib.F16)[1] == 1
    mod.tuning = _lib.TUNE_ACT16_SPLIT
    mod._fw_cache.clear()
    y2 = mod(xd).float().cpu()
    h = lambda t: t.half().float()
    ref = O.lqer_linear_forward(h(x), h(W), None, h(A), h(B), qc)
    for y in (y1, y2):
        assert float((y - ref).norm() / ref.norm()) <= 1e-3
    assert float((y1 - y2).norm() / ref.norm()) <= 3e-4
    mod.tuning = 0
    mod._fw_cache.clear()
    y4 = mod(xd[:4]).float().cpu()
    assert float((y4 - ref[:4]).norm() / ref[:4].norm()) <= 1e-3
