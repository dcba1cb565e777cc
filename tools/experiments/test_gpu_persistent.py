"""The persistent tile loop of the 128-row bf16 kernel (k_lqer_gemm_p: one workgroup per CU walks its tiles on grids of more than
one round, the ring prefetch of a tile's last steps already fetching the next tile) against the one-workgroup-per-tile launch
of the same shapes (lqer_debug_set_gemm_persistent(0)): the arithmetic and its order are the same, so the outputs must be equal
bit for bit - and against the oracle on row slices.   Run on the GPU box:  python -m pytest tests -m gpu -x -q"""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import lqer_oracle as O  # the checker

DEV = "cuda:0"


@pytest.fixture(scope="module")
def lq():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import lqer_amd

    return lqer_amd


@pytest.mark.parametrize("M,K,N,r,dtype", [
    (2048, 4096, 11008, 32, torch.float16),   # Llama-7B gate / up: 688 tiles, 2.7 rounds
    (512, 1024, 16640, 32, torch.bfloat16),   # 260 tiles: four workgroups get a second tile
    (1024, 512, 8448, 16, torch.float16),     # two ring turns per tile (the shortest the loop takes), rank 16
    (2048, 512, 8192 + 256, 32, torch.float16),
])
def test_persistent_tile_loop_equals_one_workgroup_per_tile(lq, M, K, N, r, dtype):
    from bench import MXINT_Q, make_case
    from lqer_amd import _lib

    L = _lib.lib()
    x, W, A, B = make_case(M, K, N, r, seed=M + N)
    mod = lq.LinearFlexibleLqer(K, N, bias=False, q_config=MXINT_Q, l_config={"rank": r})
    mod.load_state_dict({"weight": W, "A": A, "B": B})
    mod = mod.to(DEV).to(dtype)
    xd = x.to(dtype).to(DEV)
    try:
        L.lqer_debug_set_gemm_persistent(0)
        y0 = mod(xd).clone()
        L.lqer_debug_set_gemm_persistent(1)
        y1 = mod(xd).clone()
        y2 = mod(xd).clone()
    finally:
        L.lqer_debug_set_gemm_persistent(1)
    assert torch.isfinite(y1).all()
    assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
    assert torch.equal(y1.view(torch.int16), y2.view(torch.int16))  # run to run
    idx = torch.tensor(list(range(0, 48)) + list(range(M // 2 - 24, M // 2 + 24)) + list(range(M - 48, M)))
    xs = xd[idx.to(DEV)].float().cpu()
    ref = O.lqer_linear_forward(xs, W.to(dtype).float(), None, A.to(dtype).float(), B.to(dtype).float(), MXINT_Q)
    got = y1[idx.to(DEV)].float().cpu()
    tol = 1e-3 if dtype == torch.float16 else 4e-3  # (bf16 outputs: 8 significand bits)
    assert float((got - ref).norm() / ref.norm()) <= tol
