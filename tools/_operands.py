"""Real operands for the A/B tools: the bench's synthetic case (make_case: gaussian tokens with three x30 outlier channels,
W ~ 0.02 N(0,1), A / B on the MXINT grid or unquantized) packed by the module, so that the timed kernels see the bit patterns
(and the power draw) of the bench - random image BYTES give random block exponents, i.e. inf / NaN outputs."""
import ctypes as C
import os
import sys
from copy import deepcopy

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def real_operands(M, K, N, r, bout=1, blimbs=1, dev="cuda:0"):
    """dict: desc (LinearDesc), x (fp16 [M,K]), w / a_t / b_t (packed images), a_limbs, b_limbs, xq / xaq (images made by the
    installed library's quantizer), scr / nscr (side-GEMM scratch), y (fp16 [M,N])."""
    import lqer_amd
    from bench import MXINT_Q, _bfp, make_case
    from lqer_amd import _lib

    qc = deepcopy(MXINT_Q)
    if bout == 0:
        qc["B_out_quantizer"] = {"name": "passthrough"}
    elif bout == 2:
        qc["B_out_quantizer"] = _bfp(8, [1, -1], True)
    xc, W, A, B = make_case(M, K, N, max(r, 1), seed=0, quantize_ab=(blimbs == 1))
    if r > 0:
        mod = lqer_amd.LinearFlexibleLqer(K, N, bias=False, q_config=qc, l_config={"rank": r})
        mod.load_state_dict({"weight": W, "A": A, "B": B})
    else:
        mod = lqer_amd.LinearFlexible(K, N, bias=False, q_config=qc)
        mod.load_state_dict({"weight": W})
    mod = mod.to(dev).half()
    x = xc.half().to(dev)
    y = mod(x)
    L = _lib.lib()
    pk, desc = mod._packed, mod._desc()
    Kp, Mp, rp = L.lqer_padded_k(K), L.lqer_padded_m(M), L.lqer_padded_r(max(r, 1))
    xq = torch.empty(Mp, Kp, dtype=torch.bfloat16, device=dev)
    xaq = torch.empty(Mp, rp, dtype=torch.bfloat16, device=dev)
    nscr = max(L.lqer_lowrank_xa_scratch_bytes(C.byref(desc), M), L.lqer_linear_gemm_scratch_bytes(C.byref(desc), M), 16)
    scr = torch.empty(nscr, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    rc = L.lqer_quantize_act_xa(C.byref(desc), x.data_ptr(), _lib.F16, M, K, pk["a_t"].data_ptr() if r else None,
                                pk.get("a_limbs", 0), xq.data_ptr(), xaq.data_ptr() if r else None, scr.data_ptr(), nscr, st)
    assert rc == 0, L.lqer_last_error()
    torch.cuda.synchronize()
    return dict(mod=mod, desc=desc, x=x, w=pk["w"], a_t=pk.get("a_t"), b_t=pk.get("b_t"), a_limbs=pk.get("a_limbs", 0),
                b_limbs=pk.get("b_limbs", 0), xq=xq, xaq=xaq, scr=scr, nscr=nscr, y=y)
