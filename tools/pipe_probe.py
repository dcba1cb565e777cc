#!/usr/bin/env python3
"""Where does the GEMM's time go inside the step?  Times lqer_linear_gemm on the bench's own operands
(C2: 2048 x 4096 -> 4096, r 32) in four settings, all with HIP events on the launch stream:
  a) back to back, one event pair around N launches;
  b) back to back, one event pair per launch;
  c) the bench step (quantize_act_xa, then gemm), event pair around each gemm;
  d) like c with a 33 MB device memset in place of the quantizer (same idle gap, no xq rewrite).
usage (GPU box): python tools/pipe_probe.py [lib.so]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from lqer_amd import _lib, ops  # noqa: E402
from lqer_amd.linear import LinearFlexibleLqer  # noqa: E402


def main():
    if len(sys.argv) > 1:
        from tools.ab_gemm import load

        L = load(sys.argv[1])
    else:
        L = _lib.lib()
    dev = torch.device("cuda:0")
    M, K, N, r = 2048, 4096, 4096, 32
    qc = bench.MXINT_Q
    x, W, A, B = bench.make_case(M, K, N, r, seed=0)[:4]
    mod = LinearFlexibleLqer(K, N, bias=False, q_config=qc, l_config={"rank": r})
    mod.load_state_dict({"weight": W, "A": A, "B": B})
    mod = mod.to(dev).half()
    xd = x.half().to(dev)
    y = mod(xd).reshape(M, N)  # packs
    p, desc = mod._packed, mod._desc()
    sz = ops.linear_sizes(desc, M)
    ws = ops.workspace(dev, sz.workspace)
    Kp, Mp, rp = L.lqer_padded_k(K), L.lqer_padded_m(M), L.lqer_padded_r(r)
    xq = ws.data_ptr()
    xaq = xq + ((Mp * Kp * 2 + 255) // 256) * 256
    xscr = xaq + ((Mp * rp * 2 + 255) // 256) * 256
    nscr = L.lqer_lowrank_xa_scratch_bytes(C.byref(desc), M)
    st = torch.cuda.current_stream().cuda_stream
    junk = torch.empty(33 << 20, dtype=torch.uint8, device=dev)

    def quant():
        _lib.check(L.lqer_quantize_act_xa(C.byref(desc), xd.data_ptr(), _lib.F16, M, K, p["a_t"].data_ptr(), p["a_limbs"], xq, xaq,
                                          xscr, nscr, st), "q")

    def gemm():
        _lib.check(L.lqer_linear_gemm(C.byref(desc), xq, M, p["w"].data_ptr(), xaq, p["b_t"].data_ptr(), p["b_limbs"], None,
                                      y.data_ptr(), _lib.F16, N, xscr, 0, st), "g")

    def ev():
        return torch.cuda.Event(enable_timing=True)

    quant()
    for _ in range(20):
        gemm()
    torch.cuda.synchronize()
    n = 200
    res = {}
    for rnd in range(3):
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(n):
            gemm()
        e1.record()
        torch.cuda.synchronize()
        res.setdefault("a) back-to-back, one pair", []).append(e0.elapsed_time(e1) / n * 1e3)
        prs = []
        for _ in range(n):
            a, b = ev(), ev()
            a.record()
            gemm()
            b.record()
            prs.append((a, b))
        torch.cuda.synchronize()
        res.setdefault("b) back-to-back, pair per launch", []).append(sum(a.elapsed_time(b) for a, b in prs) / n * 1e3)
        for name, pre in (("c) after quantize_act_xa", quant), ("d) after 33 MB memset", lambda: junk.zero_())):
            prs = []
            e0, e1 = ev(), ev()
            e0.record()
            for _ in range(n):
                pre()
                a, b = ev(), ev()
                a.record()
                gemm()
                b.record()
                prs.append((a, b))
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(name, []).append(sum(a.elapsed_time(b) for a, b in prs) / n * 1e3)
            res.setdefault(name + " (whole step)", []).append(e0.elapsed_time(e1) / n * 1e3)
    for k, v in res.items():
        print(f"{k:45s} " + "  ".join(f"{t:7.2f}" for t in v) + "  us")


if __name__ == "__main__":
    main()
