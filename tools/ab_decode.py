#!/usr/bin/env python3
"""A/B timing of the decode forward (lqer_linear_forward, M <= 8: one launch) across several builds of the library in ONE
process, interleaved rounds, back-to-back calls on one stream (GPU events).
    python tools/ab_decode.py [--M 1 --K 4096 --N 4096 --r 32] lib_a.so lib_b.so ...
Operands are packed once with the default build (the images are the same across builds); results are compared between builds."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lqer_amd  # noqa: E402
from bench import MXINT_Q, make_case  # noqa: E402
from lqer_amd import _lib, ops  # noqa: E402
from tools.ab_gemm import load  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--M", type=int, default=1)
    ap.add_argument("--K", type=int, default=4096)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--r", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    M, K, N, r = a.M, a.K, a.N, a.r
    x, W, A, B = make_case(8, K, N, r, seed=0)
    mod = lqer_amd.LinearFlexibleLqer(K, N, bias=False, q_config=MXINT_Q, l_config={"rank": r})
    mod.load_state_dict({"weight": W, "A": A, "B": B})
    mod = mod.to(dev).half()
    xd = x[:M].half().to(dev)
    mod(xd)
    desc, p = mod._desc(), mod._packed
    ws = ops.workspace(dev, ops.linear_sizes(desc, M).workspace)
    st = torch.cuda.current_stream().cuda_stream
    libs = [(q, load(q)) for q in a.libs]
    ys = {q: torch.empty(M, N, dtype=torch.float16, device=dev) for q, _ in libs}

    def call(q, L):
        rc = L.lqer_linear_forward(C.byref(desc), xd.data_ptr(), _lib.F16, M, K, p["w"].data_ptr(), p["a_t"].data_ptr(), p["b_t"].data_ptr(),
                                   p["a_limbs"], p["b_limbs"], None, ys[q].data_ptr(), N, ws.data_ptr(), ws.numel(), st)
        assert rc == 0, L.lqer_last_error()

    for q, L in libs:
        for _ in range(50):
            call(q, L)
    torch.cuda.synchronize()
    ref = ys[a.libs[0]]
    for q, _ in libs[1:]:
        print(f"{os.path.basename(q)}: {'same bits as' if torch.equal(ys[q], ref) else 'DIFFERS from'} {os.path.basename(a.libs[0])}")
    times = {q: [] for q, _ in libs}
    for _ in range(a.rounds):
        for q, L in libs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                call(q, L)
            e1.record()
            torch.cuda.synchronize()
            times[q].append(e0.elapsed_time(e1) / a.iters * 1e3)
    # host cost of one call (the loop above is GPU-bound only while this stays under the kernel's duration): wall time of a burst
    # of calls that are all queued before the first kernel can have finished much
    import time

    host = {}
    for q, L in libs:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(64):
            call(q, L)
        host[q] = (time.perf_counter() - t0) / 64 * 1e6
        torch.cuda.synchronize()
    for q, _ in libs:
        t = sorted(times[q])
        print(f"{q:40s} M={M} K={K} N={N}: median {t[len(t) // 2]:7.2f} us  min {t[0]:7.2f} us per forward   host {host[q]:5.2f} us per call")


if __name__ == "__main__":
    main()
