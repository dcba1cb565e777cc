#!/usr/bin/env python3
"""Throughput of INDEPENDENT forwards (the bench's C2 operands by default) issued round robin on S HIP streams (own activation image, x A image, scratch and output
per stream) against the same forwards on one stream: how much of the quantizer (HBM-bound) and of the GEMM's prologue / store
phases a second queue hides under the other forward's main loop.
    python tools/two_stream_probe.py [--M 2048 --K 4096 --N 4096 --r 32] [--streams 2]
Results are compared between the streams and with a one-stream run (bit-identical)."""
import argparse
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lqer_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=2048)
    ap.add_argument("--K", type=int, default=4096)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--r", type=int, default=32)
    ap.add_argument("--streams", type=int, default=2)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--carve", action="store_true", help="one buffer per stream, carved into image / x A / scratch the way bench.py does")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    L = _lib.lib()
    M, K, N, r = a.M, a.K, a.N, a.r
    Kp, Mp, rp = (K + 63) // 64 * 64, (M + 255) // 256 * 256, (r + 15) // 16 * 16
    import lqer_amd
    from bench import MXINT_Q, make_case

    xc, W, A, B = make_case(M, K, N, r, seed=0, quantize_ab=True)
    mod = lqer_amd.LinearFlexibleLqer(K, N, bias=False, q_config=MXINT_Q, l_config={"rank": r})
    mod.load_state_dict({"weight": W, "A": A, "B": B})
    mod = mod.to(dev).half()
    x = xc.half().to(dev)
    mod(x)  # packs the operands
    pk = mod._packed
    wp, at, bt = pk["w"], pk["a_t"], pk["b_t"]
    desc = mod._desc()
    nscr = L.lqer_lowrank_xa_scratch_bytes(C.byref(desc), M)
    S = a.streams
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    bufs = [dict(xq=torch.empty(Mp, Kp, dtype=torch.bfloat16, device=dev), xaq=torch.empty(Mp, rp, dtype=torch.bfloat16, device=dev),
                 scr=torch.empty(max(nscr, 16), dtype=torch.uint8, device=dev), y=torch.empty(M, N, dtype=torch.float16, device=dev))
            for _ in range(S)]
    if a.carve:
        for b in bufs:
            n1, n2 = Mp * Kp * 2, (Mp * rp * 2 + 255) // 256 * 256
            big = torch.empty(n1 + n2 + max(nscr, 16), dtype=torch.uint8, device=dev)
            b["big"], b["xq"], b["xaq"], b["scr"] = big, big[:n1], big[n1:n1 + n2], big[n1 + n2:]

    def fwd(i):
        b, st = bufs[i], streams[i].cuda_stream
        rc = L.lqer_quantize_act_xa(C.byref(desc), x.data_ptr(), _lib.F16, M, K, at.data_ptr(), 1, b["xq"].data_ptr(),
                                    b["xaq"].data_ptr(), b["scr"].data_ptr(), nscr, st)
        assert rc == 0, L.lqer_last_error()
        rc = L.lqer_linear_gemm(C.byref(desc), b["xq"].data_ptr(), M, wp.data_ptr(), b["xaq"].data_ptr(), bt.data_ptr(), 1, None,
                                b["y"].data_ptr(), _lib.F16, N, b["scr"].data_ptr(), nscr, st)
        assert rc == 0, L.lqer_last_error()

    def region(ns, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n):
            fwd(k % ns)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    # sequential reference: every stream alone, synchronised
    ref = {}
    for i in range(S):
        fwd(i)
        torch.cuda.synchronize()
        ref[i] = {k: bufs[i][k].clone() for k in ("xq", "xaq", "y")}
    for i in range(1, S):
        for k in ("xq", "xaq", "y"):
            assert torch.equal(ref[0][k][:M].view(torch.int16), ref[i][k][:M].view(torch.int16)), ("sequential", i, k)
    print("non-finite outputs:", int((~torch.isfinite(ref[0]["y"])).sum()), flush=True)
    for _ in range(3):
        region(S, 50)
    bad = 0
    for i in range(S):
        for k in ("xq", "xaq", "y"):
            d = (bufs[i][k][:M].view(torch.int16) != ref[0][k][:M].view(torch.int16))
            if d.any():
                rows = d.any(dim=1).nonzero().flatten()
                cols = d.any(dim=0).nonzero().flatten()
                print(f"OVERLAPPED stream {i} {k}: {int(d.sum())} elements differ; rows {rows[:8].tolist()}..{int(rows[-1])} ({len(rows)}), "
                      f"cols {cols[:8].tolist()}..{int(cols[-1])} ({len(cols)})", flush=True)
                bad += 1
    if bad:
        sys.exit("results differ when forwards overlap on several streams")
    flops = 2.0 * M * N * (K + r) + 2.0 * M * K * r
    for rd in range(a.rounds):
        one, many = region(1, a.iters), region(S, a.iters)
        print(f"M={M} K={K} N={N} r={r}: one stream {one:.2f} us / forward ({flops / one / 1e6:.0f} TFLOP/s-equiv), "
              f"{S} streams {many:.2f} us ({flops / many / 1e6:.0f}) -> x{one / many:.3f}", flush=True)


if __name__ == "__main__":
    main()
