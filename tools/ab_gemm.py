#!/usr/bin/env python3
"""A/B timing of lqer_linear_gemm across several builds of the library in ONE process, interleaved
rounds (cdna_hip_programming.md rule 24).  Usage on the GPU box:
    python tools/ab_gemm.py [--M 2048 --K 4096 --N 4096 --r 32] lib_a.so lib_b.so ...
Operands: the bench's synthetic case packed by the module (tools/_operands.py); results are not checked here."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lqer_amd import _lib  # noqa: E402


def load(path):
    L = C.CDLL(os.path.abspath(path))
    for name, (res, args) in _lib.SIGNATURES.items():
        if not hasattr(L, name):
            continue  # (an older build without that entry point / test hook)
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--M", type=int, default=2048)
    ap.add_argument("--K", type=int, default=4096)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--r", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--bout", type=int, default=1, help="1: B_out MXINT8/16, 0: passthrough, 2: MXINT8, one block per row (pre-pass)")
    ap.add_argument("--xcd-bm", type=int, nargs="*", default=[], help="also time every build with XCD-local tile blocks of this many "
                    "token tiles (LQER_TUNE_XCD_BLOCK; 128-row kernel)")
    ap.add_argument("--no-persist", action="store_true", help="also time every build with the persistent tile loop off (lqer_debug_set_gemm_persistent)")
    ap.add_argument("--blimbs", type=int, default=1, help="bf16 limbs of B (1: MXINT8 values, 2: fp16 values)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    M, K, N, r = a.M, a.K, a.N, a.r
    from tools._operands import real_operands

    op = real_operands(M, K, N, r, bout=a.bout, blimbs=a.blimbs)
    desc, xq, wp, xaq, bt, y = op["desc"], op["xq"], op["w"], op["xaq"], op["b_t"], op["y"]
    b_limbs = op["b_limbs"]
    libs = [(p, load(p)) for p in a.libs]
    base = list(libs)
    if a.no_persist:
        libs += [(f"{p} [one workgroup per tile]", L) for p, L in base if hasattr(L, "lqer_debug_set_gemm_persistent")]
    for bm in a.xcd_bm:
        libs += [(f"{p} [xcd block {bm}]", L) for p, L in base]
    bm_of = lambda p: int(p.rsplit("[xcd block ", 1)[1][:-1]) if p.endswith("]") and "[xcd block " in p else 0
    st = torch.cuda.current_stream().cuda_stream

    scr, nscr = op["scr"], op["nscr"]

    def run(L, bm=0, persist=1):
        desc.tuning = _lib.tune_xcd_block(bm)  # per call, in the descriptor (lqer_linear_desc_t.tuning)
        if hasattr(L, "lqer_debug_set_gemm_persistent"):
            L.lqer_debug_set_gemm_persistent(persist)
        rc = L.lqer_linear_gemm(C.byref(desc), xq.data_ptr(), M, wp.data_ptr(), xaq.data_ptr() if r else None,
                                bt.data_ptr() if r else None, b_limbs, None, y.data_ptr(), _lib.F16, N, scr.data_ptr(), nscr, st)
        assert rc == 0, L.lqer_last_error()

    times = {p: [] for p, _ in libs}
    for p, L in libs:
        for _ in range(5):
            run(L, bm_of(p), 0 if p.endswith('[one workgroup per tile]') else 1)
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for p, L in libs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                run(L, bm_of(p), 0 if p.endswith('[one workgroup per tile]') else 1)
            e1.record()
            torch.cuda.synchronize()
            times[p].append(e0.elapsed_time(e1) / a.iters * 1e3)
    fl = 2.0 * M * K * N + 2.0 * M * r * N
    for p, _ in libs:
        t = sorted(times[p])
        med, mn = t[len(t) // 2], t[0]
        print(f"{os.path.basename(p):46s} median {med:8.2f} us  min {mn:8.2f} us   {fl / med / 1e6:8.1f} TFLOP/s (median)")


if __name__ == "__main__":
    main()
