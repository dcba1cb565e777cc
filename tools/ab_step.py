#!/usr/bin/env python3
"""A/B timing of the whole step (lqer_quantize_act_xa + lqer_linear_gemm: activation quantizer, side GEMM, reduce pass, fused
GEMM - what bench.py times) across several builds of the library in ONE process, interleaved rounds.
    python tools/ab_step.py [--M 2048 --K 4096 --N 4096 --r 32] lib_a.so lib_b.so ...
Operands: the bench's synthetic case packed by the module (tools/_operands.py); results are not checked here."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lqer_amd import _lib  # noqa: E402
from tools.ab_gemm import load  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--M", type=int, default=2048)
    ap.add_argument("--K", type=int, default=4096)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--r", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--also-128", action="store_true", help="time every build a second time with the 128-row tiles pinned (LQER_TUNE_TILE_ROWS_128)")
    ap.add_argument("--also-64", action="store_true", help="time every build once more with 64-row tiles forced (two workgroups per CU at large M)")
    ap.add_argument("--spin0", action="store_true", help="also time every build with the in-launch hand-offs' poll bound at 0 (every workgroup sums its own rows)")
    ap.add_argument("--also-in-gemm", action="store_true", help="time every build once more with the GEMM summing the partial tiles of x A itself (LQER_TUNE_XA_REDUCE_IN_GEMM: two launches)")
    ap.add_argument("--gap", type=int, default=0, help="tiny unrelated kernels launched between the quantizer and the GEMM (boundary-effect probe)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    M, K, N, r = a.M, a.K, a.N, a.r
    from tools._operands import real_operands

    op = real_operands(M, K, N, r)
    desc, x, xq, wp, xaq, at, bt, y = op["desc"], op["x"], op["xq"], op["w"], op["xaq"], op["a_t"], op["b_t"], op["y"]
    libs = [(p, load(p)) for p in a.libs]
    base = list(libs)
    if a.also_128:
        libs += [(p + " [128-row tiles]", L) for p, L in base]
    if a.also_64:
        libs += [(p + " [64-row tiles]", L) for p, L in base]
    if a.spin0:
        libs += [(p + " [spin 0]", L) for p, L in base]
    if a.also_in_gemm:
        libs += [(p + " [reduce in GEMM]", L) for p, L in base]
    st = torch.cuda.current_stream().cuda_stream
    scr, nscr = op["scr"], op["nscr"]

    dummy = torch.zeros(64, device=dev)

    def run(L, pin=0):
        # per-call knobs ride in the descriptor (lqer_linear_desc_t.tuning, ABI 9; older builds read a prefix of the struct)
        desc.tuning = {128: _lib.TUNE_TILE_ROWS_128, 64: _lib.TUNE_TILE_ROWS_64, -1: _lib.TUNE_DECODE_NO_POLL,
                       -2: _lib.TUNE_XA_REDUCE_IN_GEMM}.get(pin, 0)
        # (a build whose GEMM sums the partial tiles of x A itself - lqer_decode_partials / lqer_tile_partials - gets no xaq: two launches)
        part = L.lqer_decode_partials(C.byref(desc), M) or (hasattr(L, "lqer_tile_partials") and L.lqer_tile_partials(C.byref(desc), M, _lib.F16))
        xa = None if part else xaq.data_ptr()
        rc = L.lqer_quantize_act_xa(C.byref(desc), x.data_ptr(), _lib.F16, M, K, at.data_ptr(), 1, xq.data_ptr(), xa,
                                    scr.data_ptr(), nscr, st)
        assert rc == 0, L.lqer_last_error()
        for _ in range(a.gap):
            dummy.add_(1.0)
        rc = L.lqer_linear_gemm(C.byref(desc), xq.data_ptr(), M, wp.data_ptr(), xa, bt.data_ptr(), 1, None, y.data_ptr(),
                                _lib.F16, N, scr.data_ptr(), nscr, st)
        assert rc == 0, L.lqer_last_error()

    pin_of = lambda p: 128 if p.endswith("[128-row tiles]") else (64 if p.endswith("[64-row tiles]") else (-1 if p.endswith("[spin 0]") else (-2 if p.endswith("[reduce in GEMM]") else 0)))
    times = {p: [] for p, _ in libs}
    for p, L in libs:
        for _ in range(10):
            run(L, pin_of(p))
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for p, L in libs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                run(L, pin_of(p))
            e1.record()
            torch.cuda.synchronize()
            times[p].append(e0.elapsed_time(e1) / a.iters * 1e3)
    fl = 2.0 * M * K * N + 2.0 * M * K * r + 2.0 * M * r * N
    for p, _ in libs:
        t = sorted(times[p])
        med, mn = t[len(t) // 2], t[0]
        print(f"{os.path.basename(p):44s} median {med:8.2f} us  min {mn:8.2f} us per step  {fl / med / 1e6:8.1f} TFLOP/s-equiv (median)")


if __name__ == "__main__":
    main()
