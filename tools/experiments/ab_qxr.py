#!/usr/bin/env python3
"""The int8 route's activation side (lqer_quantize_act_xa with an LQER_Q_MXINT_I8 descriptor): one launch (quant_rows_xa.hip) against
the three launches it replaces (tuning LQER_TUNE_I8_SPLIT_SIDE), interleaved rounds, HIP events.
usage: python tools/ab_qxr.py [--M 2048 --K 4096 --r 32] [--lib other.so]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--lib" in sys.argv:
    from lqer_amd import _lib as _l

    _l.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
import lqer_amd  # noqa: E402
from bench import INT_Q, make_case  # noqa: E402
from lqer_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=2048)
ap.add_argument("--K", type=int, default=4096)
ap.add_argument("--r", type=int, default=32)
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--lib", default=None)
a = ap.parse_args()
dev = torch.device("cuda:0")
M, K, r, N = a.M, a.K, a.r, 256
x, W, A, B = make_case(M, K, N, r, seed=0, quantize_ab=False)
mod = lqer_amd.LinearFlexibleLqer(K, N, bias=False, q_config=INT_Q, l_config={"rank": r})
mod.load_state_dict({"weight": W, "A": A, "B": B})
mod = mod.to(dev).half()
xd = x.half().to(dev)
mod(xd)
assert mod._x_i8
L = _lib.lib()
p = mod._packed
a_t, a_limbs = (p["a_t_f16"].data_ptr(), -1) if "a_t_f16" in p else (p["a_t"].data_ptr(), p["a_limbs"])
Mp, rp = L.lqer_padded_m(M), L.lqer_padded_r(r)
st = torch.cuda.current_stream().cuda_stream
var = {}
for name, tune in (("one launch", 0), ("three launches", _lib.TUNE_I8_SPLIT_SIDE)):
    d = mod._desc()
    d.tuning = tune
    nscr = L.lqer_lowrank_xa_scratch_bytes(C.byref(d), M)
    var[name] = (d, torch.empty(ops.linear_sizes(d, M).workspace, dtype=torch.uint8, device=dev), torch.empty(Mp, rp, dtype=torch.bfloat16, device=dev),
                 torch.empty(max(nscr, 16), dtype=torch.uint8, device=dev), nscr)


def run(v):
    d, img, xaq, scr, nscr = v
    _lib.check(L.lqer_quantize_act_xa(C.byref(d), xd.data_ptr(), _lib.F16, M, K, a_t, a_limbs, img.data_ptr(), xaq.data_ptr(), scr.data_ptr(), nscr, st), "qxa")


times = {k: [] for k in var}
for _ in range(a.rounds):
    for k, v in var.items():
        for _ in range(3):
            run(v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run(v)
        e1.record()
        torch.cuda.synchronize()
        times[k].append(e0.elapsed_time(e1) / a.iters * 1e3)
by = M * K * 2 + M * (-(-K // 128) * 128)
for k, t in times.items():
    t = sorted(t)
    print(f"M={M} K={K} r={r} {k:15s} median {t[len(t) // 2]:8.1f} us  min {t[0]:8.1f} us   ({by / t[len(t) // 2] / 1e6:.2f} TB/s of x read + image written)")
