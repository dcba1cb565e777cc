#!/usr/bin/env python3
"""A/B timing of lqer_quantize_act_xa (fused activation quantize + side path) across library builds, C2 operands.
usage: python tools/ab_quant.py [--M m --K k --r r] lib_a.so lib_b.so ..."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lqer_amd import _lib
from tools.ab_gemm import load
import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=2048); ap.add_argument("--K", type=int, default=4096); ap.add_argument("--r", type=int, default=32)
ap.add_argument("--nocheck", action="store_true", help="ablation builds: skip the cross-build checksum")
ap.add_argument("libs", nargs="+")
args = ap.parse_args()
M, K, N, r = args.M, args.K, 4096, args.r
rp = -(-r // 16) * 16
dev = torch.device("cuda:0")
x = torch.randn(M, K, dtype=torch.float16, device=dev)
f8 = _lib.QFmt(1, 8, 16, 8, 127); f4 = _lib.QFmt(1, 4, 16, 8, 127)
desc = _lib.LinearDesc(K, N, r, 0, f8, f4, f8, f8, f8)
at = (0.01 * torch.randn(3, rp, K)).to(torch.bfloat16).to(dev)
xq = torch.empty(-(-M // 256) * 256, K, dtype=torch.bfloat16, device=dev)
xaq = torch.empty(-(-M // 256) * 256, rp, dtype=torch.bfloat16, device=dev)
libs = [(os.path.basename(p), load(p)) for p in args.libs]
res = {n: [] for n, _ in libs}
ref = None
for rnd in range(8):
    for n, L in libs:
        nscr = L.lqer_lowrank_xa_scratch_bytes(C.byref(desc), M)
        scr = torch.zeros(nscr, dtype=torch.uint8, device=dev)
        call = lambda: L.lqer_quantize_act_xa(C.byref(desc), x.data_ptr(), _lib.F16, M, K, at.data_ptr(), 1, xq.data_ptr(), xaq.data_ptr(), scr.data_ptr(), nscr, None)
        for _ in range(3): assert call() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): call()
        e1.record(); torch.cuda.synchronize()
        res[n].append(e0.elapsed_time(e1) / 20 * 1e3)
        chk = (xq[:M].float().sum().item(), xaq[:M].float().abs().sum().item())
        ref = ref or chk
        assert args.nocheck or abs(chk[0] - ref[0]) < 1e-3 * abs(ref[0]) + 1 and abs(chk[1] - ref[1]) < 1e-3 * ref[1], (n, chk, ref)
for n, v in res.items():
    v.sort(); print(f"{n:28s} median {v[len(v)//2]:7.2f} us  min {v[0]:7.2f} us (quantize + side path, {len(v)} rounds)")
