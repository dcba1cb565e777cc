#!/usr/bin/env python3
"""A/B timing of lqer_quantize_act_xa (activation stage + side path x A) across library builds for the configurations
that do not take the fused blocks-of-16 kernel.
usage: python tools/ab_xa.py M K r cfg lib_a.so [lib_b.so ...]     cfg = int (8-bit per-token x, whole-row A_out) |
       i8 (the same on the int8 route: int8 activation image, ONE fp16 image of A^T) |
       a16 (fp16 route, x is its own image) | opt (blocks of 16, rank > 64: unfused)"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lqer_amd import _lib
from tools.ab_gemm import load
M, K, r, cfg = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
N = 256
dev = torch.device("cuda:0")
x = torch.randn(M, K, dtype=torch.float16, device=dev)
w4 = _lib.QFmt(1, 4, 128, 8, 127)
none = _lib.QFmt(0, 0, 0, 8, 127)
if cfg == "int":
    fx = _lib.QFmt(1, 8, -1, 8, 127); desc = _lib.LinearDesc(K, N, r, 0, fx, w4, none, fx, fx); a_limbs = 2
elif cfg == "i8":
    fx = _lib.QFmt(1, 8, -1, 8, 127); desc = _lib.LinearDesc(K, N, r, 0, _lib.QFmt(3, 8, -1, 8, 127), w4, none, fx, fx); a_limbs = -1
elif cfg == "a16":
    desc = _lib.LinearDesc(K, N, r, 0, _lib.QFmt(2, 11, 0, 8, 127), w4, none, _lib.QFmt(0, 16, 0, 8, 127), none); a_limbs = 1
else:
    fx = _lib.QFmt(1, 8, 16, 8, 127); desc = _lib.LinearDesc(K, N, r, 0, fx, w4, none, fx, fx); a_limbs = 1
rp = (r + 15) // 16 * 16
at = (0.01 * torch.randn(3, rp, K)).to(torch.float16 if cfg in ("a16", "i8") else torch.bfloat16).to(dev)
xq = x if cfg == "a16" else torch.empty(M, K, dtype=torch.bfloat16, device=dev)
xaq = torch.empty(M, 3 * rp, dtype=torch.bfloat16, device=dev)
libs = [(os.path.basename(p), load(p)) for p in sys.argv[5:]]
res = {n: [] for n, _ in libs}
ref = None
for rnd in range(6):
    for n, L in libs:
        nscr = L.lqer_lowrank_xa_scratch_bytes(C.byref(desc), M)
        scr = torch.zeros(nscr, dtype=torch.uint8, device=dev)
        call = lambda: L.lqer_quantize_act_xa(C.byref(desc), x.data_ptr(), _lib.F16, M, K, at.data_ptr(), a_limbs, xq.data_ptr(), xaq.data_ptr(), scr.data_ptr(), nscr, None)
        for _ in range(3):
            rc = call(); assert rc == 0, (rc, L.lqer_last_error())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): call()
        e1.record(); torch.cuda.synchronize()
        res[n].append(e0.elapsed_time(e1) / 20 * 1e3)
        chk = xaq[:, :rp].float().abs().sum().item()
        ref = ref or chk
        assert abs(chk - ref) < 2e-3 * ref, (n, chk, ref)
for n, v in res.items():
    v.sort(); print(f"M={M} K={K} r={r} {cfg:4s} {n:28s} median {v[len(v)//2]:8.2f} us  min {v[0]:8.2f} us")
