#!/usr/bin/env python3
"""Read the per-section cycle sums of a -DLQER_STAMPS build (diagnostic).  usage: stamps.py lib_STAMPS.so"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lqer_amd import _lib
from tools.ab_gemm import load
L = load(sys.argv[1])
NW = int(sys.argv[2]) if len(sys.argv) > 2 else 8  # waves per workgroup of the build
L.lqer_debug_set_stamp_buffer.argtypes = [C.c_void_p]
M, K, N, r = 2048, 4096, 4096, 32
dev = torch.device("cuda:0")
buf = torch.zeros(256 * NW * 8, dtype=torch.int64, device=dev)
assert L.lqer_debug_set_stamp_buffer(buf.data_ptr()) == 0
from tools._operands import real_operands  # (real bit patterns: random image bytes give random block exponents -> inf / NaN)
op = real_operands(M, K, N, r)
desc, xq, wp, xaq, bt, y = op["desc"], op["xq"], op["w"], op["xaq"], op["b_t"], op["y"]
for _ in range(3):
    assert L.lqer_linear_gemm(C.byref(desc), xq.data_ptr(), M, wp.data_ptr(), xaq.data_ptr(), bt.data_ptr(), 1, None, y.data_ptr(), 1, N, None, 0, None) == 0
torch.cuda.synchronize()
b = buf.cpu().view(256, NW, 8).double()
names = ["DMA issue", "LDS reads+waits", "s2", "s3", "LOAD barrier", "COMPUTE issue", "barrier after COMPUTE", "loop overhead"]
steps = K // 64
for grp, sl in [(f"waves {a}-{a+3}", slice(a, a + 4)) for a in range(0, NW, 4)]:
    m = b[:, sl, :].mean(dim=(0, 1)) / steps
    print(grp, "cycles per k-step:", ", ".join(f"{n} {v:.0f}" for n, v in zip(names, m.tolist())), f"| total {m.sum():.0f}")
