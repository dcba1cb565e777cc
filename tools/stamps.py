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
xq = torch.randn(M, K).to(torch.bfloat16).to(dev)
wp = torch.randint(0, 256, ((N // 16) * (K // 64) * 576,), dtype=torch.uint8)
wv = wp.view(-1, 576); wv[:, 512:] = torch.randint(0, 3, (wv.shape[0], 64), dtype=torch.uint8) + 250
wp = wp.to(dev)
xaq = (0.1 * torch.randn(M, 32)).to(torch.bfloat16).to(dev)
bt = (0.1 * torch.randn(3 * N * 32)).to(torch.bfloat16).to(dev)
y = torch.empty(M, N, dtype=torch.float16, device=dev)
f8 = _lib.QFmt(1, 8, 16, 8, 127); f4 = _lib.QFmt(1, 4, 16, 8, 127)
desc = _lib.LinearDesc(K, N, r, 0, f8, f4, f8, f8, f8)
for _ in range(3):
    assert L.lqer_linear_gemm(C.byref(desc), xq.data_ptr(), M, wp.data_ptr(), xaq.data_ptr(), bt.data_ptr(), 1, None, y.data_ptr(), 1, N, None, 0, None) == 0
torch.cuda.synchronize()
b = buf.cpu().view(256, NW, 8).double()
names = ["DMA issue", "LDS reads+waits", "s2", "s3", "LOAD barrier", "COMPUTE issue", "barrier after COMPUTE", "loop overhead"]
steps = K // 64
for grp, sl in [(f"waves {a}-{a+3}", slice(a, a + 4)) for a in range(0, NW, 4)]:
    m = b[:, sl, :].mean(dim=(0, 1)) / steps
    print(grp, "cycles per k-step:", ", ".join(f"{n} {v:.0f}" for n, v in zip(names, m.tolist())), f"| total {m.sum():.0f}")
