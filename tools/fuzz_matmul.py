#!/usr/bin/env python3
"""Randomised check of the quantized attention products (lqer_matmul_q: image + fused GEMM kernels) against the two-step
route (HIP quantizers -> torch.matmul in fp32) over random batch / S1 / K / S2, dtypes and both operand layouts.
usage: python tools/fuzz_matmul.py [cases] [seed]"""
import json, os, random, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lqer_amd
from lqer_amd import functional as F
qc = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "matmul_config.json")))
n, seed = (int(sys.argv[1]) if len(sys.argv) > 1 else 60), (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
rng = random.Random(seed)
dev = torch.device("cuda:0")
bad = 0
for i in range(n):
    b = rng.choice([1, 2, 3, 5])
    S1 = rng.choice([1, 17, 64, 128, 130, 333, 512, 1000])
    K = rng.choice([16, 40, 64, 72, 128, 136, 256, 520, 1100])
    S2 = rng.choice([16, 40, 128, 256, 300, 512, 1024, 1537])
    dt = rng.choice([torch.float16, torch.float16, torch.bfloat16, torch.float32])
    tr = rng.random() < 0.5  # y as the transposed view of a [S2, K] tensor (Q K^T) or a plain [K, S2] tensor (P V)
    g = torch.Generator().manual_seed(seed * 1000 + i)
    x = (torch.randn(b, S1, K, generator=g) * 2).to(dt).to(dev)
    y = (torch.randn(b, S2, K, generator=g).transpose(1, 2) if tr else torch.randn(b, K, S2, generator=g)).to(dt).to(dev)
    # (round 4) blocks other than 16 - 32, 64, 48 or the whole row, independently per operand - take the pre-quantized images
    import copy
    qcc = copy.deepcopy(qc)
    bx, by = rng.choice([16, 16, 32, 64, 48, -1]), rng.choice([16, 16, 32, 64, -1])
    qcc["x_quantizer"]["block_size"] = [1, bx]
    qcc["w_quantizer"]["block_size"] = [1, by]
    got = lqer_amd.matmul_flexible(x, y, qcc).float()
    ref = torch.matmul(F._quantize(x, dict(qcc["x_quantizer"])).float(), F._quantize(y, dict(qcc["w_quantizer"])).float())
    err = float((got - ref).norm() / ref.norm().clamp_min(1e-30))
    tol = {torch.float16: 1e-3, torch.bfloat16: 6e-3, torch.float32: 1e-6}[dt]
    ok = err <= tol
    bad += not ok
    print(f"{i + 1:3d} b={b} S1={S1:5d} K={K:5d} S2={S2:5d} {'K^T-view' if tr else 'plain   '} {str(dt).split('.')[-1]:9s} rel-L2 {err:.2e} {'ok' if ok else 'FAIL'}")
print(f"{n - bad} / {n} within tolerance")
sys.exit(1 if bad else 0)
