#!/usr/bin/env python3
"""The fused quantized attention products (lqer_matmul_q) against the two-step route they replace (HIP quantizer kernels
-> fp32 -> cast -> torch.matmul), interleaved rounds in one process, at the BASELINE attention shape [32 b, 2048, 128].
usage: python tools/ab_matmul.py [--bh 32 --s 2048 --d 128]"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lqer_amd
from lqer_amd import functional as F
ap = argparse.ArgumentParser()
ap.add_argument("--bh", type=int, default=32); ap.add_argument("--s", type=int, default=2048); ap.add_argument("--d", type=int, default=128)
ap.add_argument("--rounds", type=int, default=8); ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
qc = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "matmul_config.json")))
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
q = torch.randn(a.bh, a.s, a.d, generator=g).half().to(dev)
k = torch.randn(a.bh, a.s, a.d, generator=g).half().to(dev)
v = torch.randn(a.bh, a.s, a.d, generator=g).half().to(dev)
p = torch.softmax(torch.randn(a.bh, a.s, a.s, generator=g), dim=-1).half().to(dev)
def two_step(x, y):
    return torch.matmul(F._quantize(x, dict(qc["x_quantizer"])), F._quantize(y, dict(qc["w_quantizer"])))
cases = {"Q K^T": (q, k.transpose(1, 2)), "P V": (p, v)}
for name, (x, y) in cases.items():
    o1, o2 = lqer_amd.matmul_flexible(x, y, qc), two_step(x, y)
    d = float((o1.float() - o2.float()).norm() / o2.float().norm())
    times = {"fused": [], "two-step": []}
    for _ in range(a.rounds):
        for which, fn in (("fused", lambda: lqer_amd.matmul_flexible(x, y, qc)), ("two-step", lambda: two_step(x, y))):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters): fn()
            e1.record(); torch.cuda.synchronize()
            times[which].append(e0.elapsed_time(e1) / a.iters * 1e3)
    f, t = sorted(times["fused"])[len(times["fused"]) // 2], sorted(times["two-step"])[len(times["two-step"]) // 2]
    by = (x.numel() + y.numel() + o1.numel()) * 2
    print(f"{name:6s} [{a.bh}, {a.s}, {a.d}]: fused {f:8.1f} us ({by / f / 1e6:6.2f} TB/s of algorithmic bytes), two-step {t:8.1f} us, "
          f"ratio {t / f:.2f}x, rel-L2 between them {d:.1e}")
