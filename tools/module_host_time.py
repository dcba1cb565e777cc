#!/usr/bin/env python3
"""Host time of one module forward (python + ctypes + launches, no synchronisation) against the GPU time of the same forward - where the
drop-in module is host-bound.   usage: python tools/module_host_time.py [--M 2048 --K 4096 --N 4096 --r 32 --q int|mx] [--profile]"""
import argparse, cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=2048); ap.add_argument("--K", type=int, default=4096); ap.add_argument("--N", type=int, default=4096)
ap.add_argument("--r", type=int, default=32); ap.add_argument("--q", default="int"); ap.add_argument("--profile", action="store_true")
a = ap.parse_args()
import lqer_amd
from bench import INT_Q, MXINT_Q, make_case
dev = torch.device("cuda:0")
qc = INT_Q if a.q == "int" else MXINT_Q
x, W, A, B = make_case(a.M, a.K, a.N, a.r, seed=0, quantize_ab=a.q != "int")
mod = lqer_amd.LinearFlexibleLqer(a.K, a.N, bias=False, q_config=qc, l_config={"rank": a.r})
mod.load_state_dict({"weight": W, "A": A, "B": B})
mod = mod.to(dev).half()
xd = x.half().to(dev)
for _ in range(50): mod(xd)
torch.cuda.synchronize()
n = 2000
t0 = time.perf_counter()
for _ in range(n): mod(xd)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{a.q} M={a.M} K={a.K} N={a.N}: issue {1e6 * (t1 - t0) / n:.1f} us per forward on the host, {1e6 * (t2 - t0) / n:.1f} us per forward until the GPU is done")
# host alone: the GPU far behind is no back-pressure below the queue depth, so time a short burst after a sync
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(64): mod(xd)
t1 = time.perf_counter()
print(f"  burst of 64 after a sync: {1e6 * (t1 - t0) / 64:.1f} us per forward on the host")
torch.cuda.synchronize()
if a.profile:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(500): mod(xd)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
