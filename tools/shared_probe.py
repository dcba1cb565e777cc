#!/usr/bin/env python3
"""q/k/v (or gate/up) sharing one quantized input against the same Linears run one by one: GPU time per group call.
    python tools/shared_probe.py [M K N r members]"""
import copy, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lqer_amd
from bench import MXINT_Q, make_case
from lqer_amd.linear import SharedActivation
M, K, N, r, nm = (int(v) for v in sys.argv[1:6]) if len(sys.argv) > 5 else (2048, 4096, 4096, 32, 3)
dev = torch.device("cuda:0")
x, W, A, B = make_case(M, K, N, r, seed=0, quantize_ab=True)
mod = lqer_amd.LinearFlexibleLqer(K, N, bias=False, q_config=MXINT_Q, l_config={"rank": r})
mod.load_state_dict({"weight": W, "A": A, "B": B}); mod = mod.to(dev).half()
xd = x.half().to(dev)
solo = [copy.deepcopy(mod) for _ in range(nm)]
grp_m = [copy.deepcopy(mod) for _ in range(nm)]
grp = SharedActivation(grp_m); assert grp.enabled
def run(ms, n):
    for _ in range(10):
        for m in ms: m(xd)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            for m in ms: m(xd)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[len(ts) // 2]
with torch.no_grad():
    for _ in range(2):
        a, b = run(solo, 30), run(grp_m, 30)
        print(f"M={M} K={K} N={N} r={r} x{nm}: one by one {a:.1f} us, shared input {b:.1f} us per group ({a - b:+.1f})")
