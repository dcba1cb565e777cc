#!/usr/bin/env python3
"""One Llama-7B decoder layer's seven projections (M = 2048 tokens, rank 32, W4A8 MXINT) through the module API, with and
without SharedActivation groups (q/k/v and gate/up quantize their common input once).  GPU time per layer, HIP events."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lqer_amd
from lqer_amd.linear import SharedActivation
from bench import MXINT_Q, make_case

dev = torch.device("cuda:0")
M, H, I, r = 2048, 4096, 11008, 32
def mk(K, N, seed):
    x, W, A, B = make_case(8, K, N, r, seed=seed)
    m = lqer_amd.LinearFlexibleLqer(K, N, bias=False, q_config=MXINT_Q, l_config={"rank": r})
    m.load_state_dict({"weight": W, "A": A, "B": B})
    return m.to(dev).half()
def layer():
    return {n: mk(*s) for n, s in {"q": (H, H, 1), "k": (H, H, 2), "v": (H, H, 3), "o": (H, H, 4), "gate": (H, I, 5), "up": (H, I, 6),
                                   "down": (I, H, 7)}.items()}
def run(L, x, x2, x3):
    L["q"](x); L["k"](x); L["v"](x); L["o"](x2); L["gate"](x2); L["up"](x2); L["down"](x3)
x = torch.randn(M, H, device=dev, dtype=torch.float16); x2 = torch.randn(M, H, device=dev, dtype=torch.float16)
x3 = torch.randn(M, I, device=dev, dtype=torch.float16)
for name, share in (("one by one", False), ("shared q/k/v and gate/up inputs", True)):
    L = layer()
    if share:
        SharedActivation([L["q"], L["k"], L["v"]]); SharedActivation([L["gate"], L["up"]])
    with torch.no_grad():
        for _ in range(3): run(L, x, x2, x3)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run(L, x, x2, x3)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
    fl = 2.0 * M * (4 * H * H + 3 * H * I) + 2.0 * M * r * (7 * H + 3 * I + 4 * H + 3 * I)
    print(f"{name:34s}: {best * 1e3:8.1f} us per layer  ({fl / best / 1e9:7.1f} TFLOP/s-equiv)")
    del L
