import sys, os, time, torch
sys.path.insert(0, "/root/repo")
import lqer_amd
from bench import make_case, MXINT_Q
dev = torch.device("cuda:0")
K = N = 4096; r = 32
x, W, A, B = make_case(64, K, N, r, seed=0)
mod = lqer_amd.LinearFlexibleLqer(K, N, bias=False, q_config=MXINT_Q, l_config={"rank": r})
mod.load_state_dict({"weight": W, "A": A, "B": B}); mod = mod.to(dev).half()
for M in (1, 4, 16, 32, 64):
    xd = x[:M].half().to(dev)
    for _ in range(5): mod(xd)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): mod(xd)
    e1.record(); torch.cuda.synchronize()
    print(f"M={M:3d}: {e0.elapsed_time(e1)/50*1e3:7.1f} us per forward")
