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

# the same forwards replayed from a captured graph (no per-launch host work)
for M in (1, 16, 64):
    xd = x[:M].half().to(dev)
    mod(xd)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        mod(xd)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        y = mod(xd)
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"M={M:3d}: {e0.elapsed_time(e1)/200*1e3:7.1f} us per forward (graph replay)")
