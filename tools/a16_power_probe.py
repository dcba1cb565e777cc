"""Is the fp16 main loop slower than the bf16 one because of its instructions or because of its data?  Times the
forward of one Linear (M=16384, 5120x5120, rank 64) for the W4A8 route and the W4A16 fp16 route, the latter with
full-precision fp16 activations and with activations snapped to few significand bits (what the A8 route multiplies)."""
import sys
import torch
sys.path.insert(0, ".")
import lqer_amd
from bench import A16_Q, INT_Q, make_case

DEV = "cuda:0"
M, K, N, r = 16384, 5120, 5120, 64
x, W, A, B = make_case(M, K, N, r, seed=0, quantize_ab=False)


def run(qc, xin, label):
    mod = lqer_amd.LinearFlexibleLqer(K, N, bias=False, q_config=qc, l_config={"rank": r})
    mod.load_state_dict({"weight": W, "A": A, "B": B})
    mod = mod.to(DEV).half()
    xd = xin.half().to(DEV)
    for _ in range(3):
        mod(xd)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        mod(xd)
    e1.record()
    torch.cuda.synchronize()
    print(f"{label:48s} {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us/forward  (fp16 route: {mod._x_f16})")


run(INT_Q, x, "W4A8 per-token (bf16 main loop)")
run(A16_Q, x, "W4A16 fp16 main loop, randn activations")
run(A16_Q, torch.round(x * 16) / 16, "W4A16 fp16 main loop, activations k/16")
run(A16_Q, torch.sign(x) * torch.exp2(torch.round(torch.log2(x.abs() + 1e-6))), "W4A16 fp16 main loop, power-of-two activations")
run(INT_Q, x, "W4A8 per-token (bf16 main loop) again")
