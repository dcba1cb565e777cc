#!/usr/bin/env python3
"""In-kernel timeline of the block-16 one-launch activation kernel (act16_fused.hip, -DLQER_CLOCKPROBE build): shader cycles from a wave's
start to the phase boundaries of its first slab, medians over all waves.
usage: python tools/clock_probe_a16.py build/abl/liblqer_cp.so [--M 2048 --K 4096 --r 32]"""
import argparse, ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("lib")
ap.add_argument("--M", type=int, default=2048)
ap.add_argument("--K", type=int, default=4096)
ap.add_argument("--N", type=int, default=256)
ap.add_argument("--r", type=int, default=32)
a = ap.parse_args()
from lqer_amd import _lib
_lib.LIB_PATH = os.path.abspath(a.lib)
import lqer_amd
from bench import MXINT_Q, make_case
from lqer_amd import ops
L = _lib.lib()
L.lqer_debug_set_a16_stamp_buffer.argtypes = [C.c_void_p]
dev = torch.device("cuda:0")
M, K, N, r = a.M, a.K, a.N, a.r
x, W, A, B = make_case(M, K, N, r, seed=0, quantize_ab=True)
mod = lqer_amd.LinearFlexibleLqer(K, N, bias=False, q_config=MXINT_Q, l_config={"rank": r})
mod.load_state_dict({"weight": W, "A": A, "B": B})
mod = mod.to(dev).half()
xd = x.half().to(dev)
mod(xd[:256])
assert "a_t_b16" in mod._packed
wgs = -(-M // 8)
buf = torch.zeros(wgs * 8 * 8, dtype=torch.int64, device=dev)
assert L.lqer_debug_set_a16_stamp_buffer(buf.data_ptr()) == 0
desc = mod._desc()
p = mod._packed
st = torch.cuda.current_stream().cuda_stream
Kp, Mp, rp = L.lqer_padded_k(K), L.lqer_padded_m(M), L.lqer_padded_r(r)
ws = torch.empty(ops.linear_sizes(desc, M).workspace, dtype=torch.uint8, device=dev)
xq = ws.data_ptr(); xaq = xq + ((Mp * Kp * 2 + 255) // 256) * 256; scr = xaq + ((Mp * rp * 2 + 255) // 256) * 256
nscr = L.lqer_lowrank_xa_scratch_bytes(C.byref(desc), M)
def launch():
    _lib.check(L.lqer_quantize_act_xa(C.byref(desc), xd.data_ptr(), _lib.F16, M, K, p["a_t_b16"].data_ptr(), -2, xq, xaq, scr, nscr, st), "q")
t0 = time.time()
while time.time() - t0 < 1.5:
    for _ in range(20): launch()
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): launch()
e1.record(); torch.cuda.synchronize()
b = buf.cpu().view(wgs, 8, 8).double()
med = lambda t: t.median().item()
names = ["row requests out", "row 0 landed + quantized", "8 rows quantized, stores issued", "fragments landed, 16 steps multiplied", "partial tiles barrier passed"]
print(f"M={M} K={K} r={r}: call {e0.elapsed_time(e1) / 20 * 1e3:.1f} us, {wgs} workgroups of 8 rows")
for i, n in enumerate(names, 1):
    print(f"  {n:40s} {med(b[:, :, i]):8.0f} cycles (min {b[:, :, i].min().item():.0f}, max {b[:, :, i].max().item():.0f})")
rt = b[:, :, 6]
print(f"  barrier passed, 100 MHz clock: first workgroup to last {(rt.max().item() - rt.min().item()) / 100:.2f} us")
