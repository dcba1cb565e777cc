#!/usr/bin/env python3
"""One Linear forward through three doors, same buffers, back to back on one box: the split C calls (bench.py's timed loop), the one C call
lqer_linear_forward (what the module issues), and the nn.Module itself - where the module's few percent over the split calls come from.
usage: python tools/fwd_paths_time.py [--q int|mx] [--M 2048 --K 4096 --N 4096 --r 32]"""
import argparse, ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=2048); ap.add_argument("--K", type=int, default=4096); ap.add_argument("--N", type=int, default=4096)
ap.add_argument("--r", type=int, default=32); ap.add_argument("--q", default="int")
a = ap.parse_args()
import lqer_amd
from lqer_amd import _lib, ops
from bench import INT_Q, MXINT_Q, make_case
dev = torch.device("cuda:0")
M, K, N, r = a.M, a.K, a.N, a.r
qc = INT_Q if a.q == "int" else MXINT_Q
x, W, A, B = make_case(M, K, N, r, seed=0, quantize_ab=a.q != "int")
mod = lqer_amd.LinearFlexibleLqer(K, N, bias=False, q_config=qc, l_config={"rank": r})
mod.load_state_dict({"weight": W, "A": A, "B": B})
mod = mod.to(dev).half()
xd = x.half().to(dev)
mod(xd)
L, desc, p, dt = _lib.lib(), mod._desc(), mod._packed, _lib.F16
a_t, a_limbs = mod._side_image(M, desc, dt)
wsb = ops.linear_sizes(desc, M).workspace
Kp, Mp, rp = L.lqer_padded_k(K), L.lqer_padded_m(M), L.lqer_padded_r(r)
act = L.lqer_act_image_bytes(C.byref(desc), M)
offs = act + ((Mp * rp * 2 + 255) // 256) * 256
nscr, gscr = L.lqer_lowrank_xa_scratch_bytes(C.byref(desc), M), L.lqer_linear_gemm_scratch_bytes(C.byref(desc), M)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev); y = torch.empty(M, N, dtype=torch.float16, device=dev)
st = torch.cuda.current_stream().cuda_stream
dref = C.byref(desc); ready = C.c_size_t(0); rref = C.byref(ready)
wp, bt, bl = p["w"].data_ptr(), p["b_t"].data_ptr(), p["b_limbs"]
def split():
    L.lqer_quantize_act_xa_prep(dref, xd.data_ptr(), dt, M, K, a_t, a_limbs, ws.data_ptr(), ws.data_ptr() + act, ws.data_ptr() + offs, nscr, ws.data_ptr() + offs, rref, st)
    L.lqer_linear_gemm_prepared(dref, ws.data_ptr(), M, wp, ws.data_ptr() + act, bt, bl, None, y.data_ptr(), dt, N, ws.data_ptr() + offs, gscr, ready.value, st)
def one():
    L.lqer_linear_forward(dref, xd.data_ptr(), dt, M, K, wp, a_t, bt, a_limbs, bl, None, y.data_ptr(), N, ws.data_ptr(), wsb, st)
def module():
    mod(xd)
def timeit(f, n=3000):
    for _ in range(200): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return 1e6 * (t2 - t0) / n, 1e6 * (t1 - t0) / n
for rep in range(3):
    for name, f in (("split C calls", split), ("lqer_linear_forward", one), ("nn.Module", module)):
        tot, host = timeit(f)
        print(f"{a.q} {name:22s} {tot:7.2f} us per forward (host issue {host:5.1f} us)")
