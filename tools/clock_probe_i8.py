#!/usr/bin/env python3
"""In-kernel timeline of the int8 GEMM (a -DLQER_CLOCKPROBE build stamps s_memtime / s_memrealtime at the start, around the
main loop and at the end of every wave): ring fill, main loop (cycles per 128-k step, sustained clock), epilogue, and the
spread of tile start times inside a launch.  Runs >= 2 s of back-to-back launches first (DVFS settles).
usage: python tools/clock_probe_i8.py build/abl/liblqer_clockprobe.so [--K 5120 --N 5120 --M 16384 --wblock 128]
(build:  cd lqer_amd/csrc && make -s -j8 EXTRA=-DLQER_CLOCKPROBE OUT=../../build/abl/liblqer_clockprobe.so OBJDIR=../../build/obj_cp)"""
import argparse, ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("lib")
ap.add_argument("--M", type=int, default=16384)
ap.add_argument("--K", type=int, default=5120)
ap.add_argument("--N", type=int, default=5120)
ap.add_argument("--r", type=int, default=64)
ap.add_argument("--wblock", type=int, default=128)
ap.add_argument("--slots", type=int, default=16, help="u64 stamps per wave the build writes (round-5 builds: 8)")
ap.add_argument("--tuning", type=lambda v: int(v, 0), default=0, help="lqer_linear_desc_t.tuning, e.g. 0x80000: the pre-pass with segment partials instead of the in-GEMM exchange")
a = ap.parse_args()
from lqer_amd import _lib
_lib.LIB_PATH = os.path.abspath(a.lib)  # the module and every helper bind the diagnostic build
import lqer_amd
from bench import INT_Q, _bfp, make_case
from lqer_amd import ops
L = _lib.lib()
L.lqer_debug_set_i8_stamp_buffer.argtypes = [C.c_void_p]
dev = torch.device("cuda:0")
M, K, N, r = a.M, a.K, a.N, a.r
qc = dict(INT_Q, w_quantizer=_bfp(4, [1, a.wblock], False))
x, W, A, B = make_case(M, K, N, r, seed=0, quantize_ab=False)
mod = lqer_amd.LinearFlexibleLqer(K, N, bias=False, q_config=qc, l_config={"rank": r})
mod.load_state_dict({"weight": W, "A": A, "B": B})
mod = mod.to(dev).half()
xd = x.half().to(dev)
y = mod(xd)
assert mod._x_i8
L.lqer_gemm_tile_rows.argtypes = [C.c_void_p, C.c_int64, C.c_int]
BMt = L.lqer_gemm_tile_rows(C.byref(mod._desc()), M, _lib.F16)  # 128- or 256-row tiles of the int8 kernel
tiles = (-(-M // BMt)) * (-(-N // 256))
S = a.slots
buf = torch.zeros(tiles * 8 * S, dtype=torch.int64, device=dev)
assert L.lqer_debug_set_i8_stamp_buffer(buf.data_ptr()) == 0
desc = mod._desc()
desc.tuning = a.tuning
p = mod._packed
st = torch.cuda.current_stream().cuda_stream
Kp, Mp, rp = L.lqer_padded_k(K), L.lqer_padded_m(M), L.lqer_padded_r(r)
ws = torch.empty(ops.linear_sizes(desc, M).workspace, dtype=torch.uint8, device=dev)
xq = ws.data_ptr(); xaq = xq + ((Mp * Kp * 2 + 255) // 256) * 256; scr = xaq + ((Mp * rp * 2 + 255) // 256) * 256
nscr = L.lqer_lowrank_xa_scratch_bytes(C.byref(desc), M); gscr = L.lqer_linear_gemm_scratch_bytes(C.byref(desc), M)
_lib.check(L.lqer_quantize_act_xa(C.byref(desc), xd.data_ptr(), _lib.F16, M, K, p["a_t"].data_ptr(), p["a_limbs"], xq, xaq, scr, nscr, st), "q")
def launch():
    _lib.check(L.lqer_linear_gemm(C.byref(desc), xq, M, p["w"].data_ptr(), xaq, p["b_t"].data_ptr(), p["b_limbs"], None, y.data_ptr(), _lib.F16, N, scr, gscr, st), "g")
t0 = time.time()
while time.time() - t0 < 2.5:
    for _ in range(20): launch()
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): launch()
e1.record(); torch.cuda.synchronize()
nb = min(tiles, 256)  # persistent grid: one workgroup per CU, the stamps are those of its last tile
b = buf.cpu().view(tiles, 8, S)[:nb].double()
steps = -(-K // 128)
cyc, rt = b[:, :, 0], b[:, :, 1]
clk = (cyc / rt * 100e6).median().item()
med = lambda t: t.median().item()
print(f"M={M} K={K} N={N} r={r} wblock={a.wblock} tuning={a.tuning:#x} tile rows {BMt}: call (pre-pass + GEMM) {e0.elapsed_time(e1) / 20 * 1e3:.1f} us, {tiles} tiles = {tiles / 256:.2f} rounds")
print(f"  main loop   {med(cyc):9.0f} cycles = {med(cyc) / steps:6.0f} per 128-k step ({BMt * 8} = MFMA-bound), {med(rt) / 100:7.2f} us; clock {clk / 1e9:.3f} GHz")
print(f"  ring fill   {med(b[:, :, 2]):9.0f} cycles")
pk = buf.cpu().view(tiles, 8, S)[:nb, :, 3]
parts = [((pk >> (16 * i)) & 0xffff).double() for i in range(4)]
for w in (0, 4):
    print(f"  wave {w}: issue of DMA + loads {parts[0][:, w].median().item():.0f}, conversion pass {parts[1][:, w].median().item():.0f}, "
          f"vmcnt(0) {parts[2][:, w].median().item():.0f}, barrier {parts[3][:, w].median().item():.0f} cycles")
print(f"  epilogue    {med(b[:, :, 4]):9.0f} cycles, {med(b[:, :, 5]) / 100:7.2f} us")
print(f"     of which staging + barrier {med(b[:, :, 6]):7.0f} cycles, math of the first two token tiles {med(b[:, :, 7]):7.0f} cycles")
if S >= 16 and tiles > 256 and BMt == 128 and not (a.tuning & 0x40C0000) and float(b[:, :, 8].max()) > 0:  # round 6, MRX: the FIRST tile's prologue, cycles from the kernel's start
    for w in (0, 4):
        print(f"  wave {w} first tile: ring requests out {med(b[:, w, 8]):.0f}, item published {med(b[:, w, 9]):.0f}, band's granules seen + tables "
              f"{med(b[:, w, 10]):.0f} (polls: median {med(b[:, w, 12]):.0f}, max {b[:, w, 12].max().item():.0f}), main loop starts {med(b[:, w, 11]):.0f}")
    t0, t1 = b[:, 0, 13], b[:, 0, 14]
    print(f"  first start to last end {(t1.max() - t0.min()).item() / 100:.2f} us")
elif S >= 16 and float(b[:, :, 8].max()) > 0:  # round 6: the exchange instantiation's prologue sections, cycles from the wave's start
    for w in (0, 4):
        print(f"  wave {w} prologue: first request out {med(b[:, w, 8]):.0f}, all requests out + tables {med(b[:, w, 9]):.0f}, row maxima computed "
              f"{med(b[:, w, 10]):.0f}, reduced + published {med(b[:, w, 11]):.0f}, main loop starts {med(b[:, w, 2]):.0f}")
    tr = b[:, :2, 12]
    print(f"  gather polls (waves 0-1): median {tr.median().item():.0f}, max {tr.max().item():.0f}, workgroups that polled {(tr.max(dim=1)[0] > 0).sum().item()} of {nb}")
    t0, t1 = b[:, 0, 13], b[:, 0, 14]
    print(f"  workgroup start spread {(t0.max() - t0.min()).item() / 100:.2f} us, end spread {(t1.max() - t1.min()).item() / 100:.2f} us, first start to last end {(t1.max() - t0.min()).item() / 100:.2f} us")
