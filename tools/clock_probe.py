#!/usr/bin/env python3
"""In-kernel clock of the GEMM main loop: a -DLQER_CLOCKPROBE build stamps s_memtime (shader cycles) and
s_memrealtime (100 MHz) around the loop of every wave.  Runs >= 2 s of back-to-back launches first (DVFS settles),
then reports cycles per k-step, the sustained clock and the launch's timeline.  usage: clock_probe.py lib_CLOCKPROBE.so [K [rank]]"""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lqer_amd import _lib
from tools.ab_gemm import load
L = load(sys.argv[1])
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
L.lqer_debug_set_stamp_buffer.argtypes = [C.c_void_p]
M, N, r = 2048, 4096, (int(sys.argv[3]) if len(sys.argv) > 3 else 32)
dev = torch.device("cuda:0")
buf = torch.zeros(256 * 8 * 8, dtype=torch.int64, device=dev)
assert L.lqer_debug_set_stamp_buffer(buf.data_ptr()) == 0
from tools._operands import real_operands  # (real bit patterns: random image bytes give random block exponents -> inf / NaN)
op = real_operands(M, K, N, r)
desc, xq, wp, xaq, bt, y = op["desc"], op["xq"], op["w"], op["xaq"], op["b_t"], op["y"]
def launch():
    assert L.lqer_linear_gemm(C.byref(desc), xq.data_ptr(), M, wp.data_ptr(), xaq.data_ptr(), bt.data_ptr(), 1, None, y.data_ptr(), 1, N, None, 0, None) == 0
t0 = time.time()
while time.time() - t0 < 2.5:
    for _ in range(200): launch()
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): launch()
e1.record(); torch.cuda.synchronize()
b = buf.cpu().view(256, 8, 8).double()
cyc, rt = b[:, :, 0], b[:, :, 1]
steps = K // 64
clk = (cyc / rt * 100e6).median().item()
print(f"K={K}: kernel {e0.elapsed_time(e1) / 100 * 1e3:.2f} us; main loop {cyc.median().item():.0f} cycles = "
      f"{cyc.median().item() / steps:.0f} cycles per k-step, {rt.median().item() / 100:.2f} us; in-kernel clock {clk / 1e9:.3f} GHz")

# timeline on the chip-wide 100 MHz counter (10 ns ticks), relative to the first wave's entry of the LAST launch
ab = buf.cpu().view(256, 8, 8)[:, :, 2:7].double()
t0 = ab[:, :, 0].min().item()
us = lambda t: (t - t0) / 100.0
q = lambda t, p: torch.quantile(t.flatten(), p).item()
names = ["entry", "main loop starts", "main loop ends", "stores issued", "stores acknowledged"]
for i, nm in enumerate(names):
    col = ab[:, :, i]
    print(f"  {nm:20s} median {us(q(col, 0.5)):7.2f} us   first {us(col.min().item()):7.2f}   last {us(col.max().item()):7.2f}")
print(f"  per wave: prologue {q(ab[:, :, 1] - ab[:, :, 0], 0.5) / 100:.2f} us, main loop {q(ab[:, :, 2] - ab[:, :, 1], 0.5) / 100:.2f} us, "
      f"convert + store issue {q(ab[:, :, 3] - ab[:, :, 2], 0.5) / 100:.2f} us, store drain {q(ab[:, :, 4] - ab[:, :, 3], 0.5) / 100:.2f} us")
rq = buf.cpu().view(256, 8, 8)[:, :, 7].double()
if rq.max().item() > 0:
    print(f"  prologue split: entry -> side operands landed {q(rq - ab[:, :, 0], 0.5) / 100:.2f} us, B_out re-quantization + bias + barrier "
          f"{q(ab[:, :, 1] - rq, 0.5) / 100:.2f} us")
