#!/usr/bin/env python3
"""Phase timeline of the one-launch decode kernel (a -DLQER_D1_STAMPS build): s_memrealtime at kernel entry, after the
activation image, after the main loop, after the combine, after the granule gather and at the end, per workgroup.
usage: python tools/d1_stamps.py build/abl/lib_d1stamps.so [K N M]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lqer_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
_probe = C.CDLL(_lib.LIB_PATH)  # (an older build of the library: bind only the entry points it has)
_lib.SIGNATURES = {k: v for k, v in _lib.SIGNATURES.items() if hasattr(_probe, k)}
import lqer_amd
from lqer_amd import ops
from bench import make_case, MXINT_Q
K, N, M = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (4096, 4096, 1)
dev = torch.device("cuda:0"); L = _lib.lib(); L.lqer_debug_set_d1_stamps.argtypes = [C.c_void_p]
x, W, A, B = make_case(8, K, N, 32, seed=0)
mod = lqer_amd.LinearFlexibleLqer(K, N, bias=False, q_config=MXINT_Q, l_config={"rank": 32})
mod.load_state_dict({"weight": W, "A": A, "B": B}); mod = mod.to(dev).half()
xd = x[:M].half().to(dev)
npd = -(-K // 256); nb = npd + (-(-N // 256) * 256) // 16
buf = torch.zeros(nb * 16, dtype=torch.int64, device=dev)
for _ in range(20): mod(xd)
L.lqer_debug_set_d1_stamps(buf.data_ptr())
for _ in range(5): mod(xd)
torch.cuda.synchronize()
b = buf.cpu().view(nb, 16).double()
t0 = b[:, 0].min()
us = (b - t0) / 100.0
prod, cons = us[:npd], us[npd:]
print(f"K={K} N={N} M={M}: {npd} producers, {nb - npd} consumers; times in us after the first workgroup's entry (median / max)")
print(f"  producers: entry {prod[:,0].median():.2f} / {prod[:,0].max():.2f}, x quantized {prod[:,6].median():.2f} / {prod[:,6].max():.2f}, "
      f"published {prod[:,5].median():.2f} / {prod[:,5].max():.2f}")
print(f"  consumers: first x block quantized {cons[:,6].median():6.2f} / {cons[:,6].max():6.2f}")
for i, name in [(0, "entry"), (1, "activation image"), (2, "main loop done"), (3, "combined"), (4, "granules gathered"), (7, "all waves agree"),
                (8, "A_out staged"), (9, "side MFMA"), (10, "B_out"), (5, "end")]:
    print(f"  consumers: {name:18s} {cons[:,i].median():6.2f} / {cons[:,i].max():6.2f}")
