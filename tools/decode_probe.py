import ctypes as C, sys, os, time, torch
sys.path.insert(0, "/root/repo")
import lqer_amd
from lqer_amd import _lib, ops
from bench import make_case, MXINT_Q
dev = torch.device("cuda:0"); L = _lib.lib()
for (K, N, M) in ((4096, 4096, 1), (4096, 4096, 8), (4096, 11008, 1), (11008, 4096, 1)):
    r = 32
    x, W, A, B = make_case(8, K, N, r, seed=0)
    mod = lqer_amd.LinearFlexibleLqer(K, N, bias=False, q_config=MXINT_Q, l_config={"rank": r})
    mod.load_state_dict({"weight": W, "A": A, "B": B}); mod = mod.to(dev).half()
    xd = x[:M].half().to(dev); y = mod(xd)
    desc, p = mod._desc(), mod._packed
    ws = ops.workspace(dev, ops.linear_sizes(desc, M).workspace)
    st = torch.cuda.current_stream().cuda_stream
    call = lambda: L.lqer_linear_forward(C.byref(desc), xd.data_ptr(), _lib.F16, M, K, p["w"].data_ptr(), p["a_t"].data_ptr(), p["b_t"].data_ptr(), p["a_limbs"], p["b_limbs"], None, y.data_ptr(), N, ws.data_ptr(), ws.numel(), st)
    for _ in range(50): call()
    torch.cuda.synchronize()
    res = []
    for rnd in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for _ in range(200): call()
        e1.record(); th = time.perf_counter() - t0; torch.cuda.synchronize()
        res.append((e0.elapsed_time(e1) / 200 * 1e3, th / 200 * 1e6))
    res.sort()
    print(f"K={K} N={N} M={M}: {res[2][0]:.2f} us per forward (GPU events), host issue {res[2][1]:.2f} us per call")
