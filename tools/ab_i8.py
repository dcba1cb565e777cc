#!/usr/bin/env python3
"""The int8 main loop against the bf16 256-row kernel on the SAME Linear (W4, per-token A8, rank r, fp16 A / B, per-row
B_out: the C4 configuration), interleaved rounds in one process: lqer_linear_gemm alone (activation images prepared
once per route), and the whole forward (quantizer + side GEMM + pre-pass + GEMM).
usage: python tools/ab_i8.py [--M 16384 --K 5120 --N 5120 --r 64 --wblock 128] [--rounds 8 --iters 10]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--lib" in sys.argv:  # another build of the library (e.g. -DLQER_I8_T16=0), bound before anything loads it
    from lqer_amd import _lib as _l

    _l.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
import lqer_amd  # noqa: E402
from bench import INT_Q, _bfp, make_case  # noqa: E402
from lqer_amd import _lib, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=16384)
    ap.add_argument("--K", type=int, default=5120)
    ap.add_argument("--N", type=int, default=5120)
    ap.add_argument("--r", type=int, default=64)
    ap.add_argument("--wblock", type=int, default=128)
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--lib", default=None, help="path of another build of liblqer_hip.so")
    ap.add_argument("--amax", action="store_true", help="with --rows: 128-row tiles with segment-partial row maxima against atomicMax cells + memset")
    ap.add_argument("--rows", action="store_true", help="compare the int8 kernel's 128-row and 256-row tiles (pinned) and the bf16 route")
    ap.add_argument("--xcd", type=int, nargs="*", default=[], help="(round 6) also the int8 route with XCD-local tile BLOCKS of this many token tiles "
                    "(LQER_TUNE_XCD_BLOCK: applied where the tile grid divides, e.g. 4 or 8 at 16 x 16 tiles)")
    ap.add_argument("--only", default=None, help="time only this variant (counter passes: one map per process)")
    ap.add_argument("--split", action="store_true", help="(round 6) also the int8 route with its activation side pinned to the three launches (LQER_TUNE_ACT8_SPLIT)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    M, K, N, r = a.M, a.K, a.N, a.r
    qc = dict(INT_Q, w_quantizer=_bfp(4, [1, a.wblock], False))
    x, W, A, B = make_case(M, K, N, r, seed=0, quantize_ab=False)
    mod = lqer_amd.LinearFlexibleLqer(K, N, bias=False, q_config=qc, l_config={"rank": r})
    mod.load_state_dict({"weight": W, "A": A, "B": B})
    mod = mod.to(dev).half()
    xd = x.half().to(dev)
    y = mod(xd)
    assert mod._x_i8, "the weight is not eligible for the int8 route"
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    p = mod._packed
    Kp, Mp, rp = L.lqer_padded_k(K), L.lqer_padded_m(M), L.lqer_padded_r(r)
    routes = {}
    variants = [("int8", mod._desc()), ("bf16", mod._desc(plain=True))]
    if a.split:
        d3 = mod._desc()
        d3.tuning = _lib.TUNE_ACT8_SPLIT
        variants = [("int8", mod._desc()), ("i8-3l", d3)]
    for t in a.xcd:
        dx = mod._desc()
        dx.tuning = (t & 0x3f) << 4
        variants.append((f"xcd{t}", dx))
    if a.only:
        variants = [v for v in variants if v[0] == a.only]
    if a.rows:  # the int8 kernel's two tile heights, pinned (same bits)
        d128, d256 = mod._desc(), mod._desc()
        d128.tuning, d256.tuning = _lib.TUNE_I8_ROWS_128, _lib.TUNE_I8_ROWS_256
        variants = [("i8r128", d128), ("i8r256", d256), ("bf16", mod._desc(plain=True))]
        if a.amax:  # ... the forms of the B_out row maxima: exchanged inside the GEMM launch (default where eligible), the pre-pass
            # with segment partials, the pre-pass with atomicMax cells behind a zero-fill launch (round 4)
            d128p, d128a = mod._desc(), mod._desc()
            d128p.tuning = _lib.TUNE_I8_ROWS_128 | _lib.TUNE_AMAX_PARTS
            d128a.tuning = _lib.TUNE_I8_ROWS_128 | _lib.TUNE_AMAX_ATOMIC
            variants = [("i8r128", d128), ("r128pa", d128p), ("r128at", d128a)]
    if a.amax and not a.rows:  # the library's own tile height; the row maxima exchanged in the launch (where eligible) / the two pre-pass forms
        dx, dp, da = mod._desc(), mod._desc(), mod._desc()
        dp.tuning, da.tuning = _lib.TUNE_AMAX_PARTS, _lib.TUNE_AMAX_ATOMIC
        variants = [("i8xch", dx), ("i8pa", dp), ("i8at", da)]
    for name, desc in variants:
        ws = torch.empty(ops.linear_sizes(desc, M).workspace, dtype=torch.uint8, device=dev)
        xq = ws.data_ptr()
        xaq = xq + ((Mp * Kp * 2 + 255) // 256) * 256
        scr = xaq + ((Mp * rp * 2 + 255) // 256) * 256
        nscr = L.lqer_lowrank_xa_scratch_bytes(C.byref(desc), M)
        gscr = L.lqer_linear_gemm_scratch_bytes(C.byref(desc), M)
        routes[name] = dict(desc=desc, ws=ws, xq=xq, xaq=xaq, scr=scr, nscr=nscr, gscr=gscr, route=L.lqer_gemm_route(C.byref(desc), M, _lib.F16))

    def quant(rt):
        # (the int8 route's side GEMM takes A^T as ONE fp16 image, a_limbs = -1 - what the module passes at these token counts)
        i8 = rt["route"] == 3 and "a_t_f16" in p
        _lib.check(L.lqer_quantize_act_xa(C.byref(rt["desc"]), xd.data_ptr(), _lib.F16, M, K, (p["a_t_f16"] if i8 else p["a_t"]).data_ptr(),
                                          -1 if i8 else p["a_limbs"], rt["xq"],
                                          rt["xaq"], rt["scr"], rt["nscr"], st), "quantize_act_xa")

    def gemm(rt):
        _lib.check(L.lqer_linear_gemm(C.byref(rt["desc"]), rt["xq"], M, p["w"].data_ptr(), rt["xaq"], p["b_t"].data_ptr(), p["b_limbs"],
                                      None, y.data_ptr(), _lib.F16, N, rt["scr"], rt["gscr"], st), "linear_gemm")

    print("routes:", {k: v["route"] for k, v in routes.items()}, "(3 = int8 tile kernel, 2 = bf16 256-row kernel)")
    outs = {}
    for name, rt in routes.items():
        quant(rt)
        gemm(rt)
        torch.cuda.synchronize()
        outs[name] = y.clone()
    i8n = "i8r128" if a.rows else ("i8xch" if a.amax else "int8")
    for t in a.xcd:
        if f"xcd{t}" in outs and "int8" in outs:
            print(f"XCD blocks of {t} token tiles bit-identical to the default map:", bool(torch.equal(outs[f"xcd{t}"], outs["int8"])))
    if "bf16" in outs and i8n in outs:
        d = (outs[i8n].float() - outs["bf16"].float()).norm() / outs["bf16"].float().norm()
        print(f"int8 vs bf16 route: rel-L2 {float(d):.2e}, differing fp16 elements {float((outs[i8n] != outs['bf16']).float().mean()):.2e}")
    if a.amax and not a.rows:
        print("in-GEMM exchange / segment partials / atomic cells bit-identical:",
              bool(torch.equal(outs["i8xch"], outs["i8at"]) and torch.equal(outs["i8xch"], outs["i8pa"])))
    elif a.rows and a.amax:
        print("in-GEMM exchange / segment partials / atomic cells bit-identical:",
              bool(torch.equal(outs["i8r128"], outs["r128at"]) and torch.equal(outs["i8r128"], outs["r128pa"])))
    elif a.rows:
        print("128-row vs 256-row int8 tiles bit-identical:", bool(torch.equal(outs["i8r128"], outs["i8r256"])),
              " default tile rows:", L.lqer_gemm_tile_rows(C.byref(mod._desc()), M, _lib.F16))
    fl = 2.0 * M * K * N + 2.0 * M * r * N
    for what, fn in (("GEMM alone (incl. B_out pre-pass)", lambda rt: gemm(rt)), ("whole forward", lambda rt: (quant(rt), gemm(rt)))):
        times = {k: [] for k in routes}
        for _ in range(a.rounds):
            for name, rt in routes.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    fn(rt)
                e1.record()
                torch.cuda.synchronize()
                times[name].append(e0.elapsed_time(e1) / a.iters * 1e3)
        for name in routes:
            t = sorted(times[name])
            med = t[len(t) // 2]
            print(f"{what:36s} {name:5s} median {med:9.1f} us  min {t[0]:9.1f} us   {fl / med / 1e6:8.1f} T(FL)OP/s")


if __name__ == "__main__":
    main()
