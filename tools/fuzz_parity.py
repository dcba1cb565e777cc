#!/usr/bin/env python3
"""Randomised parity sweep: LinearFlexibleLqer forward on the GPU vs the CPU oracle over random shapes, ranks, dtypes
and quantizer configurations (MXINT blocks of 16, OPT-style bias blocks, the INT configuration, pass-through B_out,
the INT templates as shipped = pass-through activations on the fp16 or the bf16-limb route, no side path).  Shapes are drawn to reach all three GEMM kernels (small-M, 128-row tiles, 256-row tiles).
usage: python tools/fuzz_parity.py [cases] [seed]"""
import os, random, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lqer_amd
from bench import A16_Q, INT_Q, MXINT_Q, OPT_Q, make_case
from oracle import lqer_oracle as O


def one_case(rng):
    kind = rng.choice(["small", "small", "tile", "tile", "tile", "m256"])
    if kind == "small":
        M = rng.randint(1, 64)
    elif kind == "tile":
        M = rng.randint(65, 700)
    else:
        M = rng.choice([2048, 2304, 4096])
    K = rng.choice([16, 48, 64, 100, 176, 256, 320, 520, 1000])
    N = rng.choice([16, 40, 160, 256, 300, 1024, 1500]) if kind != "m256" else rng.choice([8192, 16384 + 256])
    r = rng.choice([0, 8, 16, 32, 48, 64, 96, 128])
    cfgname = rng.choice(["mxint", "opt", "int", "bout_pass", "a16", "a16", "a16mix", "tile"])
    dtype = rng.choice([torch.float16, torch.float16, torch.bfloat16, torch.float32])
    if kind == "m256":
        K = rng.choice([64, 128, 200, 320, 520, 1000])
        r = rng.choice([0, 16, 32, 64, 96, 128])
        dtype = torch.float16
    return M, K, N, r, cfgname, dtype


def run_case(M, K, N, r, cfgname, dtype, dev):
    qc = {"mxint": MXINT_Q, "opt": OPT_Q, "int": INT_Q, "bout_pass": dict(MXINT_Q, B_out_quantizer={"name": "passthrough"}),
          "a16": A16_Q, "a16mix": dict(A16_Q, B_out_quantizer=MXINT_Q["x_quantizer"]),
          "tile": dict(MXINT_Q, w_quantizer=dict(MXINT_Q["w_quantizer"], block_size=[(M % 3 + 1) * 4, 16 * (K % 2 + 1)]))}[cfgname]
    bias = cfgname == "opt"
    case = make_case(M, K, N, max(r, 1), seed=M * 7919 + K * 31 + N, bias=bias, quantize_ab=cfgname not in ("int", "a16", "a16mix"))
    x, W, A, B = case[:4]
    b = case[4] if bias else None
    if r == 0:
        cls, lc, qcm = lqer_amd.LinearFlexible, None, dict(qc, name="flexible")
    else:
        cls, lc, qcm = lqer_amd.LinearFlexibleLqer, {"rank": r}, qc
    mod = cls(K, N, bias=bias, q_config=qcm, l_config=lc)
    sd = {"weight": W}
    if r:
        sd["A"], sd["B"] = A[:, :r].contiguous(), B[:r].contiguous()
    if bias:
        sd["bias"] = b
    mod.load_state_dict(sd)
    mod = mod.to(dev).to(dtype)
    mod.a16_native = (M + K + N) % 3 != 0  # pass-through fp16 activations: mostly the fp16 route, sometimes bf16 limbs
    xin = x.to(dtype)
    y = mod(xin.to(dev)).float().cpu()
    cast = lambda t: None if t is None else t.to(dtype).float()
    ref = O.lqer_linear_forward(xin.float(), cast(W), cast(b), cast(sd.get("A")) if r else None, cast(sd.get("B")) if r else None, qcm)
    err = float((y - ref).norm() / ref.norm().clamp_min(1e-30))
    tol = {torch.float16: 1e-3, torch.bfloat16: 6e-3, torch.float32: 3e-5}[dtype]
    return err, tol


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = random.Random(seed)
    dev = torch.device("cuda:0")
    bad = 0
    for i in range(n):
        c = one_case(rng)
        err, tol = run_case(*c, dev)
        flag = "" if err <= tol else "   <-- FAIL"
        bad += err > tol
        print(f"{i:3d} M={c[0]:5d} K={c[1]:5d} N={c[2]:6d} r={c[3]:3d} {c[4]:9s} {str(c[5])[6:]:9s} rel-L2 {err:.2e} (tol {tol:.0e}){flag}")
    print(f"{n - bad} / {n} within tolerance")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
