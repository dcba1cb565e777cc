#!/usr/bin/env python3
"""Randomised parity sweep: LinearFlexibleLqer forward on the GPU vs the CPU oracle over random shapes, ranks, dtypes
and quantizer configurations (MXINT blocks of 16, OPT-style bias blocks, the INT configuration, pass-through B_out,
the INT templates as shipped = pass-through activations on the fp16 or the bf16-limb route, no side path; round 4: 4-bit
`integer` weights, the int8 route with weight groups spread over several binades - every MODE of the int8 weight image -,
the GEMM summing the partial tiles of x A itself, and at M <= 8 the q/k/v group launch against its members; round 5: the int8
kernel from M = 128 with its tile height and pre-pass form pinned at random, weights of 5..8 bits on the limb route and - one
exponent per row and per token - on the int8 code image).  Shapes are
drawn to reach every GEMM kernel (small-M / one-launch decode, 128- and 64-row tiles, 256-row tiles bf16 and int8).
usage: python tools/fuzz_parity.py [cases] [seed]"""
import os, random, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lqer_amd
from bench import A16_Q, INT_Q, INTROW_Q, MXINT_Q, OPT_Q, make_case
from oracle import lqer_oracle as O


def one_case(rng):
    kind = rng.choice(["small", "small", "tile", "tile", "tile", "m256", "i8", "i8", "w8", "acttile"])
    if kind == "acttile":  # activation blocks that span token rows (the module's tile route): M is batch x tokens of a 3-D tensor
        return (rng.choice([6, 20, 35, 64]), rng.choice([64, 100, 176, 320]), rng.choice([16, 40, 160, 300]), rng.choice([0, 8, 16, 32]),
                rng.choice(["at4x16", "at3x32", "atallx16", "at2d8x16", "atbout"]), rng.choice([torch.float16, torch.bfloat16, torch.float32]))
    if kind == "i8":  # the int8 tile kernel: M >= 128, per-token x, weight blocks of 128 (or whole rows) whose exponents spread
        return (rng.choice([128, 130, 200, 256, 384, 512, 768, 1024, 1300]), rng.choice([128, 256, 384, 640, 1024]), rng.choice([256, 300, 512, 1024, 1296]),
                rng.choice([0, 16, 32, 64]), rng.choice(["i8spread0", "i8spread2", "i8spread4", "i8spread7", "i8row", "w8row", "w8row"]),
                rng.choice([torch.float16, torch.float16, torch.bfloat16]))
    if kind == "w8":  # weights of 5..8 bits: any M (decode, small-M, tiles), blocks of 16 / 32 / 128 / the row, MXINT or per-token x
        return (rng.choice([1, 5, 16, 40, 100, 129, 300, 520]), rng.choice([64, 100, 256, 320, 520]), rng.choice([16, 160, 256, 300, 1024]),
                rng.choice([0, 16, 32, 64]), rng.choice(["w8b16", "w6b32", "w5b16", "w7g128", "w8row", "w8opt"]),
                rng.choice([torch.float16, torch.bfloat16, torch.float32]))
    if kind == "small":
        M = rng.randint(1, 64)
    elif kind == "tile":
        M = rng.randint(65, 700)
    else:
        M = rng.choice([2048, 2304, 4096])
    K = rng.choice([16, 48, 64, 100, 176, 256, 320, 520, 1000])
    N = rng.choice([16, 40, 160, 256, 300, 1024, 1500]) if kind != "m256" else rng.choice([8192, 16384 + 256])
    r = rng.choice([0, 8, 16, 32, 48, 64, 96, 128])
    cfgname = rng.choice(["mxint", "opt", "int", "bout_pass", "a16", "a16", "a16mix", "tile", "intw", "xa_in_gemm", "group"])
    dtype = rng.choice([torch.float16, torch.float16, torch.bfloat16, torch.float32])
    if kind == "m256":
        K = rng.choice([64, 128, 200, 320, 520, 1000])
        r = rng.choice([0, 16, 32, 64, 96, 128])
        dtype = torch.float16
    return M, K, N, r, cfgname, dtype


def run_case(M, K, N, r, cfgname, dtype, dev):
    i8 = cfgname.startswith("i8")
    w8 = lambda width, block, base: dict(base, w_quantizer=dict(base["w_quantizer"], width=width, block_size=block))
    if cfgname == "group":
        return run_group(M, K, N, r, dtype, dev)
    qc = {"i8spread0": INT_Q, "i8spread2": INT_Q, "i8spread4": INT_Q, "i8spread7": INT_Q, "i8row": INTROW_Q, "xa_in_gemm": MXINT_Q,
          "intw": dict(MXINT_Q, w_quantizer=dict(name="integer", width=4, frac_width=1 + (M + K) % 4, is_signed=True)),
          "w8b16": w8(8, [1, 16], MXINT_Q), "w6b32": w8(6, [1, 32], MXINT_Q), "w5b16": w8(5, [1, 16], MXINT_Q), "w7g128": w8(7, [1, 128], INT_Q),
          "w8row": w8(8, [1, -1], INT_Q), "w8opt": w8(8, [1, 16], OPT_Q),
          "at4x16": dict(MXINT_Q, x_quantizer=dict(MXINT_Q["x_quantizer"], block_size=[4, 16])),
          "at3x32": dict(MXINT_Q, x_quantizer=dict(MXINT_Q["x_quantizer"], block_size=[3, 32])),
          "atallx16": dict(MXINT_Q, x_quantizer=dict(MXINT_Q["x_quantizer"], block_size=[-1, 16])),
          "at2d8x16": dict(MXINT_Q, x_quantizer=dict(MXINT_Q["x_quantizer"], block_size=[8, 16], skip_first_dim=False)),
          "atbout": dict(MXINT_Q, B_out_quantizer=dict(MXINT_Q["x_quantizer"], block_size=[2, 16])),
          "mxint": MXINT_Q, "opt": OPT_Q, "int": INT_Q, "bout_pass": dict(MXINT_Q, B_out_quantizer={"name": "passthrough"}),
          "a16": A16_Q, "a16mix": dict(A16_Q, B_out_quantizer=MXINT_Q["x_quantizer"]),
          "tile": dict(MXINT_Q, w_quantizer=dict(MXINT_Q["w_quantizer"], block_size=[(M % 3 + 1) * 4, 16 * (K % 2 + 1)]))}[cfgname]
    bias = cfgname in ("opt", "w8opt")
    case = make_case(M, K, N, max(r, 1), seed=M * 7919 + K * 31 + N, bias=bias,
                     quantize_ab=cfgname not in ("int", "a16", "a16mix", "w7g128", "w8row") and not i8)
    x, W, A, B = case[:4]
    if cfgname.startswith("i8spread"):  # per (row, 128-k group) scales 2^[0..spread]: PRESHIFT1 / PRESHIFT / FOLD tiles; some rows plain
        spread = int(cfgname[-1])
        g = torch.Generator().manual_seed(M + K + N)
        e = torch.randint(0, spread + 1, (N, (K + 127) // 128), generator=g)
        e[: N // 3] = 0
        W = W * torch.pow(2.0, -e.float()).repeat_interleave(128, dim=1)[:, :K]
    if cfgname == "intw":
        W = W * 40.0  # (fixed point: the codes must leave zero)
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
    from lqer_amd import _lib
    if cfgname == "xa_in_gemm":
        mod.tuning = _lib.TUNE_XA_REDUCE_IN_GEMM | (_lib.TUNE_TILE_ROWS_128 if M % 2 else 0)
    if i8 or cfgname == "w8row":  # the int8 kernel's tile height and the form of its B_out pre-pass, pinned at random (same bits)
        pick = (M * 31 + K * 7 + N) % 6
        mod.tuning = [0, _lib.TUNE_I8_ROWS_128, _lib.TUNE_I8_ROWS_256, _lib.TUNE_AMAX_ATOMIC, _lib.TUNE_AMAX_PARTS | _lib.TUNE_I8_ROWS_128,
                      _lib.TUNE_AMAX_PARTS][pick]
    xin = x.to(dtype)
    if cfgname.startswith("at") and cfgname != "at2d8x16" and M % 2 == 0:
        xin = xin.reshape(2, M // 2, K)  # [batch, tokens, features]: the tiles are anchored per batch element
    y = mod(xin.to(dev)).float().cpu()
    cast = lambda t: None if t is None else t.to(dtype).float()
    ref = O.lqer_linear_forward(xin.float(), cast(W), cast(b), cast(sd.get("A")) if r else None, cast(sd.get("B")) if r else None, qcm)
    err = float((y - ref).norm() / ref.norm().clamp_min(1e-30))
    tol = {torch.float16: 1e-3, torch.bfloat16: 6e-3, torch.float32: 3e-5}[dtype]
    if cfgname.startswith("at") and dtype != torch.float32:
        tol *= 2  # the tile route rounds x A and (x A) B to the module's dtype between the quantizers, as the reference does
    return err, tol


def run_group(M, K, N, r, dtype, dev):
    """q/k/v handed the same tokens at a decode size: the group launch against the members one by one (bit-identical), and
    member 0 against the oracle."""
    from lqer_amd.linear import SharedActivation

    M = 1 + M % 8
    K = max(64, K // 16 * 16)
    r = r if 0 < r <= 32 else 16
    dtype = torch.float16 if dtype == torch.float32 else dtype
    mods, cases = [], []
    for i, n in enumerate((N, max(16, N // 2), N)):
        case = make_case(M, K, n, r, seed=M * 7919 + K * 31 + n + i)
        x, W, A, B = case[:4]
        m = lqer_amd.LinearFlexibleLqer(K, n, bias=False, q_config=MXINT_Q, l_config={"rank": r})
        m.load_state_dict({"weight": W, "A": A, "B": B})
        mods.append(m.to(dev).to(dtype))
        cases.append((W, A, B))
    x = make_case(M, K, 16, r, seed=5)[0].to(dtype)
    xd = x.to(dev)
    alone = [m(xd).clone() for m in mods]
    SharedActivation(mods)
    together = [m(xd).clone() for m in mods]
    for a, t in zip(alone, together):
        if not torch.equal(a, t):
            return float("inf"), 0.0
    cast = lambda t: t.to(dtype).float()
    W, A, B = cases[0]
    ref = O.lqer_linear_forward(x.float(), cast(W), None, cast(A), cast(B), MXINT_Q)
    err = float((together[0].float().cpu() - ref).norm() / ref.norm().clamp_min(1e-30))
    return err, {torch.float16: 1e-3, torch.bfloat16: 6e-3}[dtype]


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
