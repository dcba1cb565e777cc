"""Generate the golden vectors under tests/golden/ by importing the REFERENCE (read-only, at
/root/reference) in the build container.  Run once, here:  python tests/golden/make_golden.py

Only data is written (inputs + the reference's outputs, as .npz).  No reference source travels.
The tests that consume these files never touch /root/reference.

The reference package needs `colorlog` (absent here) at import time only for its logger; a stub
module is injected.  `lqer.quantize` is the only sub-package imported.
"""
import logging
import os
import zlib
import sys
import types
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    stub = types.ModuleType("colorlog")

    class ColoredFormatter(logging.Formatter):
        def __init__(self, *a, **k):
            super().__init__()

    stub.ColoredFormatter = ColoredFormatter
    sys.modules.setdefault("colorlog", stub)
    sys.path.insert(0, "/root/reference/src")
    from lqer.quantize import get_quantized_layer_cls, get_quantizer  # noqa

    return get_quantized_layer_cls, get_quantizer


def bfp_cfg(width, block, skip):
    return dict(name="block_fp", width=width, exponent_width=8, exponent_bias=None, block_size=block, skip_first_dim=skip)


def f32_from_bits(bits):
    return np.array(bits, dtype=np.uint32).view(np.float32)


def main():
    get_cls, get_q = import_reference()
    bfp = get_q("block_fp")
    integer = get_q("integer")
    torch.manual_seed(1234)
    g = {}

    # ------------------------------------------------------------------ quantizer vectors
    def add_q(name, x, width, block, skip):
        y = bfp(x.clone(), width=width, exponent_width=8, exponent_bias=None, block_size=block, skip_first_dim=skip)
        g[f"q/{name}/x"] = x.numpy()
        g[f"q/{name}/y"] = y.numpy()
        g[f"q/{name}/meta"] = np.array([width, int(skip)] + list(block), dtype=np.int64)

    def outliers(t, cols=(7,)):
        t = t.clone()
        for c in cols:
            if c < t.shape[-1]:
                t[..., c] *= 30.0
        return t

    add_q("w4_b16", 0.02 * torch.randn(48, 80), 4, [1, 16], False)
    add_q("w4_b128", 0.02 * torch.randn(16, 384), 4, [1, 128], False)
    add_q("w4_row", 0.02 * torch.randn(16, 200), 4, [1, -1], False)
    add_q("w4_ragged", 0.02 * torch.randn(9, 50), 4, [1, 16], False)
    add_q("w2_b32", 0.02 * torch.randn(8, 96), 2, [1, 32], False)
    add_q("w4_tile2d", 0.02 * torch.randn(12, 40), 4, [4, 8], False)
    add_q("x8_2d", outliers(torch.randn(7, 176), (7, 100)), 8, [1, 16], True)
    add_q("x8_3d", outliers(torch.randn(2, 5, 176), (7, 100)), 8, [1, 16], True)
    add_q("x8_ragged", torch.randn(3, 50), 8, [1, 16], True)
    add_q("x8_row", outliers(torch.randn(5, 200)), 8, [1, -1], True)
    add_q("x4_2d", torch.randn(4, 64), 4, [1, 16], True)
    add_q("ab_col", 0.05 * torch.randn(64, 32), 8, [16, 1], False)
    add_q("b_row", 0.05 * torch.randn(32, 160), 8, [16, 1], False)
    add_q("bias_all", 0.01 * torch.randn(160), 8, [-1], False)
    add_q("bias_16", 0.01 * torch.randn(50), 8, [1, 16], False)
    add_q("x16_2d", torch.randn(4, 64), 16, [1, 16], True)

    # edge cases: powers of two and their fp32 neighbours as block maxima,
    # rounding ties, zero blocks, |x| <= 1e-8 pass-through, +1e-9 effect on tiny values
    rows = []
    for k in (-20, -10, -6, -5, -3, -1, 0, 1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 40):
        base = int(f32_from_bits([0])[0].view(np.uint32)) + ((k + 127) << 23)
        for j in (-1, 0, 1, 2, 3, 5, 6, 11, 12, 22, 23, 100):
            top = f32_from_bits([base + j])[0]
            r = np.zeros(16, dtype=np.float32)
            r[0] = top
            r[1] = -top * np.float32(0.5)
            r[2] = top * np.float32(0.75)
            r[3:11] = top * (np.arange(1, 9, dtype=np.float32) + np.float32(0.5)) / np.float32(128.0)  # .5 ties
            r[11] = 1e-9
            r[12] = -1e-9
            r[13] = 5e-9
            r[14] = -2e-8
            rows.append(r)
    edge = torch.from_numpy(np.stack(rows))
    add_q("edge_pow2_w8", edge, 8, [1, 16], True)
    add_q("edge_pow2_w4", edge, 4, [1, 16], False)
    z = torch.randn(6, 64)
    z[1] = 0.0
    z[2, 16:32] = 0.0
    z[3, :16] = 1e-12
    z[4] = torch.tensor([3e-9, -4e-9] * 32)
    z[5, 5] = 1e30
    add_q("zeros_tiny", z, 8, [1, 16], True)
    add_q("all_zero", torch.zeros(3, 32), 8, [1, 16], True)
    add_q("all_zero_bias", torch.zeros(24), 8, [-1], False)
    small = (torch.randint(0, 16, (64, 16)).float() + 0.5) * (2.0**-10)  # exact ties, |x| < 2^-5
    small[:, 0] = 2.0**-3
    add_q("ties_small", small, 8, [1, 16], True)
    add_q("ties_small_w4", small * 1.0, 4, [1, 16], False)

    # integer quantizer
    xi = torch.randn(5, 33) * 3
    for (w, f) in ((8, 4), (16, 9), (4, 1)):
        g[f"int/w{w}f{f}/x"] = xi.numpy()
        g[f"int/w{w}f{f}/y"] = integer(xi.clone(), w, f, True).numpy()

    # 2-D weight tiles (quantizers/utils.py:161-183; round 3 - drawn after every other vector so that those do not change):
    # tiles of R rows x L k, ragged in both dims, and a lone [L] with skip_first_dim = false = all rows x L
    add_q("w4_tile_8x16", 0.02 * torch.randn(40, 80), 4, [8, 16], False)
    add_q("w4_tile_16x32_ragged", 0.02 * torch.randn(25, 72), 4, [16, 32], False)
    add_q("w4_tile_allrows_16", 0.02 * torch.randn(20, 64), 4, [16], False)
    add_q("w4_tile_4xrow", 0.02 * torch.randn(12, 48), 4, [4, -1], False)

    # the ceil(log2) rule: for every k, the number of ulps above 2^k that still give k
    thr = []
    for k in range(-126, 128):
        bits = np.uint32((k + 127) << 23) + np.arange(0, 64, dtype=np.uint32)
        vals = torch.from_numpy(np.tile(bits.view(np.float32), 8).copy())
        lv = torch.ceil(torch.log2(vals)).numpy()[:64]
        thr.append(int(np.max(np.where(lv == k)[0])))
    g["log2_rule/k"] = np.arange(-126, 128)
    g["log2_rule/slack"] = np.array(thr)
    np.savez_compressed(os.path.join(HERE, "quantizers.npz"), **g)
    print("quantizers.npz:", len(g), "arrays")

    # ------------------------------------------------------------------ full-forward vectors
    def lowrank(W, wcfg, abcfg, r, scale=None):
        """A,B as the reference's approximators make them: truncated SVD of (W - Q(W))^T
        (approximate/lqer_svd.py:37-47; with activation scale s: lqer_act.py:84-97)."""
        wq = get_q(wcfg["name"])(W.clone(), **{k: v for k, v in wcfg.items() if k != "name"}) if wcfg["name"] != "passthrough" else W
        E = (W - wq).t().double()
        if scale is not None:
            E = scale.double()[:, None] * E
        U, S, Vh = torch.linalg.svd(E, full_matrices=False)
        A = U[:, :r]
        Bm = S[:r, None] * Vh[:r]
        if scale is not None:
            A = A / scale.double()[:, None]
        A, Bm = A.float(), Bm.float()
        if abcfg is not None:
            kw = {k: v for k, v in abcfg.items() if k != "name"}
            A, Bm = bfp(A, **kw), bfp(Bm, **kw)
        return A.contiguous(), Bm.contiguous()

    mxint_q = dict(
        name="flexible_lqer",
        is_ptq=True,
        default=False,
        x_quantizer=bfp_cfg(8, [1, 16], True),
        w_quantizer=bfp_cfg(4, [1, 16], False),
        b_quantizer=bfp_cfg(8, [-1], False),
    )
    opt_q = dict(mxint_q, b_quantizer=bfp_cfg(8, [1, 16], False))
    int_q = dict(
        name="flexible_lqer",
        is_ptq=True,
        default=False,
        x_quantizer=bfp_cfg(8, [1, -1], True),
        w_quantizer=bfp_cfg(4, [1, 128], False),
        b_quantizer=dict(name="passthrough"),
    )
    introw_q = dict(int_q, w_quantizer=bfp_cfg(4, [1, -1], False))
    # the *-int.toml templates as shipped: pass-through ("A16") activations, A_out / B_out falling back to them
    # (experiments/configs/template/llama-7b-int.toml, opt-6.7b-int.toml)
    a16_q = dict(int_q, x_quantizer=dict(name="passthrough", width=16, frac_width=12))
    a16row_q = dict(a16_q, w_quantizer=bfp_cfg(4, [1, -1], False))
    a16mix_q = dict(a16_q, B_out_quantizer=bfp_cfg(8, [1, 16], True))  # pass-through x and A_out, block_fp B_out
    abq = dict(name="block_fp", width=8, exponent_width=8, exponent_bias=None, block_size=[16, 1], skip_first_dim=False)
    # fixed-point ("integer", quantizers/integer.py:10-43) activations: A_out and B_out fall back to the x quantizer's config
    # (linear.py:115-124), so the side product is re-quantized to the same 8-bit fixed-point grid twice (round 3)
    intx_q = dict(name="flexible_lqer", is_ptq=True, default=False, x_quantizer=dict(name="integer", width=8, frac_width=4),
                  w_quantizer=bfp_cfg(4, [1, 16], False), b_quantizer=dict(name="integer", width=8, frac_width=6))
    intx5_q = dict(intx_q, x_quantizer=dict(name="integer", width=8, frac_width=5))  # saturating outlier channels

    cases = [
        # name, x shape, K, N, r, bias, q_config, A/B quantizer, act-scale
        ("m1", (1, 64), 64, 64, 16, False, mxint_q, abq, False),
        ("m7", (7, 176), 176, 160, 32, False, mxint_q, abq, True),
        ("m64", (64, 512), 512, 160, 32, True, mxint_q, abq, True),
        ("b2s5", (2, 5, 176), 176, 64, 16, True, opt_q, abq, False),
        ("r128", (16, 512), 512, 160, 128, True, opt_q, abq, False),
        ("int128", (9, 256), 256, 96, 64, False, int_q, None, True),
        ("introw", (9, 256), 256, 96, 64, False, introw_q, None, False),
        ("ragged", (5, 72), 72, 40, 16, True, mxint_q, abq, False),
        ("a16", (9, 256), 256, 96, 64, False, a16_q, None, True),
        ("a16row", (2, 5, 176), 176, 64, 16, True, a16row_q, None, False),
        ("a16mix", (70, 128), 128, 160, 32, False, a16mix_q, None, False),
        ("intx", (12, 192), 192, 112, 32, True, intx_q, abq, False),
        ("intx70", (2, 35, 128), 128, 160, 16, False, intx5_q, None, True),
        # 2-D weight tiles (round 3)
        ("tile8", (7, 176), 176, 160, 32, True, dict(mxint_q, w_quantizer=bfp_cfg(4, [8, 16], False)), abq, False),
        ("tileall", (70, 128), 128, 96, 16, False, dict(mxint_q, w_quantizer=bfp_cfg(4, [32], False)), abq, False),
        # fixed-point WEIGHTS (round 4): 4-bit `integer`, frac_width 7 - sigma 0.02 is 2.56 steps, both clamps (-8, +7) occur
        ("intw", (12, 192), 192, 112, 32, True, dict(mxint_q, w_quantizer=dict(name="integer", width=4, frac_width=7)), abq, False),
        ("intxw", (2, 35, 128), 128, 160, 16, False, dict(intx5_q, w_quantizer=dict(name="integer", width=4, frac_width=7)), None, True),
        # weights of 5..8 bits (round 5): the reference's W8A8 formats (sweep_baseline_no_lqer.sh:73-76: block_fp width 8, one block
        # per row / per token) with a side path, blocks of 128, blocks of 16 beside block-16 activations, and a 6-bit weight
        ("w8row", (9, 256), 256, 96, 32, True, dict(int_q, w_quantizer=bfp_cfg(8, [1, -1], False), b_quantizer=bfp_cfg(8, [1, -1], False)), abq, False),
        ("w8g128", (9, 256), 256, 96, 32, False, dict(int_q, w_quantizer=bfp_cfg(8, [1, 128], False)), None, True),
        ("w8b16", (7, 176), 176, 160, 32, True, dict(mxint_q, w_quantizer=bfp_cfg(8, [1, 16], False)), abq, False),
        ("w6b32", (2, 5, 192), 192, 64, 16, False, dict(mxint_q, w_quantizer=bfp_cfg(6, [1, 32], False)), abq, True),
        # activation blocks that span token rows (round 5; quantizers/utils.py:211-237 for 3-D tensors, :161-183 / :261-270 for 2-D
        # tensors with skip_first_dim = false): A_out / B_out fall back to the x quantizer, so all three are tiled
        ("acttile3d", (2, 20, 176), 176, 64, 16, True, dict(mxint_q, x_quantizer=bfp_cfg(8, [4, 16], True)), abq, False),
        ("acttile_ragged", (3, 7, 72), 72, 40, 16, False, dict(mxint_q, x_quantizer=bfp_cfg(8, [3, 32], True)), abq, True),
        ("acttile2d", (24, 128), 128, 96, 32, False, dict(mxint_q, x_quantizer=bfp_cfg(8, [8, 16], False)), abq, False),
        ("acttile2d_all", (10, 64), 64, 48, 16, True, dict(mxint_q, x_quantizer=bfp_cfg(8, [32], False)), abq, False),
        ("acttile_r_on_2d", (9, 64), 64, 48, 16, False, dict(mxint_q, x_quantizer=bfp_cfg(8, [4, 16], True)), abq, False),  # R is read per row
        ("acttile_whole", (2, 6, 64), 64, 48, 16, False, dict(mxint_q, x_quantizer=bfp_cfg(8, [-1, -1], True)), abq, False),  # one exponent per batch element
        ("acttile_bout", (2, 6, 64), 64, 48, 16, False, dict(mxint_q, B_out_quantizer=bfp_cfg(8, [2, 16], True)), abq, False),  # only B_out tiled
        # (round 6) the quantizer's DEFAULT block_size - a lone [16] with skip_first_dim = true - on a 3-D tensor: right-aligned it reads
        # [1, T, 16], one exponent per batch element, ALL T token rows and 16 columns (utils.py:56-66, :211-237); on a 2-D tensor: per row
        ("acttile_lone3d", (2, 20, 176), 176, 64, 16, True, dict(mxint_q, x_quantizer=bfp_cfg(8, [16], True)), abq, False),
        ("acttile_lone2d", (9, 64), 64, 48, 16, False, dict(mxint_q, x_quantizer=bfp_cfg(8, [16], True)), abq, False),
        # (round 6) the reference's weight-only sweep (experiments/pipeline/sweep_lqer_act_w-only.sh:74-77, the paper's "W3A16" row): 3-bit
        # weights in blocks of [1, 32], pass-through activations / bias / A / B, rank 64
        ("w3b32_a16_r64", (9, 256), 256, 96, 64, False, dict(a16_q, w_quantizer=bfp_cfg(3, [1, 32], False)), None, True),
        ("w3b32_a16_r64_3d", (2, 5, 192), 192, 160, 64, True, dict(a16_q, w_quantizer=bfp_cfg(3, [1, 32], False)), None, False),
    ]
    f = {}
    for name, xs, K, N, r, has_b, qc, abc, use_s in cases:
        torch.manual_seed(zlib.crc32(name.encode()) % 10000)
        x = outliers(torch.randn(*xs), (7, 33))
        W = 0.02 * torch.randn(N, K)
        bias = 0.01 * torch.randn(N) if has_b else None
        s = None
        if use_s:
            s = x.reshape(-1, K).abs().mean(0)
            s = s / torch.sqrt(s.min() * s.max())  # statistic_profiler/scale.py:44-51 style normalisation
        A, Bm = lowrank(W, qc["w_quantizer"], abc, r, s)
        cls = get_cls("linear", qc)
        mod = cls(K, N, bias=has_b, q_config=qc, l_config={"rank": r})
        with torch.no_grad():
            mod.weight.copy_(W)
            if has_b:
                mod.bias.copy_(bias)
            mod.A.copy_(A)
            mod.B.copy_(Bm)
            y = mod(x)
            xq = mod.x_quantizer(x)
            xA = torch.matmul(xq, mod.A)
            xAq = mod.A_out_quantizer(xA)
            xAB = torch.matmul(xAq, mod.B)
            xABq = mod.B_out_quantizer(xAB)
        f[f"{name}/x"] = x.numpy()
        f[f"{name}/W"] = W.numpy()
        f[f"{name}/A"] = A.numpy()
        f[f"{name}/B"] = Bm.numpy()
        if has_b:
            f[f"{name}/bias"] = bias.numpy()
            f[f"{name}/bq"] = mod.bias.detach().numpy()
        f[f"{name}/wq"] = mod.weight.detach().numpy()
        for k_, v_ in (("xq", xq), ("xA", xA), ("xAq", xAq), ("xAB", xAB), ("xABq", xABq), ("y", y)):
            f[f"{name}/{k_}"] = v_.numpy()
        f[f"{name}/rank"] = np.array([r])
    # LinearFlexible (no side path), reference linear.py:50-59
    qc = dict(mxint_q, name="flexible")
    cls = get_cls("linear", qc)
    torch.manual_seed(77)
    x = torch.randn(6, 96)
    mod = cls(96, 48, bias=True, q_config=qc, l_config=None)
    W, b = mod.weight.detach().clone(), mod.bias.detach().clone()
    with torch.no_grad():
        y = mod(x)
    f["flex/x"], f["flex/W"], f["flex/bias"], f["flex/y"] = x.numpy(), W.numpy(), b.numpy(), y.numpy()
    # LinearFlexible in the configuration the reference runs it with (round 5): W8A8, one block per weight row and per token, the
    # bias in the activations' format (experiments/pipeline/sweep_baseline_no_lqer.sh:50-58, :73-76); and with weight blocks of 128
    w8a8 = dict(name="flexible", is_ptq=True, default=False, x_quantizer=bfp_cfg(8, [1, -1], True), w_quantizer=bfp_cfg(8, [1, -1], False),
                b_quantizer=bfp_cfg(8, [1, -1], False))
    flex_cfgs = {"flex": qc}
    for fname, fq, shape, K_, N_, seed in (("flex_w8a8_row", w8a8, (2, 9, 320), 320, 144, 78),
                                           ("flex_w8a8_g128", dict(w8a8, w_quantizer=bfp_cfg(8, [1, 128], False)), (11, 256), 256, 96, 79)):
        cls = get_cls("linear", fq)
        torch.manual_seed(seed)
        x = outliers(torch.randn(*shape), (7, 33))
        mod = cls(K_, N_, bias=True, q_config=fq, l_config=None)
        with torch.no_grad():
            mod.weight.mul_(3.0)  # (the default init is +-1/sqrt(K): spread it over more binades)
            mod.bias.mul_(0.5)
        W, b = mod.weight.detach().clone(), mod.bias.detach().clone()
        with torch.no_grad():
            y = mod(x)
        f[f"{fname}/x"], f[f"{fname}/W"], f[f"{fname}/bias"], f[f"{fname}/y"] = x.numpy(), W.numpy(), b.numpy(), y.numpy()
        f[f"{fname}/wq"], f[f"{fname}/bq"] = mod.weight.detach().numpy(), mod.bias.detach().numpy()
        flex_cfgs[fname] = fq
    np.savez_compressed(os.path.join(HERE, "forward.npz"), **f)
    import json

    cfgs = {c[0]: c[6] for c in cases}
    cfgs.update(flex_cfgs)
    with open(os.path.join(HERE, "forward_configs.json"), "w") as fh:
        json.dump(cfgs, fh, indent=1)
    print("forward.npz:", len(f), "arrays")


if __name__ == "__main__":
    main()
