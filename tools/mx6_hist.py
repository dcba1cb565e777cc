#!/usr/bin/env python3
"""Exponent-pair histogram for the block-scaled MFMA question (DESIGN.md §4 "fp6 block-scaled MFMA", NOTEBOOK.md §8.4).

`v_mfma_scale_f32_32x32x64_f8f6f4` applies ONE E8M0 scale per 32 k; the MXINT formats of the path carry one exponent per
16 k (llama-7b.toml:82-97).  A pair of 16-blocks (e1, e2) fits one scale E = max(e1, e2) only if the block with the smaller
exponent can carry its offset d = E - e_i inside its fp6 elements:
  * activation mantissa m (7 bits) as two signed-digit limbs m = 16 hi + lo, lo in [-8, 7], hi in [0, 8]: every limb value is
    an integer of <= 3 significant bits, exact in e3m2 ("bf6") as n 2^(j-4) for j = 0..5  ->  d <= 5
    (as two plain 4-bit limbs in e2m3: d <= 2)
  * weight code c in [-7, 7]: e3m2, d <= 6 (e2m3: d <= 3)
This script counts, on the bench's synthetic operands (bench.make_x / make_weights: N(0,1) tokens with three x30 outlier
channels, N(0, 0.02^2) weights), how often d exceeds those limits, per (row, pair) and per 128-row tile x pair (the unit an
exception pass would work in).  CPU only; uses the oracle's exponent rule.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_weights, make_x  # noqa: E402
from oracle import lqer_oracle as O  # noqa: E402


def block_exps(t, block=16):
    R, C = t.shape
    amax = t.abs().reshape(R, C // block, block).amax(-1)
    e = O.ceil_log2_f32(amax.clamp_min(1e-30)).to(torch.int32)
    return torch.where(amax > 0, e, torch.full_like(e, -127)), amax


def report(name, t, limits):
    e, amax = block_exps(t)
    e1, e2 = e[:, 0::2], e[:, 1::2]
    live = (amax[:, 0::2] > 0) & (amax[:, 1::2] > 0)  # a zero block fits any scale
    d = (e1 - e2).abs()
    d = torch.where(live, d, torch.zeros_like(d))
    hist = torch.bincount(d.flatten().long(), minlength=12)[:12].tolist()
    tot = d.numel()
    print(f"{name}: {tuple(t.shape)}  (row, pair) count {tot}")
    print("  d histogram 0..11:", " ".join(f"{h / tot * 100:.3f}%" for h in hist))
    for lim, what in limits:
        bad = d > lim
        tiles = bad.reshape(-1, 128, bad.shape[1]).any(1) if bad.shape[0] % 128 == 0 else None
        print(f"  d > {lim} ({what}): {bad.float().mean() * 100:.4f}% of (row, pair)"
              + (f"; {tiles.float().mean() * 100:.3f}% of (128-row tile, pair) = {tiles.sum(1).float().mean():.1f} of {tiles.shape[1]} pairs per tile"
                 if tiles is not None else ""))


if __name__ == "__main__":
    M, K, N, r = 2048, 4096, 4096, 32
    x, g = make_x(M, K, seed=0)
    W = make_weights(g, K, N, 0)[0]
    report("x (bench tokens, fp16)", x.half().float(), [(2, "two 4-bit limbs in e2m3"), (5, "two signed-digit limbs in e3m2")])
    report("W (bench weights)", W.half().float(), [(3, "e2m3"), (6, "e3m2")])
    # a harsher activation: heavy-tailed (Student t, 3 dof) tokens with the same outlier channels
    xt = torch.distributions.StudentT(3.0).sample((M, K))
    for c in (7, 1033, 2900):
        xt[:, c] *= 30
    report("x (Student-t(3) tokens + outlier channels)", xt.half().float(), [(2, "e2m3"), (5, "e3m2")])
