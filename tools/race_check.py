#!/usr/bin/env python3
"""Run the same forward repeatedly and report where outputs differ between runs (race screen)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lqer_amd
from bench import make_case, A16_Q, INT_Q, MXINT_Q
# usage: race_check.py [K [N [M [rank [mxint|int|a16]]]]]   (rank > 32 or the int / a16 configurations: staged side-path prologue)
arg = lambda i, d: type(d)(sys.argv[i]) if len(sys.argv) > i else d
K = arg(1, 4096)
N, M, r, cfg = arg(2, K), arg(3, 2048), arg(4, 32), arg(5, "mxint")
qc = {"mxint": MXINT_Q, "int": INT_Q, "a16": A16_Q}[cfg]
x, W, A, B = make_case(M, K, N, r, seed=1, quantize_ab=cfg == "mxint")
x = x.half().cuda()
mod = lqer_amd.LinearFlexibleLqer(K, N, bias=False, q_config=qc, l_config={"rank": r})
mod.load_state_dict({"weight": W, "A": A, "B": B})
mod = mod.cuda().half()
y0 = mod(x).clone()
bad = 0
for it in range(30):
    y = mod(x)
    d = (y != y0)
    if d.any():
        bad += 1
        idx = d.nonzero()
        rows = idx[:, 0].unique()
        cols = idx[:, 1].unique()
        print(f"run {it}: {int(d.sum())} diffs; rows {rows[:8].tolist()}..(n={len(rows)}) cols {cols[:8].tolist()}..(n={len(cols)}) maxabs {float((y.float()-y0.float()).abs().max()):.4g}")
print("runs with differences:", bad, "/ 30")
perm = torch.randperm(M, device="cuda")
yp = mod(x[perm])
d = yp != y0[perm]
print("perm diffs:", int(d.sum()))
if d.any():
    idx = d.nonzero()
    print(" rows(perm pos)", idx[:, 0].unique()[:16].tolist(), " cols", idx[:, 1].unique()[:16].tolist())
# detailed pattern of one differing run
for it in range(5):
    y = mod(x)
    d = (y != y0)
    if d.any():
        idx = d.nonzero().cpu()
        import collections
        tiles = collections.Counter((int(r) // 128, int(c) // 256) for r, c in idx.tolist())
        print("tiles (tm,tn):", dict(tiles))
        (tm, tn), _ = tiles.most_common(1)[0]
        sub = d[tm * 128:(tm + 1) * 128, tn * 256:(tn + 1) * 256].cpu()
        print("rows in tile:", sub.any(1).nonzero().flatten().tolist())
        print("cols in tile:", sub.any(0).nonzero().flatten().tolist()[:40], "... n=", int(sub.any(0).sum()))
        break
