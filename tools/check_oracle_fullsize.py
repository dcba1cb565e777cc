#!/usr/bin/env python3
"""Full-size pin of the CPU oracle against the REFERENCE itself ("check the build's CPU restatement ...
at 4096^2 before it becomes the on-box comparator").  BUILD CONTAINER ONLY: imports `lqer.quantize` from /root/reference
(the way tests/golden/make_golden.py does) - nothing here travels to or runs on the GPU box, and neither the product nor
the GPU tests import this file.

    python tools/check_oracle_fullsize.py [--quick]

For every BASELINE configuration's first projection shape (C2/C3 4096^2 rank 32 MXINT, C4 5120^2 rank 64 both INT
readings, C5 4096^2 rank 128 with the OPT bias format) at M = 256 tokens of the bench's synthetic operands:
x_quantizer(x) and w_quantizer(W) bit-equal, forward rel-L2 <= 1e-6 (the two sides differ at most in the summation order
of torch's own matmul - in practice 0.0).  Exit code 0 = all equal.  Last run (round 3): see the table it prints.
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="C2 only")
    ap.add_argument("--M", type=int, default=256)
    a = ap.parse_args()
    if not os.path.isdir("/root/reference/src/lqer"):
        sys.exit("the reference is only present in the build container (/root/reference)")
    from make_golden import import_reference  # the colorlog stub + sys.path entry, nothing else

    get_cls, _ = import_reference()
    from bench import A16_Q, INT_Q, INTROW_Q, MXINT_Q, OPT_Q, UNQUANTIZED_AB, make_case
    from oracle import lqer_oracle as O

    cases = [("c2/c3 4096x4096 r32 MXINT", 4096, 4096, 32, False, MXINT_Q)]
    if not a.quick:
        cases += [("c3 4096x11008 r32 MXINT", 4096, 11008, 32, False, MXINT_Q),
                  ("c4 5120x5120 r64 INT (W block 128, A8 per token)", 5120, 5120, 64, False, INT_Q),
                  ("c4row 5120x5120 r64 INT (one W block per row)", 5120, 5120, 64, False, INTROW_Q),
                  ("c4a16 5120x5120 r64 (pass-through activations)", 5120, 5120, 64, False, A16_Q),
                  ("c5 4096x4096 r128 MXINT + bias blocks of 16", 4096, 4096, 128, True, OPT_Q)]
    torch.set_num_threads(os.cpu_count() or 1)
    bad = 0
    for name, K, N, r, has_bias, qc in cases:
        t0 = time.time()
        ops = make_case(a.M, K, N, r, seed=0, bias=has_bias, quantize_ab=not any(qc is c for c in UNQUANTIZED_AB))
        x, W, A, B = ops[:4]
        bias = ops[4] if has_bias else None
        ref_mod = get_cls("linear", qc)(K, N, bias=has_bias, q_config=qc, l_config={"rank": r})
        sd = {"weight": W.clone(), "A": A.clone(), "B": B.clone()}
        if has_bias:
            sd["bias"] = bias.clone()
        ref_mod.load_state_dict(sd)
        with torch.no_grad():
            y_ref = ref_mod(x.clone())
            xq_ref = ref_mod.x_quantizer(x.clone())
        wq_ref = ref_mod.weight.detach()  # quantized in place by the first forward (linear.py:149-153)
        qs = O.resolve_linear_quantizers(qc)
        xq = O.get_quantizer(qs["x"])(x)
        wq = O.get_quantizer(qs["w"])(W)
        y = O.lqer_linear_forward(x, W, bias, A, B, qc)
        rel = float((y - y_ref).norm() / y_ref.norm())
        ok = torch.equal(xq, xq_ref) and torch.equal(wq, wq_ref) and rel <= 1e-6
        bad += not ok
        print(f"{name:55s} M={a.M}: xq bit-equal {torch.equal(xq, xq_ref)}, wq bit-equal {torch.equal(wq, wq_ref)}, "
              f"y rel-L2 {rel:.2e}  [{'ok' if ok else 'MISMATCH'}, {time.time() - t0:.0f} s]", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
