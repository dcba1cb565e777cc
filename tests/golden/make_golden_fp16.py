"""fp16 evaluation of the reference's quantizers (the reference evaluates a model in its dtype, fp16:
/root/reference/src/lqer/runners.py:203): vectors that put a number on how often the HIP path - which upcasts to fp32 and
adds 1e-9f there (csrc/common.h) - differs from the reference's own half-precision arithmetic, where `+ 1e-9` is a no-op
and `2 ** e` is a half.  Run once, in the build container:  python tests/golden/make_golden_fp16.py
Only data is written (inputs + the reference's outputs)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference  # noqa: E402


def main():
    _, get_q = import_reference()
    bfp = get_q("block_fp")
    torch.manual_seed(4321)
    g = {}

    def add(name, x16, width, block, skip):
        y16 = bfp(x16.clone(), width=width, exponent_width=8, exponent_bias=None, block_size=block, skip_first_dim=skip)
        assert y16.dtype == torch.float16
        g[f"{name}/x"] = x16.numpy()
        g[f"{name}/y"] = y16.numpy()
        g[f"{name}/meta"] = np.array([width, int(skip)] + list(block), dtype=np.int64)

    # activations as the models see them: N(0,1) with outlier channels
    x = torch.randn(64, 512)
    x[:, 7] *= 30.0
    add("act8", x.half(), 8, [1, 16], True)
    # small magnitudes (|x| < 2^-5, where fp32 `+ 1e-9` changes the value): LLM weights ~ N(0, 0.02^2)
    add("w4", (0.02 * torch.randn(64, 512)).half(), 4, [1, 16], False)
    add("w4_128", (0.02 * torch.randn(32, 512)).half(), 4, [1, 128], False)
    add("small8", (0.01 * torch.randn(64, 512)).half(), 8, [1, 16], True)
    # exact rounding ties at small magnitude: k + 0.5 steps of a block whose maximum pins the exponent
    t = torch.zeros(16, 16)
    t[:, 0] = 2.0 ** -6                                       # block maximum: e = -6, step 2^-13 for width 8
    for i in range(1, 16):
        t[:, i] = (torch.arange(16).float() * 8 + i + 0.5) * 2.0 ** -13
    add("ties8", t.half(), 8, [1, 16], True)
    # per-token activations (the W4A8 INT configurations)
    add("row8", x.half(), 8, [1, -1], True)
    np.savez_compressed(os.path.join(HERE, "quantizers_fp16.npz"), **g)
    print("wrote", len(g), "arrays")


if __name__ == "__main__":
    main()
