"""A fixed-seed slice of the randomised sweeps (tools/fuzz_parity.py, tools/fuzz_matmul.py) inside the GPU suite: random shapes,
ranks, dtypes and quantizer configurations - every GEMM route, the int8 weight-image modes, integer weights, the group launch
against its members, attention products with blocks of 16 / 32 / 48 / 64 / row - against the CPU oracle (forward) or the
two-step route (attention products).  The tools run hundreds of cases; this keeps 36 + 24 of them under the driver's clock.
Run on the GPU box:  python -m pytest tests -m gpu -x -q"""
import os
import random
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_forward_fuzz_slice_vs_oracle():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, ROOT)
    from tools import fuzz_parity as F

    rng = random.Random(20260404)
    dev = torch.device("cuda:0")
    bad = []
    for i in range(36):
        c = F.one_case(rng)
        err, tol = F.run_case(*c, dev)
        if not err <= tol:
            bad.append((i, c, err, tol))
    assert not bad, bad


def test_attention_product_fuzz_slice():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_matmul.py"), "24", "7"], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    assert "24 / 24 within tolerance" in res.stdout
