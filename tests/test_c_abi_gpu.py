"""The C ABI used from a C++ program with no Python in the process (tests/c_abi/forward_demo.cpp): it must build against
include/lqer_hip.h + liblqer_hip.so alone, and its output must match the CPU oracle on the same inputs."""
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "forward_demo.cpp")
LIBDIR = os.path.join(ROOT, "lqer_amd")


def _build(tmp_path, compile_only=False):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    if not os.path.exists(os.path.join(LIBDIR, "liblqer_hip.so")):
        pytest.skip("liblqer_hip.so not built")
    exe = str(tmp_path / "forward_demo")
    cmd = [hipcc, "-O2", "-Wno-unused-result", "-o", exe, SRC, f"-L{LIBDIR}", "-llqer_hip", f"-Wl,-rpath,{LIBDIR}"]
    if compile_only:
        cmd = [hipcc, "-O2", "-c", "-o", exe + ".o", SRC]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    return exe


def test_demo_compiles_against_the_header(tmp_path):
    _build(tmp_path, compile_only=True)


def _lcg_f16(n, scale, state):
    out = np.empty(n, dtype=np.float32)
    for i in range(n):
        state = (state * 1664525 + 1013904223) & 0xFFFFFFFF
        out[i] = np.float32(state >> 8) / np.float32(8388608.0) - np.float32(1.0)
    return (np.float32(scale) * out).astype(np.float16), state


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["mxint", "a16"])
@pytest.mark.parametrize("M,K,N,r", [(40, 320, 300, 32), (200, 256, 512, 16)])
def test_demo_matches_oracle(tmp_path, mode, M, K, N, r):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import lqer_oracle as O

    exe = _build(tmp_path)
    out = str(tmp_path / "y.bin")
    res = subprocess.run([exe, str(M), str(K), str(N), str(r), out] + (["a16"] if mode == "a16" else []), capture_output=True, text=True,
                         timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "forward_demo" in res.stdout
    y = torch.from_numpy(np.fromfile(out, dtype=np.float16).reshape(M, N)).float()
    st = 12345
    x, st = _lcg_f16(M * K, 2.0, st)
    W, st = _lcg_f16(N * K, 0.05, st)
    A, st = _lcg_f16(K * r, 0.02, st)
    B, st = _lcg_f16(r * N, 0.02, st)
    bias, st = _lcg_f16(N, 0.1, st)
    t = lambda a, *s: torch.from_numpy(a.astype(np.float32)).reshape(*s)
    bfp = lambda w, b, skip: dict(name="block_fp", width=w, exponent_width=8, exponent_bias=None, block_size=b, skip_first_dim=skip)
    qc = dict(name="flexible_lqer", is_ptq=True, default=False, w_quantizer=bfp(4, [1, 16], False), b_quantizer=bfp(8, [1, 16], False),
              x_quantizer=bfp(8, [1, 16], True) if mode == "mxint" else dict(name="passthrough"))
    ref = O.lqer_linear_forward(t(x, M, K), t(W, N, K), t(bias, N), t(A, K, r), t(B, r, N), qc)
    err = (y - ref).norm() / ref.norm()
    assert err <= 1e-3, float(err)
