"""Criterion for a re-quantized fp32 sum whose accumulation order differs from the reference's (xAq = A_out(xq @ A),
reference quantized_layers/linear.py:154): every product is exact, so any summation order lands within D ulps of the exact
sum s; the output must equal the quantizer applied to SOME value in [s - D ulp, s + D ulp], element by element, with the
block exponent taken from the same interval of the block maximum.  Test infrastructure (uses the oracle's exponent rule)."""
import numpy as np
import torch

from oracle import lqer_oracle as O


def _q_with_exponent(v32: np.ndarray, e: np.ndarray, mbits: int) -> np.ndarray:
    """block_fp.py:55-65 in fp32 with a given block exponent (elementwise, monotone in v)."""
    v32 = v32.astype(np.float32)
    t = (np.abs(v32) + np.float32(1e-9)).astype(np.float32)
    m = np.minimum(np.rint(np.ldexp(t, (mbits - e).astype(np.int32)).astype(np.float32)), np.float32(2 ** mbits - 1))
    q = np.copysign(np.ldexp(m, (e - mbits).astype(np.int32)), v32).astype(np.float32)
    return np.where(np.abs(v32) <= np.float32(1e-8), np.float32(0), q)  # the bf16 image flushes the pass-through range


def envelope_check(s64: np.ndarray, got: np.ndarray, L: int, mbits: int, D: float):
    """Every block of `L` entries of `got` equals Q_e(v) for v in [s - D ulp, s + D ulp], e from the same interval of amax."""
    M, r = s64.shape
    ulp = np.spacing(np.abs(s64).astype(np.float32)).astype(np.float64)
    lo, hi = (s64 - D * ulp), (s64 + D * ulp)
    bad = 0
    for b0 in range(0, r, L):
        sl = slice(b0, b0 + L)
        amax = np.abs(s64[:, sl]).max(axis=1)
        aulp = np.spacing(amax.astype(np.float32)).astype(np.float64)
        cands = []
        for a in (amax - D * aulp, amax, amax + D * aulp):
            a32 = torch.from_numpy(np.maximum(a, 0).astype(np.float32))
            e = O.ceil_log2_f32(torch.where(a32 > 0, a32, torch.ones_like(a32))).numpy().astype(np.int32)
            cands.append(e)
        ok = np.zeros(M, dtype=bool)
        for e in cands:
            e2 = e[:, None]
            qlo = _q_with_exponent(lo[:, sl], e2, mbits)
            qhi = _q_with_exponent(hi[:, sl], e2, mbits)
            ok |= np.all((got[:, sl] >= np.minimum(qlo, qhi)) & (got[:, sl] <= np.maximum(qlo, qhi)), axis=1)
        ok |= amax == 0
        bad += int((~ok).sum())
    return bad
