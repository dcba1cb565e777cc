"""CPU oracle for the LQER quantized-Linear hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (eager torch-CPU, fp32) of the algorithm of the
reference's `LinearFlexibleLqer.forward` and of the number-format emulators it calls.  It is the
checker for the HIP path: only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it.  Nothing under `lqer_amd/` imports it, and the product path fails loudly
when the HIP library is missing instead of falling back to this code.

Parity pin: the reference has no tests or golden vectors of its own, so this oracle
is pinned against outputs of the reference itself, imported in the build container by
`tests/golden/make_golden.py` (the generating script is committed, the vectors are under
`tests/golden/*.npz`).  `tests/test_oracle_golden.py` checks every function here bit-for-bit
(quantizers) / to 1e-6 (forward) against those vectors on every run, with no access to
/root/reference.

Reference citations (paths relative to /root/reference/src/lqer/quantize/):
  mxint_quantize          <- quantizers/block_fp.py:7-82  (_block_fp_quantize)
  _blocks / _unblocks     <- quantizers/utils.py:42-83 (shape inference, padding),
                             :86-124 (1-D bias), :127-158 (2-D activation), :161-208 (2-D weight),
                             :211-258 (3-D activation), :261-321 (dispatch)
  integer_quantize        <- quantizers/integer.py:10-43
  passthrough             <- quantizers/passthrough.py:1
  get_quantizer           <- quantizers/__init__.py:7-18
  lqer_linear_forward     <- quantized_layers/linear.py:145-157 (PTQ branch), :50-59 (no side path)

Semantics are fp32: inputs of any float dtype are upcast to fp32 first.  (The reference evaluates
in the tensor's own dtype; SURVEY.md §4 shows its fp16 evaluation is 1.3e-2 away from its fp32 one
and overflows for |x| > 32768, so fp32 is the oracle - BASELINE.json's "CPU emulation".)
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------------
# ceil(log2(v)) exactly as torch evaluates `torch.ceil(torch.log2(v))` on fp32 (block_fp.py:58)
# ----------------------------------------------------------------------------------------------
# torch's fp32 log2 is correctly rounded near powers of two, so for v = 2^k * (1 + j*2^-23) the
# sum k + log2(1 + j*2^-23) rounds back to k while j <= J(k); then ceil() yields k instead of k+1.
# J depends on the binade p of the values just above k on the real line: J = floor(2^(p-1)*ln 2).
# Tabulated against torch for every k in [-126, 127] (tests/golden/make_golden.py, "log2_rule").
_J_TABLE = (0, 0, 1, 2, 5, 11, 22)


def _slack_ulps(k: torch.Tensor) -> torch.Tensor:
    """J(k): number of ulps above 2^k for which torch's ceil(log2(.)) still returns k."""
    ak = k.abs()
    # p = floor(log2(|k|)) for k > 0; for k < 0 the neighbours of k towards +inf have magnitude
    # just under |k|, so an exact power of two |k| belongs to the binade below.
    akf = ak.clamp(min=1).to(torch.float64)
    p = torch.floor(torch.log2(akf)).to(torch.int64)
    is_pow2 = (ak & (ak - 1)) == 0
    p = torch.where((k < 0) & is_pow2, p - 1, p)
    p = p.clamp(min=0, max=len(_J_TABLE) - 1)
    table = torch.tensor(_J_TABLE, dtype=torch.int64)
    return torch.where(ak == 0, torch.zeros_like(p), table[p])


def ceil_log2_f32(v: torch.Tensor) -> torch.Tensor:
    """Integer-valued fp32 tensor equal to torch.ceil(torch.log2(v)) for positive fp32 v,
    computed from the bit pattern (this is the rule the HIP kernels implement)."""
    assert v.dtype == torch.float32
    bits = v.contiguous().view(torch.int32).to(torch.int64)
    expf = (bits >> 23) & 0xFF
    mant = bits & 0x7FFFFF
    k = expf - 127
    e = torch.where(mant > _slack_ulps(k), k + 1, k)
    # subnormals / inf / nan: leave to libm (never reached by data the path cares about:
    # a block whose max is subnormal is entirely inside the |x| <= 1e-8 pass-through)
    odd = (expf == 0) | (expf == 255)
    if bool(odd.any()):
        e = torch.where(odd, torch.ceil(torch.log2(v.to(torch.float64))).clamp(-1e6, 1e6).to(torch.int64), e)
    return e.to(torch.float32)


# ----------------------------------------------------------------------------------------------
# blocking (quantizers/utils.py:42-321), restated with reshape/permute instead of unfold/fold
# ----------------------------------------------------------------------------------------------
def infer_block_shape(x_shape: Sequence[int], block_shape: Sequence[int]) -> List[int]:
    """Right-align `block_shape` to `x_shape`; -1 or an oversize entry means the whole dim
    (utils.py:42-67)."""
    nd = len(x_shape)
    bs = list(block_shape)
    bs = bs[-nd:] if len(bs) >= nd else [-1] * (nd - len(bs)) + bs
    return [xs if (b == -1 or b > xs) else b for xs, b in zip(x_shape, bs)]


def _pad_to(x: torch.Tensor, dims: Sequence[int], mult: Sequence[int]) -> torch.Tensor:
    """Zero-pad dims `dims` of x on the right up to multiples of `mult` (utils.py:70-83)."""
    pad = []
    for d in range(x.ndim - 1, -1, -1):
        if d in dims:
            m = mult[list(dims).index(d)]
            pad += [0, (-x.shape[d]) % m]
        else:
            pad += [0, 0]
    return F.pad(x, pad) if any(pad) else x


def _blocks(x: torch.Tensor, block_size: Sequence[int], skip_first_dim: bool, via_unfold: bool = False):
    """Return (blocked [..., nblocks, block_elems], restore_fn).  Dispatch as utils.py:261-284."""
    shape = list(x.shape)
    if x.ndim == 1:
        assert not skip_first_dim, "skip_first_dim must be False for a 1-D bias"
        (L,) = infer_block_shape(shape, block_size)
        xp = _pad_to(x, [0], [L])
        blk = xp.reshape(-1, L)
        return blk, (lambda q: q.reshape(-1)[: shape[0]])
    if x.ndim == 2 and skip_first_dim:
        # activation [tokens, hidden]: block inferred against ONE row (utils.py:127-144)
        L = infer_block_shape([1, shape[1]], block_size)[-1]
        xp = _pad_to(x, [1], [L])
        blk = xp.reshape(shape[0], -1, L)
        return blk, (lambda q: q.reshape(shape[0], -1)[:, : shape[1]])
    if x.ndim == 2:
        # weight [rows, cols]: 2-D tiles of br x bc (utils.py:161-183)
        br, bc = infer_block_shape(shape, block_size)
        xp = _pad_to(x, [0, 1], [br, bc])
        R, C = xp.shape
        if via_unfold:  # op-for-op the reference's route (cpu_baseline timing only)
            cols = F.unfold(xp[None, None], kernel_size=(br, bc), stride=(br, bc))[0]  # [br*bc, nblk]
            blk = cols.t()

            def restore_u(q):
                y = F.fold(q.t()[None], output_size=(R, C), kernel_size=(br, bc), stride=(br, bc))
                return y[0, 0, : shape[0], : shape[1]]

            return blk, restore_u
        blk = xp.reshape(R // br, br, C // bc, bc).permute(0, 2, 1, 3).reshape(-1, br * bc)

        def restore(q):
            y = q.reshape(R // br, C // bc, br, bc).permute(0, 2, 1, 3).reshape(R, C)
            return y[: shape[0], : shape[1]]

        return blk, restore
    if x.ndim == 3 and skip_first_dim:
        # activation [batch, tokens, hidden]: tiles over (tokens, hidden) per batch (utils.py:211-237)
        _, br, bc = infer_block_shape([1, shape[1], shape[2]], block_size)
        xp = _pad_to(x, [1, 2], [br, bc])
        Bn, R, C = xp.shape
        if via_unfold:
            cols = F.unfold(xp[:, None], kernel_size=(br, bc), stride=(br, bc))  # [B, br*bc, nblk]
            blk = cols.transpose(1, 2)

            def restore_u3(q):
                y = F.fold(q.transpose(1, 2), output_size=(R, C), kernel_size=(br, bc), stride=(br, bc))
                return y[:, 0, : shape[1], : shape[2]]

            return blk, restore_u3
        blk = xp.reshape(Bn, R // br, br, C // bc, bc).permute(0, 1, 3, 2, 4).reshape(Bn, -1, br * bc)

        def restore3(q):
            y = q.reshape(Bn, R // br, C // bc, br, bc).permute(0, 1, 3, 2, 4).reshape(Bn, R, C)
            return y[:, : shape[1], : shape[2]]

        return blk, restore3
    raise NotImplementedError(f"blocking of a {x.ndim}-D tensor with skip_first_dim={skip_first_dim}")


# ----------------------------------------------------------------------------------------------
# MXINT / block floating point (quantizers/block_fp.py:7-82)
# ----------------------------------------------------------------------------------------------
def mxint_quantize(
    x: torch.Tensor,
    width: int = 12,
    exponent_width: int = 8,
    exponent_bias: Optional[int] = None,
    block_size: Sequence[int] = (16,),
    skip_first_dim: bool = True,
    *,
    decompose: bool = False,
    via_unfold: bool = False,
    libm_log2: bool = False,
):
    """Shared-exponent sign-magnitude quantizer.  Returns the dequantized fp32 tensor; with
    decompose=True also (signed mantissa codes as int32 shaped like x, per-block exponents)."""
    if isinstance(block_size, int):
        block_size = [block_size]
    x = x.to(torch.float32)
    blk, restore = _blocks(x, list(block_size), skip_first_dim, via_unfold)
    amax = blk.abs().amax(dim=-1, keepdim=True)
    nz = amax != 0
    if not bool(nz.any()):
        amax_f = torch.ones_like(amax)  # block_fp.py:40-42
    else:
        amax_f = torch.where(nz, amax, amax[nz].min())  # block_fp.py:44
    mbits = width - 1
    if exponent_bias in (None, "none", "None", "NA"):
        exponent_bias = 2 ** (exponent_width - 1) - 1
    e_max = 2**exponent_width - 1 - exponent_bias
    e_min = -exponent_bias
    m_max = 2**mbits - 1
    if libm_log2:
        e = torch.ceil(torch.log2(amax_f))
    else:
        e = ceil_log2_f32(amax_f)
    e = e.clamp(e_min, e_max)
    sign = torch.sign(blk + 1e-9)  # block_fp.py:55
    mag = blk.abs() + 1e-9  # block_fp.py:57
    scale = torch.pow(2.0, e)
    m = torch.round(mag / scale * (2**mbits)).clamp(0, m_max)  # block_fp.py:61-65, round = RNE
    q = sign * scale * (m / (2**mbits))
    out = restore(q)
    tiny = x.abs() <= 1e-8  # torch.isclose(x, 0): atol 1e-8, rtol*|0| = 0  (block_fp.py:79-80)
    out = torch.where(tiny, x, out)
    if not decompose:
        return out
    codes = restore((sign * m)).to(torch.int32)
    codes = torch.where(tiny, torch.zeros_like(codes), codes)
    exps = torch.where(nz, e, torch.zeros_like(e)).to(torch.int32).squeeze(-1)
    return out, codes, exps


def integer_quantize(x: torch.Tensor, width: int, frac_width: int, is_signed: bool = True) -> torch.Tensor:
    """Fixed point: clamp(rne(x * 2^frac), lo, hi) / 2^frac  (quantizers/integer.py:10-43)."""
    x = x.to(torch.float32)
    lo, hi = (-(2 ** (width - 1)), 2 ** (width - 1) - 1) if is_signed else (0, 2**width - 1)
    s = 2**frac_width
    return torch.round(x * s).clamp(lo, hi) / s


def passthrough(x: torch.Tensor, *args, **kwargs) -> torch.Tensor:
    return x


def get_quantizer(cfg: Optional[dict], **extra):
    """q_config entry -> callable, same dispatch as quantizers/__init__.py:7-18 with the remaining
    keys bound as kwargs (linear.py:93-98).  'NA' stands for None (utils.py:58-94 of the reference)."""
    cfg = dict(cfg)
    name = cfg.pop("name")
    if name == "passthrough":
        return lambda t: t
    if name == "block_fp":
        kw = {k: cfg[k] for k in ("width", "exponent_width", "exponent_bias", "block_size", "skip_first_dim") if k in cfg}
        return lambda t: mxint_quantize(t, **kw, **extra)
    if name == "integer":
        kw = {k: cfg[k] for k in ("width", "frac_width", "is_signed") if k in cfg}
        return lambda t: integer_quantize(t, **kw)
    raise ValueError(f"quantizer {name} not supported")


# ----------------------------------------------------------------------------------------------
# the Linear forward (quantized_layers/linear.py:145-157; :50-59 without the side path)
# ----------------------------------------------------------------------------------------------
def resolve_linear_quantizers(q_config: dict) -> Dict[str, dict]:
    """x / w / b / A_out / B_out quantizer configs with the reference's fall-backs
    (linear.py:90-106 and :115-124: A_out and B_out default to the x quantizer's config)."""
    d = q_config.get("default")
    xq = q_config.get("x_quantizer", d)
    return {
        "x": xq,
        "w": q_config.get("w_quantizer", d),
        "b": q_config.get("b_quantizer", d),
        "A_out": q_config.get("A_out_quantizer", xq),
        "B_out": q_config.get("B_out_quantizer", xq),
    }


def lqer_linear_forward(
    x: torch.Tensor,
    weight: torch.Tensor,
    bias: Optional[torch.Tensor],
    A: Optional[torch.Tensor],
    B: Optional[torch.Tensor],
    q_config: dict,
    *,
    weight_is_quantized: bool = False,
    via_unfold: bool = False,
    intermediates: bool = False,
):
    """y = Q_x(x) W_q^T + b_q + Q_Bout(Q_Aout(Q_x(x) A) B)   (fp32).

    weight/bias are quantized here unless `weight_is_quantized` (the reference does it in place on
    the first call, linear.py:149-153).  A=None selects LinearFlexible (no side path)."""
    qs = resolve_linear_quantizers(q_config)
    ex = dict(via_unfold=via_unfold)
    qx = get_quantizer(qs["x"], **(ex if qs["x"]["name"] == "block_fp" else {}))
    x32 = x.to(torch.float32)
    xq = qx(x32)
    if weight_is_quantized:
        wq, bq = weight.to(torch.float32), (None if bias is None else bias.to(torch.float32))
    else:
        wq = get_quantizer(qs["w"], **(ex if qs["w"]["name"] == "block_fp" else {}))(weight.to(torch.float32))
        bq = None if bias is None else get_quantizer(qs["b"])(bias.to(torch.float32))
    out: Dict[str, torch.Tensor] = {"xq": xq, "wq": wq}
    if bq is not None:
        out["bq"] = bq
    y = F.linear(xq, wq, bq)
    if A is not None:
        qa = get_quantizer(qs["A_out"], **(ex if qs["A_out"]["name"] == "block_fp" else {}))
        qb = get_quantizer(qs["B_out"], **(ex if qs["B_out"]["name"] == "block_fp" else {}))
        xA = torch.matmul(xq, A.to(torch.float32))
        xAq = qa(xA)
        xAB = torch.matmul(xAq, B.to(torch.float32))
        xABq = qb(xAB)
        y = y + xABq
        out.update(xA=xA, xAq=xAq, xAB=xAB, xABq=xABq)
    out["y"] = y
    return out if intermediates else y


# ----------------------------------------------------------------------------------------------
# the build's packed weight format (no counterpart in the reference; restated here so the GPU
# pack kernel can be checked bit-for-bit).  See DESIGN.md "Data layout".
# ----------------------------------------------------------------------------------------------
_NIB_POS = [0, 2, 4, 6, 1, 3, 5, 7]  # nibble position (within a 32-bit word) of k = 0..7


def pack_weight_mxint4(weight: torch.Tensor, block: int, n_pad: int = 1, k_pad: int = 64):
    """W[N,K] -> (codes uint8 [Np, Kp/2], exps int8 [Np, Kp/block_eff]) with
    code = 4-bit sign-magnitude mantissa (bit 3 = sign, magnitude 0..7, +0 canonical);
    within each group of 8 consecutive k (one little-endian 32-bit word) nibble p holds
    k = p/2 (p even) or 4 + p/2 (p odd);
    exps = shared exponent e of each `block` consecutive k (value = +-magnitude * 2^(e-3)).
    block <= 0 means one block per row.  Rows/cols are zero-padded to multiples of n_pad / k_pad.
    Elements with |w| <= 1e-8 are flushed to code 0 (reference keeps them unquantized)."""
    N, K = weight.shape
    L = K if block <= 0 or block > K else block
    _, codes, exps = mxint_quantize(weight, width=4, block_size=[1, L], skip_first_dim=False, decompose=True)
    Np = -(-N // n_pad) * n_pad
    Kp = -(-K // k_pad) * k_pad
    nblk = -(-Kp // L)
    cpad = torch.zeros(Np, Kp, dtype=torch.int32)
    cpad[:N, :K] = codes
    epad = torch.zeros(Np, nblk, dtype=torch.int32)
    epad[:N, : exps.reshape(N, -1).shape[1]] = exps.reshape(N, -1)
    sm = torch.where(cpad < 0, 8 - cpad, cpad).to(torch.uint8)  # sign-magnitude nibble
    grp = sm.reshape(Np, Kp // 8, 8)
    nib = torch.zeros_like(grp)
    for k, pos in enumerate(_NIB_POS):
        nib[:, :, pos] = grp[:, :, k]
    nib = nib.reshape(Np, Kp)
    packed = nib[:, 0::2] | (nib[:, 1::2] << 4)
    return packed.contiguous(), epad.clamp(-128, 127).to(torch.int8).contiguous()


def unpack_weight_mxint4(packed: torch.Tensor, exps: torch.Tensor, N: int, K: int, block: int) -> torch.Tensor:
    L = K if block <= 0 or block > K else block
    lo = (packed & 0xF).to(torch.int32)
    hi = (packed >> 4).to(torch.int32)
    nib = torch.stack([lo, hi], dim=-1).reshape(packed.shape[0], -1, 8)
    grp = torch.stack([nib[:, :, pos] for pos in _NIB_POS], dim=-1).reshape(packed.shape[0], -1)
    c = torch.where(grp >= 8, -(grp - 8), grp).to(torch.float32)
    e = exps.to(torch.float32).repeat_interleave(L, dim=1)[:, : c.shape[1]]
    return (c * torch.pow(2.0, e - 3))[:N, :K]


def flops(M: int, K: int, N: int, r: int) -> int:
    """The reference's multiply model (experiments/hw_performance/README.md:81-106), x2 for FMA."""
    return 2 * M * K * N + 2 * M * K * r + 2 * M * r * N


# ----------------------------------------------------------------------------------------------
# quantized attention matmuls (quantized_functions/matmul.py:12-37)
# ----------------------------------------------------------------------------------------------
def matmul_flexible(x: torch.Tensor, y: torch.Tensor, q_config: dict, style: str = "matmul") -> torch.Tensor:
    """product = matmul(x_quantizer(x), w_quantizer(y)); blocks run along the LAST dim of each operand (for y that is
    not the contraction dim - llama-7b.toml:110-126).  q_config["default"] is evaluated eagerly, as in the reference."""
    xq = get_quantizer(q_config.get("x_quantizer", q_config["default"]))
    wq = get_quantizer(q_config.get("w_quantizer", q_config["default"]))
    mm = torch.matmul if style == "matmul" else torch.bmm
    return mm(xq(x), wq(y))


def bmm_flexible(x: torch.Tensor, y: torch.Tensor, q_config: dict) -> torch.Tensor:
    return matmul_flexible(x, y, q_config, style="bmm")


# ----------------------------------------------------------------------------------------------
# low-rank factors of the quantization error (approximate/lqer_svd.py:37-47, lqer_act.py:74-97, base.py:44-49)
# ----------------------------------------------------------------------------------------------
def lqer_factors(W: torch.Tensor, w_cfg: dict, rank: int, a_cfg: Optional[dict] = None, b_cfg: Optional[dict] = None,
                 scale: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """A = Q_A(S^-1 U_k), B = Q_B(diag(s_k) V_k^T) with S E^T = U diag(s) V^T, E = W - Q_w(W), S = diag(scale) or I."""
    Wf = W.float()
    err_t = (Wf - get_quantizer(w_cfg)(Wf)).t()
    if scale is not None:
        err_t = torch.diag(scale.float()) @ err_t
    U, S, Vh = torch.linalg.svd(err_t)
    A, B = U[:, :rank], torch.diag(S[:rank]) @ Vh[:rank, :]
    if scale is not None:
        A = torch.diag(scale.float()).inverse() @ A
    qa = get_quantizer(a_cfg) if a_cfg else (lambda t: t)
    qb = get_quantizer(b_cfg) if b_cfg else (lambda t: t)
    return qa(A), qb(B)
