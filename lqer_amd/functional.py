"""Quantized attention matmuls - mirror of the reference's quantized_functions package for this path
(src/lqer/quantize/quantized_functions/matmul.py:12-37, __init__.py:3-21; call sites llama_decoder.py:263,294 and
opt_decoder.py:125,190):   product = matmul(x_quantizer(x), w_quantizer(y)).

Blocks run along the last dim of each operand, which for y is NOT the contraction dim (llama-7b.toml:110-126).  With the
templates' settings (block_fp, width <= 8, blocks of 16) the whole product is ONE fused HIP GEMM (lqer_matmul_q,
csrc/matmul_q.hip): x is quantized in the GEMM's load path - read from HBM once, no quantized copy -, y through a small bf16
image, bf16 MFMA with fp32 accumulation of exact products; 4-D [bsz, heads, ..] operands are folded into one batch dim.  Other
block lengths (16 n elements, or whole rows; no template uses them) take the same product kernel behind the library's
standalone quantizer: that operand's bf16 image is written first and read as it is - the same bits.  Only operands whose
leading dims broadcast (no call site has them) run the quantizer kernels on both operands and hand the two quantized tensors to
torch.matmul / torch.bmm.  Quantizer settings outside what the kernels implement raise - there is no software fallback.
"""
from __future__ import annotations

from copy import deepcopy

import torch

import ctypes as C

from . import _lib, ops

MATMUL_MAP = {"matmul": torch.matmul, "bmm": torch.bmm}


def _quantize(t: torch.Tensor, cfg: dict) -> torch.Tensor:
    cfg = dict(cfg)
    name = cfg.get("name")
    if name == "passthrough":
        return t
    if name != "block_fp":
        raise NotImplementedError(f"lqer_amd.functional: quantizer {name!r} is not implemented on the HIP path")
    ops._need_gpu(t)
    fmt = ops.make_qfmt(cfg)
    if getattr(fmt, "act_tiles", None) is not None:
        # blocks that can span rows (incl. the quantizer's default lone [L], which the reference right-aligns to [1, S, L] on a 3-D operand,
        # quantizers/utils.py:56-66, 211-237): the HIP tile quantizer, as the reference blocks this very tensor (2-D / 3-D; else its error)
        if t.dim() not in (2, 3):
            raise RuntimeError(f"Unsupported x.ndim = {t.dim()}")  # (quantizers/utils.py:284)
        return ops.quantize_act_tiles(t, fmt)
    if fmt.block > 0 and fmt.block < t.shape[-1] and fmt.block % 16:
        raise NotImplementedError(f"lqer_amd.functional: block_size {cfg.get('block_size')} - the quantizer kernels take blocks of 16 n "
                                  "elements or whole rows along the last dim")
    return ops.quantize_mxint(t, fmt, want=("deq",))["deq"].to(t.dtype)


def _fused_fmt(cfg: dict):
    """The lqer_qfmt_t of a quantizer the library's product kernels cover (block_fp, width <= 8, blocks of 16 n elements or
    whole rows along the last dim: 16 runs fused in the load path, other lengths through the standalone quantizer's bf16
    image first - csrc/matmul_q.hip), else None."""
    if cfg.get("name") != "block_fp" or int(cfg.get("width", 12)) > 8:
        return None
    try:
        fmt = ops.make_qfmt(cfg, "x")
    except NotImplementedError:
        return None
    if getattr(fmt, "act_tiles", None) is not None:
        return None  # (blocks that can span rows: the tile quantizer + the library product, _quantize above)
    return fmt if (fmt.block <= 0 or fmt.block % 16 == 0) else None


_MAX_GRID_Z = 65535  # the kernels put the batch on grid.z


@torch.no_grad()
def _matmul_fused(x: torch.Tensor, y: torch.Tensor, fx, fy) -> torch.Tensor:
    """x [.., S1, K] @ y [.., K, S2] (equal leading dims, or both 2-D) through lqer_matmul_q.  Leading dims are folded into
    one batch dim (the llama call sites hand over 4-D [bsz, heads, ...] operands, llama_decoder.py:263,294) - as a view
    where the strides allow, e.g. the transposed view of K in Q K^T -, and batches beyond the grid limit go in chunks."""
    lead = x.shape[:-2]
    squeeze = x.dim() == 2
    x3 = x[None] if squeeze else (x.flatten(0, -3) if x.dim() > 3 else x)
    y3 = y[None] if squeeze else (y.flatten(0, -3) if y.dim() > 3 else y)
    b, S1, K = x3.shape
    S2 = y3.shape[2]
    if x3.stride(2) != 1 or x3.stride(1) < K:
        x3 = x3.contiguous()
    if y3.stride(1) != 1 and y3.stride(2) != 1:  # (the kernel reads y dense along k - the transposed view of K - or along j)
        y3 = y3.contiguous()
    # an operand whose blocks are not 16 goes through the standalone quantizer first: evenly spaced rows over the batch, y dense along j
    x_pre, y_pre = fx.block != 16, fy.block != 16
    if x_pre and b > 1 and x3.stride(0) != S1 * x3.stride(1):
        x3 = x3.contiguous()
    if y_pre and (y3.stride(2) != 1 or (b > 1 and y3.stride(0) != K * y3.stride(1))):
        y3 = y3.contiguous()
    out = torch.empty(b, S1, S2, dtype=x.dtype, device=x.device)
    L = _lib.lib()
    with torch.cuda.device(x.device):
        for b0 in range(0, b, _MAX_GRID_Z):
            xb, yb, ob = x3[b0:b0 + _MAX_GRID_Z], y3[b0:b0 + _MAX_GRID_Z], out[b0:b0 + _MAX_GRID_Z]
            nb = xb.shape[0]
            nws = L.lqer_matmul_q_workspace_bytes_fmt(nb, S1, K, S2, C.byref(fx), C.byref(fy))
            ws = ops.workspace(x.device, max(nws, 16))
            ys = yb.stride()
            _lib.check(L.lqer_matmul_q(xb.data_ptr(), yb.data_ptr(), ob.data_ptr(), ops.dtype_code(x3), nb, S1, K, S2, xb.stride(0),
                                       xb.stride(1), ys[0], ys[1], ys[2], C.byref(fx), C.byref(fy), ws.data_ptr(), ws.numel(),
                                       ops._stream(x.device)),
                       "lqer_matmul_q")
    return out[0] if squeeze else out.reshape(*lead, S1, S2)


def generic_matmul_flexible(x: torch.Tensor, y: torch.Tensor, q_config: dict, style: str = "matmul") -> torch.Tensor:
    matmul = MATMUL_MAP[style]
    # q_config["default"] is evaluated eagerly, as in the reference (matmul.py:15-16)
    x_cfg = deepcopy(q_config.get("x_quantizer", q_config["default"]))
    w_cfg = deepcopy(q_config.get("w_quantizer", q_config["default"]))
    fx, fy = _fused_fmt(x_cfg), _fused_fmt(w_cfg)
    if (fx is not None and fy is not None and x.dtype == y.dtype and x.dtype in ops._DT and x.dim() == y.dim() and x.dim() >= 2
            and (style == "matmul" or x.dim() == 3)  # (torch.bmm takes 3-D operands only: anything else raises below, as there)
            and x.shape[-1] == y.shape[-2] and x.shape[:-2] == y.shape[:-2] and x.numel() > 0 and y.numel() > 0):
        ops._need_gpu(x, y)
        return _matmul_fused(x, y, fx, fy)
    return matmul(_quantize(x, x_cfg), _quantize(y, w_cfg))


def matmul_flexible(x, y, q_config):
    return generic_matmul_flexible(x, y, q_config, style="matmul")


def bmm_flexible(x, y, q_config):
    return generic_matmul_flexible(x, y, q_config, style="bmm")


QUANTIZED_FUNCTION_MAP = {"matmul": {"flexible": matmul_flexible}, "bmm": {"flexible": bmm_flexible}}


def get_quantized_func(op: str, q_config: dict):
    assert op in QUANTIZED_FUNCTION_MAP, f"Unsupported quantized op: {op}"
    assert q_config["name"] in QUANTIZED_FUNCTION_MAP[op], f"Unsupported quantized config: {q_config}"
    return QUANTIZED_FUNCTION_MAP[op][q_config["name"]]
