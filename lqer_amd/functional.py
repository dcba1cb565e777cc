"""Quantized attention matmuls - mirror of the reference's quantized_functions package for this path
(src/lqer/quantize/quantized_functions/matmul.py:12-37, __init__.py:3-21; call sites llama_decoder.py:263,294 and
opt_decoder.py:125,190):   product = matmul(x_quantizer(x), w_quantizer(y)).

Both operands are quantized by the library's HIP quantizer kernels (blocks along the last dim of each operand, which
for y is NOT the contraction dim - llama-7b.toml:110-126); the product of the two quantized images is a plain GEMM and
goes through torch.matmul / torch.bmm (rocBLAS / hipBLASLt) in the operands' dtype, like the reference's.  Quantizer
settings outside what the kernels implement raise - there is no software fallback.
"""
from __future__ import annotations

from copy import deepcopy

import torch

from . import ops

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
    return ops.quantize_mxint(t, fmt, want=("deq",))["deq"].to(t.dtype)


def generic_matmul_flexible(x: torch.Tensor, y: torch.Tensor, q_config: dict, style: str = "matmul") -> torch.Tensor:
    matmul = MATMUL_MAP[style]
    # q_config["default"] is evaluated eagerly, as in the reference (matmul.py:15-16)
    x_cfg = deepcopy(q_config.get("x_quantizer", q_config["default"]))
    w_cfg = deepcopy(q_config.get("w_quantizer", q_config["default"]))
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
