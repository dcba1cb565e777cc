"""lqer_amd - MI355X (gfx950) implementation of the LQER quantized-Linear inference path.

Public surface mirrors the reference's `lqer.quantize` package for this path:
    get_quantized_layer_cls("linear", q_config) -> LinearFlexible | LinearFlexibleLqer
    get_quantized_func("matmul" | "bmm", q_config) -> matmul_flexible | bmm_flexible
All compute is in liblqer_hip.so (hand-written HIP); see include/lqer_hip.h and DESIGN.md.
"""
from .functional import bmm_flexible, get_quantized_func, matmul_flexible  # noqa: F401
from .linear import LinearFlexible, LinearFlexibleLqer, get_quantized_layer_cls  # noqa: F401

__all__ = ["LinearFlexible", "LinearFlexibleLqer", "get_quantized_layer_cls", "matmul_flexible", "bmm_flexible",
           "get_quantized_func"]
