"""lqer_amd - MI355X (gfx950) implementation of the LQER quantized-Linear inference path.

Public surface mirrors the reference's `lqer.quantize` package for this path:
    get_quantized_layer_cls("linear", q_config) -> LinearFlexible | LinearFlexibleLqer
All compute is in liblqer_hip.so (hand-written HIP); see include/lqer_hip.h and DESIGN.md.
"""
from .linear import LinearFlexible, LinearFlexibleLqer, get_quantized_layer_cls  # noqa: F401

__all__ = ["LinearFlexible", "LinearFlexibleLqer", "get_quantized_layer_cls"]
