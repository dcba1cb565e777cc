"""GPU approximator: the low-rank factors A, B of a Linear's quantization error.

Counterpart of the reference's WeightApproximatorLqerSvd.approximate (src/lqer/approximate/lqer_svd.py:37-47) and
WeightApproximatorLqerAct.approximate (lqer_act.py:84-97):

    E^T = (W - Q_w(W))^T                      [in, out]      (base.py:44-49)
    S E^T = U diag(s) V^T                     S = diag(scale), scale = 1 for LQER-SVD   (lqer_act.py:74-82)
    A = Q_A(S^-1 U[:, :r]),  B = Q_B(diag(s[:r]) V^T[:r])

Q_w, Q_A, Q_B run on the library's HIP quantizer kernels (the A / B quantizers' blocks run along dim 0 - block_size
[16, 1], llama-7b.toml:60-73 - which is the last dim of the transposed operand); the SVD is torch.linalg.svd on the GPU
(rocSOLVER), in fp32.  A and B themselves are not comparable across SVD implementations (sign / rotation of close
singular vectors); the tests pin the reconstruction error |E^T - A B|.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from .linear import LinearFlexibleLqer


def _quantize_along_dim0(t: torch.Tensor, cfg: Optional[dict]) -> torch.Tensor:
    """A / B quantizer: block_size [L, 1] (blocks along dim 0), [1, L] / [L] (along the last dim) or passthrough."""
    if cfg is None or cfg.get("name") == "passthrough":
        return t
    bs = cfg.get("block_size", [16])
    bs = [bs] if isinstance(bs, int) else list(bs)
    if len(bs) == 2 and bs[1] == 1 and bs[0] != 1:
        c = dict(cfg, block_size=[1, bs[0]])
        q = ops.quantize_mxint(t.t().contiguous(), ops.make_qfmt(c), want=("deq",))["deq"]
        return q.t().contiguous()
    return ops.quantize_mxint(t.contiguous(), ops.make_qfmt(cfg), want=("deq",))["deq"]


@torch.no_grad()
def lqer_factors(W: torch.Tensor, w_cfg: dict, rank: int, a_cfg: Optional[dict] = None, b_cfg: Optional[dict] = None,
                 scale: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """A [in, rank], B [rank, out] (fp32, on W's device) for one weight W [out, in].  `scale` [in] is the activation
    scale of L2QER (lqer_act.py); None gives LQER-SVD."""
    ops._need_gpu(W)
    Wf = W.float()
    Wq = ops.quantize_mxint(Wf.contiguous(), ops.make_qfmt(w_cfg), want=("deq",))["deq"]
    err_t = (Wf - Wq).t()
    if scale is not None:
        s = scale.to(Wf.device, torch.float32)
        err_t = s[:, None] * err_t
    U, S, Vh = torch.linalg.svd(err_t, full_matrices=False)
    A = U[:, :rank]
    if scale is not None:
        A = A / s[:, None]
    B = S[:rank, None] * Vh[:rank]
    return _quantize_along_dim0(A.contiguous(), a_cfg), _quantize_along_dim0(B.contiguous(), b_cfg)


@torch.no_grad()
def approximate_model(model: nn.Module, a_cfg: Optional[dict] = None, b_cfg: Optional[dict] = None,
                      scale_dict: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
    """Fill A and B of every LinearFlexibleLqer of a model prepared by models.quantize_model (weights still dense, on
    the GPU).  Returns the dictionary the reference stores as low_rank_dict.pt ({"<module>.A", "<module>.B"}).
    scale_dict maps module names to activation scales [in_features] (L2QER); missing names use plain SVD."""
    out: Dict[str, torch.Tensor] = {}
    for name, m in model.named_modules():
        if not isinstance(m, LinearFlexibleLqer) or m.rank == 0:
            continue
        if m.w_is_quantized:
            raise RuntimeError(f"{name}: weight already replaced by its quantized values (run before the first forward)")
        w_cfg = m.q_config.get("w_quantizer", m.q_config["default"])
        sc = scale_dict.get(name) if scale_dict else None
        A, B = lqer_factors(m.weight.data, w_cfg, m.rank, a_cfg, b_cfg, sc)
        m.A.data.copy_(A.to(m.A.dtype))
        m.B.data.copy_(B.to(m.B.dtype))
        m.invalidate_packed()
        out[f"{name}.A"], out[f"{name}.B"] = m.A.data.clone(), m.B.data.clone()
    return out
