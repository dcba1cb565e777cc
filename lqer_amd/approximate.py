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

import re
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from .linear import LinearFlexibleLqer
from .sweep import layer_partition


def _quantize_along_dim0(t: torch.Tensor, cfg: Optional[dict]) -> torch.Tensor:
    """A / B quantizer: block_size [L, 1] (blocks along dim 0), [1, L] / [L] (along the last dim) or passthrough."""
    if cfg is None or cfg.get("name") == "passthrough":
        return t
    bs = cfg.get("block_size", [16])
    bs = [bs] if isinstance(bs, int) else list(bs)
    if len(bs) == 2 and bs[1] == 1 and bs[0] != 1:
        c = dict(cfg, block_size=[1, bs[0]])
        q = ops.quantize_mxint(t.t().contiguous(), ops.make_qfmt(c, "w"), want=("deq",))["deq"]
        return q.t().contiguous()
    return ops.quantize_mxint(t.contiguous(), ops.make_qfmt(cfg, "w"), want=("deq",))["deq"]


@torch.no_grad()
def lqer_factors(W: torch.Tensor, w_cfg: dict, rank: int, a_cfg: Optional[dict] = None, b_cfg: Optional[dict] = None,
                 scale: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """A [in, rank], B [rank, out] (fp32, on W's device) for one weight W [out, in].  `scale` [in] is the activation
    scale of L2QER (lqer_act.py); None gives LQER-SVD."""
    ops._need_gpu(W)
    Wf = W.float()
    Wq = ops.quantize_mxint(Wf.contiguous(), ops.make_qfmt(w_cfg, "w"), want=("deq",))["deq"]
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


_LAYER_RE = re.compile(r"(?:^|\.)layers\.(\d+)\.")


def module_owners(names, world: int) -> Dict[str, int]:
    """Rank that factorizes each module when the work is split over `world` ranks: the decoder layers are dealt out in
    consecutive runs of ceil(L / G) (sweep.layer_partition, the reference's experiments/infer_device_map.py:29-37
    rule); modules outside any decoder layer belong to rank 0."""
    layer_of = {n: (int(m.group(1)) if (m := _LAYER_RE.search(n)) else None) for n in names}
    n_layers = 1 + max((l for l in layer_of.values() if l is not None), default=-1)
    owner_of_layer = {l: g for g, rng in enumerate(layer_partition(n_layers, world)) for l in rng}
    return {n: (0 if l is None else owner_of_layer[l]) for n, l in layer_of.items()}


@torch.no_grad()
def approximate_model(model: nn.Module, a_cfg: Optional[dict] = None, b_cfg: Optional[dict] = None,
                      scale_dict: Optional[Dict[str, torch.Tensor]] = None, factors_fn=None) -> Dict[str, torch.Tensor]:
    """Fill A and B of every LinearFlexibleLqer of a model prepared by models.quantize_model (weights still dense, on
    the GPU).  Returns the dictionary the reference stores as low_rank_dict.pt ({"<module>.A", "<module>.B"}).
    scale_dict maps module names to activation scales [in_features] (L2QER); missing names use plain SVD.

    With an initialised torch.distributed group (one process per GPU, every rank holding the same model) the SVDs are
    split over the ranks by decoder layer (module_owners) and each module's A, B are broadcast from their owner, so all
    ranks end with the full set - the one exchange step of this offline path (RCCL broadcast of K*r + r*N elements).
    factors_fn (tests): replaces lqer_factors."""
    import torch.distributed as dist

    factors_fn = factors_fn or lqer_factors
    world, rank = (dist.get_world_size(), dist.get_rank()) if dist.is_available() and dist.is_initialized() else (1, 0)
    mods = [(name, m) for name, m in model.named_modules() if isinstance(m, LinearFlexibleLqer) and m.rank > 0]
    owners = module_owners([n for n, _ in mods], world)
    out: Dict[str, torch.Tensor] = {}
    for name, m in mods:
        if m.w_is_quantized:
            raise RuntimeError(f"{name}: weight already replaced by its quantized values (run before the first forward)")
        if owners[name] == rank:
            w_cfg = m.q_config.get("w_quantizer", m.q_config["default"])
            sc = scale_dict.get(name) if scale_dict else None
            A, B = factors_fn(m.weight.data, w_cfg, m.rank, a_cfg, b_cfg, sc)
            m.A.data.copy_(A.to(m.A.dtype))
            m.B.data.copy_(B.to(m.B.dtype))
        if world > 1:
            dist.broadcast(m.A.data, src=owners[name])
            dist.broadcast(m.B.data, src=owners[name])
        m.invalidate_packed(weight_changed=False)  # A, B replaced: the weight stays quantized once
        out[f"{name}.A"], out[f"{name}.B"] = m.A.data.clone(), m.B.data.clone()
    return out
