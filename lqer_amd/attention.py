"""Quantized attention matmuls inside a stock HuggingFace decoder (wiring).

The reference's decoder copies replace the two matmuls of eager attention by `matmul_flexible`
(llama_decoder.py:259-297: Q K^T on [b*h, s, d] x [b*h, d, s], scaled AFTER the quantized product; softmax in fp32;
P V on [b*h, s, s] x [b*h, s, d]; opt_decoder.py:125,190 likewise with bmm).  Here the same computation is registered
as an attention implementation of the installed transformers (AttentionInterface), so no model class is copied:
`enable_quantized_attention(model, q_config)` selects it and stores the per-layer matmul configs on the attention
modules.  Operands are flattened to 3-D like the reference does (its quantizer rejects 4-D tensors); blocks run along
the last dim of each operand.
"""
from __future__ import annotations

from copy import deepcopy
from typing import Optional

import torch
import torch.nn as nn

from .functional import matmul_flexible

IMPLEMENTATION = "lqer_eager"


def _repeat_kv(t: torch.Tensor, n_rep: int) -> torch.Tensor:
    if n_rep == 1:
        return t
    b, h, s, d = t.shape
    return t[:, :, None, :, :].expand(b, h, n_rep, s, d).reshape(b, h * n_rep, s, d)


def lqer_eager_attention_forward(module: nn.Module, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                                 attention_mask: Optional[torch.Tensor], scaling: float, dropout: float = 0.0, **kwargs):
    cfg0, cfg1 = module._lqer_matmul_cfg
    key_states = _repeat_kv(key, getattr(module, "num_key_value_groups", 1))
    value_states = _repeat_kv(value, getattr(module, "num_key_value_groups", 1))
    b, h, s, d = query.shape
    t = key_states.shape[2]
    scores = matmul_flexible(query.reshape(b * h, s, d), key_states.reshape(b * h, t, d).transpose(1, 2), cfg0)
    attn_weights = scores.reshape(b, h, s, t) * scaling
    if attention_mask is not None:
        attn_weights = attn_weights + attention_mask
    attn_weights = nn.functional.softmax(attn_weights, dim=-1, dtype=torch.float32).to(query.dtype)
    attn_weights = nn.functional.dropout(attn_weights, p=dropout, training=module.training)
    out = matmul_flexible(attn_weights.reshape(b * h, s, t), value_states.reshape(b * h, t, d), cfg1)
    return out.reshape(b, h, s, d).transpose(1, 2).contiguous(), attn_weights


def _register() -> None:
    from transformers import AttentionInterface
    from transformers.masking_utils import AttentionMaskInterface, eager_mask

    AttentionInterface.register(IMPLEMENTATION, lqer_eager_attention_forward)
    AttentionMaskInterface.register(IMPLEMENTATION, eager_mask)


def enable_quantized_attention(model: nn.Module, q_config: dict) -> nn.Module:
    """Route every decoder layer's attention through matmul_flexible.  q_config["matmul"] applies to both products of
    every layer unless `model_layer_<i>` / `model_layer` carry `self_attn: {matmul_0, matmul_1}` overrides
    (llama_decoder.py:423-482); OPT models read q_config["bmm"] and `bmm_0` / `bmm_1` instead, as the reference's OPT
    decoder does (opt_decoder.py:125,190,329-339).  Model families without the attention-interface hook raise."""
    from .models import _OPT, _decoder_layers

    _register()
    layers, table = _decoder_layers(model)
    op = "bmm" if table is _OPT else "matmul"
    base = q_config[op]
    for i, layer in enumerate(layers):
        attn = layer.self_attn
        cfgs = []
        for name in (f"{op}_0", f"{op}_1"):
            cfg = base
            for key in (f"model_layer_{i}", "model_layer"):
                entry = q_config.get(key)
                if entry is not None and name in entry.get("self_attn", {}):
                    cfg = entry["self_attn"][name]
                    break
            cfgs.append(deepcopy(cfg))
        attn._lqer_matmul_cfg = tuple(cfgs)
    if not hasattr(model, "set_attn_implementation"):
        raise NotImplementedError(f"{type(model).__name__}: no attention-implementation switch in this transformers version")
    model.set_attn_implementation(IMPLEMENTATION)
    return model
