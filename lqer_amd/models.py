"""Model-swap helper: put the HIP quantized Linear into a stock HuggingFace decoder.

Counterpart of the reference's `quantize_model` (src/lqer/models/__init__.py:21) and of the wiring in
`run_evaluate_perplexity` (src/lqer/runners.py:212-222).  The reference copies whole decoder-layer classes
with the Linear constructor swapped (llama_decoder.py:62-145, opt_decoder.py:21-236) and reloads the
original weights with strict=False (llama_decoder.py:494-508); here the `nn.Linear` children of every
decoder layer are replaced *by name*, which yields the same module paths and therefore the same state-dict
keys (`model.layers.<i>.self_attn.q_proj.{weight,A,B}` ...), so HF checkpoints and the reference's
`low_rank_dict.pt` load unchanged.  Works with whatever transformers version is installed.

Config expansion follows llama_decoder.py:423-482: `q_config["linear"]` / `l_config["linear"]` apply to every
projection of every layer unless a `model_layer_<i>` entry overrides them.  The attention matmuls
(`q_config["matmul"]`, reference quantized_functions/matmul.py) are switched on separately:
lqer_amd.attention.enable_quantized_attention.
"""
from __future__ import annotations

from copy import deepcopy
from typing import Dict, Iterable, Optional, Tuple

import torch
import torch.nn as nn

from .linear import LinearFlexibleLqer, SharedActivation, get_quantized_layer_cls

# decoder-layer children that the reference quantizes, by model family
_LLAMA_LIKE = {"self_attn": ("q_proj", "k_proj", "v_proj", "o_proj"), "mlp": ("gate_proj", "up_proj", "down_proj")}
_OPT = {"self_attn": ("q_proj", "k_proj", "v_proj", "out_proj"), "": ("fc1", "fc2")}


def _decoder_layers(model: nn.Module) -> Tuple[nn.ModuleList, Dict[str, Tuple[str, ...]]]:
    base = getattr(model, "model", model)
    if hasattr(base, "layers"):  # Llama / Mistral (llama_decoder.py:494, mistral_decoder.py:594)
        return base.layers, _LLAMA_LIKE
    if hasattr(base, "decoder") and hasattr(base.decoder, "layers"):  # OPT (opt_decoder.py:383)
        return base.decoder.layers, _OPT
    raise ValueError(f"unsupported model class {type(model).__name__}: no decoder layer list found")


def _layer_cfg(cfg: Optional[dict], layer_id: int, group: str, name: str) -> Optional[dict]:
    """Per-projection config: `model_layer_<i>` override, else the `model_layer` template, else `linear`."""
    if cfg is None:
        return None
    for key in (f"model_layer_{layer_id}", "model_layer"):
        entry = cfg.get(key)
        if entry is not None:
            node = entry.get(group, entry) if group else entry
            if name in node:
                return deepcopy(node[name])
    return deepcopy(cfg["linear"])


# projections of one decoder layer that are fed the same tensor (llama_decoder.py:246-248 q/k/v, :176 gate/up;
# opt_decoder.py q/k/v)
_SHARED_INPUTS = {"self_attn": (("q_proj", "k_proj", "v_proj"),), "mlp": (("gate_proj", "up_proj"),)}


def quantize_model(model: nn.Module, q_config: dict, l_config: Optional[dict], share_inputs: bool = True) -> nn.Module:
    """Replace the projections of every decoder layer in place; weights are carried over.  share_inputs: projections
    fed by the same tensor quantize it once and share one side GEMM (linear.SharedActivation); per-projection results
    are unchanged up to the fp32 summation order of the side product."""
    layers, table = _decoder_layers(model)
    for layer_id, layer in enumerate(layers):
        for group, names in table.items():
            parent = getattr(layer, group) if group else layer
            for name in names:
                old = getattr(parent, name)
                if not isinstance(old, nn.Linear):
                    raise TypeError(f"layer {layer_id} {group}.{name} is {type(old).__name__}, expected nn.Linear")
                qc = _layer_cfg(q_config, layer_id, group, name)
                lc = _layer_cfg(l_config, layer_id, group, name)
                cls = get_quantized_layer_cls("linear", qc)
                new = cls(old.in_features, old.out_features, bias=old.bias is not None, q_config=qc, l_config=lc)
                new.to(device=old.weight.device, dtype=old.weight.dtype)
                new.load_state_dict(old.state_dict(), strict=False)  # A, B stay zero until loaded
                setattr(parent, name, new)
            if share_inputs:
                for names2 in _SHARED_INPUTS.get(group, ()):
                    mods = [getattr(parent, n, None) for n in names2]
                    if all(isinstance(m, LinearFlexibleLqer) for m in mods):
                        SharedActivation(mods)
    return model


def load_low_rank_dict(model: nn.Module, low_rank_dict: Dict[str, torch.Tensor]) -> Iterable[str]:
    """`low_rank_dict.pt` -> the A / B parameters (runners.py:220-222: cast to the model dtype, strict=False).
    Returns the keys that did not match any parameter."""
    p = next(iter(model.parameters()))
    sd = {k: v.to(dtype=p.dtype) for k, v in low_rank_dict.items()}
    missing_ok = model.load_state_dict(sd, strict=False)
    return list(missing_ok.unexpected_keys)
