"""On-disk packed checkpoint.  The reference has no such format: it re-quantizes every weight from
fp16 at each load (quantized_layers/linear.py:149-153).  Here the derived operands of every quantized Linear - 4-bit
weight panels with their block exponents (0.5625 B per weight), the exact bf16 limbs of A^T and B^T, the quantized
bias - are written once and attached at load time, so a load touches 0.28x the bytes of the fp16 checkpoint and the
dense weight / A / B tensors need not be materialised on the GPU at all.

File = one safetensors file: tensors "<module path>.<w|a_t|b_t|bias_q|header>" (uint8 / int32) for every quantized
Linear, every other parameter and buffer of the model under its state-dict key, and a metadata header
{"format": "lqer_amd.packed", "version": "1"}.  Each module header carries (version, in, out, rank, limb counts, bias
flag, the five quantizer settings); load_packed refuses a file whose header disagrees with the module it is loaded into.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn as nn

from .linear import _LinearBase

FORMAT = "lqer_amd.packed"
VERSION = "1"
_FIELDS = ("w", "a_t", "b_t", "bias_q", "header")


def _quantized_modules(model: nn.Module) -> Dict[str, _LinearBase]:
    return {name: m for name, m in model.named_modules() if isinstance(m, _LinearBase)}


def packed_state_dict(model: nn.Module) -> Dict[str, torch.Tensor]:
    """All tensors of the packed checkpoint, on the CPU.  The quantized Linears must live on the GPU (packing runs
    there); their dense weight / bias / A / B are left out, everything else is copied from state_dict()."""
    out: Dict[str, torch.Tensor] = {}
    qmods = _quantized_modules(model)
    skip = set()
    for name, m in qmods.items():
        for k, v in m.packed_state().items():
            out[f"{name}.{k}"] = v.detach().cpu().contiguous()
        for pname in ("weight", "bias", "A", "B"):
            skip.add(f"{name}.{pname}")
    for k, v in model.state_dict().items():
        if k not in skip:
            out[k] = v.detach().cpu().contiguous()
    return out


def save_packed(model: nn.Module, path: str) -> int:
    """Write the packed checkpoint; returns the number of quantized Linears written."""
    from safetensors.torch import save_file

    tensors = packed_state_dict(model)
    n = sum(1 for k in tensors if k.endswith(".header"))
    # safetensors refuses aliased storage: state_dict() of tied embeddings shares memory
    seen, uniq = {}, {}
    for k, v in tensors.items():
        key = (v.data_ptr(), v.numel(), v.dtype)
        uniq[k] = v.clone() if key in seen and v.numel() else v
        seen[key] = k
    save_file(uniq, path, metadata={"format": FORMAT, "version": VERSION, "quantized_linears": str(n)})
    return n


def load_packed(model: nn.Module, path: str, device="cuda") -> List[str]:
    """Attach a packed checkpoint to a model whose Linears have been swapped (models.quantize_model).  Quantized
    Linears receive their packed images on `device`; all other tensors go through load_state_dict(strict=False).
    Returns the state-dict keys of the model that the file did not provide (dense weight / A / B of the quantized
    Linears excepted - those are not needed any more)."""
    from safetensors import safe_open

    dev = torch.device(device)
    qmods = _quantized_modules(model)
    rest: Dict[str, torch.Tensor] = {}
    per_mod: Dict[str, Dict[str, torch.Tensor]] = {}
    with safe_open(path, framework="pt", device="cpu") as f:
        meta = f.metadata() or {}
        if meta.get("format") != FORMAT:
            raise RuntimeError(f"{path}: not a {FORMAT} file (metadata {meta})")
        if meta.get("version") != VERSION:
            raise RuntimeError(f"{path}: packed checkpoint version {meta.get('version')}, this build reads {VERSION}")
        for key in f.keys():
            mod, _, field = key.rpartition(".")
            if mod in qmods and field in _FIELDS:
                per_mod.setdefault(mod, {})[field] = f.get_tensor(key)
            else:
                rest[key] = f.get_tensor(key)
    missing_mods = [n for n in qmods if n not in per_mod]
    if missing_mods:
        raise RuntimeError(f"{path}: no packed images for {missing_mods[:4]}{'...' if len(missing_mods) > 4 else ''}")
    for name, st in per_mod.items():
        qmods[name].load_packed_state(st, dev)
    res = model.load_state_dict(rest, strict=False)
    dense = {f"{n}.{p}" for n in qmods for p in ("weight", "bias", "A", "B")}
    return [k for k in res.missing_keys if k not in dense]
