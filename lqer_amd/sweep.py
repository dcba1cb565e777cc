"""Host-side helpers for running the hot path over many independent Linear units on several GPUs.

The path shards into independent units (every Linear forward on a given input is independent of every
other - SURVEY.md §8e), so there is no collective on the data path: each rank owns a list of units,
builds them from a seed and runs them; torch.distributed (RCCL on GPUs, gloo in the CPU tests) is
used only for the timing barrier, the max-over-ranks of the elapsed time and a checksum gather.

`layer_partition` is the reference's only multi-GPU rule - ceil(L / G) consecutive decoder layers per
device (experiments/infer_device_map.py:29-37) - restated for a fixed model split over ranks.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import torch


def layer_partition(n_layers: int, world: int) -> List[range]:
    """Consecutive layers per rank: rank g gets [g*ceil(L/G), min(L, (g+1)*ceil(L/G)))."""
    if n_layers < 0 or world <= 0:
        raise ValueError("n_layers >= 0 and world > 0 required")
    per = math.ceil(n_layers / world) if n_layers else 0
    return [range(min(n_layers, g * per), min(n_layers, (g + 1) * per)) for g in range(world)]


def projection_units(shapes: Sequence[Tuple[int, int, int]], layers: Sequence[int]) -> List[Tuple[int, int, int, int]]:
    """(layer, K, N, copy) for every projection of the given layers; `shapes` = [(K, N, count per layer)]."""
    return [(l, K, N, c) for l in layers for (K, N, cnt) in shapes for c in range(cnt)]


def unit_seed(rank: int, unit_index: int) -> int:
    """Seed of the synthetic weights of one unit: distinct per rank and unit, reproducible."""
    return 1000 * rank + unit_index


def max_over_ranks(seconds: float, device: torch.device) -> float:
    """Whole-job elapsed time = max over ranks (all-reduce MAX; identity without a process group)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_checksums(value: float, device: torch.device) -> List[float]:
    """One checksum per rank, gathered on every rank (outside any timed region)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def aggregate_throughput(flops_per_rank_step: float, steps: int, world: int, elapsed_max: float) -> float:
    """Weak scaling: every rank does `flops_per_rank_step` per step; TFLOP/s of the whole job."""
    return flops_per_rank_step * world * steps / elapsed_max / 1e12


def broadcast_activation(x: torch.Tensor, src: int = 0) -> torch.Tensor:
    """The one exchange step of a multi-layer sweep: the token batch x [M, K] is generated on rank
    `src` and broadcast to every rank (RCCL over xGMI on GPUs, gloo in the CPU tests), once per distinct K, outside
    any timed region.  In place; identity without a process group."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(x, src=src)
    return x


def gather_rows(values: Sequence[float], device: torch.device) -> List[List[float]]:
    """One row of floats per rank (per-rank time, checksum, unit count ...), gathered on every rank."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [[float(v) for v in values]]
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[float(v) for v in o.tolist()] for o in out]


def sum_over_ranks(value: float, device: torch.device) -> float:
    """Whole-job total of a per-rank quantity (FLOPs per step: ranks of a layer-partitioned sweep own different counts)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def column_partition(n_out: int, world: int, granule: int = 16) -> List[Tuple[int, int]]:
    """Column-parallel split of ONE Linear (SURVEY.md §8e "single-Linear scaling"): rank g owns the output columns
    [n0, n1), cut at multiples of `granule` (16: a B_out block and a packed weight panel are 16 columns, and W's blocks run
    along K - so every rank's y[:, n0:n1] is exactly the unsharded result, no reduction).  ceil(blocks / G) consecutive
    blocks per rank, like the layer rule; trailing ranks may own fewer or no columns."""
    if n_out < 0 or world <= 0 or granule <= 0:
        raise ValueError("n_out >= 0, world > 0 and granule > 0 required")
    blocks = -(-n_out // granule)
    per = math.ceil(blocks / world) if blocks else 0
    return [(min(n_out, g * per * granule), min(n_out, (g + 1) * per * granule)) for g in range(world)]


def all_gather_columns(y_local: torch.Tensor, ranges: Sequence[Tuple[int, int]], n_out: int) -> torch.Tensor:
    """Assemble y [M, n_out] from every rank's column slice y[:, n0:n1] (all-gather over RCCL / gloo, OUTSIDE any timed
    region; slices are padded to the widest one for the collective).  Identity without a process group."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return y_local
    M = y_local.shape[0]
    width = max(n1 - n0 for n0, n1 in ranges)
    buf = torch.zeros(M, width, dtype=y_local.dtype, device=y_local.device)
    buf[:, : y_local.shape[1]] = y_local
    parts = [torch.empty_like(buf) for _ in ranges]
    dist.all_gather(parts, buf)
    out = torch.empty(M, n_out, dtype=y_local.dtype, device=y_local.device)
    for (n0, n1), p in zip(ranges, parts):
        out[:, n0:n1] = p[:, : n1 - n0]
    return out
