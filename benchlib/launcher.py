"""`python bench.py --gpus N` without a launcher, and the CPU / gloo dry run of the multi-rank host logic."""
from __future__ import annotations

import json
import os
import sys
import time

import torch

from .workloads import WORKLOADS, flops, make_x


def launch_ranks(n, argv, script):
    """Run `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a CHILD process (this process
    has not touched the GPU and never will - never an exec), pass its stdout - rank 0's JSON line - and stderr through
    unchanged, return its exit code.  The rule the ranks then follow is the reference's consecutive-layers-per-device split
    (experiments/infer_device_map.py:29-37)."""
    import socket
    import subprocess

    with socket.socket() as s:  # a free rendezvous port: two benches on one node must not meet
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(script), *argv]
    print("# bench.py: launching %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr)
    return subprocess.run(cmd, env=env).returncode


def dry_run_cpu(args, rank, world):
    """--dry-run-cpu: everything of a multi-rank run that is not a kernel, on gloo / CPU tensors - layer partition (or the
    column split of --shard n), broadcast of the token batch once per distinct K, barrier-bracketed timed region, max / sum
    over ranks, gather, ONE JSON line on rank 0.  The step is a no-op (the hot path has no CPU form), so `value` is null and
    the line says so."""
    from lqer_amd import sweep

    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    dev = torch.device("cpu")
    desc_txt, M, r, has_bias, qc, shapes, layers = WORKLOADS[args.workload]
    if args.layers > 0:
        layers = args.layers
    M = min(M, 64)  # the plumbing does not depend on the token count
    steps = args.steps or 2
    shard_n = getattr(args, "shard", "none") == "n"
    strong = shard_n or (args.sweep == "strong") or (args.sweep == "auto" and layers > 1)
    if strong and not shard_n and layers == 1 and world > 1:
        sys.exit("--sweep strong needs a model workload (c3/c4/c5): a single Linear has no layers to split")
    my_layers = range(layers) if shard_n else (sweep.layer_partition(layers, world)[rank] if strong else range(layers))
    xs, broadcast_ms = {}, 0.0
    for K in sorted({K for K, _, _ in shapes}):
        xd = make_x(M, K, seed=0)[0].half() if rank == 0 else torch.full((M, K), float("nan"), dtype=torch.float16)
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        sweep.broadcast_activation(xd, src=0)
        broadcast_ms += (time.perf_counter() - t0) * 1e3
        xs[K] = xd
    units = sweep.projection_units(shapes, my_layers)
    col_ranges = None
    if shard_n:  # column-parallel: rank g owns the columns [n0, n1) of every Linear (multiples of 16: B_out blocks stay whole)
        col_ranges = {N: sweep.column_partition(N, world)[rank] for _, N, _ in shapes}
        units = [(l, K, col_ranges[N][1] - col_ranges[N][0], c) for l, K, N, c in units]
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        pass  # (no kernel without a GPU)
    if dist is not None:
        dist.barrier()
    elapsed_rank = time.perf_counter() - t0
    elapsed = sweep.max_over_ranks(elapsed_rank, dev)
    xsum = float(sum(v.float().sum().item() for v in xs.values()))
    gathered = sweep.gather_rows([elapsed_rank * 1e3 / steps, xsum, float(len(my_layers))], dev)
    flops_all = sweep.sum_over_ranks(float(sum(flops(M, K, N, r) for _, K, N, _ in units)), dev)
    if rank == 0:
        print(json.dumps({
            "metric": "W4A8+rank-r Linear GEMM TFLOPS-equiv", "value": None, "unit": "TFLOP/s-equiv", "n_gpus": world,
            "steps": steps, "warmup": 0, "ms_per_step": round(elapsed / steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": None, "data": "synthetic",
            "dry_run": "cpu/gloo: launcher, partition, broadcast, timed-region protocol and gather only - no kernel ran",
            "config": {"workload": desc_txt + " [dry run, M=%d]" % M, "tokens_per_step": M, "rank": r,
                       "layers_per_rank": [int(row[2]) for row in gathered],
                       "column_shard": None if col_ranges is None else {str(N): list(rg) for N, rg in col_ranges.items()}},
            "broadcast_ms": round(broadcast_ms, 3), "flops_per_step_all_ranks": flops_all,
            "rank_ms_per_step": [round(row[0], 4) for row in gathered],
            "rank_checksums": [round(row[1], 3) for row in gathered]}))
    if dist is not None:
        dist.destroy_process_group()
    return 0
