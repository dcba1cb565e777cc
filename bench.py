#!/usr/bin/env python3
"""Benchmark of the LQER quantized-Linear hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c4row|c5|c4a16|c2int|c2introw|c3int|c2w8a8|c2w8a8m8k|d1|d16|d1a16|d1layer]
                    [--sweep auto|weak|strong] [--shard none|n] [--no-configs] [--no-cpu-baseline] [--no-check] [--no-module]
                    [--graph G] [--prewarm-ms T] [--dry-run-cpu]

A step = one pass of the hot path (x fp16 in -> activation quantize -> rank-r side GEMM -> fused W4A8 GEMM -> y fp16
out) over one batch of synthetic tokens for every Linear unit this rank owns, inputs resident in HBM.  Default
workload = BASELINE.json configs[1]: one LqerLinear 4096 -> 4096, rank 32, W4A8 MXINT (block 16), M = 2048 tokens.
On the default line (one GPU, workload c2) the JSON also carries `configs`: every other BASELINE configuration at its full token
count - c3 (Llama-7B, all 32 layers x 7 projections), c3int (the same shapes with the reference's INT template: int8 MFMA), c4
(Llama-13B W4A8-INT rank 64, M = 16384: 8 of its 40 layers), c5 (OPT-6.7B rank 128, all 32 layers x 6 projections) - each with its
own packed images per Linear, timed region, per-shape dominant-kernel time / fraction of peak, the sustained shader clock and
oracle parity, so that every BASELINE configuration is under the driver's clock.  With --gpus N > 1 and no --workload the
headline is BASELINE's multi-GPU configuration (c4, layers split over the ranks) and `replicas_c2` the N independent C2 Linears.

Multi-GPU (one rank per GPU).  Either the driver launches the ranks (`python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N`: RANK / LOCAL_RANK / WORLD_SIZE in the environment), or a plain `python bench.py
--gpus N` starts them itself: before anything touches the GPU it runs that very launcher as a CHILD process (never an
exec), relays the child's output - rank 0's one JSON line - and exits with its return code.  `--dry-run-cpu` runs the
same host logic (partition, broadcast, barrier-bracketed timed region, max over ranks, gather, one JSON line) on gloo
without a GPU and without a kernel: the plumbing check of tests/test_multirank_cpu.py.  The path shards into independent Linear
units.  Rank 0 generates the token batch x once per distinct K and BROADCASTS it (RCCL, outside the timed region, timed
separately as `broadcast_ms`); every rank builds the weights of its own units from a seed; per-rank times and checksums
are GATHERED after the timed region.  No collective on the data path.
  * model sweeps (c3 / c4 / c5): the model's decoder layers are split over the ranks by the reference's rule - rank g
    owns layers [g ceil(L/G), (g+1) ceil(L/G)) (experiments/infer_device_map.py:29-37, lqer_amd.sweep.layer_partition)
    - total work is fixed: "scaling": "strong".  --sweep weak gives every rank the whole model instead.
  * single-Linear workloads (c2, d1, d16): every rank runs its own Linear of that shape on the broadcast batch (N
    independent units - e.g. the same projection of N layers): "scaling": "weak".  `--shard n` instead splits that ONE
    Linear column-parallel (optional row): rank g packs and runs W[n0:n1], B[:, n0:n1] - cuts at multiples
    of 16, exact with no reduction - and y[:, n0:n1] is all-gathered outside the timed region: "scaling": "strong".

Prints ONE JSON line on rank 0 (driver contract) with these extra objects: "roofline" (dominant kernel: algorithmic
FLOPs / its HIP-event time inside the timed region, against the dense MFMA peak of the main loop's operand type),
"cpu_baseline" (the CPU oracle timed on this box's host cores on a bounded sample), "module" (the same K steps timed
through the drop-in nn.Module, `mod(x)` - the boundary the reference's callers use; `value` is the C-ABI figure),
"uninstrumented" (the K steps once more without the roofline's event pairs) and "parity_rel_l2" (row slices of the
outputs the timed kernels just wrote, against the CPU oracle).

Timing: setup (packing, plans, broadcast), an untimed device clock ramp of --prewarm-ms (300 ms: after idling the GPU
needs tens of milliseconds of load to reach the clocks it then holds, and the default C2 run is only ~5 ms long), the W
untimed warm-up steps, then exactly K timed steps between barrier + synchronize on both sides, max over ranks.

The pieces live in benchlib/: workloads.py (configurations, synthetic operands), launcher.py (self-launch, CPU dry run),
runner.py (setup + timed regions of one workload), roofline.py, hipevents.py, cpu_baseline.py.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# (re-exported: tests, tools and __graft_entry__.smoke() import the configurations and operand generators from here)
from benchlib.cpu_baseline import cpu_baseline  # noqa: E402,F401
from benchlib.hipevents import HipEvent  # noqa: E402,F401
from benchlib.launcher import dry_run_cpu, launch_ranks  # noqa: E402
from benchlib.workloads import (A16_Q, BF16_MFMA_PEAK_TFLOPS, HBM_PEAK_GBS, INT8_MFMA_PEAK_TOPS, INT_Q, INTROW_Q,  # noqa: E402,F401
                                LLAMA13B, MXINT_Q, OPT_Q, UNQUANTIZED_AB, W3A16_Q, W8A8_Q, WORKLOADS, _bfp, _snap_mxint8_dim0, check_rows, flops,
                                make_case, make_weights, make_x)

# every other BASELINE configuration, carried by the default line: (workload, steps, warm-up steps, decoder layers: 0 = the whole
# model).  c3 / c3int / c5 run at FULL depth (32 layers, own packed images per Linear: ~25 ms per step), c4 (M = 16384: 5 ms per
# layer) at 8 of its 40 layers - VERDICT r4 item 5
CONFIG_LAYERS = (("c3", 10, 3, 0), ("c3int", 10, 3, 0), ("c4", 5, 1, 8), ("c5", 10, 3, 0))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 50; 4 for the M=16384 model sweeps)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default 10; 1 for the M=16384 sweeps)")
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: c2 (BASELINE configs[1]) on one GPU; on several GPUs c4 - BASELINE's multi-GPU configuration (configs[3]: "
                         "Llama-13B, layers split over the ranks, strong scaling) - with the c2 replicas as the secondary `replicas_c2`")
    ap.add_argument("--sweep", default="auto", choices=["auto", "weak", "strong"],
                    help="multi-GPU: strong = the model's layers split over the ranks (default for c3/c4/c5), weak = every "
                         "rank runs the full unit list (default for single-Linear workloads)")
    ap.add_argument("--shard", default="none", choices=["none", "n"],
                    help="n: single-Linear workloads column-parallel - rank g owns the output columns [n0, n1) (cuts at multiples "
                         "of 16: exact, no reduction), y all-gathered outside the timed region")
    ap.add_argument("--layers", type=int, default=0, help="override the model's decoder layer count (profiling runs: --layers 1)")
    ap.add_argument("--no-configs", action="store_true",
                    help="default line only: skip the `configs` object (one decoder layer of c3 / c4 / c5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the oracle comparison of the timed outputs")
    ap.add_argument("--no-module", action="store_true", help="skip the second timed region through the nn.Module")
    ap.add_argument("--tuning", type=lambda v: int(v, 0), default=0,
                    help="lqer_linear_desc_t.tuning for every Linear (LQER_TUNE_* bits of include/lqer_hip.h: kernel-variant A/B, same bits), "
                         "e.g. 0x800000 = the block-16 activation side as two launches, 0x200000 = the int8 route's as three")
    ap.add_argument("--prewarm-ms", type=float, default=300.0, help="untimed device clock ramp before the warm-up steps (0 = none)")
    ap.add_argument("--graph", type=int, default=0, metavar="G",
                    help="capture G consecutive steps in one hipGraph and replay it steps/G times (launch-bound decode sizes; "
                         "G ~ the number of Linears a model pushes a token through)")
    ap.add_argument("--rotate", type=int, default=None, metavar="R",
                    help="decode workloads (M <= 64): walk R distinct copies of the packed operands round robin, so that the "
                         "weight stream comes from HBM and not from the 256 MB Infinity Cache (default 48 = 451 MB of 4096 x 4096 "
                         "images; 0 = one resident weight); the resident figure is reported beside it")
    ap.add_argument("--no-two-streams", action="store_true",
                    help="skip the secondary figure with the step's independent forwards issued alternately on two HIP streams")
    ap.add_argument("--shared-weights", action="store_true",
                    help="model sweeps: re-run ONE packed image per projection shape for every layer (the round-1/2 behaviour: "
                         "~60 MB of weights that never leave the Infinity Cache) instead of one distinct copy per Linear of the "
                         "model (default: a Llama-7B rank walks 3.6 GB of packed operands per step, as the model does)")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="no GPU, no kernel: the multi-rank host logic alone (launcher, partition, gloo broadcast / gather, "
                         "timed-region protocol, the JSON line) - value is null")
    return ap.parse_args(argv)


def config_layers(ctx, args):
    """`configs` of the default line: one decoder layer of c3 / c4 / c5 at the BASELINE token count - own packed images per
    Linear, >= 3 timed steps, per-shape dominant-kernel time and fraction of peak (HIP events inside the timed region),
    oracle parity of the buffers just timed.  Failures are reported in place: they never cost the headline."""
    from benchlib.runner import Opts, run_workload

    out = {}
    for name, steps, warm, nlayers in CONFIG_LAYERS:
        t0 = time.perf_counter()
        try:
            rec = run_workload(ctx, Opts(workload=name, steps=steps, warmup=warm, layers=nlayers, check=not args.no_check, module=False,
                                         two_streams=False, cpu_base=False, prewarm_ms=100.0, uninstrumented=False))
            rl = rec["roofline"]
            out[name] = {"workload": rec["config"]["workload"], "value": rec["value"], "unit": rec["unit"],
                         "ms_per_step": rec["ms_per_step"], "steps": rec["steps"], "warmup": rec["warmup"], "dtype": rec["dtype"],
                         "tokens_per_step": rec["config"]["tokens_per_step"], "kernel": rl["kernel"], "frac": rl["frac"],
                         "peak": rl["peak"], "achieved": rl["achieved"], "roofline_unit": rl["unit"],
                         "avg_launch_us": rl["avg_launch_us"], "launches": rl["launches"], "per_shape": rl["per_shape"],
                         "sustained_mhz": rl.get("sustained_mhz"), "frac_at_sustained_clock": rl.get("frac_at_sustained_clock"),
                         "inside_load": rl.get("inside_load"),
                         "layers": rec["config"]["layers_per_rank"][0], "weights": rec["config"]["weights"],
                         "parity_rel_l2": rec["parity_rel_l2"], "parity_rows": rec["parity_rows"]}
        except (Exception, SystemExit) as e:  # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        out[name]["wall_s"] = round(time.perf_counter() - t0, 1)
        print(f"# configs[{name}]: {out[name]}", file=sys.stderr)
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: start the ranks ourselves, as a child process, BEFORE any GPU call of this process
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], __file__))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world
    # no --workload: one GPU -> c2 (BASELINE configs[1]); several -> BASELINE's own multi-GPU configuration, c4 (configs[3]: layers
    # split over the ranks by infer_device_map.py:29-37, strong scaling), the c2 replicas (weak scaling: comparable with the
    # one-GPU headline) as a secondary record
    default_workload = args.workload is None
    if default_workload:
        args.workload = "c2" if (world == 1 or args.shard != "none") else "c4"  # (--shard n splits ONE Linear: c2)
    if args.dry_run_cpu:
        return dry_run_cpu(args, rank, world)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the hot path has no CPU fallback)")
    # LQER_BENCH_REHEARSE=1: every rank on cuda:0 with gloo carrying the collectives (RCCL refuses two ranks on one device) -
    # the N > 1 code path with real kernels on a one-GPU box; the figures of such a run mean nothing
    rehearse = os.environ.get("LQER_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # (LQER_BENCH_FORCE_DIST=1 under a launcher: a single rank still goes through RCCL - the rehearsal a one-GPU box allows)
    if world > 1 or (os.environ.get("LQER_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ):
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from benchlib.runner import Ctx, Opts, run_workload

    ctx = Ctx(rank=rank, world=world, dev=dev, dist=dist)
    opts = Opts(workload=args.workload, steps=args.steps, warmup=args.warmup, layers=args.layers, sweep=args.sweep, shard=args.shard,
                check=not args.no_check, module=not args.no_module, two_streams=not args.no_two_streams,
                cpu_base=not args.no_cpu_baseline, prewarm_ms=args.prewarm_ms, graph=args.graph, rotate=args.rotate,
                shared_weights=args.shared_weights, tuning=args.tuning)
    out = run_workload(ctx, opts)
    default_line = (world == 1 and args.workload == "c2" and args.layers == 0 and args.shard == "none" and not args.graph
                    and not args.no_configs)
    replicas = None
    if world > 1 and default_workload and args.shard == "none" and not args.graph:
        # every rank its own C2 Linear on the broadcast batch (N independent units): the weak-scaling figure next to the headline
        rep = run_workload(ctx, Opts(workload="c2", steps=None, warmup=None, check=not args.no_check, module=False, two_streams=False,
                                     cpu_base=False, prewarm_ms=100.0, uninstrumented=False))
        if rank == 0:
            replicas = {k: rep[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "scaling", "tokens_per_s", "parity_rel_l2",
                                            "rank_ms_per_step")}
            replicas["workload"] = rep["config"]["workload"]
            replicas["frac"] = rep["roofline"]["frac"]
    if rank == 0:
        if default_line:
            out["configs"] = config_layers(ctx, args)
        if replicas is not None:
            out["replicas_c2"] = replicas
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
