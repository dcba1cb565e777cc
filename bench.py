#!/usr/bin/env python3
"""Benchmark of the LQER quantized-Linear hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c4row|c5|c4a16|d1|d16|d1a16]
                    [--sweep auto|weak|strong] [--no-cpu-baseline] [--no-check] [--no-module] [--graph G] [--prewarm-ms T]
                    [--dry-run-cpu]

A step = one pass of the hot path (x fp16 in -> activation quantize -> rank-r side GEMM -> fused W4A8 GEMM -> y fp16
out) over one batch of synthetic tokens for every Linear unit this rank owns, inputs resident in HBM.  Default
workload = BASELINE.json configs[1]: one LqerLinear 4096 -> 4096, rank 32, W4A8 MXINT (block 16), M = 2048 tokens.

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
    independent units - e.g. the same projection of N layers): "scaling": "weak".

Prints ONE JSON line on rank 0 (driver contract) with these extra objects: "roofline" (dominant kernel: algorithmic
FLOPs / its HIP-event time inside the timed region, against the dense MFMA peak of the main loop's operand type),
"cpu_baseline" (the CPU oracle timed on this box's host cores on a bounded sample), "module" (the same K steps timed
through the drop-in nn.Module, `mod(x)` - the boundary the reference's callers use; `value` is the C-ABI figure) and
"parity_rel_l2" (row slices of the outputs the timed kernels just wrote, against the CPU oracle).

Timing: setup (packing, plans, broadcast), an untimed device clock ramp of --prewarm-ms (300 ms: after idling the GPU
needs tens of milliseconds of load to reach the clocks it then holds, and the default C2 run is only ~5 ms long), the W
untimed warm-up steps, then exactly K timed steps between barrier + synchronize on both sides, max over ranks.
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


def _bfp(width, block, skip):
    return dict(name="block_fp", width=width, exponent_width=8, exponent_bias=None, block_size=block, skip_first_dim=skip)


# reference experiments/configs/template/llama-7b.toml:78-105 (W4A8 MXINT, blocks of 16)
MXINT_Q = dict(name="flexible_lqer", is_ptq=True, default=False, x_quantizer=_bfp(8, [1, 16], True),
               w_quantizer=_bfp(4, [1, 16], False), b_quantizer=_bfp(8, [-1], False))
# opt-6.7b.toml:98-102: bias in blocks of 16
OPT_Q = dict(MXINT_Q, b_quantizer=_bfp(8, [1, 16], False))

# reference sweep_lqer_act_int.sh:83 / llama-7b-int.toml (W block 128, A/B unquantized fp16) with the 8-bit
# per-token activation format BASELINE.json's "W4A8 L2QER-INT" pins: block_fp(8, [1,-1])
INT_Q = dict(name="flexible_lqer", is_ptq=True, default=False, x_quantizer=_bfp(8, [1, -1], True),
             w_quantizer=_bfp(4, [1, 128], False), b_quantizer=dict(name="passthrough"))
# the same with one weight block per row (llama-7b-int.toml:87, block_size [1, -1])
INTROW_Q = dict(INT_Q, w_quantizer=_bfp(4, [1, -1], False))

# the INT templates as shipped (llama-7b-int.toml q_config.linear): pass-through fp16 activations ("W4A16"), A_out and
# B_out falling back to the same pass-through (linear.py:115-124), A/B unquantized
A16_Q = dict(INT_Q, x_quantizer=dict(name="passthrough", width=16, frac_width=12))
UNQUANTIZED_AB = (INT_Q, INTROW_Q, A16_Q)

BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
INT8_MFMA_PEAK_TOPS = 5000.0    # 2x bf16 per clock (same guide, "Matrix cores", I8 row)
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md, HBM3E

LLAMA13B = [(5120, 5120, 4), (5120, 13824, 2), (13824, 5120, 1)]
WORKLOADS = {
    # name: (description, M, rank, bias, q_config, [(K, N, count per layer)], decoder layers of the model)
    "c2": ("LqerLinear 4096x4096 rank32 W4A8-MXINT16 M=2048 (BASELINE configs[1])", 2048, 32, False, MXINT_Q, [(4096, 4096, 1)], 1),
    "c3": ("Llama-7B 7 projections x 32 layers rank32 W4A8-MXINT16 M=2048 (BASELINE configs[2])", 2048, 32, False, MXINT_Q,
           [(4096, 4096, 4), (4096, 11008, 2), (11008, 4096, 1)], 32),
    "c4": ("Llama-13B 7 projections x 40 layers rank64 W4(block128)A8(per-token) M=16384 (BASELINE configs[3])", 16384, 64, False,
           INT_Q, LLAMA13B, 40),
    "c4row": ("Llama-13B 7 projections x 40 layers rank64 W4(one block per row, llama-7b-int.toml:87)A8(per-token) M=16384", 16384, 64,
              False, INTROW_Q, LLAMA13B, 40),
    "c5": ("OPT-6.7B 6 projections x 32 layers rank128 W4A8-MXINT16 M=2048 (BASELINE configs[4])", 2048, 128, True, OPT_Q,
           [(4096, 4096, 4), (4096, 16384, 1), (16384, 4096, 1)], 32),
    "c4a16": ("Llama-13B 7 projections x 40 layers rank64 W4(block128)A16 (the reference's INT template as shipped) M=16384",
              16384, 64, False, A16_Q, LLAMA13B, 40),
    "d1a16": ("LqerLinear 4096x4096 rank32 W4(block128)A16 M=1 (decode)", 1, 32, False, A16_Q, [(4096, 4096, 1)], 1),
    # decode sizes (HBM-bound on the packed weight; roofline quoted in GB/s): the small-M kernel
    "d1": ("LqerLinear 4096x4096 rank32 W4A8-MXINT16 M=1 (decode)", 1, 32, False, MXINT_Q, [(4096, 4096, 1)], 1),
    "d16": ("LqerLinear 4096x4096 rank32 W4A8-MXINT16 M=16 (decode)", 16, 32, False, MXINT_Q, [(4096, 4096, 1)], 1),
}


def flops(M, K, N, r):
    """Reference multiply model (experiments/hw_performance/README.md:81-106) x 2."""
    return 2 * M * K * N + 2 * M * K * r + 2 * M * r * N


def _snap_mxint8_dim0(t):
    """t -> the 8-bit MXINT grid with blocks of 16 along dim 0 (the reference approximator's A / B format,
    llama-7b.toml:60-73).  On a GPU box this is the library's own HIP quantizer; without a GPU (CPU-only tools and
    tests) the CPU oracle's.  Setup of synthetic inputs only - any values would do."""
    if torch.cuda.is_available():
        from lqer_amd import ops

        fmt = ops.make_qfmt(_bfp(8, [1, 16], True), "x")
        return ops.quantize_mxint(t.t().contiguous().cuda(), fmt, want=("deq",))["deq"].t().contiguous().cpu()
    from oracle import lqer_oracle as O

    return O.mxint_quantize(t, width=8, block_size=[16, 1], skip_first_dim=False)


def make_x(M, K, seed=0):
    """Synthetic token batch of SURVEY.md §8d: x ~ N(0,1) with three x30 outlier channels."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K, generator=g)
    for c in (7, 1033, 2900):
        if c < K:
            x[:, c] *= 30.0
    return x, g


def make_weights(g, K, N, r, bias=False, quantize_ab=True):
    """W ~ N(0, 0.02^2); A, B ~ 0.01 N(0,1), snapped to the 8-bit MXINT grid for the MXINT configurations, left
    unquantized for the INT ones (llama-7b-int.toml:61-68); optional bias ~ 0.01 N(0,1)."""
    W = 0.02 * torch.randn(N, K, generator=g)
    A = B = None
    if r > 0:
        A = 0.01 * torch.randn(K, r, generator=g)
        B = 0.01 * torch.randn(r, N, generator=g)
        if quantize_ab:
            A, B = _snap_mxint8_dim0(A), _snap_mxint8_dim0(B)
    return (W, A, B, 0.01 * torch.randn(N, generator=g)) if bias else (W, A, B)


def make_case(M, K, N, r, seed=0, bias=False, quantize_ab=True):
    """(x, W, A, B[, bias]) from one seed (tests, tools, smoke)."""
    x, g = make_x(M, K, seed)
    return (x,) + make_weights(g, K, N, r, bias, quantize_ab)


def cpu_baseline(M, K, N, r, q_config, reps=3):
    """The CPU oracle (a port of the reference's eager-torch emulation, routed through the same pad/unfold/fold blocking
    ops as the reference) timed on the host cores; steady state, i.e. the one-time weight quantization (reference
    linear.py:149-153) is done before the clock starts.  Bounded sample: at most 2048 tokens of the workload's first
    projection shape (rows are independent, the emulation's cost is linear in M)."""
    from oracle import lqer_oracle as O

    host = os.cpu_count() or 1
    cores = min(host, int(os.environ.get("LQER_CPU_THREADS", "16")))  # tools/cpu_scan.py: no gain beyond 16 threads
    torch.set_num_threads(cores)
    Ms = min(M, 2048)
    x, W, A, B = make_case(Ms, K, N, r, seed=0, quantize_ab=not any(q_config is c for c in UNQUANTIZED_AB))
    x = x.half().float()
    wq = O.get_quantizer(q_config["w_quantizer"])(W)
    O.lqer_linear_forward(x, wq, None, A, B, q_config, weight_is_quantized=True, via_unfold=True)  # warm-up
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        O.lqer_linear_forward(x, wq, None, A, B, q_config, weight_is_quantized=True, via_unfold=True)
        times.append(time.perf_counter() - t0)
    best = min(times)
    return {"value": round(flops(Ms, K, N, r) / best / 1e12, 4), "unit": "TFLOP/s-equiv", "cores": cores, "host_cores": host,
            "kind": "port", "ms": round(best * 1e3, 2), "ms_all_reps": [round(t * 1e3, 2) for t in times],
            "tokens_per_s": round(Ms / best, 1),
            "sample": f"M={Ms} of {M} tokens, K={K} N={N} r={r} (first projection shape), fp32 eager torch-CPU, "
                      f"min of {reps} after warm-up, weights pre-quantized"}


def check_rows(M, every=False):
    """Rows compared with the oracle after the timed region: the first and the last 96 (first / last row tile) - every row of a
    single-Linear workload of up to 2048 tokens (the headline: 0.3 s more of the oracle)."""
    n = M if every and M <= 2048 else min(96, M)
    return torch.tensor(sorted(set(range(n)) | set(range(M - n, M))))


class HipEvent:
    """A timing event recorded straight through the HIP runtime on the launch stream (the roofline sample brackets single
    kernel launches inside the timed region; torch.cuda.Event is the same call with default flags).  LQER_BENCH_EVENT_FLAGS
    selects the creation flags: default 0x20000000 = hipEventDisableSystemFence (the event's release stays at device scope -
    nothing on the host reads what the bracketed kernel wrote; measured on the driver's 20-step C2 run with 10 sampled
    launches: 982 TFLOP/s-equiv against 958-961 with hipEventDefault and 960 with hipEventReleaseToDevice, calibrated pair
    overhead 3.4 vs 4.6-5.2 us, same kernel durations), 0 = hipEventDefault."""
    _hip = None

    def __init__(self, flags):
        import ctypes as C

        if HipEvent._hip is None:
            h = C.CDLL("libamdhip64.so")
            h.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
            h.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
            h.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
            HipEvent._hip = h
        self.h = C.c_void_p()
        rc = HipEvent._hip.hipEventCreateWithFlags(C.byref(self.h), flags)
        assert rc == 0, f"hipEventCreateWithFlags: {rc}"

    def record(self, stream):
        HipEvent._hip.hipEventRecord(self.h, stream)

    def elapsed_time(self, other):
        import ctypes as C

        ms = C.c_float()
        rc = HipEvent._hip.hipEventElapsedTime(C.byref(ms), self.h, other.h)
        assert rc == 0, f"hipEventElapsedTime: {rc}"
        return ms.value


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --standalone --nproc-per-node N
    bench.py <same arguments>` as a child process (this process has not touched the GPU and never will), pass its stdout
    - rank 0's JSON line - and stderr through unchanged, return its exit code.  The rule the ranks then follow is the
    reference's consecutive-layers-per-device split (experiments/infer_device_map.py:29-37)."""
    import socket
    import subprocess

    with socket.socket() as s:  # a free rendezvous port: two benches on one node must not meet
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    print("# bench.py: launching %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr)
    return subprocess.run(cmd, env=env).returncode


def dry_run_cpu(args, rank, world):
    """--dry-run-cpu: everything of a multi-rank run that is not a kernel, on gloo / CPU tensors - layer partition,
    broadcast of the token batch once per distinct K, barrier-bracketed timed region, max / sum over ranks, gather, ONE
    JSON line on rank 0.  The step is a no-op (the hot path has no CPU form), so `value` is null and the line says so."""
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
    strong = (args.sweep == "strong") or (args.sweep == "auto" and layers > 1)
    if strong and layers == 1 and world > 1:
        sys.exit("--sweep strong needs a model workload (c3/c4/c5): a single Linear has no layers to split")
    my_layers = sweep.layer_partition(layers, world)[rank] if strong else range(layers)
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
                       "layers_per_rank": [int(row[2]) for row in gathered]},
            "broadcast_ms": round(broadcast_ms, 3), "flops_per_step_all_ranks": flops_all,
            "rank_ms_per_step": [round(row[0], 4) for row in gathered],
            "rank_checksums": [round(row[1], 3) for row in gathered]}))
    if dist is not None:
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 50; 4 for the M=16384 model sweeps)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default 10; 1 for the M=16384 sweeps)")
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--sweep", default="auto", choices=["auto", "weak", "strong"],
                    help="multi-GPU: strong = the model's layers split over the ranks (default for c3/c4/c5), weak = every "
                         "rank runs the full unit list (default for single-Linear workloads)")
    ap.add_argument("--layers", type=int, default=0, help="override the model's decoder layer count (profiling runs: --layers 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the oracle comparison of the timed outputs")
    ap.add_argument("--no-module", action="store_true", help="skip the second timed region through the nn.Module")
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
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: start the ranks ourselves, as a child process, BEFORE any GPU call of this process
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world
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

    import ctypes as C

    import lqer_amd
    from lqer_amd import _lib, ops, sweep

    desc_txt, M, r, has_bias, qc, shapes, layers = WORKLOADS[args.workload]
    if args.layers > 0:
        layers = args.layers
        desc_txt += f" [--layers {layers}]"
    big = M >= 8192
    if args.steps is None:
        # (decode sizes: a 50-step region is 0.4 ms, of which the first launch's latency and the closing synchronize are ~10 %)
        args.steps = 4 if big else (400 if M <= 64 else 50)
    if args.warmup is None:
        args.warmup = 1 if big else 10
    strong = (args.sweep == "strong") or (args.sweep == "auto" and layers > 1)
    if strong and layers == 1 and world > 1:
        sys.exit("--sweep strong needs a model workload (c3/c4/c5): a single Linear has no layers to split")
    my_layers = sweep.layer_partition(layers, world)[rank] if strong else range(layers)
    layers_here = len(my_layers)
    quantize_ab = not any(qc is c for c in UNQUANTIZED_AB)

    # ---- the token batch: generated on rank 0, broadcast once per distinct K (RCCL over xGMI), outside the timed region
    xs, broadcast_ms = {}, 0.0
    for K in sorted({K for K, _, _ in shapes}):
        if rank == 0:
            xd = make_x(M, K, seed=0)[0].half().to(dev)
        else:
            xd = torch.empty(M, K, dtype=torch.float16, device=dev)
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            sweep.broadcast_activation(xd, src=0)
            torch.cuda.synchronize()
            broadcast_ms += (time.perf_counter() - t0) * 1e3
        xs[K] = xd

    # ---- this rank's units: one module per distinct projection shape, weights from a per-rank seed; a model sweep re-runs
    # each shape `count x layers owned` times per step (layers differ in values, not in cost)
    mods = []
    for i, (K, N, cnt) in enumerate(shapes):
        g = torch.Generator().manual_seed(sweep.unit_seed(rank, i))
        wts = make_weights(g, K, N, r, bias=has_bias, quantize_ab=quantize_ab)
        mod = lqer_amd.LinearFlexibleLqer(K, N, bias=has_bias, q_config=qc, l_config={"rank": r})
        sd = {"weight": wts[0], "A": wts[1], "B": wts[2]}
        if has_bias:
            sd["bias"] = wts[3]
        mod.load_state_dict(sd)
        mod = mod.to(dev).half()
        y = mod(xs[K])  # packs the operands (one-time, like the reference's first forward)
        mods.append((mod, xs[K], K, N, cnt * layers_here, y, wts))
    torch.cuda.synchronize()

    # decode sizes: a model walks ~3.6 GB of DISTINCT weights per token, so one 9.4 MB image re-run from the Infinity Cache
    # says little - R copies of the Linear (own packed images; same values, same output buffer) are walked round robin
    rotate = args.rotate if args.rotate is not None else (48 if (M <= 64 and len(shapes) == 1 and layers == 1) else 0)
    if rotate and not (M <= 64 and len(mods) == 1):
        sys.exit("--rotate is for the single-Linear decode workloads")
    rot_mods = list(mods)
    if rotate > 1:
        import copy

        rot_mods += [(copy.deepcopy(mods[0][0]),) + mods[0][1:] for _ in range(rotate - 1)]
        for m in rot_mods[1:]:
            m[0](m[1])  # (its launch cache)
        torch.cuda.synchronize()

    L = _lib.lib()
    stream = torch.cuda.current_stream(dev).cuda_stream
    gemm_events = []
    launch_no = [0]
    # bracket every n-th launch of the dominant kernel with HIP events: at least 8 samples inside the timed region
    # whatever --steps is (the driver's 20-step C2 run: every 2nd launch), at most every 10th (a pair costs ~12 us of gaps)
    EV_TOTAL = args.steps * sum(m[4] for m in mods)  # timed launches of the dominant kernel
    EV_EVERY = max(1, min(10, EV_TOTAL // 8))
    # (short regions - the driver's 20-step C2 run: exactly 8 samples, evenly spread, instead of every 2nd launch = 10)
    ev_sample = (lambda i: i % EV_EVERY == 0) if EV_EVERY >= 10 or EV_TOTAL < 8 else \
        (lambda i: i == 0 or (i * 8) // EV_TOTAL != ((i - 1) * 8) // EV_TOTAL)

    # per-module launch constants (descriptor, workspace carving), built once: decode-size steps are host-bound
    plans = []
    distinct = layers > 1 and not args.shared_weights and rotate <= 1
    distinct_keep = []  # (the cloned images stay alive for the run)
    ws = ops.workspace(dev, max(ops.linear_sizes(mod._desc(), M).workspace for mod, *_ in mods))  # one buffer for all
    for mod, xd, K, N, reps, y, _ in rot_mods:
        desc = mod._desc()
        if mod._x_i8 and L.lqer_gemm_route(C.byref(desc), M, _lib.F16) != _lib.ROUTE_TILE256_I8:
            desc = mod._desc(plain=True)  # token counts the int8 tile kernel does not serve: the bf16 kernels, same buffers
        p = mod._packed
        Kp, Mp = L.lqer_padded_k(K), L.lqer_padded_m(M)
        xl, al = ops.desc_limbs(desc)  # bf16 limbs of the activation / x A images (1, 1 unless pass-through)
        xq = ws.data_ptr()
        xaq = xq + ((Mp * Kp * 2 * xl + 255) // 256) * 256
        if mod._x_f16 and K % 64 == 0 and (M % 256 == 0 or M <= 64):
            xq = xd.data_ptr()  # fp16 route: a dense, aligned fp16 tensor is its own activation image (include/lqer_hip.h)
        rp = L.lqer_padded_r(r)
        xscr = xaq + ((Mp * rp * 2 * al + 255) // 256) * 256
        nscr = L.lqer_lowrank_xa_scratch_bytes(C.byref(desc), M)
        gscr = L.lqer_linear_gemm_scratch_bytes(C.byref(desc), M)
        if L.lqer_decode_partials(C.byref(desc), M):
            xaq, gscr = None, nscr  # decode route: the GEMM reduces the partial tiles of x A left in the scratch itself
        a_t, a_limbs = p["a_t"].data_ptr(), p["a_limbs"]
        if mod._x_i8 and "a_t_f16" in p and L.lqer_gemm_route(C.byref(desc), M, _lib.F16) == _lib.ROUTE_TILE256_I8:
            a_t, a_limbs = p["a_t_f16"].data_ptr(), -1  # the int8 route's side GEMM: A as one fp16 image (as the module passes it)
        # model sweeps: every Linear of the model owns its packed operands (same values, distinct addresses - layers differ
        # in values, not in cost, but a weight that is re-read from the Infinity Cache 32 times is not what a model does)
        copies = []
        if distinct and reps > 1:
            a_key = "a_t_f16" if a_limbs == -1 else "a_t"
            for _ in range(reps - 1):
                cw, ca, cb = p["w"].clone(), p[a_key].clone(), p["b_t"].clone()
                cbias = p["bias"].clone() if p.get("bias") is not None else None
                distinct_keep.append((cw, ca, cb, cbias))
                copies.append((cw.data_ptr(), ca.data_ptr(), cb.data_ptr(), ops._ptr(cbias)))
        plans.append(dict(desc=desc, dref=C.byref(desc), x=xd.data_ptr(), a_t=a_t, a_limbs=a_limbs, xq=xq, xaq=xaq, copies=copies,
                          ws=ws.data_ptr(), ws_bytes=ws.numel(),
                          xscr=xscr, nscr=nscr, w=p["w"].data_ptr(),
                          b_t=p["b_t"].data_ptr(), b_limbs=p["b_limbs"], bias=ops._ptr(p.get("bias")), y=y.data_ptr(),
                          gscr=gscr, K=K, N=N, reps=reps, route=L.lqer_gemm_route(C.byref(desc), M, _lib.F16)))

    # M <= 8 with block_fp activations in blocks of 16: lqer_linear_forward issues ONE launch (not inside a captured graph)
    one_launch = (M <= 8 and r > 0 and  # (capturable since round 3: the kernel's granule tag carries its dispatch id)
                  all(L.lqer_decode_partials(pl["dref"], M) and pl["a_limbs"] == 1 for pl in plans))

    # the C-ABI calls of a step with their arguments bound once per stream (the launch stream, or the capture stream of
    # --graph): at decode sizes the Python that assembles 16 arguments per call costs as much as the kernel it launches
    fwd, qxa, gemm = L.lqer_linear_forward, L.lqer_quantize_act_xa, L.lqer_linear_gemm
    bound = {}
    # event pairs for the sampled launches, created ahead of the timed region (creating one costs more host time than a
    # decode-size kernel runs)
    ev_flags = int(os.environ.get("LQER_BENCH_EVENT_FLAGS", "0x20000000"), 0)
    new_pair = lambda: (HipEvent(ev_flags), HipEvent(ev_flags))
    ev_pool = [new_pair() for _ in range(64)]

    def calls_for(st, pls=None):
        pls = plans if pls is None else pls
        if (st, id(pls)) not in bound:
            rows = []
            for pl in pls:
                K, N = pl["K"], pl["N"]
                fa = (pl["dref"], pl["x"], _lib.F16, M, K, pl["w"], pl["a_t"], pl["b_t"], pl["a_limbs"], pl["b_limbs"], pl["bias"],
                      pl["y"], N, pl["ws"], pl["ws_bytes"], st)
                qa = (pl["dref"], pl["x"], _lib.F16, M, K, pl["a_t"], pl["a_limbs"], pl["xq"], pl["xaq"], pl["xscr"], pl["nscr"], st)
                ga = (pl["dref"], pl["xq"], M, pl["w"], pl["xaq"], pl["b_t"], pl["b_limbs"], pl["bias"], pl["y"], _lib.F16, N,
                      pl["xscr"], pl["gscr"], st)
                per_unit = [(fa, qa, ga)]
                for cw, ca, cb, cbias in pl["copies"]:  # the other Linears of this shape: own weight / A / B / bias images
                    per_unit.append(((pl["dref"], pl["x"], _lib.F16, M, K, cw, ca, cb, pl["a_limbs"], pl["b_limbs"], cbias,
                                      pl["y"], N, pl["ws"], pl["ws_bytes"], st),
                                     (pl["dref"], pl["x"], _lib.F16, M, K, ca, pl["a_limbs"], pl["xq"], pl["xaq"], pl["xscr"],
                                      pl["nscr"], st),
                                     (pl["dref"], pl["xq"], M, cw, pl["xaq"], cb, pl["b_limbs"], cbias, pl["y"], _lib.F16, N,
                                      pl["xscr"], pl["gscr"], st)))
                rows.append((pl["reps"], K, N, per_unit))
            bound[(st, id(pls))] = rows
        return bound[(st, id(pls))]

    rot_no = [0]
    resident = [False]  # True: every step re-runs plan 0 (the weight stays in the Infinity Cache)

    def step(timed: bool, stream=stream):
        rows = calls_for(stream)
        if rotate > 1:
            rows = rows[:1] if resident[0] else rows[rot_no[0] % rotate: rot_no[0] % rotate + 1]
            rot_no[0] += 1
        for reps, K, N, per_unit in rows:
            for u in range(reps):
                fa, qa, ga = per_unit[u % len(per_unit)]
                ev = timed and ev_sample(launch_no[0])  # counts timed launches only: the first one is always sampled
                if timed:
                    launch_no[0] += 1
                if one_launch:
                    # up to 8 tokens the whole forward is ONE launch (csrc/decode1.hip) behind lqer_linear_forward - the entry
                    # point of INTEGRATION.md; the events bracket that launch
                    if ev:
                        e0, e1 = ev_pool.pop() if ev_pool else new_pair()
                        e0.record(stream)
                    rc = fwd(*fa)
                    if rc:
                        _lib.check(rc, "linear_forward")
                    if ev:
                        e1.record(stream)
                        gemm_events.append((e0, e1, K, N))
                    continue
                # the two calls of lqer_linear_forward, issued separately so that the dominant kernel can be
                # bracketed with HIP events on the launch stream
                rc = qxa(*qa)
                if rc:
                    _lib.check(rc, "quantize_act_xa")
                if ev:
                    e0, e1 = ev_pool.pop() if ev_pool else new_pair()
                    e0.record(stream)
                rc = gemm(*ga)
                if rc:
                    _lib.check(rc, "linear_gemm")
                if ev:
                    e1.record(stream)
                    gemm_events.append((e0, e1, K, N))

    mrot_no = [0]

    def step_module():
        if rotate > 1:
            mod, xd = rot_mods[mrot_no[0] % rotate][:2]
            mrot_no[0] += 1
            return mod(xd)
        for mod, xd, K, N, reps, _, _ in mods:
            for _ in range(reps):
                mod(xd)

    def timed_region(fn, steps):
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(steps)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    # Device clock ramp (setup, like the packing above): after idling the GPU needs tens of milliseconds of load to
    # reach the clocks it then holds - a 60-step run (5 ms) would measure the ramp, not the kernels (C2: 850 vs 970
    # TFLOP/s-equiv on the same box).  Untimed; the W warm-up steps and the K timed steps follow unchanged.
    if layers_here > 0:
        t_ramp = time.perf_counter()
        while time.perf_counter() - t_ramp < args.prewarm_ms * 1e-3:
            for _ in range(1 if big else 20):
                step(False)
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    graph = None
    if args.graph:
        # launch-bound steps (decode sizes: three ~3 us kernels): capture one step in a hipGraph and replay it.  The
        # kernels cannot be bracketed with events inside a graph, so the roofline sample is taken from ungraphed
        # launches after the timed region.
        if args.steps % args.graph:
            sys.exit("--steps must be a multiple of --graph")
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(args.graph):
                step(False, torch.cuda.current_stream(dev).cuda_stream)
        graph.replay()
        torch.cuda.synchronize()

    def run_abi(steps):
        if graph is not None:
            for _ in range(steps // args.graph):
                graph.replay()
        else:
            for _ in range(steps):
                step(True)

    elapsed_rank = timed_region(run_abi, args.steps)
    if graph is not None:
        for _ in range(min(args.steps, 4 * EV_EVERY)):
            step(True)
        torch.cuda.synchronize()
    elapsed = sweep.max_over_ranks(elapsed_rank, dev)

    # decode workloads: the same steps once more on ONE resident weight (what rounds 1-2 reported: an upper bound)
    resident_fig = None
    if rotate > 1 and graph is None:
        rot_events, gemm_events = gemm_events, []
        resident[0] = True
        for _ in range(args.warmup):
            step(False)
        el_res = sweep.max_over_ranks(timed_region(run_abi, args.steps), dev)
        resident[0] = False
        res_events, gemm_events = gemm_events, rot_events
        resident_fig = {"ms_per_step": round(el_res / args.steps * 1e3, 4), "events": res_events}

    # second timed region: the same K steps through the drop-in module (torch.empty, descriptor cache, ctypes marshalling
    # included) - the boundary the reference's callers use
    module = None
    if not args.no_module:
        for _ in range(max(1, args.warmup // 2)):
            step_module()
        el_mod = sweep.max_over_ranks(timed_region(lambda n: [step_module() for _ in range(n)], args.steps), dev)
        module = {"ms_per_step": round(el_mod / args.steps * 1e3, 4), "vs_c_abi": round(el_mod / elapsed, 4)}
        if M <= 64 and graph is None:
            # decode sizes are host-bound through the module (torch.empty + ctypes per call ~8 us on a ~8 us kernel): the way a
            # serving loop runs them is ONE captured graph per token step - here G module forwards (one per rotated weight)
            # captured by lqer_amd.graph.GraphedCallable and replayed; the one-launch decode route is capturable
            from lqer_amd.graph import GraphedCallable

            G = max(rotate, 1) * max(1, 48 // max(rotate, 1))
            while args.steps % G:
                G -= 1
            gm = GraphedCallable(lambda: [step_module() for _ in range(G)][-1], warmup=2)
            gm()
            el_g = sweep.max_over_ranks(timed_region(lambda n: [gm() for _ in range(n // G)], args.steps), dev)
            module.update(graph_ms_per_step=round(el_g / args.steps * 1e3, 4), graph_vs_c_abi=round(el_g / elapsed, 4),
                          graph_forwards_per_replay=G)

    # third timed region (model workloads): the same Linears the way the model runs them - q/k/v and
    # gate/up receive ONE tensor, so its activation image and one side GEMM over the members' concatenated A are made once
    # per group (lqer_amd.linear.SharedActivation; same quantizers and GEMM kernels, results as member by member).  The
    # headline `value` stays the conservative one: every Linear quantizes its own input, as the reference's modules do.
    model_shared = None
    # (secondary figures are single-rank only: a rank that fails or owns no layer would leave the others in a collective)
    if not args.no_module and layers > 1 and layers_here > 0 and world == 1:
        import copy

        from lqer_amd.linear import SharedActivation

        try:
            units = []  # per shape: (group members, solo module or None, solo calls per layer, x)
            for mod, xd, K, N, reps, _, _ in mods:
                cnt = reps // layers_here
                gsz = 3 if (cnt >= 3 and K == N) else (2 if cnt == 2 else 0)
                solo_n = cnt - gsz
                copies = [copy.deepcopy(mod) for _ in range(max(gsz - 1, 0) + (1 if solo_n else 0))]
                solo = copies.pop() if solo_n else None
                members = []
                if gsz:
                    grp = SharedActivation([mod] + copies)
                    members = grp.members if grp.enabled else []
                    if not grp.enabled:
                        solo, solo_n = mod, cnt
                units.append((members, solo, solo_n, xd))
            if any(members for members, _, _, _ in units):
                def step_shared():
                    for _ in range(layers_here):
                        for members, solo, solo_n, xd in units:
                            for m in members:
                                m(xd)
                            for _ in range(solo_n):
                                solo(xd)

                for _ in range(max(1, args.warmup // 2)):
                    step_shared()
                el_sh = sweep.max_over_ranks(timed_region(lambda n: [step_shared() for _ in range(n)], args.steps), dev)
                model_shared = {"ms_per_step": round(el_sh / args.steps * 1e3, 4), "vs_c_abi": round(el_sh / elapsed, 4),
                                "groups_per_layer": [len(members) for members, _, _, _ in units if members]}

        except Exception as e:  # (a secondary figure must not cost the bench line)
            model_shared = {"error": f"{type(e).__name__}: {e}"[:200]}

    # secondary figure: the forwards of this workload are INDEPENDENT units - issued alternately on two HIP streams
    # (own activation / x A images, scratch and output per stream) the second queue's quantizer, side GEMM and store phases
    # run under the other forward's main loop.  A throughput figure for sweeps and serving batches; `value` stays the
    # one-stream figure (a model's Linears form a chain: 8d sums their times)
    two_streams = None
    if not args.no_two_streams and layers_here > 0 and M > 64 and graph is None and world == 1:
        try:
            # (two explicit streams: the legacy default stream this script otherwise launches on serialises with every other stream)
            s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
            ws2 = torch.empty(ws.numel(), dtype=torch.uint8, device=dev)
            plans2, ys2 = [], []
            for pl in plans:
                q = dict(pl)
                for k in ("xq", "xaq", "xscr", "ws"):
                    if q[k] is not None and ws.data_ptr() <= q[k] < ws.data_ptr() + ws.numel():
                        q[k] = q[k] - ws.data_ptr() + ws2.data_ptr()
                ys2.append(torch.empty(M, q["N"], dtype=torch.float16, device=dev))
                q["y"] = ys2[-1].data_ptr()
                plans2.append(q)
            unit_no, on_b = [0], set()

            def step_two():
                ra, rb = calls_for(s1.cuda_stream), calls_for(s2.cuda_stream, plans2)
                for i, ((reps, K, N, pa), (_, _, _, pb)) in enumerate(zip(ra, rb)):
                    for u in range(reps):
                        if unit_no[0] % 2:
                            on_b.add((i, u % len(pa)))
                        _, qa, ga = (pa if unit_no[0] % 2 == 0 else pb)[u % len(pa)]
                        unit_no[0] += 1
                        rc = qxa(*qa) or gemm(*ga)
                        if rc:
                            _lib.check(rc, "two-stream step")

            s1.wait_stream(torch.cuda.current_stream(dev))
            s2.wait_stream(torch.cuda.current_stream(dev))
            for _ in range(max(2, args.warmup // 2)):
                step_two()
            torch.cuda.synchronize()
            for i, ((mod, xd, K, N, reps, y, _), y2) in enumerate(zip(mods, ys2)):
                # both queues produce the bits of the one-stream run (units of a plan that share one image set)
                if any(pi == i for pi, _ in on_b):  # (every image set of a plan holds the same values)
                    assert torch.equal(y.view(torch.int16), y2.view(torch.int16)), "two-stream outputs differ"
            el_two = sweep.max_over_ranks(timed_region(lambda n: [step_two() for _ in range(n)], args.steps), dev)
            two_streams = {"ms_per_step": round(el_two / args.steps * 1e3, 4), "vs_one_stream": round(elapsed / el_two, 4)}
            del ws2, ys2
        except Exception as e:  # (a secondary figure must not cost the bench line)
            two_streams = {"error": f"{type(e).__name__}: {e}"[:200]}

    # ---- gather (outside the timed regions): per-rank elapsed time, a checksum of the first unit's output
    ysum = float(mods[0][5].float().sum().item()) if layers_here > 0 else 0.0
    gathered = sweep.gather_rows([elapsed_rank * 1e3 / args.steps, ysum, float(layers_here)], dev)

    flops_rank = sum(flops(M, K, N, r) * reps for _, _, K, N, reps, _, _ in mods)
    flops_all = sweep.sum_over_ranks(float(flops_rank), dev)
    ms_per_step = elapsed / args.steps * 1e3
    value = flops_all * args.steps / elapsed / 1e12

    # ---- parity of what was just timed: row slices of every unit's output buffer against the CPU oracle (rank 0)
    parity = None
    if not args.no_check and rank == 0:
        from oracle import lqer_oracle as O  # the checker - after the timed regions, never inside them

        idx = check_rows(M, every=len(mods) == 1)
        worst = 0.0
        for mod, xd, K, N, reps, y, wts in mods:
            h = lambda t: None if t is None else t.half().float()
            ref = O.lqer_linear_forward(xd[idx.to(dev)].float().cpu(), h(wts[0]), h(wts[3]) if has_bias else None, h(wts[1]), h(wts[2]), qc)
            got = y[idx.to(dev)].float().cpu()
            worst = max(worst, float((got - ref).norm() / ref.norm()))
        parity = worst
        print(f"# parity vs CPU oracle ({len(idx)} rows x {len(mods)} shapes): rel-L2 {parity:.3e}", file=sys.stderr)

    if rank == 0:
        # dominant kernel = the fused GEMM; algorithmic FLOPs per launch = 2MKN + 2MrN
        # an event pair around a kernel also measures the gap between the first event and the kernel's start: the
        # same pair around nothing, recorded right behind a kernel, gives that overhead (median of 32), which is
        # subtracted - the result agrees with the kernel durations of the rocprofv3 trace of the same command
        cal = []
        for _ in range(32):
            _lib.check(L.lqer_quantize_act_mxint(mods[0][1].data_ptr(), _lib.F16, min(32, M), mods[0][2], mods[0][2],
                                                 C.byref(ops.make_qfmt(MXINT_Q["x_quantizer"], "x")), ops.workspace(dev, 1 << 20).data_ptr(), stream), "cal")
            c0, c1 = new_pair()
            c0.record(stream)
            c1.record(stream)
            cal.append((c0, c1))
        torch.cuda.synchronize()
        ev_overhead_ms = sorted(c0.elapsed_time(c1) for c0, c1 in cal)[len(cal) // 2]
        tot_ms, tot_fl, n_launch = 0.0, 0.0, 0
        for e0, e1, K, N in gemm_events:
            tot_ms += max(e0.elapsed_time(e1) - ev_overhead_ms, 1e-6)
            tot_fl += 2.0 * M * K * N + 2.0 * M * r * N
            n_launch += 1
        ach = tot_fl / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
        routes = sorted({pl["route"] for pl in plans})
        kname = {_lib.ROUTE_SMALLM: "k_decode1 (whole forward)" if one_launch else "k_lqer_gemm_smallm", _lib.ROUTE_TILE128: "k_lqer_gemm", _lib.ROUTE_TILE256: "k_lqer_gemm_m256",
                 _lib.ROUTE_TILE256_I8: "k_lqer_gemm_i8"}
        int8 = routes == [_lib.ROUTE_TILE256_I8]  # every GEMM of the step ran the int8 MFMA main loop
        peak = INT8_MFMA_PEAK_TOPS if int8 else BF16_MFMA_PEAK_TFLOPS
        kernels = "+".join(kname.get(rt, str(rt)) for rt in routes)
        # HBM-side bytes per launch of the dominant kernel come from separate rocprofv3 --pmc passes over this very
        # command (tools/pmc_bench.sh; a profiler cannot run inside this process); the committed summary of the
        # workload is quoted, with its ratio to the algorithmic bytes
        traffic, traffic_ratio, traffic_source = None, None, None
        for rnd in ("r03", "r02"):
            tfile = os.path.join(ROOT, "profiles", f"{rnd}_traffic_{args.workload}.json")
            if os.path.exists(tfile):
                with open(tfile) as fh:
                    tj = json.load(fh)
                traffic, traffic_ratio = tj.get("traffic_bytes_per_launch"), tj.get("ratio_to_algorithmic")
                traffic_source = (f"profiles/{rnd}_traffic_{args.workload}.json - separate rocprofv3 --pmc passes over this command "
                                  "(tools/pmc_bench.sh), committed; NOT measured in this run")
                break
        roofline = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TOP/s" if int8 else "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": traffic, "traffic_over_algorithmic": traffic_ratio,
                    "traffic_source": traffic_source, "kernel": kernels, "avg_launch_us": round(tot_ms / max(n_launch, 1) * 1e3, 2), "launches": n_launch,
                    "event_pair_overhead_us": round(ev_overhead_ms * 1e3, 2), "event_flags": hex(ev_flags),
                    "frac_of_int8_peak": round(ach / INT8_MFMA_PEAK_TOPS, 4)}
        if M <= 64:
            # small-M kernel: HBM-bound.  Algorithmic bytes per launch: packed W (0.5625 B per weight)
            # + B^T limbs + bias + the activation image + xAq + y
            tot_by = 0.0
            for _, _, K, N in gemm_events:
                Kp, Np, rp = -(-K // 64) * 64, -(-N // 256) * 256, -(-r // 16) * 16
                tot_by += Np * Kp * 0.5625 + Np * rp * 2 + M * Kp * 2 + M * rp * 2 + M * N * 2 + (Np * 4 if has_bias else 0)  # (one copy of every image: algorithmic)
                if one_launch:
                    tot_by += rp * Kp * 2  # the whole forward: A^T as well (x in place of its image: the same bytes)
            gbs = tot_by / (tot_ms * 1e-3) / 1e9 if tot_ms > 0 else 0.0
            roofline = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_over_algorithmic": traffic_ratio,
                        "traffic_source": traffic_source, "kernel": kernels, "avg_launch_us": round(tot_ms / max(n_launch, 1) * 1e3, 2), "launches": n_launch,
                        "weights_rotated": rotate if rotate > 1 else 1, "event_flags": hex(ev_flags),
                        "event_pair_overhead_us": round(ev_overhead_ms * 1e3, 2),
                        # (a ~6 us kernel: the subtracted pair overhead is half of what the pair measures, and rocprofv3's own
                        # kernel durations carry 1.5-3 us of instrumentation at this size - profiles/README.md; the bound that
                        # needs no calibration is ms_per_step, one launch + one launch gap per forward)
                        "avg_launch_us_upper_bound": round(ms_per_step * 1e3, 2) if one_launch else None}
            if resident_fig is not None:  # the same launches on ONE weight that stays in the Infinity Cache (an upper bound)
                rms = sum(max(e0.elapsed_time(e1) - ev_overhead_ms, 1e-6) for e0, e1, _, _ in resident_fig["events"])
                rn = max(len(resident_fig["events"]), 1)
                rgbs = (tot_by / max(n_launch, 1)) * rn / (rms * 1e-3) / 1e9 if rms > 0 else 0.0
                roofline["resident_weight"] = {"avg_launch_us": round(rms / rn * 1e3, 2), "achieved": round(rgbs, 1),
                                               "frac": round(rgbs / HBM_PEAK_GBS, 4), "launches": rn,
                                               "ms_per_step": resident_fig["ms_per_step"]}
        out = {
            "metric": "W4A8+rank-r Linear GEMM TFLOPS-equiv",
            "value": round(value, 2),
            "unit": "TFLOP/s-equiv",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            # the arithmetic type of the main loop's MFMA operands
            "dtype": "int8" if (M > 64 and int8) else ("f16" if mods[0][0]._x_f16 else "bf16"),
            "data": "synthetic",
            "config": {"workload": desc_txt + (f" [{rotate} distinct packed weights walked round robin: {rotate * 9.4:.0f} MB > the 256 MB "
                                                "Infinity Cache]" if rotate > 1 else ""), "tokens_per_step": M, "rank": r,
                       "formats": "x %s, W MXINT4/%s, A_out,B_out as x, y fp16" % (
                           "MXINT8/%s" % qc["x_quantizer"]["block_size"][-1] if qc["x_quantizer"]["name"] == "block_fp"
                           else ("fp16 pass-through (fp16 MFMA main loop)" if mods[0][0]._x_f16 else "fp16 pass-through (2 bf16 limbs)"),
                           qc["w_quantizer"]["block_size"][-1]),
                       "boundary": ("C ABI (lqer_linear_forward per Linear: one launch)" if one_launch else
                                    "C ABI (lqer_quantize_act_xa + lqer_linear_gemm per Linear, pre-built plans)") +
                                   "; the nn.Module figure is in `module`",
                       "sharding": ("decoder layers split over the ranks, ceil(L/G) consecutive layers each (infer_device_map.py:29-37)"
                                    if strong else "every rank runs its own Linear unit(s) of the workload") +
                                   "; x broadcast from rank 0 and per-rank results gathered outside the timed region, no data-path collective",
                       "layers_per_rank": [int(row[2]) for row in gathered],
                       "weights": ("one packed image set per Linear of the model: %.2f GB walked per step on rank 0" % (
                           sum(sum(t.numel() * t.element_size() for t in c if t is not None) for c in distinct_keep) / 1e9
                           + sum(m[0]._packed["w"].numel() for m in mods) / 1e9) if distinct else
                           ("one packed image set per projection shape, re-run for every layer" if layers > 1 else "one Linear"))},
            "tokens_per_s": round(M * args.steps / elapsed * (1 if strong else world), 1),
            "launch": ("hipGraph replay, %d steps per graph" % args.graph) if graph is not None else "direct launches",
            "prewarm_ms": args.prewarm_ms,
            "broadcast_ms": round(broadcast_ms, 3),
            "roofline": roofline,
            "module": module,
            # (model workloads) q/k/v and gate/up sharing one quantized input, as the model runs them; `value` does not use it
            "model_shared_inputs": model_shared if model_shared is None or "error" in model_shared else dict(
                model_shared, value=round(flops_all / (model_shared["ms_per_step"] * 1e-3) / 1e12, 2)),
            # independent forwards alternating on two HIP streams (throughput of sweeps / serving batches; not `value`)
            "two_streams": two_streams if two_streams is None or "error" in two_streams else dict(
                two_streams, value=round(flops_all / (two_streams["ms_per_step"] * 1e-3) / 1e12, 2)),
            "parity_rel_l2": None if parity is None else float(f"{parity:.3e}"),
            "parity_rows": None if parity is None else int(len(check_rows(M, every=len(mods) == 1))),
            "rank_ms_per_step": [round(row[0], 4) for row in gathered],
            "rank_checksums": [round(row[1], 3) for row in gathered],
        }
        if world == 1 and not args.no_cpu_baseline:
            K0, N0, _ = shapes[0]
            out["cpu_baseline"] = cpu_baseline(M, K0, N0, r, qc)
        if parity is not None:
            assert parity <= 1e-3, f"parity of the timed outputs vs the CPU oracle: rel-L2 {parity:.3e} > 1e-3"
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
