"""The `cpu_baseline` leg of the bench line: the CPU oracle timed on the box's host cores (the only place outside tests/ and
smoke() that runs the oracle as a thing measured - as the reported baseline, never as the product)."""
from __future__ import annotations

import os
import time

import torch

from .workloads import UNQUANTIZED_AB, flops, make_case


def _time_oracle(O, x, wq, A, B, q_config, reps):
    O.lqer_linear_forward(x, wq, None, A, B, q_config, weight_is_quantized=True, via_unfold=True)  # warm-up
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        O.lqer_linear_forward(x, wq, None, A, B, q_config, weight_is_quantized=True, via_unfold=True)
        times.append(time.perf_counter() - t0)
    return times


def cpu_baseline(M, K, N, r, q_config, reps=3):
    """The CPU oracle (a port of the reference's eager-torch emulation, routed through the same pad/unfold/fold blocking
    ops as the reference) timed on the host cores; steady state, i.e. the one-time weight quantization (reference
    linear.py:149-153) is done before the clock starts.  Bounded sample: at most 2048 tokens of the workload's first
    projection shape (rows are independent, the emulation's cost is linear in M).  Timed twice: on LQER_CPU_THREADS
    (default 16: tools/cpu_scan.py found no gain beyond - the emulation is a chain of memory-bound elementwise passes) and
    on every core the box reports ("all cores, count printed"); `value` is the better of the two."""
    from oracle import lqer_oracle as O

    host = os.cpu_count() or 1
    few = min(host, int(os.environ.get("LQER_CPU_THREADS", "16")))
    Ms = min(M, 2048)
    x, W, A, B = make_case(Ms, K, N, r, seed=0, quantize_ab=not any(q_config is c for c in UNQUANTIZED_AB))
    x = x.half().float()
    wq = O.get_quantizer(q_config["w_quantizer"])(W)
    runs = {}
    before = torch.get_num_threads()
    for cores in sorted({few, host}):
        torch.set_num_threads(cores)
        runs[cores] = _time_oracle(O, x, wq, A, B, q_config, reps)
    torch.set_num_threads(before)
    best_cores = min(runs, key=lambda c: min(runs[c]))
    best = min(runs[best_cores])
    fig = lambda t: round(flops(Ms, K, N, r) / t / 1e12, 4)
    return {"value": fig(best), "unit": "TFLOP/s-equiv", "cores": best_cores, "host_cores": host,
            "kind": "port", "ms": round(best * 1e3, 2), "ms_all_reps": [round(t * 1e3, 2) for t in runs[best_cores]],
            "by_threads": {str(c): {"value": fig(min(t)), "ms": round(min(t) * 1e3, 2)} for c, t in runs.items()},
            "tokens_per_s": round(Ms / best, 1),
            "sample": f"M={Ms} of {M} tokens, K={K} N={N} r={r} (first projection shape), fp32 eager torch-CPU, "
                      f"min of {reps} after warm-up per thread count ({', '.join(str(c) for c in runs)} threads), weights pre-quantized"}
