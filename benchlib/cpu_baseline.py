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
    projection shape (rows are independent, the emulation's cost is linear in M).  `value`: on LQER_CPU_THREADS threads
    (default 16: tools/cpu_scan.py found no gain beyond - the emulation is a chain of memory-bound elementwise passes);
    `by_threads` also carries a run on every core the box reports ("all cores, count printed")."""
    from oracle import lqer_oracle as O

    host = os.cpu_count() or 1
    few = min(host, int(os.environ.get("LQER_CPU_THREADS", "16")))
    Ms = min(M, 2048)
    x, W, A, B = make_case(Ms, K, N, r, seed=0, quantize_ab=not any(q_config is c for c in UNQUANTIZED_AB))
    x = x.half().float()
    wq = O.get_quantizer(q_config["w_quantizer"])(W)
    before = torch.get_num_threads()
    torch.set_num_threads(few)
    times = _time_oracle(O, x, wq, A, B, q_config, reps)
    best = min(times)
    fig = lambda t, rows: round(flops(rows, K, N, r) / t / 1e12, 4)
    by_threads = {str(few): {"value": fig(best, Ms), "ms": round(best * 1e3, 2), "rows": Ms}}
    if host != few:
        # every core the box reports: on a GPU box whose CPU share is a fraction of the host (16 of 256 here) that
        # oversubscribes the share - measured 18 s against 0.2 s per forward - so this leg runs a smaller sample (rows are independent, the
        # emulation is linear in M) once after a warm-up, and is reported beside the figure, never as it
        Ma = min(Ms, 128)
        torch.set_num_threads(host)
        t_all = min(_time_oracle(O, x[:Ma].contiguous(), wq, A, B, q_config, 1))
        by_threads[str(host)] = {"value": fig(t_all, Ma), "ms": round(t_all * 1e3, 2), "rows": Ma}
    torch.set_num_threads(before)
    return {"value": fig(best, Ms), "unit": "TFLOP/s-equiv", "cores": few, "host_cores": host,
            "kind": "port", "ms": round(best * 1e3, 2), "ms_all_reps": [round(t * 1e3, 2) for t in times],
            "by_threads": by_threads, "tokens_per_s": round(Ms / best, 1),
            "sample": f"M={Ms} of {M} tokens, K={K} N={N} r={r} (first projection shape), fp32 eager torch-CPU, "
                      f"min of {reps} after warm-up on {few} threads (`by_threads`: also every reported core, on a smaller sample), "
                      "weights pre-quantized"}
