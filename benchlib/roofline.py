"""The `roofline` object of the bench line: algorithmic work of the dominant kernel per launch / its HIP-event time inside
the timed region, against the chip peak of the main loop's operand type."""
from __future__ import annotations

import ctypes as C
import json
import os

import torch

from .workloads import BF16_MFMA_PEAK_TFLOPS, HBM_PEAK_GBS, INT8_MFMA_PEAK_TOPS, MXINT_Q

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROFILE_ROUNDS = ("r06", "r05", "r04", "r03", "r02")  # newest committed PMC summary wins
NOMINAL_MHZ = 2400.0  # the clock the peaks are priced at (MI355X_MICROARCH.md: max clock)


def event_pair_overhead_ms(L, _lib, ops, dev, stream, cal_x, new_pair, M):
    """An event pair around a kernel also measures the gap between the first event and the kernel's start: the same pair
    around nothing, recorded right behind a kernel, gives that overhead (median of 32), which is subtracted - the result
    agrees with the kernel durations of the rocprofv3 trace of the same command (profiles/README.md)."""
    xd, K = cal_x
    cal = []
    fmt = ops.make_qfmt(MXINT_Q["x_quantizer"], "x")
    scratch = ops.workspace(dev, 1 << 20)
    for _ in range(32):
        _lib.check(L.lqer_quantize_act_mxint(xd.data_ptr(), _lib.F16, min(32, M), K, K, C.byref(fmt), scratch.data_ptr(), stream), "cal")
        c0, c1 = new_pair()
        c0.record(stream)
        c1.record(stream)
        cal.append((c0, c1))
    torch.cuda.synchronize()
    return sorted(c0.elapsed_time(c1) for c0, c1 in cal)[len(cal) // 2]


def _traffic(workload):
    """HBM-side bytes per launch of the dominant kernel come from separate rocprofv3 --pmc passes over this very command
    (tools/pmc_bench.sh; a profiler cannot run inside this process): the committed summary of the workload is quoted."""
    for rnd in PROFILE_ROUNDS:
        tfile = os.path.join(ROOT, "profiles", f"{rnd}_traffic_{workload}.json")
        if os.path.exists(tfile):
            with open(tfile) as fh:
                tj = json.load(fh)
            return (tj.get("traffic_bytes_per_launch"), tj.get("ratio_to_algorithmic"),
                    f"profiles/{rnd}_traffic_{workload}.json - separate rocprofv3 --pmc passes over this command "
                    "(tools/pmc_bench.sh), committed; NOT measured in this run")
    return None, None, None


def _kernel_names(routes, one_launch, _lib):
    kname = {_lib.ROUTE_SMALLM: "k_decode1 (whole forward)" if one_launch else "k_lqer_gemm_smallm", _lib.ROUTE_TILE128: "k_lqer_gemm",
             _lib.ROUTE_TILE256: "k_lqer_gemm_m256", _lib.ROUTE_I8: "k_lqer_gemm_i8"}
    return "+".join(kname.get(rt, str(rt)) for rt in routes)


# The dominant kernel's MAIN-LOOP clock, from the committed stamp builds (s_memtime / s_memrealtime around the main loop of every wave:
# tools/clock_probe_i8.py, tools/clock_probe.py).  `sustained_mhz` (the probe beside back-to-back launches) averages a whole launch -
# prologue and epilogue run without the matrix pipe and clock ~2.1 GHz - so it reads above these; quoted, not re-measured in a bench run.
MAIN_LOOP_CLOCK_MHZ = {
    "c2int": (1598, "profiles/r06_i8_timeline.txt (2048 x 4096 x 4096: 1,183 cycles per 128-k step at 1.598 GHz)"),
    "c2introw": (1598, "profiles/r06_i8_timeline.txt (the same kernel and shape)"),
    "c3int": (1560, "profiles/r06_i8_timeline.txt (1.598 GHz at K = 4096, 1.488 GHz at K = 11008, ~2.09 GHz on the three-round N = 11008 "
                    "launch's last tile: the one-round shapes' launch-weighted mean)"),
    "c4": (1586, "tools/clock_probe_i8.py --M 16384 --K 5120 --N 5120 --r 64 on the round-6 build (2,096 cycles per step at 1.586 GHz; "
                 "profiles/r04_i8_timeline_block128.txt: 1.61)"),
    "c4row": (1711, "profiles/r04_i8_timeline_row.txt (one block per row: 2,128 cycles per step at 1.711 GHz)"),
    "c2": (1570, "profiles/r04_c2_gemm_timeline.txt (1,133 cycles per k-step at 1.570 GHz; the kernel's main loop has not changed since)"),
}


def mfma_roofline(gemm_events, ev_overhead_ms, M, r, routes, int8, one_launch, _lib, workload, ev_flags):
    """Dominant kernel = the fused GEMM; algorithmic FLOPs per launch = 2MKN + 2MrN.  `per_shape`: the same
    per projection shape (K, N) of the workload."""
    peak = INT8_MFMA_PEAK_TOPS if int8 else BF16_MFMA_PEAK_TFLOPS
    tot_ms, tot_fl, n_launch = 0.0, 0.0, 0
    shapes = {}
    for e0, e1, K, N in gemm_events:
        ms = max(e0.elapsed_time(e1) - ev_overhead_ms, 1e-6)
        fl = 2.0 * M * K * N + 2.0 * M * r * N
        tot_ms += ms
        tot_fl += fl
        n_launch += 1
        s = shapes.setdefault((K, N), [0.0, 0.0, 0])
        s[0] += ms
        s[1] += fl
        s[2] += 1
    ach = tot_fl / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
    traffic, traffic_ratio, traffic_source = _traffic(workload)
    per_shape = [{"K": K, "N": N, "launches": n, "avg_launch_us": round(ms / n * 1e3, 2),
                  "achieved": round(fl / (ms * 1e-3) / 1e12, 1), "frac": round(fl / (ms * 1e-3) / 1e12 / peak, 4)}
                 for (K, N), (ms, fl, n) in sorted(shapes.items())]
    return {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TOP/s" if int8 else "TFLOP/s",
            "frac": round(ach / peak, 4), "traffic": traffic, "traffic_over_algorithmic": traffic_ratio,
            "traffic_source": traffic_source, "kernel": _kernel_names(routes, one_launch, _lib),
            "avg_launch_us": round(tot_ms / max(n_launch, 1) * 1e3, 2), "launches": n_launch, "per_shape": per_shape,
            "event_pair_overhead_us": round(ev_overhead_ms * 1e3, 2), "event_flags": hex(ev_flags),
            "frac_of_int8_peak": round(ach / INT8_MFMA_PEAK_TOPS, 4),
            "main_loop_clock_mhz": (MAIN_LOOP_CLOCK_MHZ.get(workload) or (None, None))[0],
            "main_loop_clock_source": (MAIN_LOOP_CLOCK_MHZ.get(workload) or (None, None))[1]}


def hbm_roofline(gemm_events, ev_overhead_ms, M, r, has_bias, routes, one_launch, _lib, workload, ev_flags, rotate, ms_per_step,
                 resident_fig):
    """Small-M kernel: HBM-bound.  Algorithmic bytes per launch: packed W (0.5625 B per weight) + B^T limbs +
    bias + the activation image + xAq + y (one copy of every image)."""
    tot_ms, tot_by, n_launch = 0.0, 0.0, 0
    for e0, e1, K, N in gemm_events:
        tot_ms += max(e0.elapsed_time(e1) - ev_overhead_ms, 1e-6)
        n_launch += 1
        Kp, Np, rp = -(-K // 64) * 64, -(-N // 256) * 256, -(-r // 16) * 16
        tot_by += Np * Kp * 0.5625 + Np * rp * 2 + M * Kp * 2 + M * rp * 2 + M * N * 2 + (Np * 4 if has_bias else 0)
        if one_launch:
            tot_by += rp * Kp * 2  # the whole forward: A^T as well (x in place of its image: the same bytes)
    gbs = tot_by / (tot_ms * 1e-3) / 1e9 if tot_ms > 0 else 0.0
    traffic, traffic_ratio, traffic_source = _traffic(workload)
    rl = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
          "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_over_algorithmic": traffic_ratio,
          "traffic_source": traffic_source, "kernel": _kernel_names(routes, one_launch, _lib),
          "avg_launch_us": round(tot_ms / max(n_launch, 1) * 1e3, 2), "launches": n_launch,
          "weights_rotated": rotate if rotate > 1 else 1, "event_flags": hex(ev_flags),
          "event_pair_overhead_us": round(ev_overhead_ms * 1e3, 2),
          # (a ~6 us kernel: the subtracted pair overhead is half of what the pair measures, and rocprofv3's own
          # kernel durations carry 1.5-3 us of instrumentation at this size - profiles/README.md; the bound that
          # needs no calibration is ms_per_step, one launch + one launch gap per forward)
          "avg_launch_us_upper_bound": round(ms_per_step * 1e3, 2) if one_launch else None}
    if one_launch and n_launch:
        # Three clocks measure this ~7-us kernel - all three are in the line: `frac` from the events minus the calibrated
        # pair overhead (the kindest), the trace of a profiled run (r05_kernel_stats_d1.csv: 9.59 us, carries the tracer's per-dispatch cost),
        # and the kernel's own stamps (first wave's start to last wave's end, a -DLQER_D1_STAMPS build: ~7.0 us at M = 1, committed).
        by = tot_by / n_launch
        rl["frac_by_clock"] = {
            "events_minus_pair_overhead": rl["frac"],
            "in_kernel_span": {"us": 7.0, "frac": round(by / 7.0e-6 / 1e9 / HBM_PEAK_GBS, 4),
                               "source": "profiles/r05_decode_lds_ring.txt (stamps build, M = 1; not re-measured in this run)"},
            "rocprofv3_trace": {"us": 9.59, "frac": round(by / 9.59e-6 / 1e9 / HBM_PEAK_GBS, 4),
                                "source": "profiles/r05_kernel_stats_d1.csv (profiled run; not re-measured in this run)"},
            "whole_step": {"us": round(ms_per_step * 1e3, 2), "frac": round(by / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
    if resident_fig is not None:  # the same launches on ONE weight that stays in the Infinity Cache (an upper bound)
        rms = sum(max(e0.elapsed_time(e1) - ev_overhead_ms, 1e-6) for e0, e1, _, _ in resident_fig["events"])
        rn = max(len(resident_fig["events"]), 1)
        rgbs = (tot_by / max(n_launch, 1)) * rn / (rms * 1e-3) / 1e9 if rms > 0 else 0.0
        rl["resident_weight"] = {"avg_launch_us": round(rms / rn * 1e3, 2), "achieved": round(rgbs, 1),
                                 "frac": round(rgbs / HBM_PEAK_GBS, 4), "launches": rn,
                                 "ms_per_step": resident_fig["ms_per_step"]}
    return rl
