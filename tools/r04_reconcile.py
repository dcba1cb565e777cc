#!/usr/bin/env python3
"""Side by side: the dominant kernel's duration at C2 by HIP events (bench.py's roofline sample) in an unprofiled run, in a run
under rocprofv3 --kernel-trace, and in a second unprofiled run - all on one box, back to back - against the trace's own
durations of the SAME launches the profiled run's events bracketed (the timed region's launches, not the whole process).
usage: tools/r04_reconcile.py <dir with reconcile_{plain1,profiled,plain2}.json, reconcile_kernel_stats.csv, reconcile_kernel_trace.csv>"""
import csv
import json
import sys

d = sys.argv[1]
runs = {k: json.load(open(f"{d}/reconcile_{k}.json")) for k in ("plain1", "profiled", "plain2")}
print("C2, driver arguments (--gpus 1 --steps 20 --warmup 5), one box, back to back")
for k, r in runs.items():
    rl = r["roofline"]
    print(f"  {k:9s} value {r['value']:8.2f} TFLOP/s-equiv  ms/step {r['ms_per_step']:.4f}  {rl['kernel']} by events: {rl['avg_launch_us']:6.2f} us "
          f"({rl['launches']} launches, pair overhead {rl['event_pair_overhead_us']} us)  frac {rl['frac']}")
rows = list(csv.DictReader(open(f"{d}/reconcile_kernel_stats.csv")))
for row in rows:
    if "k_lqer_gemm" in row["Name"]:
        print(f"  trace, all {row['Calls']} launches of the process (clock ramp, warm-up, both timed regions): average "
              f"{float(row['AverageNs']) / 1e3:.2f} us, min {float(row['MinNs']) / 1e3:.2f}, max {float(row['MaxNs']) / 1e3:.2f}")
tr = [r for r in csv.DictReader(open(f"{d}/reconcile_kernel_trace.csv")) if "k_lqer_gemm" in r["Kernel_Name"]]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr]
steps = runs["profiled"]["steps"]
# the process ends with: timed region (20, with event pairs) + uninstrumented region (20); the 20 before the last 20 are the timed ones
if len(dur) >= 2 * steps:
    timed = dur[-2 * steps:-steps]
    last = dur[-steps:]
    print(f"  trace, the {steps} launches of the TIMED region (the ones the events sample): average {sum(timed) / len(timed):.2f} us; "
          f"the {steps} of the event-free region behind it: {sum(last) / len(last):.2f} us")
    ev = runs["profiled"]["roofline"]["avg_launch_us"]
    print(f"  profiled run: events {ev:.2f} us vs trace of the same region {sum(timed) / len(timed):.2f} us -> ratio {ev / (sum(timed) / len(timed)):.3f}")
p = (runs["plain1"]["roofline"]["avg_launch_us"] + runs["plain2"]["roofline"]["avg_launch_us"]) / 2
print(f"  unprofiled (mean of the two runs) {p:.2f} us vs profiled by the same events {runs['profiled']['roofline']['avg_launch_us']:.2f} us: "
      f"the tracer itself costs x{runs['profiled']['roofline']['avg_launch_us'] / p:.3f} on this kernel")
