#!/bin/bash
# Round 6: A/B of the zero-fill hand-over (lqer_quantize_act_xa_prep -> lqer_linear_gemm_prepared) on one box, alternating runs.
set -e
O=gpurun_out/$1; mkdir -p $O
pick='import json,sys
r=json.load(sys.stdin); print(sys.argv[1], r["value"], r["ms_per_step"], "module", (r.get("module") or {}).get("ms_per_step"), [ (p["K"],p["N"],p["avg_launch_us"]) for p in r["roofline"]["per_shape"]])'
for rep in 1 2 3; do
  timeout -k 10 300 python bench.py --workload c3int --no-cpu-baseline --no-two-streams 2>/dev/null | python -c "$pick" "c3int hand-over" >> $O/amax2.txt
  LQER_BENCH_NO_PREP=1 timeout -k 10 300 python bench.py --workload c3int --no-cpu-baseline --no-two-streams 2>/dev/null | python -c "$pick" "c3int memset  " >> $O/amax2.txt
done
cat $O/amax2.txt
