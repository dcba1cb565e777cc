#!/bin/bash
# Round 6: the B_out row maxima of the multi-round int8 launches - atomicMax cells + zero fill (by memset, or by the one-launch activation
# kernel of the same lqer_linear_forward) against segment partials (32 cells per row beyond N = 4096: LQER_TUNE_AMAX_PARTS).
# usage: tools/r06_amax.sh <outdir>
set -e
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_int8.py tests/test_gpu_act8_fused.py tests/test_gpu_fullsize.py -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
pick='import json,sys
r=json.load(sys.stdin); print(sys.argv[1], r["value"], r["ms_per_step"], "module", (r.get("module") or {}).get("ms_per_step"), [ (p["K"],p["N"],p["avg_launch_us"]) for p in r["roofline"]["per_shape"]])'
for rep in 1 2; do
for t in 0 0x80000 0x40000; do
  timeout -k 10 300 python bench.py --workload c3int --no-cpu-baseline --no-two-streams --tuning $t 2>/dev/null | python -c "$pick" "c3int tuning=$t" >> $O/amax.txt
done
done
for t in 0 0x40000; do
  timeout -k 10 300 python bench.py --workload c4 --no-cpu-baseline --no-two-streams --tuning $t 2>/dev/null | python -c "$pick" "c4 tuning=$t" >> $O/amax.txt
done
cat $O/amax.txt
