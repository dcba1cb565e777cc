#!/bin/bash
# Round 6: the multi-round exchange of the B_out row maxima (the default) against the pre-pass launch (LQER_TUNE_AMAX_NO_MRX, bench.py --tuning 0x4000000),
# c3int, alternating runs on one box; then a kernel trace with it.   usage: tools/r06_mrx.sh <outdir>
set -e
O=gpurun_out/$1; mkdir -p $O
pick='import json,sys
r=json.load(sys.stdin); print(sys.argv[1], r["value"], r["ms_per_step"], [ (p["K"],p["N"],p["avg_launch_us"]) for p in r["roofline"]["per_shape"]])'
for rep in 1 2 3; do
  for t in 0 0x4000000; do
    timeout -k 10 300 python bench.py --workload c3int --no-cpu-baseline --no-two-streams --no-module --tuning $t 2>/dev/null | python -c "$pick" "c3int tuning=$t" >> $O/mrx.txt
  done
done
cat $O/mrx.txt
tools/r06_trace_cmd.sh $1 c3int_mrx $GRAFT_REPO_ROOT/bench.py --workload c3int --no-cpu-baseline --no-two-streams --no-module --layers 4 | cut -c1-150 | head -8
