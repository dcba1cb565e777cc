#!/bin/bash
# kernel mix of a model workload's three timed regions (C ABI, module, shared inputs): rocprofv3 --kernel-trace --stats on bench.py
# usage (GPU box): tools/r02_sharedtrace.sh c3
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sharedtrace; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/bench.py --workload $1 --layers 2 --no-cpu-baseline --no-check --steps 20 > $O/bench_$1.json 2> $O/bench_$1.err
f=$(find $O/t -name '*kernel_stats.csv' | head -1)
python3 - $f <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    if int(row["Calls"]) >= 40:
        print(f'{row["Name"][:90]:90s} calls {row["Calls"]:>6s} avg {float(row["AverageNs"])/1e3:8.2f} us')
PY
rm -rf $O/t
