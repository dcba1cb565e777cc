#!/bin/bash
# per-kernel durations of the whole step (tools/ab_step.py) for each build given: one rocprofv3 --kernel-trace --stats pass per build
# usage (GPU box): tools/r02_steptrace.sh build/abl/lib_a.so lqer_amd/liblqer_hip.so ...   [ARGS="--M 2048 --K 4096"]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/steptrace; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for lib in "$@"; do
  n=$(basename $lib .so)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_$n -- python3 $R/tools/ab_step.py $ARGS --rounds 3 --iters 40 $R/$lib > /dev/null 2>&1
  python3 - $O $n <<'PY'
import csv, glob, sys
O, n = sys.argv[1], sys.argv[2]
for f in glob.glob(f"{O}/tr_{n}/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if int(row["Calls"]) >= 100:
            print(f"{n:16s} {row['Name'][:70]:70s} calls {row['Calls']:>5s} avg {float(row['AverageNs']) / 1e3:7.2f} us")
PY
  rm -rf $O/tr_$n
done
