#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench workload (no PMC): kernel_stats_<w>.csv + the bench line under the tracer
# usage: tools/r05_trace.sh <outdir under gpurun_out> <workload> [bench args...]
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; w=$2; shift 2
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -- python3 $R/bench.py --workload $w "$@" --no-cpu-baseline --no-module --no-two-streams \
    > $O/bench_under_rocprof_$w.json 2> $O/trace_$w.err
cp $(find $O/trace_$w -name '*kernel_stats.csv' | head -1) $O/kernel_stats_$w.csv
rm -rf $O/trace_$w
echo "trace $w done"
