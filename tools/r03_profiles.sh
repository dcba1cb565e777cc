#!/bin/bash
# Round-3 evidence on the GPU box: rocprofv3 --kernel-trace --stats of the bench command per workload, then the --pmc
# passes (tools/pmc_bench.sh) per workload.  usage: tools/r03_profiles.sh <outdir under gpurun_out> <workload>...
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for w in "$@"; do
  extra="--layers 1 --steps 4 --warmup 1"
  [ $w = c2 ] && extra="--steps 50 --warmup 10"
  [ $w = d1 -o $w = d16 ] && extra="--steps 96 --warmup 48"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -- python3 $R/bench.py --workload $w $extra --no-cpu-baseline --no-module --no-two-streams \
      > $O/bench_under_rocprof_$w.json 2> $O/trace_$w.err
  cp $(find $O/trace_$w -name '*kernel_stats.csv' | head -1) $O/kernel_stats_$w.csv
  rm -rf $O/trace_$w
  echo "trace $w done"
  (cd $R && tools/pmc_bench.sh $w $(basename $O)/pmc_$w)
  cp $O/pmc_$w/summary_$w.json $O/traffic_$w.json
  rm -rf $O/pmc_$w/pass*/
  echo "pmc $w done"
done
