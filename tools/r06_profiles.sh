#!/bin/bash
# Round-6 evidence on the GPU box, per workload: rocprofv3 --kernel-trace --stats of the bench command, then the --pmc passes
# (tools/pmc_bench.sh: FETCH_SIZE and WRITE_SIZE alone in their own passes, SQ / TCC sets in two more).
# usage: tools/r06_profiles.sh <outdir under gpurun_out> <workload>...
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for w in "$@"; do
  extra="--layers 1 --steps 4 --warmup 1"
  case $w in
    c2|c2int|c2introw|c2w8a8|c2w8a8m8k) extra="--steps 50 --warmup 10 --no-configs";;
    d1|d16) extra="--steps 96 --warmup 48";;
    d1layer) extra="--steps 50 --warmup 10";;
  esac
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -- python3 $R/bench.py --workload $w $extra --no-cpu-baseline --no-module --no-two-streams \
      > $O/bench_under_rocprof_$w.json 2> $O/trace_$w.err
  cp $(find $O/trace_$w -name '*kernel_stats.csv' | head -1) $O/kernel_stats_$w.csv
  rm -rf $O/trace_$w
  echo "trace $w done"
  case $w in d1|d16|d1layer) continue;; esac
  (cd $R && tools/pmc_bench.sh $w $(basename $O)/pmc_$w)
  cp $O/pmc_$w/summary_$w.json $O/traffic_$w.json
  rm -rf $O/pmc_$w/pass*/
  echo "pmc $w done"
done
