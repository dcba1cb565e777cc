#!/bin/bash
# Round-4 evidence on the GPU box, per workload: rocprofv3 --kernel-trace --stats of the bench command, then the --pmc passes
# (tools/pmc_bench.sh).  For c2 additionally the RECONCILIATION the round-3 verdict asked for: the driver's command
# unprofiled, profiled, unprofiled - back to back on this one box - with the dominant kernel's time by HIP events (all three
# runs) and by the trace (the profiled one) side by side (tools/r04_reconcile.py -> reconcile_c2.txt).
# usage: tools/r04_profiles.sh <outdir under gpurun_out> <workload>...
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for w in "$@"; do
  extra="--layers 1 --steps 4 --warmup 1"
  [ $w = c2 ] && extra="--steps 50 --warmup 10 --no-configs"
  [ $w = d1 -o $w = d16 ] && extra="--steps 96 --warmup 48"
  if [ $w = c2 ]; then
    drv="--gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline --no-module --no-two-streams"
    python3 $R/bench.py $drv > $O/reconcile_plain1.json 2> /dev/null
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_rec -- python3 $R/bench.py $drv > $O/reconcile_profiled.json 2> $O/trace_rec.err
    cp $(find $O/trace_rec -name '*kernel_stats.csv' | head -1) $O/reconcile_kernel_stats.csv
    cp $(find $O/trace_rec -name '*kernel_trace.csv' | head -1) $O/reconcile_kernel_trace.csv
    rm -rf $O/trace_rec
    python3 $R/bench.py $drv > $O/reconcile_plain2.json 2> /dev/null
    python3 $R/tools/r04_reconcile.py $O > $O/reconcile_c2.txt
    rm -f $O/reconcile_kernel_trace.csv
    echo "reconcile c2 done"
  fi
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
