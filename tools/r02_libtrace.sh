#!/bin/bash
# kernel mix of `bench.py --workload W --layers 2` for another build of the library (A/B of side kernels inside a model workload):
# usage (GPU box): tools/r02_libtrace.sh <workload> <lib.so> ...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/libtrace; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
W=$1; shift
for lib in "$@"; do
  n=$(basename $lib .so)
  cat > $O/run_$n.py <<PY
import os, sys, runpy
sys.path.insert(0, "$R")
from lqer_amd import _lib
_lib.LIB_PATH = os.path.abspath("$R/$lib")
sys.argv = ["bench.py", "--workload", "$W", "--layers", "2", "--no-cpu-baseline", "--no-check", "--no-module", "--no-two-streams", "--steps", "20"]
runpy.run_path("$R/bench.py", run_name="__main__")
PY
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$n -- python3 $O/run_$n.py > $O/bench_$n.json 2> $O/bench_$n.err
  f=$(find $O/t_$n -name '*kernel_stats.csv' | head -1)
  python3 - $f $n <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    if int(row["Calls"]) >= 40:
        print(f'{sys.argv[2]:14s} {row["Name"][:80]:80s} calls {row["Calls"]:>6s} avg {float(row["AverageNs"])/1e3:8.2f} us')
PY
  tail -1 $O/bench_$n.json | cut -c1-120
  rm -rf $O/t_$n
done
