#!/bin/bash
# kernel durations of the int8 route (tools/ab_i8.py) for the default build and for another build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/i8trace; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() {
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$1 -- python3 $R/tools/ab_i8.py --rounds 3 --iters 5 $2 $3 $4 $5 $6 > /dev/null 2>&1
  f=$(find $O/$1 -name '*kernel_stats.csv' | head -1)
  echo "== $1"; python3 - $f <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    n = row["Name"]
    if any(k in n for k in ("gemm_i8", "bout_amax", "gemm_m256")): print(f'{n[:70]:70s} calls {row["Calls"]:>5s} avg {float(row["AverageNs"])/1e3:9.1f} us')
PY
  rm -rf $O/$1
}
run t16 "$@"
run m32 --lib $R/build/abl/lib_i8m32.so "$@"
