#!/bin/bash
# PMC counters of the int8 GEMM kernels (tools/ab_i8.py) for the default build and for build/abl/lib_i8m32.so
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/i8pmc; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() {
  n=$1; shift
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/$n -- python3 $R/tools/ab_i8.py --rounds 2 --iters 4 "$@" > /dev/null 2>&1
  python3 - $O/$n $n <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "gemm_i8" in k:
            a = acc[k.split("(")[0][-40:]][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for k, c in acc.items():
    print(sys.argv[2], k, {n: round(v[0] / v[1]) for n, v in sorted(c.items())})
PY
  rm -rf $O/$n
}
run t16 "$@"
run m32 --lib $R/build/abl/lib_i8m32.so "$@"
