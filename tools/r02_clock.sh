#!/bin/bash
# effective clock of the fused GEMM under tools/ab_gemm.py for each build: GRBM_GUI_ACTIVE / 8 XCDs / kernel duration
# (MI355X_MICROARCH.md, DVFS give-back) - one --pmc pass and one --kernel-trace pass per build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/clock; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for lib in "$@"; do
  n=$(basename $lib .so)
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $O/pmc_$n -- python3 $R/tools/ab_gemm.py --rounds 3 --iters 40 $R/$lib > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_$n -- python3 $R/tools/ab_gemm.py --rounds 3 --iters 40 $R/$lib > /dev/null 2>&1
  python3 - $O $n <<'PY'
import csv, glob, sys, collections
O, n = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(f"{O}/pmc_{n}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_lqer_gemm" in row["Kernel_Name"]:
            a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
dur = None
for f in glob.glob(f"{O}/tr_{n}/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_lqer_gemm" in row["Name"]: dur = float(row["AverageNs"])
c = {k: v[0] / v[1] for k, v in acc.items()}
print(n, "avg kernel %.2f us" % (dur / 1e3), "| GRBM_GUI_ACTIVE/8 = %.0f cycles -> %.3f GHz" % (c["GRBM_GUI_ACTIVE"] / 8, c["GRBM_GUI_ACTIVE"] / 8 / dur),
      "| MFMA busy %.0f cyc/SIMD" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024), "| VALU insts %.0f" % c["SQ_INSTS_VALU"], "| wave cycles %.0f" % c["SQ_WAVE_CYCLES"])
PY
  rm -rf $O/pmc_$n $O/tr_$n
done
