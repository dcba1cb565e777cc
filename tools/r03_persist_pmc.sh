#!/bin/bash
# Persistent tile loop of the 128-row bf16 kernel (tools/experiments/gemm_w4a8_persistent.patch) against one workgroup per tile at
# Llama's 4096 -> 11008 (688 tiles, 2.7 rounds): durations from --kernel-trace, one small --pmc pass (separate runs).
# usage (GPU box): tools/r03_persist_pmc.sh <lib with the persistent kernel>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/persist_pmc; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 $R/tools/ab_gemm.py --no-persist --N 11008 --rounds 3 --iters 40 $R/$1 > $O/ab.log 2>&1
echo "trace done"
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/pmc -- python3 $R/tools/ab_gemm.py --no-persist --N 11008 --rounds 2 --iters 20 $R/$1 > /dev/null 2>&1
echo "pmc done"
python3 - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(f"{O}/pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = "persistent" if "k_lqer_gemm_p" in row["Kernel_Name"] else ("per tile" if "k_lqer_gemm<" in row["Kernel_Name"] else None)
        if k:
            a = acc[k][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
dur = {}
for f in glob.glob(f"{O}/tr/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_lqer_gemm_p" in row["Name"]: dur["persistent"] = (float(row["AverageNs"]) / 1e3, int(row["Calls"]))
        elif "k_lqer_gemm<" in row["Name"]: dur["per tile"] = (float(row["AverageNs"]) / 1e3, int(row["Calls"]))
for k in ("per tile", "persistent"):
    c = {n: v[0] / v[1] for n, v in acc[k].items()}
    print(f"{k:10s} avg {dur.get(k, (0, 0))[0]:.2f} us over {dur.get(k, (0, 0))[1]} launches | " + " | ".join(f"{n} {v:.4g}" for n, v in sorted(c.items())))
    if c.get("SQ_BUSY_CYCLES"):
        print(f"           MFMA busy / SIMD = {c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024:.0f} cycles, GRBM_GUI_ACTIVE / 8 = {c['GRBM_GUI_ACTIVE'] / 8:.0f} cycles -> MFMA pipe busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (c['GRBM_GUI_ACTIVE'] / 8):.1%}; VALU instructions per wave {c['SQ_INSTS_VALU'] / (688 * 8):.0f} (per tile)")
PY
rm -rf $O/tr $O/pmc
