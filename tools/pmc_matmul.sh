#!/bin/bash
# PMC passes over tools/ab_matmul.py (the quantized attention products). usage (GPU box): tools/pmc_matmul.sh <outdir under gpurun_out>
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/pass$i -- python3 $R/tools/ab_matmul.py --rounds 2 --iters 3 > /dev/null 2>&1
done
python3 - $O <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "qmatmul" in k or "qmm_bimage" in k:
            a = acc[k[:60]][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    print(k)
    for c, (s, n) in sorted(d.items()):
        print(f"   {c:28s} {s / n:16.1f}")
PY
rm -rf $O/pass*
