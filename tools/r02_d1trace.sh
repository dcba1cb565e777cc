#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/d1trace; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/tools/decode_probe.py > /dev/null 2>&1
f=$(find $O/t -name '*kernel_stats.csv' | head -1)
python3 - $f <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    if "decode1" in row["Name"] or "smallm" in row["Name"] or "quant_xa16" in row["Name"]:
        print(f'{row["Name"][:60]:60s} calls {row["Calls"]:>6s} avg {float(row["AverageNs"])/1e3:8.2f} us min {float(row["MinNs"])/1e3:8.2f} max {float(row["MaxNs"])/1e3:8.2f}')
PY
rm -rf $O/t
