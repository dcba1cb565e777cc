#!/bin/bash
# kernel durations and launch-to-launch periods of the one-launch decode forward for several builds, from rocprofv3's kernel trace
# usage: tools/r04_d1trace.sh <outdir under gpurun_out> <M> lib1.so lib2.so ...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; M=$2; shift 2
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $(dirname $lib))
  rocprofv3 --kernel-trace --output-format csv -d $O/tr_$tag -- python3 $R/tools/ab_decode.py --M $M --rounds 2 --iters 300 $R/$lib > $O/tr_$tag.log 2>&1
  f=$(find $O/tr_$tag -name '*kernel_trace.csv' | head -1)
  python3 - "$f" "$tag" >> $O/d1trace.txt <<'PY'
import csv, sys, statistics
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_decode1" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
per = [(int(b["Start_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3 for a, b in zip(rows, rows[1:])]
per = [p for p in per if p < 50]
print(f"{sys.argv[2]:10s} launches {len(rows)}  duration median {statistics.median(dur):.2f} us  period median {statistics.median(per):.2f} us  gap {statistics.median(per) - statistics.median(dur):.2f} us  vgpr {rows[0].get('VGPR_Count', '?')} lds {rows[0].get('LDS_Block_Size', '?')} scratch {rows[0].get('Scratch_Size', '?')}")
PY
  rm -rf $O/tr_$tag
done
cat $O/d1trace.txt
