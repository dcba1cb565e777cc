#!/bin/bash
# kernel durations of the activation quantizer builds under rocprofv3 (A/B tool, C2 operands)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/qtrace; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for lib in "$@"; do
  n=$(basename $lib .so)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -- python3 $R/tools/ab_quant.py --nocheck $R/$lib > /dev/null 2>&1
  f=$(find $O/$n -name '*kernel_stats.csv' | head -1)
  echo "== $n"; grep -E "k_quant|k_xa_reduce" $f | awk -F'","' '{print substr($1,1,40), $2, $4}'
  rm -rf $O/$n
done
