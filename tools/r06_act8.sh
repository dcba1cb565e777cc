#!/bin/bash
# Round 6: the int8 route's one-launch activation kernel - its tests, then the forward with it against the three launches (same build,
# LQER_TUNE_ACT8_SPLIT).  usage: tools/r06_act8.sh <outdir> [notest]
set -e
O=gpurun_out/$1; mkdir -p $O
if [ "$2" != "notest" ]; then
timeout -k 10 900 python -m pytest tests/test_gpu_act8_fused.py tests/test_gpu_int8.py -x -q > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
fi
for shape in "4096 4096" "11008 4096" "4096 11008"; do
  set -- $shape
  echo "== K=$1 N=$2" >> $O/ab.txt
  timeout -k 10 300 python tools/ab_i8.py --split --M 2048 --K $1 --N $2 --r 32 2>&1 | grep "median" >> $O/ab.txt
done
cat $O/ab.txt
