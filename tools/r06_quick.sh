#!/bin/bash
# Round 6 quick loop: parity tests of the int8 route, the in-kernel timeline of the clock-probe build, GEMM / forward timing (the
# one-launch activation side against the three launches, same build).   usage: tools/r06_quick.sh <outdir> [pytest args]
set -e
O=gpurun_out/$1; shift; mkdir -p $O
timeout -k 10 900 python -m pytest ${@:-tests/test_gpu_int8.py tests/test_gpu_act8_fused.py tests/test_gpu_tile_rows.py tests/test_gpu_fullsize.py} -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
tools/r06_timeline.sh $(basename $O) > /dev/null
cat $O/timeline.txt
for shape in "4096 4096" "11008 4096" "4096 11008"; do
  set -- $shape
  echo "== K=$1 N=$2" >> $O/ab.txt
  timeout -k 10 300 python tools/ab_i8.py --split --M 2048 --K $1 --N $2 --r 32 2>&1 | grep "median" >> $O/ab.txt
done
cat $O/ab.txt
