#!/bin/bash
# Round 6: the int8 kernel's exchange instantiation against the round-5 build (build/abl/liblqer_r5*.so = HEAD of round 5), one box:
# int8 parity tests, in-kernel timelines (clock-probe builds), A/B of GEMM and whole forward.   usage: tools/r06_i8_ab.sh <outdir>
set -e
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_int8.py tests/test_gpu_tile_rows.py tests/test_gpu_fullsize.py -x -q > $O/pytest_i8.log 2>&1 || { tail -40 $O/pytest_i8.log; exit 1; }
tail -3 $O/pytest_i8.log
for shape in "4096 4096" "11008 4096" "4096 11008"; do
  set -- $shape
  for lib in build/abl/liblqer_r5_cp.so build/abl/liblqer_cp.so; do
    echo "== $lib K=$1 N=$2" >> $O/timeline.txt
    timeout -k 10 300 python tools/clock_probe_i8.py $lib --M 2048 --K $1 --N $2 --r 32 >> $O/timeline.txt 2>&1
  done
done
cat $O/timeline.txt
for shape in "4096 4096" "11008 4096" "4096 11008"; do
  set -- $shape
  for lib in build/abl/liblqer_r5.so lqer_amd/liblqer_hip.so; do
    echo "== $lib K=$1 N=$2" >> $O/ab.txt
    timeout -k 10 300 python tools/ab_i8.py --lib $lib --M 2048 --K $1 --N $2 --r 32 --amax >> $O/ab.txt 2>&1
  done
done
cat $O/ab.txt
