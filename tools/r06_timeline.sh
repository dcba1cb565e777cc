#!/bin/bash
# Round 6: in-kernel timeline of the int8 GEMM's exchange instantiation (clock-probe build, prologue sections + gather polls).
# usage: tools/r06_timeline.sh <outdir> [lib ...]
set -e
O=gpurun_out/$1; shift; mkdir -p $O
LIBS=${@:-build/abl/liblqer_cp.so}
for lib in $LIBS; do
  for shape in "4096 4096" "11008 4096"; do
    set -- $shape
    echo "== $lib K=$1 N=$2" >> $O/timeline.txt
    timeout -k 10 300 python tools/clock_probe_i8.py $lib --M 2048 --K $1 --N $2 --r 32 2>&1 | grep -v amdgpu.ids >> $O/timeline.txt
  done
done
cat $O/timeline.txt
