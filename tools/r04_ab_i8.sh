#!/bin/bash
# A/B of the int8 GEMM builds on one box: tools/ab_i8.py for every --lib given, at the three Llama-13B shapes (+ one block per row)
# usage: tools/r04_ab_i8.sh OUT.log lib1.so lib2.so ...
out=$1; shift
for s in "5120 5120 128" "5120 13824 128" "13824 5120 128" "5120 5120 -1"; do
  set -- $s "$@"; K=$1; N=$2; WB=$3; shift 3
  for lib in "$@"; do
    echo "== K=$K N=$N wblock=$WB lib=$lib" >> $out
    python tools/ab_i8.py --K $K --N $N --wblock $WB --rounds 6 --iters 8 --lib $lib 2>&1 | grep -E "rel-L2|int8 " >> $out
  done
done
