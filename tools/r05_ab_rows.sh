#!/bin/bash
# int8 kernel: 128-row vs 256-row tiles (pinned) vs the bf16 route at the Llama-7B shapes, M = 2048, rank 32 (c2int / c3int)
# usage: tools/r05_ab_rows.sh OUT.log
out=$1
for s in "4096 4096 128" "4096 11008 128" "11008 4096 128" "4096 4096 -1"; do
  set -- $s; K=$1; N=$2; WB=$3
  echo "== M=2048 K=$K N=$N r=32 wblock=$WB" >> $out
  python tools/ab_i8.py --M 2048 --K $K --N $N --r 32 --wblock $WB --rounds 6 --iters 10 --rows 2>&1 | grep -E "rel-L2|bit-identical|i8r|bf16 " >> $out
done
