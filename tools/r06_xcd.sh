#!/bin/bash
# Round 6: XCD-local tile BLOCKS in the int8 kernel (LQER_TUNE_XCD_BLOCK) against the default map - wall time of
# GEMM and forward interleaved in one process, then FETCH_SIZE / WRITE_SIZE / L2 hits per launch of the GEMM under either map
# (separate --pmc passes, one map per process).   usage: tools/r06_xcd.sh <outdir>
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for shape in "2048 4096 4096 32" "2048 11008 4096 32" "16384 5120 5120 64" "16384 5120 13824 64" "16384 13824 5120 64"; do
  set -- $shape
  echo "== M=$1 K=$2 N=$3 r=$4" >> $O/xcd.txt
  timeout -k 10 300 python3 $R/tools/ab_i8.py --M $1 --K $2 --N $3 --r $4 --xcd 4 8 --rounds 6 2>&1 | grep "median\|XCD blocks" >> $O/xcd.txt
done
cat $O/xcd.txt
for shape in "2048 4096 4096 32" "16384 5120 5120 64"; do
  set -- $shape
  for v in int8 xcd4 xcd8; do
    for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
      tag=$(echo $set | tr ' ' '_')
      timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/pmc_${1}_${2}_${v}_$tag -- python3 $R/tools/ab_i8.py --M $1 --K $2 --N $3 --r $4 --xcd 4 8 --only $v --rounds 2 --iters 4 > /dev/null 2>&1 || true
    done
    echo "== M=$1 K=$2 N=$3 map $v" >> $O/xcd_pmc.txt
    python3 $R/tools/pmc_summary.py $O k_lqer_gemm_i8 2>/dev/null | sed "s/^/   /" > /dev/null
    for set in FETCH_SIZE WRITE_SIZE TCC_HIT_sum_TCC_MISS_sum; do python3 $R/tools/pmc_summary.py $O/pmc_${1}_${2}_${v}_$set k_lqer_gemm_i8 >> $O/xcd_pmc.txt 2>/dev/null || true; done
    rm -rf $O/pmc_${1}_${2}_${v}_*
  done
done
cat $O/xcd_pmc.txt
