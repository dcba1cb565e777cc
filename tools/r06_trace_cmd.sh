#!/bin/bash
# rocprofv3 --kernel-trace --stats over one python command (no PMC): kernel stats csv into gpurun_out/<outdir>/<tag>_kernel_stats.csv
# usage: tools/r06_trace_cmd.sh <outdir> <tag> <python script and args...>
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; tag=$2; shift 2
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$tag -- python3 "$@" > $O/${tag}_out.txt 2> $O/${tag}_err.txt
cp $(find $O/trace_$tag -name '*kernel_stats.csv' | head -1) $O/${tag}_kernel_stats.csv
rm -rf $O/trace_$tag
cut -d, -f1-8 $O/${tag}_kernel_stats.csv | head -14
