#!/bin/bash
# Round-6 evidence beside the bench lines: in-kernel timelines (clock-probe build), then per workload the kernel trace and the --pmc passes.
# usage: tools/r06_evidence.sh <outdir under gpurun_out> <workload>...
set -e
R=$GRAFT_REPO_ROOT
O=gpurun_out/$1; shift; mkdir -p $O
{
  echo "Round 6: in-kernel timeline of the int8 GEMM's exchange instantiation (python tools/clock_probe_i8.py build/abl/liblqer_cp.so --M 2048 --K K --N N --r 32;"
  echo "-DLQER_CLOCKPROBE build of the FINAL sources; medians over the 256 workgroups; cycles from a wave's own start)"
  for shape in "4096 4096" "11008 4096" "4096 11008"; do
    set -- $shape "$@"; K=$1; N=$2; shift 2
    timeout -k 10 300 python tools/clock_probe_i8.py build/abl/liblqer_cp.so --M 2048 --K $K --N $N --r 32 2>&1 | grep -v amdgpu.ids
  done
} > $O/i8_timeline.txt
{
  echo "Round 6: in-kernel timeline of the int8 route's one-launch activation kernel (python tools/clock_probe_a8.py build/abl/liblqer_cp.so --K K)"
  for K in 4096 11008; do timeout -k 10 300 python tools/clock_probe_a8.py build/abl/liblqer_cp.so --K $K 2>&1 | grep -v amdgpu.ids; done
} > $O/act8_timeline.txt
cat $O/i8_timeline.txt $O/act8_timeline.txt
tools/r06_profiles.sh $(basename $O) "$@"
