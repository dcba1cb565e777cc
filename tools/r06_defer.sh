#!/bin/bash
# Round 6: the B_out re-quantization of the 128-row bf16 kernel under the main loop (DEFER) against in front of it (LQER_TUNE_BOUT_IN_PROLOGUE).
# usage: tools/r06_defer.sh <outdir> [pytest files]
set -e
O=gpurun_out/$1; shift; mkdir -p $O
timeout -k 10 900 python -m pytest ${@:-tests/test_gpu_tile_rows.py tests/test_gpu_side_path.py tests/test_gpu_fullsize.py tests/test_gpu_act16_fused.py} -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 300 python tools/clock_probe.py build/abl/liblqer_cp.so 4096 32 2>&1 | grep -v amdgpu.ids > $O/cp_c2.txt
cat $O/cp_c2.txt
pick='import json,sys
r=json.load(sys.stdin); print(sys.argv[1], r["value"], r["ms_per_step"], [ (p["K"],p["N"],p["avg_launch_us"]) for p in r["roofline"]["per_shape"]])'
for rep in 1 2 3; do
  for t in 0 0x2000000; do
    timeout -k 10 300 python bench.py --workload c2 --no-cpu-baseline --no-two-streams --no-configs --no-module --tuning $t 2>/dev/null | python -c "$pick" "c2 tuning=$t" >> $O/defer.txt
  done
done
for t in 0 0x2000000; do
  timeout -k 10 300 python bench.py --workload c3 --no-cpu-baseline --no-two-streams --no-module --tuning $t 2>/dev/null | python -c "$pick" "c3 tuning=$t" >> $O/defer.txt
  timeout -k 10 300 python bench.py --workload c5 --no-cpu-baseline --no-two-streams --no-module --tuning $t 2>/dev/null | python -c "$pick" "c5 tuning=$t" >> $O/defer.txt
done
cat $O/defer.txt
