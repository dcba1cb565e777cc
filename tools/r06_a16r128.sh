#!/bin/bash
# Round 6: the one-launch block-16 activation kernel at rank 128 (c5: OPT-6.7B) against k_quant_xa128 + k_xa_reduce4 (LQER_TUNE_ACT16_SPLIT)
set -e
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_act16_fused.py tests/test_gpu_fullsize.py -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
pick='import json,sys
r=json.load(sys.stdin); print(sys.argv[1], r["value"], r["ms_per_step"], [ (p["K"],p["N"],p["avg_launch_us"]) for p in r["roofline"]["per_shape"]])'
for rep in 1 2 3; do
  for t in 0 0x800000; do
    timeout -k 10 300 python bench.py --workload c5 --no-cpu-baseline --no-two-streams --no-module --tuning $t 2>/dev/null | python -c "$pick" "c5 tuning=$t" >> $O/a16r128.txt
  done
done
cat $O/a16r128.txt
