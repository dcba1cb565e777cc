#!/bin/bash
# Round-6 state check on the GPU box: GPU tests, then the bench lines - the driver's command first (its JSON carries `configs`:
# full-depth c3 / c3int / c5 and 8 layers of c4), then one line per workload.   usage: tools/r06_state.sh <outdir under gpurun_out>
set -e
O=gpurun_out/$1; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
S=$SECONDS
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2_driver.json 2> $O/bench_c2_driver.err
echo "driver line done in $((SECONDS - S)) s"
python bench.py --no-configs > $O/bench_c2.json 2> $O/bench_c2.err
for w in c2int c2introw c3int c3 c4 c4row c5 c2w8a8 c2w8a8m8k c3w3a16 d1 d16 d1layer; do
  python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err
  echo "$w done"
done
