#!/bin/bash
# rocprofv3 --pmc passes over bench.py for one workload (one counter set per pass; FETCH_SIZE / WRITE_SIZE alone, as
# MI355X_MICROARCH.md prescribes), then the per-kernel summary -> profiles/.
# usage (on the GPU box, from the repository root):  tools/pmc_bench.sh <workload> <outdir under gpurun_out> [extra bench args]
W=$1; OUT=$2; shift 2
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$OUT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/$OUT/pass$i -- python3 $R/bench.py --workload $W --layers 1 --steps 2 --warmup 1 \
      --prewarm-ms 0 --no-cpu-baseline --no-check --no-module --no-two-streams "$@" > $R/gpurun_out/$OUT/pass$i.log 2>&1
done
python3 $R/tools/pmc_traffic.py $R/gpurun_out/$OUT $W > $R/gpurun_out/$OUT/summary_$W.json
