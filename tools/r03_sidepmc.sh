#!/bin/bash
# Round 3: what bounds C4's side kernels (k_bout_amax, k_xa_partial, k_quant_row8)?  Two rocprofv3 --pmc passes over one
# layer of the c4 workload with texture-addresser / L1 / memory-instruction counters (no --kernel-trace: counters alone),
# summarised per kernel by tools/pmc_traffic.py.   usage (GPU box, repository root): tools/r03_sidepmc.sh <outdir under gpurun_out> [workload]
OUT=$1; W=${2:-c4}
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$OUT
i=0
for set in "TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" \
           "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"; do
  i=$((i+1))
  # (a counter set the hardware cannot collect at once makes rocprofv3 abort and the child hang: bounded)
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/$OUT/pass$i -- python3 $R/bench.py --workload $W --layers 1 --steps 2 --warmup 1 \
      --prewarm-ms 0 --no-cpu-baseline --no-check --no-module --no-two-streams > $R/gpurun_out/$OUT/pass$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 $R/tools/pmc_traffic.py $R/gpurun_out/$OUT $W > $R/gpurun_out/$OUT/summary_$W.json
rm -rf $R/gpurun_out/$OUT/pass*/
