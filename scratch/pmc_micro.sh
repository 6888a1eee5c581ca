#!/bin/bash
# PMC passes over the wgrad micro-benchmark variants; one rocprofv3 run per pass and variant.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
P1="SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD"
P2="TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum"
P3="TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum"
P4="TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY"
for v in "$@"; do
  i=0
  for P in "$P1" "$P2" "$P3" "$P4"; do
    i=$((i+1))
    timeout -s KILL 90 rocprofv3 --pmc $P -d $R/gpurun_out/pmc2_${v}_$i -o p --output-format csv -- $R/scratch/wgrad_micro_$v
  done
done
