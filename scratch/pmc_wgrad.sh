#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in FULL NO_COMPUTE NO_STAGE; do $R/scratch/wgrad_micro_$v; done
P1="SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_CYCLES"
i=0
for P in "$P1" "$P2"; do i=$((i+1)); timeout -s KILL 90 rocprofv3 --pmc $P -d $R/gpurun_out/pmc_wg_$i -o p --output-format csv -- $R/scratch/wgrad_micro_FULL > /dev/null; done
