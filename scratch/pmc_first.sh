#!/bin/bash
# PMC passes over the first-convolution micro-benchmark; one rocprofv3 run per pass.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
P1="SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_CYCLES"
P3="SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM"
i=0
for P in "$P1" "$P2" "$P3"; do i=$((i+1)); timeout -s KILL 90 rocprofv3 --pmc $P -d $R/gpurun_out/pmc_first_$i -o p --output-format csv -- $R/scratch/first_micro_FULL > /dev/null; done
