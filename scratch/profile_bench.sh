#!/bin/bash
# rocprofv3 passes over bench.py (round 3): for the headline precision (fp32) and for bf16 --
#   kernel stats of the inference leg alone and with the training leg, HBM fetch / write counters and matrix-core counters in
#   runs of their own (with the training leg, so the training kernels have figures too), and a single-stream kernel trace
#   (DN_TUNE_NO_WGRAD_STREAM=1: with the weight gradients on a second stream, overlapping kernels stretch each other).
# Results under gpurun_out/; scratch/make_profiles.py turns them into profiles/.   usage: profile_bench.sh [fp32|bf16]
set -e
R=$GRAFT_REPO_ROOT
DT=${1:-fp32}
cd /tmp && export TMPDIR=/tmp
B="--dtype $DT --no-cpu-baseline --no-legs"
O=$R/gpurun_out/r03_$DT
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_infer -o p --output-format csv -- python3 $R/bench.py --steps 10 $B --no-train > $O/prof_infer_bench.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_full -o p --output-format csv -- python3 $R/bench.py --steps 10 $B > $O/prof_full_bench.log 2>&1
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- python3 $R/bench.py --steps 4 --warmup 1 $B > $O/pmc_fetch.log 2>&1
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o p --output-format csv -- python3 $R/bench.py --steps 4 --warmup 1 $B > $O/pmc_write.log 2>&1
# matrix-core utilisation per kernel: MFMA-busy cycles (summed over SIMDs) / (CU-busy cycles x 4 SIMDs)
timeout -s KILL 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $O/pmc_mfma -o p --output-format csv -- python3 $R/bench.py --steps 4 --warmup 1 $B > $O/pmc_mfma.log 2>&1
export DN_TUNE_NO_WGRAD_STREAM=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_full_1stream -o p --output-format csv -- python3 $R/bench.py --steps 10 $B > $O/prof_full_1stream_bench.log 2>&1
unset DN_TUNE_NO_WGRAD_STREAM
grep -h '"metric"' $O/prof_infer_bench.log | tail -1 | cut -c1-300
# keep what make_profiles.py reads small enough to travel back: the per-dispatch traces are not needed
find $O -name "*kernel_trace.csv" -delete
ls -la $O/*/ | head -40
