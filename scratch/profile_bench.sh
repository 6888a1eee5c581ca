#!/bin/bash
# rocprofv3 passes over bench.py: kernel stats (inference only, then with the training leg), then HBM fetch / write
# counters in runs of their own (with the training leg, so the training kernels have traffic figures too).
# Results under gpurun_out/; scratch/make_profiles.py turns them into profiles/.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-legs"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_infer -o p --output-format csv -- python3 $R/bench.py --steps 10 $B --no-train > $R/gpurun_out/prof_infer_bench.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_full -o p --output-format csv -- python3 $R/bench.py --steps 10 $B > $R/gpurun_out/prof_full_bench.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o p --output-format csv -- python3 $R/bench.py --steps 4 --warmup 1 $B > $R/gpurun_out/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o p --output-format csv -- python3 $R/bench.py --steps 4 --warmup 1 $B > $R/gpurun_out/pmc_write.log 2>&1
grep -h '"metric"' $R/gpurun_out/prof_infer_bench.log | tail -1
grep -h '"metric"' $R/gpurun_out/prof_full_bench.log | tail -1
# matrix-core utilisation per kernel: MFMA-busy cycles (summed over SIMDs) / (CU-busy cycles x 4 SIMDs)
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $R/gpurun_out/pmc_mfma -o p --output-format csv -- python3 $R/bench.py --steps 4 --warmup 1 $B > $R/gpurun_out/pmc_mfma.log 2>&1
# per-kernel durations without the second stream (weight gradients overlapping the BatchNorm passes stretch both): the time base
# of profiles/*_hbm_bandwidth_per_kernel.csv
export DN_TUNE_NO_WGRAD_STREAM=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_full_1stream -o p --output-format csv -- python3 $R/bench.py --steps 10 $B > $R/gpurun_out/prof_full_1stream_bench.log 2>&1
unset DN_TUNE_NO_WGRAD_STREAM
