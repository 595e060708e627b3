#!/bin/bash
# where the waves of one microbenchmarked kernel spend their cycles (SQ wait / active buckets, LDS conflicts), one --pmc pass
# usage: run_pmc_waits.sh <microbench kernel> [reps]
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
K=$1; N=${2:-30}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS \
  --output-format csv -d $R/gpurun_out/pmc_waits/$K -o r -- python3 $R/tools/microbench.py $K $N > $R/gpurun_out/pmc_waits_$K.log 2>&1
find $R/gpurun_out/pmc_waits/$K -name "*counter_collection.csv"
