#!/bin/bash
# L2 hit / miss / read requests of one microbenchmarked kernel (one --pmc pass, 3 TCC counters)
# usage: run_pmc_l2.sh <microbench kernel> [reps]
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
K=$1; N=${2:-30}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_READ_sum --output-format csv -d $R/gpurun_out/pmc_l2/$K -o r \
  -- python3 $R/tools/microbench.py $K $N > $R/gpurun_out/pmc_l2_$K.log 2>&1
find $R/gpurun_out/pmc_l2/$K -name "*counter_collection.csv"
