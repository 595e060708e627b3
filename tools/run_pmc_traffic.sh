#!/bin/bash
# HBM traffic of one microbenchmarked kernel: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes
# usage: run_pmc_traffic.sh <microbench kernel> [reps]
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
K=$1; N=${2:-30}
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_traffic/$K/$c -o r -- python3 $R/tools/microbench.py $K $N \
    > $R/gpurun_out/pmc_traffic_${K}_$c.log 2>&1
done
find $R/gpurun_out/pmc_traffic/$K -name "*counter_collection.csv"
