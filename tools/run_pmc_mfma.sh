#!/bin/bash
# rocprofv3 --pmc pass (kernel trace only beside it) over three microbenchmarks; CSVs land in gpurun_out/pmc_mfma/<label>/
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $R/gpurun_out/pmc_mfma_counters.txt 2>&1 || true
for lab in gemm_fc1 gemm_gu vit_attn prefill_attn; do
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv \
    -d $R/gpurun_out/pmc_mfma/$lab -o r -- python3 $R/tools/microbench.py $lab 40 > $R/gpurun_out/pmc_mfma_$lab.log 2>&1
done
find $R/gpurun_out/pmc_mfma -name "*.csv" | head -20
