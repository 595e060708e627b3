#!/bin/bash
# Round-2 MFMA-pipe utilisation passes (rocprofv3 --pmc beside --kernel-trace only) of the large-M GEMM tiles and of the LDS-DMA
# prefill attention; reduced by tools/pmc_mfma.py into profiles/pmc_mfma.json
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
cd /tmp
for lab in "gemm_7b_down 7b_down" "gemm_7b_gate_up 7b_gate"; do
  set -- $lab
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv \
    -d $R/gpurun_out/pmc_mfma_r02/$1 -o r -- python3 $R/tools/gemm_big.py "${2/_/ }" > $R/gpurun_out/pmc_mfma_r02_$1.log 2>&1
done
MB_LS=32768 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv \
  -d $R/gpurun_out/pmc_mfma_r02/prefill_attn -o r -- python3 $R/tools/prefill_attn_long.py > $R/gpurun_out/pmc_mfma_r02_prefill_attn.log 2>&1
find $R/gpurun_out/pmc_mfma_r02 -name "*counter_collection.csv"
