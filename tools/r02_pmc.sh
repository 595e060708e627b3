#!/bin/bash
# Round-2 PMC passes: HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes) of the 7B gate/up GEMV and of the long-cache decode attention
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
cd /tmp
run() {  # name env kernel reps
  for c in FETCH_SIZE WRITE_SIZE; do
    env $2 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_r02/$1/$c -o r -- python3 $R/tools/microbench.py $3 $4 > $R/gpurun_out/pmc_r02_$1_$c.log 2>&1
  done
}
MB_MODEL=7b run gu7b "MB_MODEL=7b" dec_gate_up 3
run da32k_7b "MB_HEADS=28,4 MB_L=32768" decode_attn_long 6
run da32k_2b "MB_HEADS=12,2 MB_L=32768" decode_attn_long 6
run da131k_7b "MB_HEADS=28,4 MB_L=131072" decode_attn_long 4
find $R/gpurun_out/pmc_r02 -name "*counter_collection.csv"
