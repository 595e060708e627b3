#!/bin/bash
# Round-3 wait-state passes of the decode attention on the linear planes (one --pmc pass each; summaries: tools/pmc_waits.py -> profiles/pmc_waits.json)
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
cd /tmp
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS"
run() {  # name env kernel reps
  env $2 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_waits_r03/$1 -o r -- python3 $R/tools/microbench.py $3 $4 > $R/gpurun_out/pmc_waits_r03_$1.log 2>&1 || exit 1
}
run da2k "MB_LIN=1" decode_attn 40
run da32k_7b "MB_HEADS=28,4 MB_L=32768" decode_attn_long 6
run da131k_7b "MB_HEADS=28,4 MB_L=131072" decode_attn_long 4
cd $R
f() { find gpurun_out/pmc_waits_r03/$1 -name "*counter_collection.csv" | head -1; }
python3 tools/pmc_waits.py "decode_attn_split (2B window, linear planes)" decode_attn_split_kernel $(f da2k)
python3 tools/pmc_waits.py "decode_attn_stream 28q/4kv x 32768 (linear planes)" decode_attn_stream_kernel $(f da32k_7b)
python3 tools/pmc_waits.py "decode_attn_stream 28q/4kv x 131072 (linear planes)" decode_attn_stream_kernel $(f da131k_7b)
