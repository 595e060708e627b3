#!/bin/bash
# Round-3 PMC passes: HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes) of the decode attention on the cache's linear planes --
# the bounded window of configs[1] and the long caches of the roofline points.  Summaries: tools/pmc_traffic.py -> profiles/pmc_traffic.json
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
cd /tmp
run() {  # name env kernel reps
  for c in FETCH_SIZE WRITE_SIZE; do
    env $2 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_r03/$1/$c -o r -- python3 $R/tools/microbench.py $3 $4 > $R/gpurun_out/pmc_r03_$1_$c.log 2>&1 || exit 1
  done
}
run da2k "MB_LIN=1" decode_attn 40
run da32k_7b "MB_HEADS=28,4 MB_L=32768" decode_attn_long 6
run da32k_2b "MB_HEADS=12,2 MB_L=32768" decode_attn_long 6
run da131k_7b "MB_HEADS=28,4 MB_L=131072" decode_attn_long 4
cd $R
f() { find gpurun_out/pmc_r03/$1/$2 -name "*counter_collection.csv" | head -1; }
python3 tools/pmc_traffic.py decode_attn_split_kernel $(f da2k FETCH_SIZE) $(f da2k WRITE_SIZE) 2212812 "tools/r03_pmc.sh (microbench decode_attn, linear planes)" "Qwen2-VL-2B"
python3 tools/pmc_traffic.py decode_attn_combine_kernel $(f da2k FETCH_SIZE) $(f da2k WRITE_SIZE) 0 "tools/r03_pmc.sh (microbench decode_attn, linear planes)" "Qwen2-VL-2B"
python3 tools/pmc_traffic.py decode_attn_stream_kernel $(f da32k_7b FETCH_SIZE) $(f da32k_7b WRITE_SIZE) 67502080 "tools/r03_pmc.sh (MB_HEADS=28,4 MB_L=32768 microbench decode_attn_long, linear planes)" "decode_attn 28q/4kv x 32768 keys"
python3 tools/pmc_traffic.py decode_attn_combine_kernel $(f da32k_7b FETCH_SIZE) $(f da32k_7b WRITE_SIZE) 0 "tools/r03_pmc.sh (MB_HEADS=28,4 MB_L=32768 microbench decode_attn_long, linear planes)" "decode_attn 28q/4kv x 32768 keys"
python3 tools/pmc_traffic.py decode_attn_stream_kernel $(f da32k_2b FETCH_SIZE) $(f da32k_2b WRITE_SIZE) 33947648 "tools/r03_pmc.sh (MB_HEADS=12,2 MB_L=32768 microbench decode_attn_long, linear planes)" "decode_attn 12q/2kv x 32768 keys"
python3 tools/pmc_traffic.py decode_attn_combine_kernel $(f da32k_2b FETCH_SIZE) $(f da32k_2b WRITE_SIZE) 0 "tools/r03_pmc.sh (MB_HEADS=12,2 MB_L=32768 microbench decode_attn_long, linear planes)" "decode_attn 12q/2kv x 32768 keys"
python3 tools/pmc_traffic.py decode_attn_stream_kernel $(f da131k_7b FETCH_SIZE) $(f da131k_7b WRITE_SIZE) 270008320 "tools/r03_pmc.sh (MB_HEADS=28,4 MB_L=131072 microbench decode_attn_long, linear planes)" "decode_attn 28q/4kv x 131072 keys"
