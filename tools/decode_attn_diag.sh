#!/bin/bash
# Where the long-cache decode attention's time goes: the streaming split kernel with and without its cos/sin loads + rotation
# (SVLM_DA_DIAG=1), with no arithmetic at all (2, 3), and without the combine launch (SVLM_DA_COMBINE_DS=-1).  Needs the
# diagnostic library (python tools/build_diag_lib.py).  Usage: tools/decode_attn_diag.sh > out.txt
export SVLM_LIB_PATH=streaming-vlm_amd/build/libsvlm_hip_diag.so
for c in "12,2,32768 128" "28,4,32768 192" "12,2,131072 192" "28,4,131072 512"; do
  for diag in 0 1 2 3; do
    for comb in 0 -1; do
      echo "== case $c diag=$diag combine_ds=$comb"
      SVLM_DA_DIAG=$diag SVLM_DA_COMBINE_DS=$comb python tools/decode_attn_sweep.py $c || exit 1
    done
  done
done
