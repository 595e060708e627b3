export SVLM_LIB_PATH=streaming-vlm_amd/build/libsvlm_hip_diag.so
for c in "12,2,32768 128" "28,4,32768 192" "12,2,131072 192" "28,4,131072 512"; do
  for diag in 8; do
      echo "== case $c diag=$diag split only"
      SVLM_DA_DIAG=$diag SVLM_DA_COMBINE_DS=-1 python tools/decode_attn_sweep.py $c || exit 1
  done
done
