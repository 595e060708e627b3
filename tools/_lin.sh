set -e
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "decode_attn or prefill_attn" 2>&1 | tail -5
for c in "12,2,2150 32,48,64" "28,4,4600 48,64" "12,2,8192 128,192" "12,2,32768 128,192,256" "28,4,32768 128,192,256,384" "12,2,131072 192,256,512" "28,4,131072 256,512"; do
  python tools/decode_attn_sweep.py $c
  python tools/decode_attn_sweep.py $c lin
done
export SVLM_LIB_PATH=streaming-vlm_amd/build/libsvlm_hip_diag.so
for c in "12,2,2150 48" "12,2,32768 128" "28,4,32768 192" "28,4,131072 512"; do
  echo "split only:"; SVLM_DA_COMBINE_DS=-1 python tools/decode_attn_sweep.py $c; SVLM_DA_COMBINE_DS=-1 python tools/decode_attn_sweep.py $c lin
done
