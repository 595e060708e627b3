// Greedy sampling with HF repetition penalty, fused with the device-side token feedback that
// lets a whole decode step replay from a HIP graph without a host round trip.
//
// Reference: generate/streaming_generate_qwen.py:73-109 -- fp32 copy of the last logits row (:73),
// logits processors (:75; RepetitionPenaltyLogitsProcessor over the FULL current ids:
// score<0 ? score*p : score/p), argmax (:99), cat to input_ids (:104).
#include "common.h"

// seen[v] = 1 for every id in ids[0..n)
__global__ void mark_seen_kernel(const int* __restrict__ ids, int n, unsigned char* __restrict__ seen, int V) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int t = ids[i];
    if (t >= 0 && t < V) seen[t] = 1;
  }
}

// Stage 1: AM_BLOCKS workgroups scan disjoint slices of the (penalised, suppressed) logits and leave
// one (value, index) candidate each in `part`.  Stage 2 (one wave) picks the winner (lowest index on
// ties, like torch.argmax) and does the device-side token feedback:
// state[0] = kv_len, state[1] = cur (index in tok_buf of the token fed to the current forward; -1 during
// prefill).  Writes tok_buf[cur+1] = argmax, marks it seen, cur += 1, kv_len += advance_kv.
#define AM_BLOCKS 64

__device__ __forceinline__ void am_better(float& best, int& bi, float x, int v) {
  if (x > best || (x == best && v < bi)) { best = x; bi = v; }
}

__global__ __launch_bounds__(256) void argmax_stage1_kernel(const float* __restrict__ logits, int V,
                                                            const unsigned char* __restrict__ seen, float penalty,
                                                            const int* __restrict__ suppress, int n_suppress,
                                                            float* __restrict__ part_val, int* __restrict__ part_idx) {
  const int per = (V + AM_BLOCKS - 1) / AM_BLOCKS;
  const int lo = blockIdx.x * per, hi = min(V, lo + per);
  int sup[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) sup[s] = s < n_suppress ? suppress[s] : -1;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int v = lo + threadIdx.x; v < hi; v += 256) {
    float x = logits[v];
    if (seen && seen[v]) x = x < 0.f ? x * penalty : x / penalty;
#pragma unroll
    for (int s = 0; s < 8; ++s)
      if (sup[s] == v) x = -INFINITY;
    am_better(best, bi, x, v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    am_better(best, bi, ob, oi);
  }
  __shared__ float sb[4];
  __shared__ int si[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { sb[wave] = best; si[wave] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) am_better(best, bi, sb[w], si[w]);
    part_val[blockIdx.x] = best;
    part_idx[blockIdx.x] = bi;
  }
}

__global__ __launch_bounds__(64) void argmax_stage2_kernel(const float* __restrict__ part_val, const int* __restrict__ part_idx,
                                                           unsigned char* seen, int* tok_buf, int* state, int advance_kv) {
  float best = part_val[threadIdx.x];
  int bi = part_idx[threadIdx.x];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    am_better(best, bi, ob, oi);
  }
  if (threadIdx.x == 0) {
    if (bi == 0x7fffffff) bi = 0;  // all -inf / NaN row: mirror torch.argmax's index 0
    const int cur = state[1] + 1;
    tok_buf[cur] = bi;
    state[1] = cur;
    state[0] += advance_kv;
    if (seen) seen[bi] = 1;
  }
}

extern "C" int svlm_mark_seen(const int* ids, int n, void* seen, int V, void* stream) {
  SVLM_CHECK_ARG(n >= 0 && V > 0, "svlm_mark_seen: bad n=%d V=%d", n, V);
  if (n == 0) return SVLM_OK;
  mark_seen_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(ids, n, (unsigned char*)seen, V);
  return svlm_check_launch("svlm_mark_seen");
}

extern "C" long long svlm_argmax_ws_bytes(void) { return AM_BLOCKS * 8; }

extern "C" int svlm_penalty_argmax(const float* logits, int V, void* seen, float penalty, const int* suppress, int n_suppress,
                                   int* tok_buf, int* state, int advance_kv, void* ws, void* stream) {
  SVLM_CHECK_ARG(V > 0 && penalty > 0.f && n_suppress >= 0 && n_suppress <= 8, "svlm_penalty_argmax: bad V=%d penalty=%f n_suppress=%d", V, penalty, n_suppress);
  SVLM_CHECK_ARG(ws != nullptr, "svlm_penalty_argmax: workspace of svlm_argmax_ws_bytes() bytes required");
  float* pv = (float*)ws;
  int* pi = (int*)(pv + AM_BLOCKS);
  argmax_stage1_kernel<<<AM_BLOCKS, 256, 0, (hipStream_t)stream>>>(logits, V, (const unsigned char*)seen, penalty, suppress, n_suppress, pv, pi);
  int rc = svlm_check_launch("svlm_penalty_argmax(stage1)");
  if (rc) return rc;
  argmax_stage2_kernel<<<1, 64, 0, (hipStream_t)stream>>>(pv, pi, (unsigned char*)seen, tok_buf, state, advance_kv);
  return svlm_check_launch("svlm_penalty_argmax(stage2)");
}
