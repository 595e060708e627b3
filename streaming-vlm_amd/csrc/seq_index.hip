// Device-side sequence bookkeeping: M-RoPE position ids of a whole (pruned) id sequence, computed ON THE
// DEVICE from the ids and the per-span grid table -- no host walk over the ids, no (3, L) host-to-device copy per chunk.
//
// Reference: get_rope_index (inference/qwen2/pos_emb.py:69-133; float variant inference/qwen2_5/pos_emb.py:107-160), which
// walks the id list in Python (.tolist() / .index() / .item() per vision span) on EVERY forward.  Same result, three phases in
// one workgroup:
//   A  flags + block scan: rank of every video token, index of the r-th video token, count of vision spans
//      (<|vision_start|> followed by a video token, pos_emb.py:74-77);
//   B  the span chain, one thread: for span v, ed = first video token at or after st (.index(video_token_id, st), :88),
//      text_len = ed - st, base = text_len + st_idx, st_idx' = max of the span + 1, st' = ed + t*h*w (:110-124) -- a few scalar
//      operations per span, the only sequential part;
//   C  every token looks its run up by binary search over the span starts and writes its three ids: text runs arange + offset,
//      vision tokens (t, h, w) + base; rows [L, L + n_extra) continue the last text run (the tokens about to be generated).
// Qwen2.5: positions are fp32 and the temporal id is ((t * second_per_grid_t) * tokens_per_second) + text_len + st_idx, operation
// for operation as the reference computes it (:121-133).
#include "common.h"

#define SI_THREADS 1024

struct SiSpan {           // one vision span and the text run in front of it
  int st, ed, nv, gh, gw;
  float tl_f, idx_f;      // fp32 chain (Qwen2.5): text_len and st_idx in front of the run (added in the reference's order)
  int base_i, idx_i;      // integer chain (Qwen2)
};

__global__ __launch_bounds__(SI_THREADS) void rope_index_kernel(const int* __restrict__ ids, int L, const int* __restrict__ grids, int n_grids,
                                                                int merge, int video_id, int vstart_id, int* __restrict__ pos_i,
                                                                float* __restrict__ pos_f, int stride, float spg, float tps, int n_extra,
                                                                int* __restrict__ vid_pos, SiSpan* __restrict__ spans, int* __restrict__ status) {
  __shared__ int wsum[SI_THREADS / 64];
  __shared__ int s_carry, s_nvid, s_nspan, s_tail_st, s_tail_i;
  __shared__ float s_tail_f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { s_carry = 0; s_nvid = 0; }
  __syncthreads();
  // ---- A: video-token ranks (blocked scan, SI_THREADS ids per round), index of the r-th video token, span count
  int nspan_local = 0;
  for (int base = 0; base < L; base += SI_THREADS) {
    const int i = base + tid;
    const int t = i < L ? ids[i] : -1;
    const int f = (t == video_id) ? 1 : 0;
    if (i + 1 < L && t == vstart_id && ids[i + 1] == video_id) ++nspan_local;
    int x = f;                                        // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const int carry = s_carry;
    const int rank = carry + woff + x - f;            // video tokens in front of i
    if (f) vid_pos[rank] = i;
    __syncthreads();
    if (tid == SI_THREADS - 1) s_carry = carry + woff + x;
    __syncthreads();
  }
  atomicAdd(&s_nvid, nspan_local);
  __syncthreads();
  const int n_vtok = s_carry;                         // video tokens in the sequence
  // ---- B: the span chain
  if (tid == 0) {
    int st = 0, vr = 0, idx_i = 0, err = 0, n = 0;    // vr = video tokens in front of st
    float idx_f = 0.f;
    const int n_vid = s_nvid;
    for (int v = 0; v < n_vid; ++v) {
      if (v >= n_grids) { err = 2; break; }           // more spans than grid rows
      if (vr >= n_vtok) { err = 3; break; }           // "video segment without <|video_pad|> tokens"
      const int ed = vid_pos[vr];
      const int t = grids[3 * v], gh = grids[3 * v + 1] / merge, gw = grids[3 * v + 2] / merge;
      const int nv = t * gh * gw;
      if (ed + nv > L) { err = 4; break; }            // span runs past the sequence
      const int text_len = ed - st;
      SiSpan s;
      s.st = st; s.ed = ed; s.nv = nv; s.gh = gh; s.gw = gw;
      s.idx_i = idx_i; s.base_i = idx_i + text_len;
      s.idx_f = idx_f; s.tl_f = (float)text_len;
      spans[n++] = s;
      // next start index: max over the span + 1 (every id is (x + text_len) + st_idx, in that order: qwen2_5/pos_emb.py:131-133)
      const int mx = max(t, max(gh, gw));
      idx_i = s.base_i + mx;
      const float tmax = (((float)(t - 1) * spg) * tps + s.tl_f) + idx_f;
      const float smax = ((float)(max(gh, gw) - 1) + s.tl_f) + idx_f;
      idx_f = fmaxf(tmax, smax) + 1.0f;
      st = ed + nv;
      vr += nv;                                       // the span's own tokens are video tokens: the next search starts behind them
    }
    s_nspan = n; s_tail_st = st; s_tail_i = idx_i; s_tail_f = idx_f;
    *status = err;
  }
  __syncthreads();
  const int n_span = s_nspan, tail_st = s_tail_st;
  const int tail_i = s_tail_i;
  const float tail_f = s_tail_f;
  // ---- C: every row writes its three ids
  for (int i = tid; i < L + n_extra; i += SI_THREADS) {
    int pi0, pi1, pi2;
    float pf0, pf1, pf2;
    if (i >= tail_st) {                               // trailing text run (and the rows of the tokens to come)
      pi0 = pi1 = pi2 = tail_i + (i - tail_st);
      pf0 = pf1 = pf2 = (float)(i - tail_st) + tail_f;
    } else {
      int lo = 0, hi = n_span - 1;                    // last span whose run starts at or before i
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (spans[mid].st <= i) lo = mid; else hi = mid - 1;
      }
      const SiSpan s = spans[lo];
      if (i < s.ed) {
        pi0 = pi1 = pi2 = s.idx_i + (i - s.st);
        pf0 = pf1 = pf2 = (float)(i - s.st) + s.idx_f;
      } else {
        const int k = i - s.ed;
        const int ti = k / (s.gh * s.gw), hi2 = (k / s.gw) % s.gh, wi = k % s.gw;
        pi0 = s.base_i + ti; pi1 = s.base_i + hi2; pi2 = s.base_i + wi;
        pf0 = (((float)ti * spg) * tps + s.tl_f) + s.idx_f;
        pf1 = ((float)hi2 + s.tl_f) + s.idx_f;
        pf2 = ((float)wi + s.tl_f) + s.idx_f;
      }
    }
    if (pos_f) {
      pos_f[i] = pf0; pos_f[(size_t)stride + i] = pf1; pos_f[2 * (size_t)stride + i] = pf2;
    } else {
      pos_i[i] = pi0; pos_i[(size_t)stride + i] = pi1; pos_i[2 * (size_t)stride + i] = pi2;
    }
  }
}

extern "C" long long svlm_rope_index_ws_bytes(int max_len, int max_spans) {
  if (max_len <= 0 || max_spans < 0) return SVLM_EINVAL;
  return (long long)max_len * 4 + (long long)(max_spans + 1) * (long long)sizeof(SiSpan) + 16;
}

// ids (L) int32, grids (n_grids, 3) int32 [t, h, w in patches], both on the device; pos3 (3, stride) int32 or posf3 fp32 (exactly one);
// rows [L, L + n_extra) continue the trailing text run; *status (in ws) = 0 or the reference's error case.  One workgroup.
extern "C" int svlm_rope_index(const int* ids, int L, const int* grids, int n_grids, int merge, int video_token_id, int vision_start_token_id,
                               int* pos3, float* posf3, int stride, float second_per_grid_t, float tokens_per_second, int n_extra, void* ws,
                               long long ws_bytes, void* stream) {
  SVLM_CHECK_ARG((pos3 != nullptr) != (posf3 != nullptr), "svlm_rope_index: pass exactly one of pos3 / posf3");
  SVLM_CHECK_ARG(L >= 0 && n_grids >= 0 && merge > 0 && n_extra >= 0 && L + n_extra <= stride, "svlm_rope_index: bad L=%d n_extra=%d stride=%d", L, n_extra, stride);
  SVLM_CHECK_ARG(ws != nullptr && ws_bytes >= svlm_rope_index_ws_bytes(L > 0 ? L : 1, n_grids), "svlm_rope_index: workspace too small");
  int* status = (int*)ws;
  int* vid_pos = status + 4;
  SiSpan* spans = (SiSpan*)(vid_pos + (L > 0 ? L : 1));
  rope_index_kernel<<<1, SI_THREADS, 0, (hipStream_t)stream>>>(ids, L, grids, n_grids, merge, video_token_id, vision_start_token_id, pos3, posf3,
                                                              stride, second_per_grid_t, tokens_per_second, n_extra, vid_pos, spans, status);
  return svlm_check_launch("svlm_rope_index");
}
