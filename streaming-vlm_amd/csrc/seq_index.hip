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

// ================================================================ eviction plan on the device (second half)
// The span finder (utils/get_qwen_range.py:15-86) and the per-chunk eviction policy (process_past_kv, inference.py:87-172; the
// BASELINE sink/window cut of SURVEY Appendix A) as ONE single-workgroup kernel over the device copy of the ids: every
// get_qwen_range is a parallel pattern match + ordered compaction of the hits + a two-pointer pairing walk on one thread; every
// prune / move is a parallel copy between two id buffers; the decisions in between are the reference's integer arithmetic on one
// thread, broadcast through LDS.  Output: the edited ids, their length, and the op list [(1 = prune, s, e, 0) | (2 = move, s, e,
// dst)] -- the "eviction indices" the parity bar speaks of -- for the host's page accounting (driver.py applies the SAME list to
// the KV pool's slot table).
#define EV_THREADS 1024
#define EV_MAX_OPS 16
enum { EV_USER = 0, EV_PREV = 1, EV_USER_TEXT = 2, EV_ASSISTANT = 3, EV_VISION = 4 };
enum { TK_IM_START = 0, TK_IM_END, TK_USER, TK_ASSISTANT, TK_VSTART, TK_VEND, TK_VPAD, TK_LF, TK_PREV0, TK_PREV1, TK_TIME, TK_N };

struct EvTok { int t[TK_N]; };

struct EvShared {
  int wsum[EV_THREADS / 64];
  int carry;
  int n_s, n_e;            // hit counts of the current scan
  int s, e, ok;            // result of the current range query
  int L;                   // current length
  int n_ops;
  int op[4];               // the edit being applied: type, a, b, c
  int go;                  // control-flow broadcast
};

// ordered compaction of the positions i in [0, L) with flag(i) into out[]; returns the count (in *cnt, LDS)
template <typename F>
__device__ void ev_compact(EvShared& sh, int L, int* __restrict__ out, int* cnt, F&& flag) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) sh.carry = 0;
  __syncthreads();
  for (int base = 0; base < L; base += EV_THREADS) {
    const int i = base + tid;
    const int f = (i < L && flag(i)) ? 1 : 0;
    int x = f;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) sh.wsum[wave] = x;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += sh.wsum[w];
    const int carry = sh.carry;
    if (f) out[carry + woff + x - 1] = i;
    __syncthreads();
    if (tid == EV_THREADS - 1) sh.carry = carry + woff + x;
    __syncthreads();
  }
  if (tid == 0) *cnt = sh.carry;
  __syncthreads();
}

// k-th closed span of `label` in ids[0, L) (negative k counts from the end); result in sh.s / sh.e / sh.ok.  All threads call it.
__device__ void ev_range(EvShared& sh, const int* __restrict__ ids, int L, const EvTok& tk, int label, int k, bool contain_lf,
                         int* __restrict__ hs, int* __restrict__ he) {
  // start / end patterns of utils/get_qwen_range.py:37-62
  int sp[4], ls, ep;
  switch (label) {
    case EV_USER: sp[0] = tk.t[TK_IM_START]; sp[1] = tk.t[TK_USER]; ls = 2; ep = tk.t[TK_IM_END]; break;
    case EV_PREV: sp[0] = tk.t[TK_IM_START]; sp[1] = tk.t[TK_PREV0]; sp[2] = tk.t[TK_PREV1]; sp[3] = tk.t[TK_LF]; ls = 4; ep = tk.t[TK_IM_END]; break;
    case EV_USER_TEXT: sp[0] = tk.t[TK_TIME]; ls = 1; ep = tk.t[TK_VSTART]; break;
    case EV_ASSISTANT: sp[0] = tk.t[TK_IM_START]; sp[1] = tk.t[TK_ASSISTANT]; ls = 2; ep = tk.t[TK_IM_END]; break;
    default: sp[0] = tk.t[TK_VSTART]; ls = 1; ep = tk.t[TK_VEND]; break;
  }
  ev_compact(sh, L, hs, &sh.n_s, [&](int i) {
    if (i + ls > L) return false;
    bool m = ids[i] == sp[0];
    for (int j = 1; j < ls; ++j) m = m && ids[i + j] == sp[j];
    return m;
  });
  ev_compact(sh, L, he, &sh.n_e, [&](int i) { return ids[i] == ep; });
  if (threadIdx.x == 0) {
    // left-to-right pairing (get_qwen_range.py:64-80): a segment opens at the first start at or after `cur`, closes at the first end
    // pattern at or after start + len(start), the scan resumes behind the end pattern; an unterminated segment ends the scan
    const int ns = sh.n_s, ne = sh.n_e;
    int a = 0, b = 0, cur = 0, count = 0;
    // first pass: count the spans (negative k needs the total)
    int total = 0;
    {
      int a2 = 0, b2 = 0, c2 = 0;
      for (;;) {
        while (a2 < ns && hs[a2] < c2) ++a2;
        if (a2 >= ns) break;
        const int s = hs[a2];
        while (b2 < ne && he[b2] < s + ls) ++b2;
        if (b2 >= ne) break;
        ++total;
        c2 = he[b2] + 1;
      }
    }
    const int want = k < 0 ? total + k : k;
    sh.ok = 0;
    if (want >= 0 && want < total) {
      for (;;) {
        while (a < ns && hs[a] < cur) ++a;
        const int s = hs[a];
        while (b < ne && he[b] < s + ls) ++b;
        int e = he[b];
        if (count == want) {
          if (contain_lf && e + 1 < L && ids[e + 1] == tk.t[TK_LF]) ++e;
          if (label == EV_USER_TEXT) --e;                 // `Time` .. the token before <|vision_start|> (:84-85)
          sh.s = s; sh.e = e; sh.ok = 1;
          break;
        }
        ++count;
        cur = he[b] + 1;
      }
    }
  }
  __syncthreads();
}

// the edit in sh.op applied from `src` to `dst` (all threads); sh.L updated; the op is appended to the list
__device__ void ev_apply(EvShared& sh, const int* __restrict__ src, int* __restrict__ dst, int* __restrict__ ops) {
  const int L = sh.L, type = sh.op[0], a = sh.op[1], b = sh.op[2], c = sh.op[3];
  __syncthreads();
  if (type == 1) {                                          // prune the closed interval [a, b]  (inference.py:50-61)
    const int n = b - a + 1;
    for (int i = threadIdx.x; i < L; i += EV_THREADS) {
      if (i < a) dst[i] = src[i];
      else if (i > b) dst[i - n] = src[i];
    }
    if (threadIdx.x == 0) sh.L = L - n;
  } else {                                                  // move [a, b] to directly after c  (inference.py:100-108), c < a
    const int n = b - a + 1;
    for (int i = threadIdx.x; i < L; i += EV_THREADS) {
      int j;
      if (i <= c) j = i;
      else if (i < a) j = i + n;                             // the rows between the destination and the span shift right
      else if (i <= b) j = c + 1 + (i - a);
      else j = i;
      dst[j] = src[i];
    }
  }
  if (threadIdx.x == 0) {
    const int k = sh.n_ops;
    if (k < EV_MAX_OPS) { ops[4 * k] = type; ops[4 * k + 1] = a; ops[4 * k + 2] = b; ops[4 * k + 3] = c; }
    sh.n_ops = k + 1;
  }
  __syncthreads();
}

// out: [0] = status (0 ok; 1 a range the policy needs is missing; 2 too many ops), [1] = n_ops, [2] = new length, [3] = which buffer holds
// the result (0 = ids, 1 = tmp), [4 ..] = ops (EV_MAX_OPS x 4)
__global__ __launch_bounds__(EV_THREADS) void evict_plan_kernel(int* __restrict__ ids, int* __restrict__ tmp, int L0, int policy, int round,
                                                                int text_round, int visual_round, int text_sink, int text_sw, int a_start_bias,
                                                                int a_end_bias, int sink, int window, int kv_len, EvTok tk,
                                                                int* __restrict__ hs, int* __restrict__ he, int* __restrict__ out) {
  __shared__ EvShared sh;
  int* ops = out + 4;
  int* cur = ids;
  int* oth = tmp;
  if (threadIdx.x == 0) { sh.L = L0; sh.n_ops = 0; sh.ok = 0; out[0] = 0; }
  __syncthreads();
  auto fail = [&](int code) { if (threadIdx.x == 0 && out[0] == 0) out[0] = code; };
  auto edit = [&](int type, int a, int b, int c) {          // thread-uniform arguments
    if (threadIdx.x == 0) { sh.op[0] = type; sh.op[1] = a; sh.op[2] = b; sh.op[3] = c; }
    __syncthreads();
    ev_apply(sh, cur, oth, ops);
    int* t = cur; cur = oth; oth = t;
  };
  if (policy == 0) {
    // BASELINE sink / window: while L_kv > S + W: prune(S, L_kv - W - 1), the end snapped forward to <|vision_end|> when it lies
    // inside a vision span (driver.py:sink_window_evict / snap_cut_end)
    int kv = kv_len;
    for (int guard = 0; guard < EV_MAX_OPS && kv > sink + window; ++guard) {
      const int end0 = kv - window - 1;
      const int t = cur[end0];
      int end = end0;
      if (t == tk.t[TK_VSTART] || t == tk.t[TK_VPAD]) {
        ev_compact(sh, sh.L, he, &sh.n_e, [&](int i) { return i >= end0 && cur[i] == tk.t[TK_VEND]; });
        if (sh.n_e == 0) { fail(1); break; }
        end = he[0];
      }
      edit(1, sink, end, 0);
      kv -= end - sink + 1;
    }
  } else if (policy == 1) {
    // structural policy, inference.py:87-172 (index arithmetic only; the conversation-history strings stay with the host)
    if (round >= text_round) {
      ev_range(sh, cur, sh.L, tk, EV_ASSISTANT, 0, true, hs, he);
      const int a_ok = sh.ok, a_s = sh.s, a_e = sh.e;
      ev_range(sh, cur, sh.L, tk, EV_PREV, 0, false, hs, he);
      const int p_ok = sh.ok, p_e = sh.e;
      if (!a_ok || !p_ok) fail(1);
      else {
        const int src_s = a_s + a_start_bias;
        const int src_e = a_e - a_end_bias - (cur[a_e] == tk.t[TK_LF] ? 1 : 0);
        if (src_s <= src_e) edit(2, src_s, src_e, p_e - 1);
        if (visual_round > text_round) {
          ev_range(sh, cur, sh.L, tk, EV_USER_TEXT, -text_round, false, hs, he);
          if (!sh.ok) fail(1); else edit(1, sh.s, sh.e, 0);
        }
        ev_range(sh, cur, sh.L, tk, EV_ASSISTANT, -text_round, true, hs, he);
        if (!sh.ok) fail(1); else edit(1, sh.s, sh.e, 0);
      }
    }
    if (round >= visual_round && visual_round < text_round) {
      ev_range(sh, cur, sh.L, tk, EV_VISION, 0, true, hs, he);
      if (!sh.ok) fail(1); else edit(1, sh.s, sh.e, 0);
    }
    if (round >= max(visual_round, text_round)) {
      ev_range(sh, cur, sh.L, tk, EV_USER, 0, true, hs, he);
      if (!sh.ok) fail(1); else edit(1, sh.s, sh.e, 0);
    }
    if (round > 0 && (text_sink >= 0 || text_sw >= 0)) {
      ev_range(sh, cur, sh.L, tk, EV_PREV, 0, true, hs, he);
      if (!sh.ok) fail(1);
      else {
        const int cut_s = text_sink >= 0 ? sh.s + text_sink + 4 : sh.s;
        const int cut_e = text_sw >= 0 ? sh.e - text_sw - 1 : sh.e;
        if (cut_s <= cut_e) edit(1, cut_s, cut_e, 0);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (sh.n_ops > EV_MAX_OPS && out[0] == 0) out[0] = 2;
    out[1] = sh.n_ops;
    out[2] = sh.L;
    out[3] = cur == ids ? 0 : 1;
  }
}

extern "C" long long svlm_evict_plan_ws_bytes(int max_len) { return max_len <= 0 ? SVLM_EINVAL : (long long)max_len * 4 * 3 + (4 + 4 * EV_MAX_OPS) * 4; }

// policy 0: sink/window on kv_len rows; policy 1: structural (round i).  text_sink / text_sliding_window < 0 = None.  tokens[11] =
// {<|im_start|>, <|im_end|>, user, assistant, <|vision_start|>, <|vision_end|>, <|video_pad|>, "\n", previous, " text", Time} (HOST array).
// ws layout: [out: 4 + 64 ints][tmp: max_len][hits: 2 x max_len]; the edited ids end up in `ids` or in ws tmp (out[3]).
extern "C" int svlm_evict_plan(int* ids, int L, int policy, int round, int text_round, int visual_round, int text_sink, int text_sliding_window,
                               int assistant_start_bias, int assistant_end_bias, int sink, int window, int kv_len, const int* tokens,
                               void* ws, long long ws_bytes, void* stream) {
  SVLM_CHECK_ARG(L > 0 && (policy == 0 || policy == 1) && tokens != nullptr && ws != nullptr && ws_bytes >= svlm_evict_plan_ws_bytes(L),
                 "svlm_evict_plan: bad L=%d policy=%d or workspace too small", L, policy);
  SVLM_CHECK_ARG(policy == 1 || (kv_len > 0 && kv_len <= L && sink >= 0 && window > 0), "svlm_evict_plan: bad kv_len=%d sink=%d window=%d", kv_len, sink, window);
  EvTok tk;
  for (int i = 0; i < TK_N; ++i) tk.t[i] = tokens[i];
  int* out = (int*)ws;
  int* tmp = out + 4 + 4 * EV_MAX_OPS;
  int* hs = tmp + L;
  int* he = hs + L;
  evict_plan_kernel<<<1, EV_THREADS, 0, (hipStream_t)stream>>>(ids, tmp, L, policy, round, text_round, visual_round, text_sink, text_sliding_window,
                                                              assistant_start_bias, assistant_end_bias, sink, window, kv_len, tk, hs, he, out);
  return svlm_check_launch("svlm_evict_plan");
}
