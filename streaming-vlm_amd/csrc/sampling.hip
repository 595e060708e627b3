// Token choice with HF's logits processors, fused with the device-side token feedback that
// lets a whole decode step replay from a HIP graph without a host round trip.
//
// Reference: generate/streaming_generate_qwen.py:73-109 -- fp32 copy of the last logits row (:73),
// logits processors (:75; RepetitionPenaltyLogitsProcessor over the FULL current ids:
// score<0 ? score*p : score/p; then the warpers HF builds from the generation config: temperature,
// top-k, top-p), then argmax (:99) or softmax + multinomial (:95-97, the reference's DEFAULT:
// do_sample=True at inference.py:446), cat to input_ids (:104).
//
// Sampling on the device: plain temperature sampling is the Gumbel-max trick -- argmax_i(s_i / T + g_i) with iid standard
// Gumbel noise g_i is an exact draw from softmax(s / T) -- so it rides in the argmax kernels the greedy path already has;
// top-k / top-p go through one single-workgroup kernel (radix select of the k-th score, bitonic sort of the survivors,
// nucleus cut, inverse-CDF draw).  Noise comes from Philox4x32-10 keyed by a per-call seed in device memory and counted by
// (generated-token index, vocabulary index), so a captured graph replays fresh noise at every step.
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
                                                            float* __restrict__ part_val, int* __restrict__ part_idx,
                                                            float inv_temp, const unsigned* __restrict__ rng,
                                                            const int* __restrict__ state) {
  const unsigned step = rng ? (unsigned)(state[1] + 1) : 0u;          // index of the token being chosen
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
    if (rng) x = x * inv_temp + svlm_gumbel_noise(rng, step, (unsigned)v);
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
  argmax_stage1_kernel<<<AM_BLOCKS, 256, 0, (hipStream_t)stream>>>(logits, V, (const unsigned char*)seen, penalty, suppress, n_suppress, pv, pi,
                                                                   1.0f, nullptr, nullptr);
  int rc = svlm_check_launch("svlm_penalty_argmax(stage1)");
  if (rc) return rc;
  argmax_stage2_kernel<<<1, 64, 0, (hipStream_t)stream>>>(pv, pi, (unsigned char*)seen, tok_buf, state, advance_kv);
  return svlm_check_launch("svlm_penalty_argmax(stage2)");
}

// ---------------------------------------------------------------- top-k / top-p sampling (one workgroup)
// HF order (generation/logits_process.py): repetition penalty -> temperature -> top-k (keep every score >= the k-th largest)
// -> top-p (sort ascending, drop the tokens whose cumulative probability is <= 1 - top_p, always keep the largest) -> softmax ->
// multinomial.  This kernel serves 1 < top_k < SF_CAP (ties at the threshold beyond the list's SF_CAP entries are cut by index);
// top_k = 0 or >= SF_CAP goes to sample_nucleus_kernel below.
#define SF_CAP 2048
#define SF_THREADS 1024

__device__ __forceinline__ unsigned sf_key(float x) {        // order-preserving float -> uint (larger score = larger key)
  const unsigned u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Walk the processed scores of the whole vocabulary once: 16 elements per thread and step (four 16-B loads of logits and four
// 4-B loads of `seen` issued before anything is consumed -- a one-workgroup scan is latency-bound otherwise: 150 dependent
// round trips per thread cost 50+ us per pass), fn(v, score) per element.
// SUP = false leaves the (at most 8) suppressed ids in: a caller that only COUNTS may prefer to correct for them afterwards
// instead of paying 8 compares on each of 150k elements (the scan runs on one CU and is VALU-bound).
template <bool SUP = true, typename F>
__device__ __forceinline__ void sf_scan(const float* __restrict__ logits, const unsigned char* __restrict__ seen, int V, float penalty,
                                        const int (&sup)[8], float inv_temp, F&& fn) {
  const int tid = threadIdx.x;
  auto one = [&](int v, float x, unsigned char sn) {
    if (sn) x = x < 0.f ? x * penalty : x / penalty;
    if constexpr (SUP) {
#pragma unroll
      for (int s = 0; s < 8; ++s)
        if (sup[s] == v) x = -INFINITY;
    }
    fn(v, x * inv_temp);
  };
  const int V4 = (V % 4 == 0) ? V / 4 : 0;             // vector path needs 16-B aligned rows of 4
  for (int base = 0; base < V4; base += 4 * SF_THREADS) {
    f32x4_t lx[4];
    unsigned ls[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = min(base + u * SF_THREADS + tid, V4 - 1);
      lx[u] = *reinterpret_cast<const f32x4_t*>(logits + 4 * q);
      ls[u] = seen ? *reinterpret_cast<const unsigned*>(seen + 4 * q) : 0u;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = base + u * SF_THREADS + tid;
      if (q < V4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) one(4 * q + e, lx[u][e], (unsigned char)(ls[u] >> (8 * e)));
      }
    }
  }
  for (int v = 4 * V4 + tid; v < V; v += SF_THREADS) one(v, logits[v], seen ? seen[v] : (unsigned char)0);
}

#define SF_BINS 2048        // top 11 bits of the order-preserving key: sign, exponent, 2 mantissa bits (quarter-octave bins)

__global__ __launch_bounds__(SF_THREADS) void sample_filtered_kernel(const float* __restrict__ logits, int V,
                                                                     const unsigned char* __restrict__ seen, float penalty,
                                                                     const int* __restrict__ suppress, int n_suppress, float inv_temp,
                                                                     int top_k, float top_p, const unsigned* __restrict__ rng,
                                                                     unsigned char* seen_w, int* tok_buf, int* state, int advance_kv) {
  __shared__ unsigned hist[SF_BINS];
  __shared__ unsigned long long cand[SF_CAP];          // (key << 32) | ~index : sorts by score, then lowest index first
  __shared__ float fsum[SF_CAP];
  __shared__ unsigned sh_bin, sh_n, sh_above, sh_done;
  const int tid = threadIdx.x;
  int sup[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) sup[s] = s < n_suppress ? suppress[s] : -1;
  const int k_eff = min(top_k, V);
  // ---- radix descent, 11 bits per level (normally ONE level): the narrowest key range [lo_key, inf) that still holds the k largest
  // scores and fits the candidate list
  if (tid == 0) { sh_n = 0u; sh_bin = 0u; sh_above = 0u; sh_done = 0u; }
  __syncthreads();
  unsigned prefix = 0u;                                // exact high bits of the threshold key found so far
  for (int level = 0; level < 3; ++level) {
    const int shift = level == 0 ? 21 : (level == 1 ? 10 : 0);
    const unsigned dmask = level == 2 ? 1023u : 2047u;
    const unsigned hmask = level == 0 ? 0u : (0xFFFFFFFFu << (level == 1 ? 21 : 10));
    for (int i = tid; i < SF_BINS; i += SF_THREADS) hist[i] = 0u;
    __syncthreads();
    // (suppressed ids are counted here as if they were alive and asked for on top of k below: the range found is at most
    // n_suppress ranks too wide, and the collect pass drops them)
    sf_scan<false>(logits, seen, V, penalty, sup, inv_temp, [&](int, float sc) {
      const unsigned key = sf_key(sc);
      if ((key & hmask) == prefix) atomicAdd(&hist[(key >> shift) & dmask], 1u);
    });
    __syncthreads();
    for (int off = 1; off < SF_BINS; off <<= 1) {     // suffix counts: hist[d] <- keys of this level with digit >= d
      unsigned a0 = 0u, a1 = 0u;
      const int i0 = tid, i1 = tid + SF_THREADS;
      if (i0 + off < SF_BINS) a0 = hist[i0 + off];
      if (i1 + off < SF_BINS) a1 = hist[i1 + off];
      __syncthreads();
      hist[i0] += a0; hist[i1] += a1;
      __syncthreads();
    }
    const unsigned above = sh_above;                   // keys strictly above the range this level splits
    __syncthreads();                                   // every thread has read sh_above before the finder below rewrites it
    const unsigned k_cnt = (unsigned)min(k_eff + n_suppress, V);
    const unsigned want = k_cnt > above ? k_cnt - above : 1u;
    for (int i = tid; i < SF_BINS; i += SF_THREADS) {
      const unsigned nxt = i + 1 < SF_BINS ? hist[i + 1] : 0u;
      if (hist[i] >= want && nxt < want) {             // the digit whose bin holds the k-th largest (suffix counts fall with the digit)
        sh_bin = (unsigned)i;
        sh_above = above + nxt;
        sh_done = (above + hist[i] <= (unsigned)SF_CAP || level == 2) ? 1u : 0u;
      }
    }
    __syncthreads();
    prefix |= sh_bin << shift;
    if (sh_done) break;
  }
  const unsigned lo_key = prefix;
  // ---- collect: every key >= lo_key goes to the list (the k largest are among them); -inf never does
  sf_scan<false>(logits, seen, V, penalty, sup, inv_temp, [&](int v, float sc) {
    const unsigned key = sf_key(sc);
    if (key >= lo_key && sc > -INFINITY) {
      bool dead = false;
#pragma unroll
      for (int s = 0; s < 8; ++s) dead |= sup[s] == v;
      if (dead) return;
      const unsigned slot = atomicAdd(&sh_n, 1u);
      if (slot < SF_CAP) cand[slot] = ((unsigned long long)key << 32) | (unsigned)(~(unsigned)v);
    }
  });
  __syncthreads();
  const int n_list = (int)min(sh_n, (unsigned)SF_CAP);
  int P = 64;                                          // sort size: next power of two
  while (P < n_list) P <<= 1;
  for (int i = n_list + tid; i < P; i += SF_THREADS) cand[i] = 0ull;
  __syncthreads();
  // ---- bitonic sort, descending
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < P; i += SF_THREADS) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = cand[i], b = cand[ixj];
          const bool desc = (i & k) == 0;
          if (desc ? (a < b) : (a > b)) { cand[i] = b; cand[ixj] = a; }
        }
      }
      __syncthreads();
    }
  }
  // ---- top-k: the k largest plus every tie of the k-th (HF keeps all scores >= the k-th largest)
  int n = min(k_eff, n_list);
  if (n > 0) {
    const unsigned kth = (unsigned)(cand[n - 1] >> 32);
    while (n < n_list && (unsigned)(cand[n] >> 32) == kth) ++n;
  }
  // ---- softmax weights over the survivors, e_i = exp(s_i - s_0)
  auto key_score = [](unsigned key) -> float {
    const unsigned u = (key & 0x80000000u) ? (key & 0x7FFFFFFFu) : ~key;
    return __uint_as_float(u);
  };
  const float s0 = n > 0 ? key_score((unsigned)(cand[0] >> 32)) : 0.f;
  for (int i = tid; i < P; i += SF_THREADS) fsum[i] = i < n ? __expf(key_score((unsigned)(cand[i] >> 32)) - s0) : 0.f;
  __syncthreads();
  for (int off = 1; off < P; off <<= 1) {              // inclusive prefix sums
    float a0 = 0.f, a1 = 0.f;
    const int i0 = tid, i1 = tid + SF_THREADS;
    if (i0 < P && i0 >= off) a0 = fsum[i0 - off];
    if (i1 < P && i1 >= off) a1 = fsum[i1 - off];
    __syncthreads();
    if (i0 < P) fsum[i0] += a0;
    if (i1 < P) fsum[i1] += a1;
    __syncthreads();
  }
  if (tid == 0) {
    int tok = 0;
    if (n > 0) {
      const float total = fsum[n - 1];
      // top-p: entry i (descending) is dropped iff (mass of i and everything smaller) / total <= 1 - top_p
      int keep = n;
      if (top_p < 1.0f) {
        int lo = 1, hi = n;                            // keep in [1, n]: smallest count whose NEXT tail mass is <= 1 - top_p
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;              // would entry `mid` be dropped?  tail(mid) = total - prefix(mid - 1)
          const float tail = total - fsum[mid - 1];
          if (tail <= (1.0f - top_p) * total) hi = mid; else lo = mid + 1;
        }
        keep = lo;
      }
      const float w = fsum[keep - 1];
      const float u = svlm_philox_uniform(rng, (unsigned)(state[1] + 1), 0xFFFFFFFFu) * w;
      int lo = 0, hi = keep - 1;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (fsum[mid] > u) hi = mid; else lo = mid + 1;
      }
      tok = (int)(~(unsigned)(cand[lo] & 0xFFFFFFFFull));
    }
    const int cur = state[1] + 1;
    tok_buf[cur] = tok;
    state[1] = cur;
    state[0] += advance_kv;
    if (seen_w) seen_w[tok] = 1;
  }
}

// Nucleus sampling over the WHOLE vocabulary (no top-k in front, or one wider than SF_CAP): no sort and no cap.  Both filters are a
// threshold on the score -- top-k by count, top-p by probability mass -- found by the same 8-bit radix descent over the
// order-preserving keys (the mass descent sums exp(s - max) in 2^-40 fixed point, so the sums do not depend on the order the
// atomics land in), and the draw is a Gumbel-max over the survivors: exact for a nucleus of any size.
__global__ __launch_bounds__(SF_THREADS) void sample_nucleus_kernel(const float* __restrict__ logits, int V,
                                                                    const unsigned char* __restrict__ seen, float penalty,
                                                                    const int* __restrict__ suppress, int n_suppress, float inv_temp,
                                                                    int top_k, float top_p, const unsigned* __restrict__ rng,
                                                                    unsigned char* seen_w, int* tok_buf, int* state, int advance_kv) {
  __shared__ unsigned long long mass[256];
  __shared__ unsigned cnt[256];
  __shared__ float red_f[SF_THREADS / 64];
  __shared__ int red_i[SF_THREADS / 64];
  __shared__ unsigned sh_prefix, sh_want;
  __shared__ unsigned long long sh_above, sh_target;
  __shared__ float sh_max;
  const int tid = threadIdx.x;
  int sup[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) sup[s] = s < n_suppress ? suppress[s] : -1;
  // ---- max score (for exp) ...
  float mx = -INFINITY;
  sf_scan(logits, seen, V, penalty, sup, inv_temp, [&](int, float sc) { mx = fmaxf(mx, sc); });
  mx = wave_max(mx);
  if ((tid & 63) == 0) red_f[tid >> 6] = mx;
  __syncthreads();
  if (tid == 0) {
    float m = red_f[0];
    for (int w = 1; w < SF_THREADS / 64; ++w) m = fmaxf(m, red_f[w]);
    sh_max = m;
  }
  __syncthreads();
  const float smax = sh_max;
  // ---- ... top-k threshold by count (only when a k wider than the sort kernel's list was asked for)
  unsigned tau = 0u;
  if (top_k > 0 && top_k < V) {
    if (tid == 0) { sh_prefix = 0u; sh_want = (unsigned)top_k; }
    __syncthreads();
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      if (tid < 256) cnt[tid] = 0u;
      __syncthreads();
      const unsigned prefix = sh_prefix;
      const unsigned mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
      sf_scan(logits, seen, V, penalty, sup, inv_temp, [&](int, float sc) {
        const unsigned key = sf_key(sc);
        if ((key & mask) == prefix) atomicAdd(&cnt[(key >> shift) & 255u], 1u);
      });
      __syncthreads();
      if (tid == 0) {
        unsigned want = sh_want, b = 255u;
        for (;; --b) {
          if (cnt[b] >= want || b == 0u) break;
          want -= cnt[b];
        }
        sh_prefix = prefix | (b << shift);
        sh_want = want;
      }
      __syncthreads();
    }
    tau = sh_prefix;
  }
  // ---- top-p threshold by mass among the keys >= tau: an entry survives iff the mass of the strictly larger ones is < top_p * Z
  if (top_p < 1.0f) {
    if (tid == 0) { sh_prefix = 0u; sh_above = 0ull; }
    __syncthreads();
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      if (tid < 256) mass[tid] = 0ull;
      __syncthreads();
      const unsigned prefix = sh_prefix;
      const unsigned mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
      sf_scan(logits, seen, V, penalty, sup, inv_temp, [&](int, float sc) {
        const unsigned key = sf_key(sc);
        if (key >= tau && (key & mask) == prefix) {
          const unsigned long long m = (unsigned long long)(__expf(sc - smax) * 1099511627776.0f);        // 2^40 fixed point
          if (m) atomicAdd(&mass[(key >> shift) & 255u], m);
        }
      });
      __syncthreads();
      if (tid == 0) {
        if (pass == 0) {
          unsigned long long z = 0ull;
          for (int b = 0; b < 256; ++b) z += mass[b];
          sh_target = (unsigned long long)((double)top_p * (double)z);
        }
        unsigned long long acc = sh_above;
        unsigned b = 255u;
        for (;; --b) {                                 // first bin (from the top) in which the running mass reaches the target
          if (acc + mass[b] >= sh_target || b == 0u) break;
          acc += mass[b];
        }
        sh_above = acc;
        sh_prefix = prefix | (b << shift);
      }
      __syncthreads();
    }
    tau = max(tau, sh_prefix);
  }
  // ---- Gumbel-max over the survivors
  const unsigned step = (unsigned)(state[1] + 1);
  float best = -INFINITY;
  int bi = 0x7fffffff;
  sf_scan(logits, seen, V, penalty, sup, inv_temp, [&](int v, float sc) {
    if (sf_key(sc) >= tau && sc > -INFINITY) am_better(best, bi, sc + svlm_gumbel_noise(rng, step, (unsigned)v), v);
  });
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    am_better(best, bi, ob, oi);
  }
  if ((tid & 63) == 0) { red_f[tid >> 6] = best; red_i[tid >> 6] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < SF_THREADS / 64; ++w) am_better(best, bi, red_f[w], red_i[w]);
    if (bi == 0x7fffffff) bi = 0;
    const int cur = state[1] + 1;
    tok_buf[cur] = bi;
    state[1] = cur;
    state[0] += advance_kv;
    if (seen_w) seen_w[bi] = 1;
  }
}

extern "C" int svlm_penalty_sample(const float* logits, int V, void* seen, float penalty, const int* suppress, int n_suppress,
                                   float temperature, int top_k, float top_p, const unsigned* rng, int* tok_buf, int* state,
                                   int advance_kv, void* ws, void* stream) {
  SVLM_CHECK_ARG(V > 0 && penalty > 0.f && n_suppress >= 0 && n_suppress <= 8, "svlm_penalty_sample: bad V=%d penalty=%f n_suppress=%d", V, penalty, n_suppress);
  SVLM_CHECK_ARG(temperature > 0.f && top_k >= 0 && top_p > 0.f && top_p <= 1.f && rng != nullptr && ws != nullptr,
                 "svlm_penalty_sample: bad temperature=%f top_k=%d top_p=%f", temperature, top_k, top_p);
  hipStream_t st = (hipStream_t)stream;
  if (top_k == 1) {                                    // one survivor: the argmax
    return svlm_penalty_argmax(logits, V, seen, penalty, suppress, n_suppress, tok_buf, state, advance_kv, ws, stream);
  }
  if (top_k == 0 && top_p >= 1.f) {                    // plain temperature sampling: Gumbel-max in the argmax kernels
    float* pv = (float*)ws;
    int* pi = (int*)(pv + AM_BLOCKS);
    argmax_stage1_kernel<<<AM_BLOCKS, 256, 0, st>>>(logits, V, (const unsigned char*)seen, penalty, suppress, n_suppress, pv, pi,
                                                     1.0f / temperature, rng, state);
    int rc = svlm_check_launch("svlm_penalty_sample(stage1)");
    if (rc) return rc;
    argmax_stage2_kernel<<<1, 64, 0, st>>>(pv, pi, (unsigned char*)seen, tok_buf, state, advance_kv);
    return svlm_check_launch("svlm_penalty_sample(stage2)");
  }
  if (top_k == 0 || top_k >= SF_CAP) {                 // the survivors of top-p alone (or of a very wide k) are not bounded: threshold form
    sample_nucleus_kernel<<<1, SF_THREADS, 0, st>>>(logits, V, (const unsigned char*)seen, penalty, suppress, n_suppress, 1.0f / temperature,
                                                    top_k, top_p, rng, (unsigned char*)seen, tok_buf, state, advance_kv);
    return svlm_check_launch("svlm_penalty_sample(nucleus)");
  }
  sample_filtered_kernel<<<1, SF_THREADS, 0, st>>>(logits, V, (const unsigned char*)seen, penalty, suppress, n_suppress, 1.0f / temperature,
                                                   top_k, top_p, rng, (unsigned char*)seen, tok_buf, state, advance_kv);
  return svlm_check_launch("svlm_penalty_sample(filtered)");
}
