// Fused decode-step (T = 1) kernels: the small per-layer ops around each weight-streaming GEMV are
// folded into its prologue / epilogue, because at batch 1 every separate launch costs ~4-5 us of
// dependent memory latency while the work it does is nanoseconds.
//
//   svlm_dec_qkv       RMSNorm(x) -> [q|k|v] = W x + b -> q to a buffer, k and v STRAIGHT into the
//                      KV pool slot of the new token (un-rotated)
//                      (qwen2/language_forward.py:183,80-82 + StreamingCache.update, streaming_cache.py:72-73)
//   svlm_dec_gate_up   RMSNorm(x) -> h = silu(Wg x) * (Wu x)            (Qwen2MLP, language_forward.py:200-201)
//   svlm_dec_lm_head   final RMSNorm -> last-row logits (fp32 copy of the bf16 value) -> repetition penalty /
//                      EOS suppression -> per-workgroup argmax candidates   (language_forward.py:315,
//                      model_forward.py:243, streaming_generate_qwen.py:73-99)
//   svlm_argmax_finish winner over the candidates + device-side token feedback
// The o_proj and down_proj GEMVs keep using svlm_gemv_bf16 with its residual epilogue.
//
// All are HBM-bound weight streams: 16-B loads straight to VGPRs, the normalised activation row is
// staged once per workgroup in LDS (bf16, rounded exactly where the eager module rounds).
#include "common.h"

extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];

// The activation row is loaded ONCE, 8 elements per thread per chunk (XC chunks of 2048 cover K), together with the norm
// gain, BEFORE the weight preloads are issued: vmcnt retires in order, so a wait for x placed behind the weight loads
// would wait for the head of the weight stream as well.
template <int XC>
struct XRow {
  u32x4_t x[XC], g[XC];
};
template <int XC>
__device__ __forceinline__ void load_x(const bf16_t* __restrict__ x, const bf16_t* __restrict__ ln_w, int K, XRow<XC>& xr) {
#pragma unroll
  for (int i = 0; i < XC; ++i) {
    const int c = (i * 256 + threadIdx.x) * 8;
    const int cc = c < K ? c : 0;
    xr.x[i] = *reinterpret_cast<const u32x4_t*>(x + cc);
    xr.g[i] = *reinterpret_cast<const u32x4_t*>(ln_w + cc);
  }
}

// xs[0..K) = bf16( w * bf16(x * rsqrt(mean(x^2) + eps)) ), rounded exactly where the eager module rounds.  256 threads.
// LDS-only barriers: the weight loads issued before this call stay in flight across them.
template <int XC>
__device__ __forceinline__ void stage_x(const XRow<XC>& xr, float eps, int K, bf16_t* xs) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  float f[XC][8];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < XC; ++i) {
    unpack8(xr.x[i], f[i]);
    if ((i * 256 + tid) * 8 < K) {
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += f[i][j] * f[i][j];
    }
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  lds_barrier();
  const float r = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)K + eps);
#pragma unroll
  for (int i = 0; i < XC; ++i) {
    const int c = (i * 256 + tid) * 8;
    if (c < K) {
      float g[8];
      unpack8(xr.g[i], g);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[i][j] = g[j] * rbf(f[i][j] * r);
      *reinterpret_cast<u32x4_t*>(xs + c) = pack8(f[i]);
    }
  }
  lds_barrier();
}

// The first PRE K-steps (512 elements each) of a wave's weight rows are requested BEFORE the activation row is
// normalised and staged: weights do not depend on x, so the HBM latency of the stream's head overlaps the norm.
// PRE = 3 covers the whole row at K = 1536 (Qwen2-VL-2B); longer rows continue in fully pre-issued batches of PRE.
template <int NR, int PRE>
struct WPre {
  u32x4_t w[PRE][NR];
};
template <int NR, int PRE>
__device__ __forceinline__ void preload_w(const bf16_t* const (&wr)[NR], int base, int K, WPre<NR, PRE>& pre) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int it = 0; it < PRE; ++it) {
    const int c = base + it * 512 + lane * 8;
    const int cc = c < K ? c : 0;                            // clamped (valid) address; its product is masked out
#pragma unroll
    for (int r = 0; r < NR; ++r) pre.w[it][r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wr[r] + cc));
  }
}

template <int NR, int PRE>
__device__ __forceinline__ void fma_batch(const WPre<NR, PRE>& pre, const bf16_t* xs, int base, int K, float (&acc)[NR]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int it = 0; it < PRE; ++it) {
    const int c = base + it * 512 + lane * 8;
    float xf[8];
    unpack8(*reinterpret_cast<const u32x4_t*>(xs + (c < K ? c : 0)), xf);
    if (c >= K) {
#pragma unroll
      for (int i = 0; i < 8; ++i) xf[i] = 0.f;
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      float wf[8];
      unpack8(pre.w[it][r], wf);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[r] = fmaf(wf[i], xf[i], acc[r]);
    }
  }
}

// acc[r] = W[row[r]] . xs   for NR rows of one wave; `pre` holds K-steps [0, PRE) already in flight
template <int NR, int PRE>
__device__ __forceinline__ void wave_dots(const bf16_t* const (&wr)[NR], const bf16_t* xs, int K, float (&acc)[NR], WPre<NR, PRE>& pre) {
#pragma unroll
  for (int r = 0; r < NR; ++r) acc[r] = 0.f;
  fma_batch<NR, PRE>(pre, xs, 0, K, acc);
  for (int base = PRE * 512; base < K; base += PRE * 512) {
    preload_w<NR, PRE>(wr, base, K, pre);
    __builtin_amdgcn_sched_barrier(0);                       // every load of the batch is issued before its first FMA
    fma_batch<NR, PRE>(pre, xs, base, K, acc);
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) acc[r] = wave_sum(acc[r]);
}

// ---------------------------------------------------------------- QKV + KV append
template <int ROWS, int PRE, int XC>
__global__ __launch_bounds__(256) void dec_qkv_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ ln_w, float eps,
                                                      const bf16_t* __restrict__ W, int ldw, const bf16_t* __restrict__ bias,
                                                      bf16_t* __restrict__ q_out, bf16_t* __restrict__ k_planes,
                                                      bf16_t* __restrict__ v_planes, const int* __restrict__ slot_of,
                                                      const int* __restrict__ len_dev, int len_host, int N, int K, int qd, int kd,
                                                      int D, int n_slots, bf16_t* __restrict__ k_lin, bf16_t* __restrict__ v_lin,
                                                      int lin_rows) {
  bf16_t* xs = reinterpret_cast<bf16_t*>(dyn_smem);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n0 = (blockIdx.x * 4 + wave) * ROWS;
  const bf16_t* wr[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) wr[r] = W + (size_t)min(n0 + r, N - 1) * ldw;
  // x first, then everything that does not depend on it: head of the weight stream, the slot, the bias
  XRow<XC> xr;
  load_x<XC>(x, ln_w, K, xr);
  __builtin_amdgcn_sched_barrier(0);
  WPre<ROWS, PRE> pre;
  preload_w<ROWS, PRE>(wr, 0, K, pre);
  const int row = len_dev ? *len_dev : len_host;
  const int slot = slot_of[row];
  bf16_t bpre[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) bpre[r] = bias[min(n0 + r, N - 1)];
  __builtin_amdgcn_sched_barrier(0);
  stage_x<XC>(xr, eps, K, xs);
  if (n0 >= N) return;
  float acc[ROWS];
  wave_dots<ROWS, PRE>(wr, xs, K, acc, pre);
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int n = n0 + r;
      if (n >= N) break;
      const bf16_t v = f2bf(acc[r] + bf2f(bpre[r]));
      if (n < qd) {
        q_out[n] = v;
      } else {
        const int j = n - qd;
        const int jj = j < kd ? j : j - kd;
        bf16_t* plane = j < kd ? k_planes : v_planes;
        plane[((size_t)(jj / D) * n_slots + slot) * D + jj % D] = v;
        if (k_lin != nullptr) {
          // ... and into the cache's linear planes at logical row `row` (svlm_decode_attn_lin): the value as it is, the key UN-rotated in
          // its tile position (rows at or above *lin_state are rotated by the reader) -- the decode attention then needs no slot table
          const int h = jj / D, d = jj % D;
          if (j < kd) k_lin[((size_t)h * (lin_rows >> 4) + (row >> 4)) * 2048 + (d >> 5) * 512 + (((d >> 3) & 3) * 16 + (row & 15)) * 8 + (d & 7)] = v;
          else v_lin[((size_t)h * lin_rows + row) * D + d] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------- gate/up + SwiGLU
template <int ROWS, int PRE, int XC>
__global__ __launch_bounds__(256) void dec_gate_up_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ ln_w, float eps,
                                                          const bf16_t* __restrict__ W, int ldw, bf16_t* __restrict__ h, int I, int K) {
  bf16_t* xs = reinterpret_cast<bf16_t*>(dyn_smem);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n0 = (blockIdx.x * 4 + wave) * ROWS;
  const bf16_t* wr[2 * ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const int n = min(n0 + r, I - 1);
    wr[r] = W + (size_t)n * ldw;               // gate row n
    wr[ROWS + r] = W + (size_t)(I + n) * ldw;  // up row n
  }
  XRow<XC> xr;
  load_x<XC>(x, ln_w, K, xr);
  __builtin_amdgcn_sched_barrier(0);
  WPre<2 * ROWS, PRE> pre;
  preload_w<2 * ROWS, PRE>(wr, 0, K, pre);
  __builtin_amdgcn_sched_barrier(0);
  stage_x<XC>(xr, eps, K, xs);
  if (n0 >= I) return;
  float acc[2 * ROWS];
  wave_dots<2 * ROWS, PRE>(wr, xs, K, acc, pre);
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int n = n0 + r;
      if (n >= I) break;
      const float g = rbf(acc[r]), u = rbf(acc[ROWS + r]);
      h[n] = f2bf(apply_act(g, SVLM_ACT_SILU) * u);
    }
  }
}

// ---------------------------------------------------------------- lm_head + penalty + argmax candidates
template <int ROWS, int PRE, int XC>
__global__ __launch_bounds__(256) void dec_lm_head_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ ln_w, float eps,
                                                          const bf16_t* __restrict__ W, int ldw, float* __restrict__ logits,
                                                          const unsigned char* __restrict__ seen, float penalty,
                                                          const int* __restrict__ suppress, int n_suppress,
                                                          float* __restrict__ part_val, int* __restrict__ part_idx, int V, int K,
                                                          float inv_temp, const unsigned* __restrict__ rng,
                                                          const int* __restrict__ state) {
  bf16_t* xs = reinterpret_cast<bf16_t*>(dyn_smem);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n0 = (blockIdx.x * 4 + wave) * ROWS;
  const bf16_t* wr[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) wr[r] = W + (size_t)min(n0 + r, V - 1) * ldw;
  XRow<XC> xr;
  load_x<XC>(x, ln_w, K, xr);
  __builtin_amdgcn_sched_barrier(0);
  WPre<ROWS, PRE> pre;
  preload_w<ROWS, PRE>(wr, 0, K, pre);
  __builtin_amdgcn_sched_barrier(0);
  stage_x<XC>(xr, eps, K, xs);
  float best = -INFINITY;
  int bi = 0x7fffffff;
  if (n0 < V) {
    float acc[ROWS];
    wave_dots<ROWS, PRE>(wr, xs, K, acc, pre);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int n = n0 + r;
      if (n >= V) break;
      float v = rbf(acc[r]);
      if (lane == 0) logits[n] = v;
      if (seen && seen[n]) v = v < 0.f ? v * penalty : v / penalty;
      for (int s = 0; s < n_suppress; ++s)
        if (suppress[s] == n) v = -INFINITY;
      // temperature sampling = Gumbel-max over the processed scores (sampling.hip); every lane draws the same noise
      if (rng) v = v * inv_temp + svlm_gumbel_noise(rng, (unsigned)(state[1] + 1), (unsigned)n);
      if (v > best || (v == best && n < bi)) { best = v; bi = n; }
    }
  }
  __shared__ float sb[4];
  __shared__ int si[4];
  if (lane == 0) { sb[wave] = best; si[wave] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (sb[w] > best || (sb[w] == best && si[w] < bi)) { best = sb[w]; bi = si[w]; }
    part_val[blockIdx.x] = best;
    part_idx[blockIdx.x] = bi;
  }
}

__global__ __launch_bounds__(1024) void argmax_finish_kernel(const float* __restrict__ part_val, const int* __restrict__ part_idx,
                                                             int n_parts, unsigned char* seen, int* tok_buf, int* state, int advance_kv) {
  float best = -INFINITY;
  int bi = 0x7fffffff;
  // the candidates are the output of the launch in front: every load here is a round trip to another CU's stores (~2 us each behind a
  // kernel boundary), so ALL of them are requested before the first compare -- 16-B loads, six per thread (24k candidates), the rest
  // (a vocabulary beyond 196k, an unaligned tail) in the plain loop
  constexpr int VL = 6;
  const int n4 = ((reinterpret_cast<uintptr_t>(part_val) | reinterpret_cast<uintptr_t>(part_idx)) & 15) == 0 ? n_parts >> 2 : 0;
  f32x4_t vv[VL];
  typedef int i32x4_t __attribute__((ext_vector_type(4)));
  i32x4_t vi[VL];
  if (n4 > 0) {                     // (kernel-uniform)
#pragma unroll
    for (int u = 0; u < VL; ++u) {
      const int j = min((int)threadIdx.x + u * 1024, n4 - 1);
      vv[u] = reinterpret_cast<const f32x4_t*>(part_val)[j];
      vi[u] = reinterpret_cast<const i32x4_t*>(part_idx)[j];
    }
  }
#pragma unroll
  for (int u = 0; u < VL; ++u) {
    if ((int)threadIdx.x + u * 1024 < n4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = vv[u][e];
        const int n = vi[u][e];
        if (v > best || (v == best && n < bi)) { best = v; bi = n; }
      }
    }
  }
#pragma unroll 4
  for (int i = min(n4, VL * 1024) * 4 + threadIdx.x; i < n_parts; i += 1024) {
    const float v = part_val[i];
    const int n = part_idx[i];
    if (v > best || (v == best && n < bi)) { best = v; bi = n; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  __shared__ float sb[16];
  __shared__ int si[16];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { sb[wave] = best; si[wave] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w)
      if (sb[w] > best || (sb[w] == best && si[w] < bi)) { best = sb[w]; bi = si[w]; }
    if (bi == 0x7fffffff) bi = 0;
    const int cur = state[1] + 1;
    tok_buf[cur] = bi;
    state[1] = cur;
    state[0] += advance_kv;
    if (seen) seen[bi] = 1;
  }
}

// ================================================================ launchers
static inline bool smem_ok(int K) { return K > 0 && K % 8 == 0 && K <= 8192; }

// (PRE, XC) by row length: the whole row in one pre-issued batch at K <= 1536, batches of 4 K-steps beyond;
// one 2048-element chunk of x per thread-pass
#define DEC_DISPATCH(KERNEL, ROWS, K, ...)                                  \
  do {                                                                      \
    if ((K) <= 1536) KERNEL<ROWS, 3, 1> __VA_ARGS__;                        \
    else if ((K) <= 2048) KERNEL<ROWS, 4, 1> __VA_ARGS__;                   \
    else if ((K) <= 4096) KERNEL<ROWS, 4, 2> __VA_ARGS__;                   \
    else KERNEL<ROWS, 4, 4> __VA_ARGS__;                                    \
  } while (0)

extern "C" int svlm_dec_qkv_lin(const void* x, const void* ln_w, float eps, const void* W, int ldw, const void* bias, void* q_out,
                                void* k_planes, void* v_planes, const int* slot_of, const int* len_dev, int len_host, int K, int qd,
                                int kd, int D, int n_slots, void* k_lin, void* v_lin, int lin_rows, void* stream);
extern "C" int svlm_dec_qkv(const void* x, const void* ln_w, float eps, const void* W, int ldw, const void* bias, void* q_out,
                            void* k_planes, void* v_planes, const int* slot_of, const int* len_dev, int len_host, int K, int qd,
                            int kd, int D, int n_slots, void* stream) {
  return svlm_dec_qkv_lin(x, ln_w, eps, W, ldw, bias, q_out, k_planes, v_planes, slot_of, len_dev, len_host, K, qd, kd, D, n_slots, nullptr, nullptr,
                          0, stream);
}

// k_lin / v_lin (optional, together): the layer's linear planes of lin_rows rows (svlm_decode_attn_lin); the new token's K row goes there
// too, UN-rotated, at its logical row, and its V row as it is.
extern "C" int svlm_dec_qkv_lin(const void* x, const void* ln_w, float eps, const void* W, int ldw, const void* bias, void* q_out,
                                void* k_planes, void* v_planes, const int* slot_of, const int* len_dev, int len_host, int K, int qd,
                                int kd, int D, int n_slots, void* k_lin, void* v_lin, int lin_rows, void* stream) {
  SVLM_CHECK_ARG((k_lin == nullptr) == (v_lin == nullptr) && (k_lin == nullptr || (D == 128 && lin_rows > 0 && lin_rows % 16 == 0 && (len_dev != nullptr || len_host < lin_rows))),
                 "svlm_dec_qkv_lin: k_lin / v_lin come together, head_dim 128, lin_rows=%d a multiple of 16 above the row", lin_rows);
  SVLM_CHECK_ARG(smem_ok(K) && ldw >= K && ldw % 8 == 0, "svlm_dec_qkv: bad K=%d ldw=%d", K, ldw);
  SVLM_CHECK_ARG(qd > 0 && kd > 0 && D > 0 && kd % D == 0 && n_slots > 0 && bias != nullptr, "svlm_dec_qkv: bad qd=%d kd=%d D=%d", qd, kd, D);
  const int N = qd + 2 * kd;
  // one row per wave: two rows per wave were measured on the 7B shape (4608 x 3584) and are slower (9.96 vs 8.8 us)
  DEC_DISPATCH(dec_qkv_kernel, 1, K, <<<(N + 3) / 4, 256, K * 2, (hipStream_t)stream>>>(
      (const bf16_t*)x, (const bf16_t*)ln_w, eps, (const bf16_t*)W, ldw, (const bf16_t*)bias, (bf16_t*)q_out, (bf16_t*)k_planes,
      (bf16_t*)v_planes, slot_of, len_dev, len_host, N, K, qd, kd, D, n_slots, (bf16_t*)k_lin, (bf16_t*)v_lin, lin_rows));
  return svlm_check_launch("svlm_dec_qkv");
}

extern "C" int svlm_dec_gate_up(const void* x, const void* ln_w, float eps, const void* W, int ldw, void* h, int I, int K, void* stream) {
  SVLM_CHECK_ARG(smem_ok(K) && ldw >= K && ldw % 8 == 0 && I > 0, "svlm_dec_gate_up: bad I=%d K=%d ldw=%d", I, K, ldw);
#ifndef GU_ROWS
#define GU_ROWS 1                 // gate rows (and as many up rows) per wave: 1 measured against 2 on MI355X -- 2B 10.83 -> 10.43 us, 7B 44.0 -> 42.8 us
#endif
  DEC_DISPATCH(dec_gate_up_kernel, GU_ROWS, K, <<<(I + 4 * GU_ROWS - 1) / (4 * GU_ROWS), 256, K * 2, (hipStream_t)stream>>>(
      (const bf16_t*)x, (const bf16_t*)ln_w, eps, (const bf16_t*)W, ldw, (bf16_t*)h, I, K));
  return svlm_check_launch("svlm_dec_gate_up");
}

#ifndef LM_ROWS
#define LM_ROWS 2                 // vocabulary rows per wave: 2 measured against 4 (2B: 78.8 -> 75.7 us; 7B unchanged) and 1 / 3 / 8
#endif
extern "C" long long svlm_dec_lm_head_ws_bytes(int V) { return V <= 0 ? SVLM_EINVAL : (long long)((V + 4 * LM_ROWS - 1) / (4 * LM_ROWS)) * 8; }

static int dec_lm_head_launch(const void* x, const void* ln_w, float eps, const void* W, int ldw, float* logits, const void* seen,
                              float penalty, const int* suppress, int n_suppress, void* ws, int V, int K, float inv_temp,
                              const unsigned* rng, const int* state, void* stream) {
  SVLM_CHECK_ARG(smem_ok(K) && ldw >= K && ldw % 8 == 0 && V > 0, "svlm_dec_lm_head: bad V=%d K=%d ldw=%d", V, K, ldw);
  SVLM_CHECK_ARG(penalty > 0.f && n_suppress >= 0 && n_suppress <= 8 && ws != nullptr && logits != nullptr, "svlm_dec_lm_head: bad sampling args");
  const int nb = (V + 4 * LM_ROWS - 1) / (4 * LM_ROWS);
  float* pv = (float*)ws;
  int* pi = (int*)(pv + nb);
  DEC_DISPATCH(dec_lm_head_kernel, LM_ROWS, K, <<<nb, 256, K * 2, (hipStream_t)stream>>>(
      (const bf16_t*)x, (const bf16_t*)ln_w, eps, (const bf16_t*)W, ldw, logits, (const unsigned char*)seen, penalty, suppress,
      n_suppress, pv, pi, V, K, inv_temp, rng, state));
  return svlm_check_launch("svlm_dec_lm_head");
}

extern "C" int svlm_dec_lm_head(const void* x, const void* ln_w, float eps, const void* W, int ldw, float* logits, const void* seen,
                                float penalty, const int* suppress, int n_suppress, void* ws, int V, int K, void* stream) {
  return dec_lm_head_launch(x, ln_w, eps, W, ldw, logits, seen, penalty, suppress, n_suppress, ws, V, K, 1.0f, nullptr, nullptr, stream);
}

// The same launch with temperature sampling folded in: the candidates are argmax(score / T + Gumbel noise), an exact draw from
// softmax(score / T) (sampling.hip); rng = {seed lo, seed hi} and state = {kv_len, cur} live in device memory.
extern "C" int svlm_dec_lm_head_sample(const void* x, const void* ln_w, float eps, const void* W, int ldw, float* logits, const void* seen,
                                       float penalty, const int* suppress, int n_suppress, void* ws, int V, int K, float temperature,
                                       const unsigned* rng, const int* state, void* stream) {
  SVLM_CHECK_ARG(temperature > 0.f && rng != nullptr && state != nullptr, "svlm_dec_lm_head_sample: bad temperature=%f / null rng or state", temperature);
  return dec_lm_head_launch(x, ln_w, eps, W, ldw, logits, seen, penalty, suppress, n_suppress, ws, V, K, 1.0f / temperature, rng, state, stream);
}

extern "C" int svlm_argmax_finish(const void* ws, int V, void* seen, int* tok_buf, int* state, int advance_kv, void* stream) {
  SVLM_CHECK_ARG(V > 0 && ws != nullptr, "svlm_argmax_finish: bad args");
  const int nb = (V + 4 * LM_ROWS - 1) / (4 * LM_ROWS);
  const float* pv = (const float*)ws;
  const int* pi = (const int*)(pv + nb);
  argmax_finish_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(pv, pi, nb, (unsigned char*)seen, tok_buf, state, advance_kv);
  return svlm_check_launch("svlm_argmax_finish");
}
