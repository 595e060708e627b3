// Decode-step (M = 1) weight-streaming GEMV:  y[N] = epi(W[N,K] . x[K]).
// HBM-bound: every weight byte is read exactly once, straight into VGPRs with 16-B loads
// (no LDS round trip: the operand is not shared between waves), fp32 accumulation,
// wave-shuffle reduction, one output per row.  ROWS rows per wave share each x chunk.
//
// Replaces the decode-step Linear calls of qwen2/language_forward.py:80-82,161, Qwen2MLP (:201)
// and the last-row lm_head of qwen2/model_forward.py:243.
#include "common.h"

template <int ROWS>
__global__ __launch_bounds__(256) void gemv_bf16_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ W, int ldw,
                                                        const bf16_t* __restrict__ bias, const bf16_t* residual,
                                                        bf16_t* y, float* __restrict__ y_f32, int N, int K, int act) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = (blockIdx.x * 4 + wave) * ROWS;
  if (n0 >= N) return;
  const bf16_t* wr[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    int n = n0 + r;
    n = n < N ? n : N - 1;
    wr[r] = W + (size_t)n * ldw;
  }
  float acc[ROWS], epi[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    acc[r] = 0.f;
    // residual / bias are requested up front so that their latency hides under the weight stream
    const int n = min(n0 + r, N - 1);
    epi[r] = (residual ? bf2f(residual[n]) : 0.f);
    acc[r] = 0.f;
  }
  float bia[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) bia[r] = bias ? bf2f(bias[min(n0 + r, N - 1)]) : 0.f;

  const int nfull = K / 512;
#pragma unroll 2
  for (int it = 0; it < nfull; ++it) {
    const int c = it * 512 + lane * 8;
    u32x4_t xv = *reinterpret_cast<const u32x4_t*>(x + c);
    u32x4_t wv[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) wv[r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wr[r] + c));
    float xf[8];
    unpack8(xv, xf);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      float wf[8];
      unpack8(wv[r], wf);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[r] = fmaf(wf[i], xf[i], acc[r]);
    }
  }
  {
    const int c = nfull * 512 + lane * 8;
    if (c < K) {
      u32x4_t xv = *reinterpret_cast<const u32x4_t*>(x + c);
      float xf[8];
      unpack8(xv, xf);
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        u32x4_t wv = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wr[r] + c));
        float wf[8];
        unpack8(wv, wf);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[r] = fmaf(wf[i], xf[i], acc[r]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < ROWS; ++r) acc[r] = wave_sum(acc[r]);
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int n = n0 + r;
      if (n >= N) break;
      float v = acc[r] + bia[r];
      v = apply_act(rbf(v), act);
      if (residual) v = rbf(v + epi[r]);
      if (y) y[n] = f2bf(v);
      if (y_f32) y_f32[n] = v;
    }
  }
}

// Long-K variant: the 4 waves of a workgroup each take a quarter of K for the SAME `ROWS` rows, so a
// skinny output (N = hidden) with a long reduction (K = intermediate) still puts N*4/ROWS waves in flight.
template <int ROWS>
__global__ __launch_bounds__(256) void gemv_bf16_ksplit_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ W, int ldw,
                                                               const bf16_t* __restrict__ bias, const bf16_t* residual,
                                                               bf16_t* y, float* __restrict__ y_f32, int N, int K, int act) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * ROWS;
  const int kq = ((K / 8 + 3) / 4) * 8;               // elements per wave (multiple of 8)
  const int k_lo = wave * kq, k_hi = min(K, k_lo + kq);
  const bf16_t* wr[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) wr[r] = W + (size_t)min(n0 + r, N - 1) * ldw;
  float acc[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) acc[r] = 0.f;
  // epilogue operands requested up front (their latency hides under the weight stream)
  float epi = 0.f, bia = 0.f;
  if (threadIdx.x < ROWS) {
    const int n = min(n0 + (int)threadIdx.x, N - 1);
    if (residual) epi = bf2f(residual[n]);
    if (bias) bia = bf2f(bias[n]);
  }
#pragma unroll 4
  for (int c = k_lo + lane * 8; c < k_hi; c += 512) {
    float xf[8];
    unpack8(*reinterpret_cast<const u32x4_t*>(x + c), xf);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      float wf[8];
      unpack8(__builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wr[r] + c)), wf);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[r] = fmaf(wf[i], xf[i], acc[r]);
    }
  }
  __shared__ float part[4][ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const float s = wave_sum(acc[r]);
    if (lane == 0) part[wave][r] = s;
  }
  __syncthreads();
  if (threadIdx.x < ROWS) {
    const int r = threadIdx.x, n = n0 + r;
    if (n < N) {
      float v = part[0][r] + part[1][r] + part[2][r] + part[3][r] + bia;
      v = apply_act(rbf(v), act);
      if (residual) v = rbf(v + epi);
      if (y) y[n] = f2bf(v);
      if (y_f32) y_f32[n] = v;
    }
  }
}

extern "C" int svlm_gemv_bf16(const void* x, const void* W, int ldw, const void* bias, const void* residual, void* y,
                              float* y_f32, int N, int K, int act, void* stream) {
  SVLM_CHECK_ARG(N > 0 && K > 0 && K % 8 == 0 && ldw % 8 == 0 && ldw >= K, "svlm_gemv_bf16: bad shape N=%d K=%d ldw=%d", N, K, ldw);
  SVLM_CHECK_ARG(act >= 0 && act <= 3, "svlm_gemv_bf16: unknown activation %d", act);
  SVLM_CHECK_ARG(y || y_f32, "svlm_gemv_bf16: no output buffer");
  hipStream_t s = (hipStream_t)stream;
  if (N <= 8192 && K >= 4096) {        // skinny output, long reduction (down_proj): split K over the workgroup's waves
    gemv_bf16_ksplit_kernel<2><<<(N + 1) / 2, 256, 0, s>>>((const bf16_t*)x, (const bf16_t*)W, ldw, (const bf16_t*)bias,
                                                 (const bf16_t*)residual, (bf16_t*)y, y_f32, N, K, act);
    return svlm_check_launch("svlm_gemv_bf16(ksplit)");
  }
  // rows per wave: keep >= ~2 waves of work per SIMD on 256 CUs, amortise x over up to 4 rows
  if (N >= 16384) {
    gemv_bf16_kernel<4><<<(N + 15) / 16, 256, 0, s>>>((const bf16_t*)x, (const bf16_t*)W, ldw, (const bf16_t*)bias,
                                                     (const bf16_t*)residual, (bf16_t*)y, y_f32, N, K, act);
  } else if (N >= 4096) {
    gemv_bf16_kernel<2><<<(N + 7) / 8, 256, 0, s>>>((const bf16_t*)x, (const bf16_t*)W, ldw, (const bf16_t*)bias,
                                                   (const bf16_t*)residual, (bf16_t*)y, y_f32, N, K, act);
  } else {
    gemv_bf16_kernel<1><<<(N + 3) / 4, 256, 0, s>>>((const bf16_t*)x, (const bf16_t*)W, ldw, (const bf16_t*)bias,
                                                   (const bf16_t*)residual, (bf16_t*)y, y_f32, N, K, act);
  }
  return svlm_check_launch("svlm_gemv_bf16");
}
