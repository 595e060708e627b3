// Decode-step (M = 1) weight-streaming GEMV:  y[N] = epi(W[N,K] . x[K]).
// HBM-bound: every weight byte is read exactly once, straight into VGPRs with 16-B loads
// (no LDS round trip: the operand is not shared between waves), fp32 accumulation,
// wave-shuffle reduction, one output per row.  ROWS rows per wave share each x chunk.
//
// Replaces the decode-step Linear calls of qwen2/language_forward.py:80-82,161, Qwen2MLP (:201)
// and the last-row lm_head of qwen2/model_forward.py:243.
#include "common.h"
#include <stdlib.h>

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
  float acc[ROWS];
  bf16_t epi_raw[ROWS], bia_raw[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    acc[r] = 0.f;
    // residual / bias are requested up front, branch-free (null -> x[0]) and converted only in the epilogue, so that
    // their latency hides under the weight stream
    const int n = min(n0 + r, N - 1);
    epi_raw[r] = (residual ? residual : x)[residual ? n : 0];
    bia_raw[r] = (bias ? bias : x)[bias ? n : 0];
  }
  // all 16-B loads of a batch are in flight before the first FMA; lanes past K load a clamped address and see x = 0
  constexpr int PRE = ROWS >= 4 ? 3 : 4;
  for (int base = 0; base < K; base += PRE * 512) {
    u32x4_t xv[PRE], wv[PRE][ROWS];
#pragma unroll
    for (int it = 0; it < PRE; ++it) {
      const int c = base + it * 512 + lane * 8;
      const int cc = c < K ? c : 0;
      xv[it] = *reinterpret_cast<const u32x4_t*>(x + cc);
#pragma unroll
      for (int r = 0; r < ROWS; ++r) wv[it][r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wr[r] + cc));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < PRE; ++it) {
      const int c = base + it * 512 + lane * 8;
      float xf[8];
      unpack8(xv[it], xf);
      if (c >= K) {
#pragma unroll
        for (int i = 0; i < 8; ++i) xf[i] = 0.f;
      }
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        float wf[8];
        unpack8(wv[it][r], wf);
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
      float v = acc[r] + (bias ? bf2f(bia_raw[r]) : 0.f);
      v = apply_act(rbf(v), act);
      if (residual) v = rbf(v + bf2f(epi_raw[r]));
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
  // epilogue operands requested up front, unconditionally (a null pointer is redirected to x so that the load stays
  // branch-free and is only WAITED for in the epilogue: its latency hides under the weight stream)
  const int ne = min(n0 + (int)(threadIdx.x % ROWS), N - 1);
  const bf16_t epi_raw = (residual ? residual : x)[residual ? ne : 0];
  const bf16_t bia_raw = (bias ? bias : x)[bias ? ne : 0];
  // Every 16-B load of a batch is issued before the first FMA (a lane-variant `c < k_hi` loop bound makes the compiler
  // wait for each iteration's loads before it issues the next ones: 2-3 KB in flight per wave instead of ~15).
  // Lanes past the end of the wave's K range load a clamped (valid) address and multiply by a zeroed x chunk.
  constexpr int PRE = 5;
  for (int base = k_lo; base < k_hi; base += PRE * 512) {
    u32x4_t xv[PRE], wv[PRE][ROWS];
#pragma unroll
    for (int it = 0; it < PRE; ++it) {
      const int c = base + it * 512 + lane * 8;
      const int cc = c < k_hi ? c : k_lo;
      xv[it] = *reinterpret_cast<const u32x4_t*>(x + cc);
#pragma unroll
      for (int r = 0; r < ROWS; ++r) wv[it][r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wr[r] + cc));
    }
    __builtin_amdgcn_sched_barrier(0);      // keep the unpack/FMA of early chunks from sinking between the load issues
#pragma unroll
    for (int it = 0; it < PRE; ++it) {
      const int c = base + it * 512 + lane * 8;
      float xf[8];
      unpack8(xv[it], xf);
      if (c >= k_hi) {
#pragma unroll
        for (int i = 0; i < 8; ++i) xf[i] = 0.f;
      }
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        float wf[8];
        unpack8(wv[it][r], wf);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[r] = fmaf(wf[i], xf[i], acc[r]);
      }
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
      float v = part[0][r] + part[1][r] + part[2][r] + part[3][r] + (bias ? bf2f(bia_raw) : 0.f);
      v = apply_act(rbf(v), act);
      if (residual) v = rbf(v + bf2f(epi_raw));
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
#ifndef KS_ROWS
#define KS_ROWS 2
#endif
    gemv_bf16_ksplit_kernel<KS_ROWS><<<(N + KS_ROWS - 1) / KS_ROWS, 256, 0, s>>>((const bf16_t*)x, (const bf16_t*)W, ldw, (const bf16_t*)bias,
                                                 (const bf16_t*)residual, (bf16_t*)y, y_f32, N, K, act);
    return svlm_check_launch("svlm_gemv_bf16(ksplit)");
  }
  // rows per wave: keep >= ~2 waves of work per SIMD on 256 CUs, amortise x over up to 4 rows
  static const int force_rows = svlm_env("SVLM_GEMV_ROWS") ? atoi(svlm_env("SVLM_GEMV_ROWS")) : 0;     // tuning aid
  if (force_rows == 4 || (force_rows == 0 && N >= 16384)) {
    gemv_bf16_kernel<4><<<(N + 15) / 16, 256, 0, s>>>((const bf16_t*)x, (const bf16_t*)W, ldw, (const bf16_t*)bias,
                                                     (const bf16_t*)residual, (bf16_t*)y, y_f32, N, K, act);
  } else if (force_rows == 2 || (force_rows == 0 && (N >= 4096 || (long long)N * K >= (8LL << 20)))) {      // enough rows, or rows long enough to want 2 per wave
    gemv_bf16_kernel<2><<<(N + 7) / 8, 256, 0, s>>>((const bf16_t*)x, (const bf16_t*)W, ldw, (const bf16_t*)bias,
                                                   (const bf16_t*)residual, (bf16_t*)y, y_f32, N, K, act);
  } else {
    gemv_bf16_kernel<1><<<(N + 3) / 4, 256, 0, s>>>((const bf16_t*)x, (const bf16_t*)W, ldw, (const bf16_t*)bias,
                                                   (const bf16_t*)residual, (bf16_t*)y, y_f32, N, K, act);
  }
  return svlm_check_launch("svlm_gemv_bf16");
}
