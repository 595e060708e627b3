// FP8 (OCP e4m3, gfx950) MFMA GEMM for the ViT of BASELINE configs[4]:  C[M,N] = epi((A8 . W8^T) * sa[m] * sw[n])
//   A8 (M,K) fp8 with one fp32 scale per ROW (dynamic, from svlm_quant_rows_fp8), W8 (N,K) fp8 with one fp32 scale per output
//   channel (static, quantised once when the weights are loaded);
//   epi(y): y = bf16(y + bias[n]); y = act(y); y = bf16(y + residual[m,n]) -- the epilogue of svlm_gemm_bf16.
// The reference has no fp8 path (its ViT GEMMs are third-party bf16 cuBLAS calls: qwen2/vision_forward.py:14,33,43-49,57,80);
// the arithmetic is pinned by the oracle's linear_fp8 (oracle/model.py), which quantises with the same recipe.
//
// Structure: the LDS-DMA kernel of gemm.hip byte for byte -- (32*TM) x 128 tile, 128-BYTE rows per K-step (= 128 fp8 instead of
// 64 bf16), 3-stage global_load_lds ring, XOR-swizzled 16-B chunks, counted vmcnt + raw barrier -- so one K-step stages the same
// bytes and feeds TWICE the MFMA work (4 x v_mfma_f32_16x16x32_fp8_fp8 k-steps per tile instead of 2 bf16 ones): the K loop of
// these skinny GEMMs is bound by the L2 -> LDS stream (gemm.hip), and fp8 halves the bytes per FLOP.
#include "common.h"
#include <stdlib.h>

#define F8_BN 128
#define F8_BK 128           // fp8 elements per K-step = 128 bytes per row
extern __shared__ __attribute__((aligned(16))) unsigned char gemm8_dyn_smem[];

#define GLDS8(gp, lp) \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp), (__attribute__((address_space(3))) void*)(lp), 16, 0, 0)

template <int TM, int NS>
__global__ __launch_bounds__(256) void gemm_fp8_kernel(const unsigned char* __restrict__ A, int lda, const float* __restrict__ a_scale,
                                                       const unsigned char* __restrict__ W, int ldw, const float* __restrict__ w_scale,
                                                       const bf16_t* __restrict__ bias, const bf16_t* residual, int ldr,
                                                       bf16_t* C, int ldc, float* __restrict__ partial,
                                                       int M, int N, int K, int k_per_split, int act, int gm, int gn, int splits) {
  constexpr int BM = 32 * TM;
  constexpr int A_INST = BM / 32;
  constexpr int W_INST = F8_BN / 32;
  constexpr int NPT = A_INST + W_INST;
  constexpr int STAGE_B = (BM + F8_BN) * 128;
  unsigned char* smem = gemm8_dyn_smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int bid = blockIdx.x;
  int tm, tn, split;
  if (splits > 1) {
    split = bid % splits;
    const int t = bid / splits;
    tn = t % gn;
    tm = t / gn;
  } else {
    split = 0;
    const int npx = (gn + 7) / 8;
    const int local = bid / 8;
    tn = (bid % 8) * npx + local / gm;
    tm = local % gm;
    if (tn >= gn) return;
  }
  const int m0 = tm * BM, n0 = tn * F8_BN;
  const int k_begin = split * k_per_split;
  const int k_end = min(K, k_begin + k_per_split);
  const int nk = (k_end - k_begin) / F8_BK;                // K % 128 == 0 and k_per_split % 128 == 0
  const int rsub = lane >> 3, ppos = lane & 7;
  const unsigned char* a_src[A_INST];
  const unsigned char* w_src[W_INST];
#pragma unroll
  for (int i = 0; i < A_INST; ++i) {
    const int row = (wave * A_INST + i) * 8 + rsub;
    a_src[i] = A + (size_t)min(m0 + row, M - 1) * lda + k_begin + ((ppos ^ (row & 7)) * 16);
  }
#pragma unroll
  for (int i = 0; i < W_INST; ++i) {
    const int row = (wave * W_INST + i) * 8 + rsub;
    w_src[i] = W + (size_t)min(n0 + row, N - 1) * ldw + k_begin + ((ppos ^ (row & 7)) * 16);
  }
  auto issue = [&](int kt, int stage) {
    unsigned char* sa = smem + stage * STAGE_B;
    unsigned char* sw = sa + BM * 128;
#pragma unroll
    for (int i = 0; i < A_INST; ++i) GLDS8(a_src[i] + kt * F8_BK, sa + (wave * A_INST + i) * 1024);
#pragma unroll
    for (int i = 0; i < W_INST; ++i) GLDS8(w_src[i] + kt * F8_BK, sw + (wave * W_INST + i) * 1024);
  };
  f32x4_t acc[TM][4];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;
  auto compute = [&](int stage) {
    const unsigned char* sa = smem + stage * STAGE_B;
    const unsigned char* sw = sa + BM * 128;
#ifndef GEMM_FP8_NONSCALED
    // One block-scaled MFMA per 128-deep K step: v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands runs at TWICE the bf16 rate
    // (the non-scaled 16x16x32 fp8 form below: the bf16 rate, MI355X_MICROARCH.md "Matrix cores").  Block scales are all 2^0 (E8M0
    // 127): the per-row x per-channel fp32 scales stay in the epilogue, so the arithmetic is the non-scaled kernel's -- exact fp8
    // products, fp32 accumulation.  A dot product does not care in which order k is walked as long as both operands walk it alike:
    // lane (fr, fq) takes the 16-byte chunks fq and 4 + fq of its row (two conflict-free ds_read_b128 of the swizzled image) as
    // the 32 fp8 of its k block -- the four lane groups cover disjoint k ranges in both operands.
    typedef int i8x_t __attribute__((ext_vector_type(8)));
    typedef int i4x_t __attribute__((ext_vector_type(4)));
    i8x_t af[TM], wf[4];
    const int p0 = ((fq) ^ (fr & 7)) * 16, p1 = ((4 + fq) ^ (fr & 7)) * 16;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const unsigned char* r = sa + (wm * 16 * TM + i * 16 + fr) * 128;
      const i4x_t lo = *reinterpret_cast<const i4x_t*>(r + p0), hi = *reinterpret_cast<const i4x_t*>(r + p1);
      af[i] = i8x_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const unsigned char* r = sw + (wn * 64 + jj * 16 + fr) * 128;
      const i4x_t lo = *reinterpret_cast<const i4x_t*>(r + p0), hi = *reinterpret_cast<const i4x_t*>(r + p1);
      wf[jj] = i8x_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
        acc[i][jj] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[jj], af[i], acc[i][jj], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
#else
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      // A dot product does not care in which order k is walked as long as both operands walk it alike: lane (fr, fq) takes the
      // 16-B chunk 4 j + fq of its row with ONE ds_read_b128 (the bank-conflict-free swizzle of the bf16 kernel) and feeds its low
      // 8 bytes to MFMA k-step 2 j and its high 8 bytes to k-step 2 j + 1 -- the four lane groups still cover disjoint k ranges.
      typedef long l2_t __attribute__((ext_vector_type(2)));
      l2_t af[TM], wf[4];
      const int pos = ((4 * j + fq) ^ (fr & 7)) * 16;
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const l2_t*>(sa + (wm * 16 * TM + i * 16 + fr) * 128 + pos);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) wf[jj] = *reinterpret_cast<const l2_t*>(sw + (wn * 64 + jj * 16 + fr) * 128 + pos);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
            acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wf[jj][h], af[i][h], acc[i][jj], 0, 0, 0);
    }
#endif
  };
  constexpr int AHEAD = NS - 1;
#pragma unroll
  for (int t = 0; t < AHEAD; ++t)
    if (t < nk) issue(t, t);
  for (int kt = 0; kt < nk; ++kt) {
    const int newer = min(AHEAD - 1, nk - 1 - kt);
    if (newer >= 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * NPT) : "memory");
    else if (newer == 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (kt + AHEAD < nk) issue(kt + AHEAD, (kt + AHEAD) % NS);
    compute(kt % NS);
  }
  // epilogue: lane holds m = fr (column of D), n = 4*fq + r (rows of D); scales first, then the bf16 epilogue of svlm_gemm_bf16
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * 16 * TM + i * 16 + fr;
    if (m >= M) continue;
    const float sa_m = a_scale[m];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + fq * 4;
      if (n >= N) continue;
      const f32x4_t sw4 = *reinterpret_cast<const f32x4_t*>(w_scale + n);
      float y[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) y[r] = acc[i][j][r] * (sa_m * sw4[r]);
      if (partial) {
        *reinterpret_cast<f32x4_t*>(partial + ((size_t)split * M + m) * N + n) = f32x4_t{y[0], y[1], y[2], y[3]};
        continue;
      }
      if (bias) {
        u32x2_t bv = *reinterpret_cast<const u32x2_t*>(bias + n);
        y[0] += lo_bf(bv[0]); y[1] += hi_bf(bv[0]); y[2] += lo_bf(bv[1]); y[3] += hi_bf(bv[1]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) y[r] = apply_act(rbf(y[r]), act);
      if (residual) {
        u32x2_t rv = *reinterpret_cast<const u32x2_t*>(residual + (size_t)m * ldr + n);
        y[0] = y[0] + lo_bf(rv[0]); y[1] = y[1] + hi_bf(rv[0]); y[2] = y[2] + lo_bf(rv[1]); y[3] = y[3] + hi_bf(rv[1]);
      }
      u32x2_t o;
      o[0] = pack2(y[0], y[1]);
      o[1] = pack2(y[2], y[3]);
      *reinterpret_cast<u32x2_t*>(C + (size_t)m * ldc + n) = o;
    }
  }
}

// ---------------------------------------------------------------- row quantiser: bf16 (rows, cols) -> fp8 e4m3 + one fp32 scale per row
// scale = max|row| / 448 (1 for an all-zero row); q = rne_fp8(x / scale).  One wave per row, 16-B loads, 8-B stores.
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const bf16_t* __restrict__ x, int ldx, unsigned char* __restrict__ q, int ldq,
                                                            float* __restrict__ scale, int rows, int cols) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const bf16_t* xr = x + (size_t)row * ldx;
  float mx = 0.f;
  for (int c = lane * 8; c < cols; c += 512) {
    float f[8];
    unpack8(*reinterpret_cast<const u32x4_t*>(xr + c), f);
#pragma unroll
    for (int i = 0; i < 8; ++i) mx = fmaxf(mx, fabsf(f[i]));
  }
  mx = wave_max(mx);
  const float s = mx > 0.f ? mx / 448.0f : 1.0f;
  if (lane == 0) scale[row] = s;
  unsigned char* qr = q + (size_t)row * ldq;
  for (int c = lane * 8; c < cols; c += 512) {
    float f[8];
    unpack8(*reinterpret_cast<const u32x4_t*>(xr + c), f);
    u32x2_t o;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int w = 0;
      w = __builtin_amdgcn_cvt_pk_fp8_f32(f[4 * h] / s, f[4 * h + 1] / s, w, false);
      w = __builtin_amdgcn_cvt_pk_fp8_f32(f[4 * h + 2] / s, f[4 * h + 3] / s, w, true);
      o[h] = (unsigned)w;
    }
    *reinterpret_cast<u32x2_t*>(qr + c) = o;
  }
}

extern "C" int svlm_quant_rows_fp8(const void* x, int ldx, void* q, int ldq, float* scale, int rows, int cols, void* stream) {
  SVLM_CHECK_ARG(rows >= 0 && cols > 0 && cols % 8 == 0 && ldx % 8 == 0 && ldq % 8 == 0 && ldx >= cols && ldq >= cols,
                 "svlm_quant_rows_fp8: bad shape rows=%d cols=%d ldx=%d ldq=%d (cols and leading dims must be multiples of 8)", rows, cols, ldx, ldq);
  if (rows == 0) return SVLM_OK;
  quant_rows_fp8_kernel<<<(rows + 3) / 4, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, ldx, (unsigned char*)q, ldq, scale, rows, cols);
  return svlm_check_launch("svlm_quant_rows_fp8");
}

// split-K reduce kernels of gemm.hip (the fp8 kernel writes SCALED fp32 partials, so the bf16 reduce epilogues apply as they are)
int svlm_gemm_reduce_launch(const float* partial, int splits, const void* bias, const void* residual, int ldr, void* C, int ldc, int M, int N,
                            int act, const void* norm_w, const void* norm_b, float eps, void* XN, int ldxn, void* stream, void* XN8 = nullptr,
                            int ldxn8 = 0, float* xn_scale = nullptr);

static int gemm_fp8_impl(const void* A8, int lda, const float* a_scale, const void* W8, int ldw, const float* w_scale, const void* bias,
                         const void* residual, int ldr, void* C, int ldc, int M, int N, int K, int act, void* ws, long long ws_bytes,
                         const void* norm_w, const void* norm_b, float eps, void* XN, int ldxn, void* XN8, int ldxn8, float* xn_scale, void* stream);

extern "C" int svlm_gemm_fp8(const void* A8, int lda, const float* a_scale, const void* W8, int ldw, const float* w_scale, const void* bias,
                             const void* residual, int ldr, void* C, int ldc, int M, int N, int K, int act, void* ws, long long ws_bytes,
                             const void* norm_w, const void* norm_b, float eps, void* XN, int ldxn, void* stream) {
  return gemm_fp8_impl(A8, lda, a_scale, W8, ldw, w_scale, bias, residual, ldr, C, ldc, M, N, K, act, ws, ws_bytes, norm_w, norm_b, eps, XN, ldxn,
                       nullptr, 0, nullptr, stream);
}

// svlm_gemm_fp8 whose fused norm ALSO leaves the normalised rows as the next GEMM's fp8 operand (e4m3 + one fp32 scale per row, the
// recipe of svlm_quant_rows_fp8 on the bf16 values of XN): the quantiser launch between two Linears of the fp8 tower disappears.
extern "C" int svlm_gemm_fp8_normq(const void* A8, int lda, const float* a_scale, const void* W8, int ldw, const float* w_scale, const void* bias,
                                   const void* residual, int ldr, void* C, int ldc, int M, int N, int K, int act, void* ws, long long ws_bytes,
                                   const void* norm_w, const void* norm_b, float eps, void* XN, int ldxn, void* XN8, int ldxn8, float* xn_scale,
                                   void* stream) {
  SVLM_CHECK_ARG(norm_w != nullptr && XN != nullptr && XN8 != nullptr && xn_scale != nullptr && ldxn8 % 8 == 0 && ldxn8 >= N,
                 "svlm_gemm_fp8_normq: needs a norm, its bf16 output, an fp8 output with 8-B aligned rows and a scale vector (ldxn8=%d N=%d)", ldxn8, N);
  return gemm_fp8_impl(A8, lda, a_scale, W8, ldw, w_scale, bias, residual, ldr, C, ldc, M, N, K, act, ws, ws_bytes, norm_w, norm_b, eps, XN, ldxn,
                       XN8, ldxn8, xn_scale, stream);
}

static int gemm_fp8_impl(const void* A8, int lda, const float* a_scale, const void* W8, int ldw, const float* w_scale, const void* bias,
                         const void* residual, int ldr, void* C, int ldc, int M, int N, int K, int act, void* ws, long long ws_bytes,
                         const void* norm_w, const void* norm_b, float eps, void* XN, int ldxn, void* XN8, int ldxn8, float* xn_scale, void* stream) {
  SVLM_CHECK_ARG(M >= 0 && N > 0 && K > 0 && K % F8_BK == 0 && N % 4 == 0, "svlm_gemm_fp8: bad shape M=%d N=%d K=%d (K %% 128 == 0, N %% 4 == 0)", M, N, K);
  SVLM_CHECK_ARG(lda % 16 == 0 && ldw % 16 == 0 && lda >= K && ldw >= K && ldc % 4 == 0 && ldc >= N && (!residual || ldr % 4 == 0),
                 "svlm_gemm_fp8: leading dims must keep 16-B row alignment (lda=%d ldw=%d ldc=%d ldr=%d)", lda, ldw, ldc, ldr);
  SVLM_CHECK_ARG(act >= 0 && act <= 3 && a_scale != nullptr && w_scale != nullptr, "svlm_gemm_fp8: bad activation %d or missing scales", act);
  SVLM_CHECK_ARG(norm_w == nullptr || (XN != nullptr && N % 8 == 0 && ldc % 8 == 0 && ldxn % 8 == 0 && ldxn >= N && eps > 0.f),
                 "svlm_gemm_fp8: the fused norm needs an output with 16-B aligned rows and N %% 8 == 0 (N=%d ldc=%d ldxn=%d)", N, ldc, ldxn);
  if (M == 0) return SVLM_OK;
  hipStream_t st = (hipStream_t)stream;
  const int gn = (N + F8_BN - 1) / F8_BN;
  // plan: 64-row tiles -- 72 KB of LDS, so TWO workgroups share a CU and one's DMA wait hides under the other's MFMAs; measured on
  // MI355X (tools/gemm_fp8_vs_bf16.py, ViT shapes): 64-row tiles 559 / 733 TFLOP/s on qkv at M = 1024 / 8192 where 128-row tiles
  // (96 KB, one workgroup per CU) reach 416 / 507 and the bf16 kernel 466 / 551.  Split K when the tile grid is still below ~one
  // round of the 256 CUs (ViT proj / fc2 at one frame: N = 1280).
  int bm = 64;
  int splits = 1;
  const long long tiles = (long long)((M + bm - 1) / bm) * gn;
  if (ws != nullptr && tiles < 160 && K >= 1024) {
    splits = (int)((256 + tiles - 1) / tiles);
    if (splits > K / 512) splits = K / 512;
    if (splits > 8) splits = 8;
    while (splits > 1 && (long long)splits * M * N * 4 > ws_bytes) --splits;
  }
  if (const char* fs = svlm_env("SVLM_GEMM8_SPLITS")) { const int v = atoi(fs); if (v >= 1 && (v == 1 || (ws && (long long)v * M * N * 4 <= ws_bytes))) splits = v; }
  if (const char* fb = svlm_env("SVLM_GEMM8_BM")) { const int v = atoi(fb); if (v == 64 || v == 128) bm = v; }
  if (norm_w != nullptr && splits == 1 && ws != nullptr && K >= 1024 && (long long)2 * M * N * 4 <= ws_bytes && N <= 4096) splits = 2;   // the reduce carries the norm
  int kps = K;
  if (splits > 1) {
    kps = ((K + splits - 1) / splits + F8_BK - 1) / F8_BK * F8_BK;
    splits = (K + kps - 1) / kps;
  }
  const int gm = (M + bm - 1) / bm;
  float* partial = splits > 1 ? (float*)ws : nullptr;
  const int nblocks = splits > 1 ? gn * gm * splits : 8 * ((gn + 7) / 8) * gm;
  constexpr int L2 = 3 * (64 + F8_BN) * 128, L4 = 3 * (128 + F8_BN) * 128;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_fp8_kernel<2, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, L2);
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_fp8_kernel<4, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, L4);
    if (e1 != hipSuccess || e2 != hipSuccess) {
      svlm_set_error("svlm_gemm_fp8: cannot reserve %d B of LDS: %s", L4, hipGetErrorString(e1 != hipSuccess ? e1 : e2));
      return SVLM_ELAUNCH;
    }
    attr_done = true;
  }
  if (bm == 64)
    gemm_fp8_kernel<2, 3><<<nblocks, 256, L2, st>>>((const unsigned char*)A8, lda, a_scale, (const unsigned char*)W8, ldw, w_scale, (const bf16_t*)bias,
                                                  (const bf16_t*)residual, ldr, (bf16_t*)C, ldc, partial, M, N, K, kps, act, gm, gn, splits);
  else
    gemm_fp8_kernel<4, 3><<<nblocks, 256, L4, st>>>((const unsigned char*)A8, lda, a_scale, (const unsigned char*)W8, ldw, w_scale, (const bf16_t*)bias,
                                                  (const bf16_t*)residual, ldr, (bf16_t*)C, ldc, partial, M, N, K, kps, act, gm, gn, splits);
  int rc = svlm_check_launch("svlm_gemm_fp8");
  if (rc) return rc;
  if (splits > 1 || norm_w != nullptr)
    return svlm_gemm_reduce_launch(partial, splits, bias, residual, ldr, C, ldc, M, N, act, norm_w, norm_b, eps, XN, ldxn, stream, XN8, ldxn8, xn_scale);
  return SVLM_OK;
}
