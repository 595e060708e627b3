// bf16 MFMA GEMM with fused epilogues:  C[M,N] = epi(A[M,K] . W[N,K]^T)
//   epi(acc) : y = bf16(acc + bias[n]);  y = act(y);  y = bf16(y + residual[m,n])   (each step optional)
// Both operands are K-contiguous (torch Linear layout), so A and W rows feed the MFMA
// fragments directly.  The product is computed transposed (D[n][m] = sum_k W[n][k] A[m][k],
// W as the MFMA "A" operand) so that every lane ends up with 4 CONSECUTIVE n for one m and
// the epilogue stores 8 bytes per lane instead of 2.
//
// Replaces (reference call sites, all third-party GEMMs): qwen2/vision_forward.py:14,33,57 and the
// VisionMlp / PatchMerger linears (:43-49,80), qwen2/language_forward.py:80-82,161 and Qwen2MLP (:201).
//
// Tile: (32*TM) x 128 x 64, 256 threads = 4 waves as 2(m) x 2(n); per wave TM x 4 tiles of
// v_mfma_f32_16x16x32_bf16.  LDS rows padded to 144 B (conflict-free ds_read_b128 over 16 rows),
// global->register prefetch of tile k+1 overlaps the MFMAs of tile k.
#include "common.h"

#define GEMM_BN 128
#define GEMM_BK 64
#define GEMM_LD 72  // padded LDS row stride in bf16 (144 B)

template <int TM>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const bf16_t* __restrict__ A, int lda,
                                                        const bf16_t* __restrict__ W, int ldw,
                                                        const bf16_t* __restrict__ bias,
                                                        const bf16_t* residual, int ldr,
                                                        bf16_t* C, int ldc, int M, int N, int K, int act) {
  constexpr int BM = 32 * TM;
  constexpr int A_PASSES = BM / 32;
  constexpr int W_PASSES = GEMM_BN / 32;
  __shared__ __attribute__((aligned(16))) bf16_t smem[(BM + GEMM_BN) * GEMM_LD];
  bf16_t* As = smem;
  bf16_t* Ws = smem + BM * GEMM_LD;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * GEMM_BN;
  const int lrow = tid >> 3, lchunk = tid & 7;  // loader: 8 threads cover one 128-B row segment

  u32x4_t ra[A_PASSES], rw[W_PASSES];
  auto load_tile = [&](int k0) {
    const int k = k0 + lchunk * 8;
    const bool kin = k < K;
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) {
      int r = m0 + p * 32 + lrow;
      r = r < M ? r : M - 1;
      ra[p] = kin ? *reinterpret_cast<const u32x4_t*>(A + (size_t)r * lda + k) : u32x4_t{0, 0, 0, 0};
    }
#pragma unroll
    for (int p = 0; p < W_PASSES; ++p) {
      int r = n0 + p * 32 + lrow;
      r = r < N ? r : N - 1;
      rw[p] = kin ? *reinterpret_cast<const u32x4_t*>(W + (size_t)r * ldw + k) : u32x4_t{0, 0, 0, 0};
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p)
      *reinterpret_cast<u32x4_t*>(As + (p * 32 + lrow) * GEMM_LD + lchunk * 8) = ra[p];
#pragma unroll
    for (int p = 0; p < W_PASSES; ++p)
      *reinterpret_cast<u32x4_t*>(Ws + (p * 32 + lrow) * GEMM_LD + lchunk * 8) = rw[p];
  };

  f32x4_t acc[TM][4];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = (K + GEMM_BK - 1) / GEMM_BK;
  load_tile(0);
  store_tile();
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_tile((kt + 1) * GEMM_BK);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t af[TM], wf[4];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[i] = *reinterpret_cast<const bf16x8_t*>(As + (wm * 16 * TM + i * 16 + fr) * GEMM_LD + ks * 32 + fq * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        wf[j] = *reinterpret_cast<const bf16x8_t*>(Ws + (wn * 64 + j * 16 + fr) * GEMM_LD + ks * 32 + fq * 8);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (kt + 1 < nk) {
      store_tile();
      __syncthreads();
    }
  }

  // epilogue: lane holds m = fr (column of D), n = 4*fq + r (rows of D)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * 16 * TM + i * 16 + fr;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + fq * 4;
      if (n >= N) continue;
      float y[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
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

extern "C" int svlm_gemm_bf16(const void* A, int lda, const void* W, int ldw, const void* bias, const void* residual, int ldr,
                              void* C, int ldc, int M, int N, int K, int act, void* stream) {
  SVLM_CHECK_ARG(M >= 0 && N > 0 && K > 0, "svlm_gemm_bf16: bad shape M=%d N=%d K=%d", M, N, K);
  SVLM_CHECK_ARG(K % 8 == 0 && N % 4 == 0, "svlm_gemm_bf16: K=%d must be a multiple of 8 and N=%d of 4", K, N);
  SVLM_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0 && (!residual || ldr % 4 == 0),
                 "svlm_gemm_bf16: leading dims must keep 16-B row alignment (lda=%d ldw=%d ldc=%d ldr=%d)", lda, ldw, ldc, ldr);
  SVLM_CHECK_ARG(lda >= K && ldw >= K && ldc >= N, "svlm_gemm_bf16: leading dim smaller than row length");
  SVLM_CHECK_ARG(act >= 0 && act <= 3, "svlm_gemm_bf16: unknown activation %d", act);
  if (M == 0) return SVLM_OK;
  const int gn = (N + GEMM_BN - 1) / GEMM_BN;
  // small-M shapes use 64-row tiles so that the grid still covers the chip
  const bool small = (M <= 64) || ((long long)((M + 127) / 128) * gn < 256 && M % 128 != 0 && M % 128 <= 64) ||
                     ((long long)((M + 127) / 128) * gn < 128);
  if (small) {
    dim3 grid(gn, (M + 63) / 64);
    gemm_bf16_kernel<2><<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, (const bf16_t*)bias,
                                                              (const bf16_t*)residual, ldr, (bf16_t*)C, ldc, M, N, K, act);
  } else {
    dim3 grid(gn, (M + 127) / 128);
    gemm_bf16_kernel<4><<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, (const bf16_t*)bias,
                                                              (const bf16_t*)residual, ldr, (bf16_t*)C, ldc, M, N, K, act);
  }
  return svlm_check_launch("svlm_gemm_bf16");
}
