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
// Tile (32*TM) x 128 x 64, 256 threads = 4 waves as 2(m) x 2(n); per wave TM x 4 tiles of
// v_mfma_f32_16x16x32_bf16.  Two LDS buffers (rows padded to 144 B: conflict-free ds_read_b128 over
// 16 rows) and a two-deep register ring: the global loads of tile k+3 are issued while tile k is
// computed, so every load has two full iterations to land and there is one barrier per K-tile.
// The path's shapes are skinny (M = 275..1024, N = 1280..5120) with long K (up to 8960): when the
// tile grid cannot cover the 256 CUs the K range is split over grid.z, fp32 partial slabs go to a
// caller-provided workspace and a second kernel sums them and applies the epilogue.
#include "common.h"
#include <stdlib.h>

#define GEMM_BN 128
#define GEMM_BK 64
#define GEMM_LD 64  // LDS row = 128 B, 16-B chunk c of row r lives at chunk position c ^ (r & 7): conflict-free for the
                    // ds_read_b128 lane groups of gfx950 ({0-3,12-15,20-27}, ...) and for the 8-lane ds_write_b128 groups

extern __shared__ __attribute__((aligned(16))) unsigned char gemm_dyn_smem[];

template <int TM>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const bf16_t* __restrict__ A, int lda,
                                                        const bf16_t* __restrict__ W, int ldw,
                                                        const bf16_t* __restrict__ bias,
                                                        const bf16_t* residual, int ldr,
                                                        bf16_t* C, int ldc, float* __restrict__ partial,
                                                        int M, int N, int K, int k_per_split, int act, int gm, int gn, int splits) {
  constexpr int BM = 32 * TM;
  constexpr int A_PASSES = BM / 32;
  constexpr int W_PASSES = GEMM_BN / 32;
  constexpr int STAGE = (BM + GEMM_BN) * GEMM_LD;
  bf16_t* smem = reinterpret_cast<bf16_t*>(gemm_dyn_smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // ---- workgroup -> (tile, K-split).  Workgroups are dealt round-robin over the 8 XCDs (private 4 MiB L2 each), so
  // bid % 8 names the XCD group.  Speed only, never correctness:
  //   split-K : the split index varies fastest -> an XCD only ever touches ONE K-slice of A and W (fits its L2)
  //   no split: each XCD owns a contiguous range of n-tiles and walks m fastest -> a W tile is fetched once and
  //             reused by all m-tiles from L2, the (smaller) A panel is what gets re-streamed
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
    if (tn >= gn) return;          // whole workgroup: the grid is padded to 8 * npx * gm
  }
  const int m0 = tm * BM, n0 = tn * GEMM_BN;
  const int lrow = tid >> 3, lchunk = tid & 7;  // loader: 8 threads cover one 128-B row segment
  const int k_begin = split * k_per_split;
  const int k_end = min(K, k_begin + k_per_split);
  const int nk = (k_end - k_begin + GEMM_BK - 1) / GEMM_BK;

  const bf16_t* a_ptr[A_PASSES];
  const bf16_t* w_ptr[W_PASSES];
#pragma unroll
  for (int p = 0; p < A_PASSES; ++p) a_ptr[p] = A + (size_t)min(m0 + p * 32 + lrow, M - 1) * lda + lchunk * 8;
#pragma unroll
  for (int p = 0; p < W_PASSES; ++p) w_ptr[p] = W + (size_t)min(n0 + p * 32 + lrow, N - 1) * ldw + lchunk * 8;

  u32x4_t ra0[A_PASSES], rw0[W_PASSES], ra1[A_PASSES], rw1[W_PASSES];
  bool kin0 = true, kin1 = true;       // per ring slot: does this thread's chunk lie inside [k_begin, k_end)?
  auto load_tile = [&](int kt, u32x4_t (&ra)[A_PASSES], u32x4_t (&rw)[W_PASSES], bool& kin_slot) {
    // UNCONDITIONAL loads from clamped addresses (a predicated load becomes a branch, and hipcc then drains the whole
    // ring with vmcnt(0) instead of a counted wait); out-of-range chunks are zeroed by a select afterwards
    const int kt_c = min(kt, nk - 1);
    const int k0 = k_begin + kt_c * GEMM_BK;
    const bool kin = kt < nk && (k0 + lchunk * 8) < k_end;     // tiles past the end are stored as zeros
    const int koff = (k0 + lchunk * 8) < k_end ? k0 : k_end - 8 - lchunk * 8;
    kin_slot = kin;                    // the zeroing select happens at STORE time so the loads stay in flight
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) ra[p] = *reinterpret_cast<const u32x4_t*>(a_ptr[p] + koff);
#pragma unroll
    for (int p = 0; p < W_PASSES; ++p) rw[p] = *reinterpret_cast<const u32x4_t*>(w_ptr[p] + koff);
  };
  auto store_tile = [&](int buf, const u32x4_t (&ra)[A_PASSES], const u32x4_t (&rw)[W_PASSES], bool kin) {
    bf16_t* As = smem + buf * STAGE;
    bf16_t* Ws = As + BM * GEMM_LD;
    const u32x4_t z = u32x4_t{0, 0, 0, 0};
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) *reinterpret_cast<u32x4_t*>(As + (p * 32 + lrow) * GEMM_LD + ((lchunk ^ (lrow & 7)) * 8)) = kin ? ra[p] : z;
#pragma unroll
    for (int p = 0; p < W_PASSES; ++p) *reinterpret_cast<u32x4_t*>(Ws + (p * 32 + lrow) * GEMM_LD + ((lchunk ^ (lrow & 7)) * 8)) = kin ? rw[p] : z;
  };

  f32x4_t acc[TM][4];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  auto compute = [&](int buf) {
    const bf16_t* As = smem + buf * STAGE;
    const bf16_t* Ws = As + BM * GEMM_LD;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t af[TM], wf[4];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[i] = *reinterpret_cast<const bf16x8_t*>(As + (wm * 16 * TM + i * 16 + fr) * GEMM_LD + (((ks * 4 + fq) ^ (fr & 7)) * 8));
#pragma unroll
      for (int j = 0; j < 4; ++j)
        wf[j] = *reinterpret_cast<const bf16x8_t*>(Ws + (wn * 64 + j * 16 + fr) * GEMM_LD + (((ks * 4 + fq) ^ (fr & 7)) * 8));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
  };

  // prologue: tile 0 -> LDS[0]; tiles 1, 2 in flight in the register ring
  load_tile(0, ra0, rw0, kin0);
  load_tile(1, ra1, rw1, kin1);
  store_tile(0, ra0, rw0, kin0);
  load_tile(2, ra0, rw0, kin0);
  __syncthreads();
  // steady state, unrolled by two so that the ring slots are named registers.  The body is branch-free: tiles past
  // the end of the K range are zeros (select at store time), so an odd tile count just computes one null tile.
  for (int kt = 0; kt < nk; kt += 2) {
    // even step: LDS[0] holds tile kt; ring slot 1 holds tile kt+1, slot 0 holds tile kt+2
    store_tile(1, ra1, rw1, kin1);
    load_tile(kt + 3, ra1, rw1, kin1);
    __builtin_amdgcn_sched_barrier(0);     // keep the prefetch ABOVE the MFMAs: hipcc otherwise sinks the loads below them
    compute(0);
    lds_barrier();
    // odd step: LDS[1] holds tile kt+1; slot 0 holds tile kt+2, slot 1 holds tile kt+3
    store_tile(0, ra0, rw0, kin0);
    load_tile(kt + 4, ra0, rw0, kin0);
    __builtin_amdgcn_sched_barrier(0);
    compute(1);
    lds_barrier();
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
      if (partial) {  // split-K: raw fp32 slab, the epilogue runs in the reduce kernel
        *reinterpret_cast<f32x4_t*>(partial + ((size_t)split * M + m) * N + n) = acc[i][j];
        continue;
      }
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

// ------------------------------------------------------------------------------------------------------------
// LDS-DMA variant (K % 64 == 0): tiles go global -> LDS with `global_load_lds_dwordx4`, never through VGPRs, so
// the ds_write pass (≈80 B/clk/CU, slower than the MFMAs it feeds) disappears and three stages fit a deeper
// prefetch: tile t+2 is in flight while tile t is computed.  The LDS image of a wave-instruction is lane-linear
// (lane i -> base + 16 i), so the XOR swizzle is applied to the per-lane SOURCE address (chunk c = p ^ (row & 7)
// lands at position p) and again on the fragment reads.  Waits are counted by hand (vmcnt(NPT) leaves exactly the
// newest tile in flight) and the barrier is raw: __syncthreads() would drain the DMA queue.
#define GLDS(gp, lp) \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp), (__attribute__((address_space(3))) void*)(lp), 16, 0, 0)

// GEMM_DIAG (compile-time, tools only): 1 = K loop without the DMA issue (LDS reads + MFMA + barrier only),
// 2 = K loop without the compute (DMA + waits + barrier only) -- splits the per-K-step cost into its two halves.
#ifndef GEMM_DIAG
#define GEMM_DIAG 0
#endif
template <int TM, int NS, int WM = 2>
__global__ __launch_bounds__(128 * WM) void gemm_glds_kernel(const bf16_t* __restrict__ A, int lda,
                                                        const bf16_t* __restrict__ W, int ldw,
                                                        const bf16_t* __restrict__ bias,
                                                        const bf16_t* residual, int ldr,
                                                        bf16_t* C, int ldc, float* __restrict__ partial,
                                                        int M, int N, int K, int k_per_split, int act, int gm, int gn, int splits) {
  // 2 WM waves as WM (m) x 2 (n), TM x 4 MFMA tiles each: WM = 2 is the 4-wave workgroup of the (32 TM) x 128 tiles; WM = 4 an
  // 8-wave workgroup on a 256 x 128 tile (TM = 4): 85 flop per staged byte against 64 / 44 for the 128- / 64-row tiles, two
  // waves per SIMD from ONE workgroup per CU
  constexpr int BM = 16 * TM * WM;
  constexpr int A_INST = TM;                    // 1-KiB DMA pieces (8 rows x 128 B) per wave per tile: A (BM / 8 pieces over 2 WM waves)
  constexpr int W_INST = 8 / WM;                // ... and W (16 pieces)
#if GEMM_DIAG == 3                               // DMA-only loop that streams W alone: is the DMA stream bound per byte or per step?
  constexpr int NPT = W_INST;
#else
  constexpr int NPT = A_INST + W_INST;          // DMA instructions per wave per tile
#endif
  constexpr int STAGE_B = (BM + GEMM_BN) * 128; // bytes per stage
  unsigned char* smem = gemm_dyn_smem;

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
  const bool swiglu = act == SVLM_ACT_SWIGLU;     // 64 output columns per tile: gate | up halves of the 128 tile columns
  const int m0 = tm * BM, n0 = tn * (swiglu ? GEMM_BN / 2 : GEMM_BN);
  const int k_begin = split * k_per_split;
  const int k_end = min(K, k_begin + k_per_split);
  const int nk = (k_end - k_begin) / GEMM_BK;    // K % 64 == 0 and k_per_split % 64 == 0

  // per-lane source pointers of this wave's DMA pieces (row = piece*8 + lane/8, position p = lane%8 holds chunk p^(row&7))
  const int rsub = lane >> 3, ppos = lane & 7;
  const bf16_t* a_src[A_INST];
  const bf16_t* w_src[W_INST];
#pragma unroll
  for (int i = 0; i < A_INST; ++i) {
    const int row = (wave * A_INST + i) * 8 + rsub;
    a_src[i] = A + (size_t)min(m0 + row, M - 1) * lda + k_begin + ((ppos ^ (row & 7)) * 8);
  }
#pragma unroll
  for (int i = 0; i < W_INST; ++i) {
    const int row = (wave * W_INST + i) * 8 + rsub;
    // SwiGLU pairing: tile rows 0..63 are gate rows n0.., tile rows 64..127 the matching up rows N + n0.. (N = output columns)
    const int wrow = swiglu ? (row < 64 ? min(n0 + row, N - 1) : N + min(n0 + row - 64, N - 1)) : min(n0 + row, N - 1);
    w_src[i] = W + (size_t)wrow * ldw + k_begin + ((ppos ^ (row & 7)) * 8);
  }
  auto issue = [&](int kt, int stage) {
    unsigned char* sa = smem + stage * STAGE_B;
    unsigned char* sw = sa + BM * 128;
#if GEMM_DIAG != 3
#pragma unroll
    for (int i = 0; i < A_INST; ++i) GLDS(a_src[i] + kt * GEMM_BK, sa + (wave * A_INST + i) * 1024);
#endif
#pragma unroll
    for (int i = 0; i < W_INST; ++i) GLDS(w_src[i] + kt * GEMM_BK, sw + (wave * W_INST + i) * 1024);
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
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t af[TM], wf[4];
      const int pos = ((ks * 4 + fq) ^ (fr & 7)) * 16;      // rows of a fragment differ by multiples of 16 -> row&7 == fr&7
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(sa + (wm * 16 * TM + i * 16 + fr) * 128 + pos);
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(sw + (wn * 64 + j * 16 + fr) * 128 + pos);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
  };

  // NS-stage ring, NS-1 tiles in flight.  Measured with the GEMM_DIAG builds (tools/gemm_shapes.py, ViT qkv 1024x3840x1280): whole
  // kernel 24.6 us, K loop WITHOUT any MFMA 24.1 us, WITHOUT the DMA 18.4 us -- the loop is bound by the L2/fabric -> LDS stream
  // (157 MB per GEMM at 128x128 tiles), not by MFMA or LDS reads.  A 4-stage ring (3 tiles in flight) made it SLOWER (31.0 us):
  // more bytes in flight do not help a bandwidth-bound stream and cost the second resident workgroup at BM = 64.  NS = 3.
  constexpr int AHEAD = NS - 1;
#pragma unroll
  for (int t = 0; t < AHEAD; ++t)
    if (t < nk) issue(t, t);
#if GEMM_DIAG == 1
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_barrier" ::: "memory");
    compute(kt % 2);
  }
#else
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed when at most the NEWER tiles' pieces are outstanding (min(AHEAD - 1, nk - 1 - kt) tiles of NPT pieces);
    // after the barrier everyone has passed compute(kt-1), whose stage (kt + AHEAD) % NS may be refilled
    const int newer = min(AHEAD - 1, nk - 1 - kt);
    if (newer >= 3) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(3 * NPT) : "memory");
    else if (newer == 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * NPT) : "memory");
    else if (newer == 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (kt + AHEAD < nk) issue(kt + AHEAD, (kt + AHEAD) % NS);
#if GEMM_DIAG != 2 && GEMM_DIAG != 3
    compute(kt % NS);
#endif
  }
#endif

  if (swiglu) {
    // the up half (waves wn = 1) hands its bf16-rounded values to the gate half through the staging LDS, which is free now
    float* U = reinterpret_cast<float*>(smem);                       // [BM][64]
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (wn == 1) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x4_t u;
          const int nb = min(n0 + j * 16 + fq * 4, N - 4);
          float ub[4] = {0.f, 0.f, 0.f, 0.f};
          if (bias) {                                        // up bias = second half of the [gate; up] bias vector
            const u32x2_t bv = *reinterpret_cast<const u32x2_t*>(bias + N + nb);
            ub[0] = lo_bf(bv[0]); ub[1] = hi_bf(bv[0]); ub[2] = lo_bf(bv[1]); ub[3] = hi_bf(bv[1]);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) u[r] = rbf(acc[i][j][r] + ub[r]);
          *reinterpret_cast<f32x4_t*>(U + (wm * 16 * TM + i * 16 + fr) * 64 + j * 16 + fq * 4) = u;
        }
    }
    __syncthreads();
    if (wn == 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * 16 * TM + i * 16 + fr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = n0 + j * 16 + fq * 4;
          if (m >= M || n >= N) continue;
          const f32x4_t u = *reinterpret_cast<const f32x4_t*>(U + (wm * 16 * TM + i * 16 + fr) * 64 + j * 16 + fq * 4);
          float h[4], gb[4] = {0.f, 0.f, 0.f, 0.f};
          if (bias) {
            const u32x2_t bv = *reinterpret_cast<const u32x2_t*>(bias + n);
            gb[0] = lo_bf(bv[0]); gb[1] = hi_bf(bv[0]); gb[2] = lo_bf(bv[1]); gb[3] = hi_bf(bv[1]);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) h[r] = apply_act(rbf(acc[i][j][r] + gb[r]), SVLM_ACT_SILU) * u[r];
          u32x2_t o;
          o[0] = pack2(h[0], h[1]);
          o[1] = pack2(h[2], h[3]);
          *reinterpret_cast<u32x2_t*>(C + (size_t)m * ldc + n) = o;
        }
      }
    }
    return;
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
      if (partial) {
        *reinterpret_cast<f32x4_t*>(partial + ((size_t)split * M + m) * N + n) = acc[i][j];
        continue;
      }
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

// C = epi(sum_z partial[z]) -- one thread per 4 consecutive n
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const float* __restrict__ partial, int splits,
                                                                 const bf16_t* __restrict__ bias, const bf16_t* residual, int ldr,
                                                                 bf16_t* C, int ldc, int M, int N, int act) {
  const int n4 = N / 4;
  const size_t total = (size_t)M * n4;
  const size_t slab = (size_t)M * N;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int m = (int)(i / n4), n = (int)(i % n4) * 4;
    f32x4_t s = *reinterpret_cast<const f32x4_t*>(partial + (size_t)m * N + n);
    for (int z = 1; z < splits; ++z) {
      const f32x4_t p = *reinterpret_cast<const f32x4_t*>(partial + z * slab + (size_t)m * N + n);
      s[0] += p[0]; s[1] += p[1]; s[2] += p[2]; s[3] += p[3];
    }
    float y[4] = {s[0], s[1], s[2], s[3]};
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

// Split-K reduce fused with the RMSNorm that follows a residual-stream GEMM (o_proj -> post_attention_layernorm, down_proj ->
// the next layer's input_layernorm; qwen2/language_forward.py:183,195-200): one workgroup per row keeps the reduced bf16 row in
// registers, so the row is normalised without a second launch and without re-reading it.  Same roundings as
// gemm_splitk_reduce_kernel followed by rmsnorm_kernel.
#define RN_IT 2       // 2048-column passes of the workgroup: N <= 4096
__global__ __launch_bounds__(256) void gemm_splitk_reduce_norm_kernel(const float* __restrict__ partial, int splits,
                                                                      const bf16_t* __restrict__ bias, const bf16_t* residual, int ldr,
                                                                      bf16_t* C, int ldc, int M, int N, int act,
                                                                      const bf16_t* __restrict__ norm_w, const bf16_t* __restrict__ norm_b,
                                                                      float eps, bf16_t* __restrict__ XN, int ldxn,
                                                                      unsigned char* __restrict__ XN8, int ldxn8, float* __restrict__ xn_scale) {
  // XN8 != nullptr (fp8 ViT tower): the normalised row also leaves as e4m3 + one fp32 scale per row, the recipe of
  // quant_rows_fp8_kernel (gemm_fp8.hip) applied to the bf16 values XN holds -- the next GEMM's operand without a quantiser launch.
  // one workgroup per row, 8 columns per thread and pass: all slab loads of a thread are independent and issued together.
  // norm_b == nullptr: RMSNorm (Qwen2RMSNorm); else LayerNorm with bias (the ViT's norm1 / norm2, one rounding at the end).
  const int m = blockIdx.x, tid = threadIdx.x;
  const size_t slab = (size_t)M * N;
  float y[RN_IT][8];
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < RN_IT; ++it) {
    const int n = it * 2048 + tid * 8;
    if (n < N) {
      float acc[8];
      const float* p0 = partial + (size_t)m * N + n;
      f32x4_t a = *reinterpret_cast<const f32x4_t*>(p0), b = *reinterpret_cast<const f32x4_t*>(p0 + 4);
      for (int z = 1; z < splits; ++z) {
        const f32x4_t pa = *reinterpret_cast<const f32x4_t*>(p0 + z * slab), pb = *reinterpret_cast<const f32x4_t*>(p0 + z * slab + 4);
        a += pa; b += pb;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc[i] = a[i]; acc[4 + i] = b[i]; }
      if (bias) {
        float bf[8];
        unpack8(*reinterpret_cast<const u32x4_t*>(bias + n), bf);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += bf[i];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = apply_act(rbf(acc[i]), act);
      if (residual) {
        float rf[8];
        unpack8(*reinterpret_cast<const u32x4_t*>(residual + (size_t)m * ldr + n), rf);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += rf[i];
      }
      const u32x4_t o = pack8(acc);
      *reinterpret_cast<u32x4_t*>(C + (size_t)m * ldc + n) = o;
      unpack8(o, y[it]);                                   // the bf16 values the norm kernel would read back
#pragma unroll
      for (int i = 0; i < 8; ++i) ss += y[it][i] * y[it][i];
    }
  }
  __shared__ float red[4], red2[4];
  if (norm_b != nullptr) {                  // LayerNorm: mean first, then the centred second moment (as layernorm_kernel does)
    float s1 = 0.f;
#pragma unroll
    for (int it = 0; it < RN_IT; ++it)
      if (it * 2048 + tid * 8 < N) {
#pragma unroll
        for (int i = 0; i < 8; ++i) s1 += y[it][i];
      }
    s1 = wave_sum(s1);
    if ((tid & 63) == 0) red[tid >> 6] = s1;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)N;
    float s2 = 0.f;
#pragma unroll
    for (int it = 0; it < RN_IT; ++it)
      if (it * 2048 + tid * 8 < N) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = y[it][i] - mean; s2 += d * d; }
      }
    s2 = wave_sum(s2);
    if ((tid & 63) == 0) red2[tid >> 6] = s2;
    __syncthreads();
    const float rstd = rsqrtf((red2[0] + red2[1] + red2[2] + red2[3]) / (float)N + eps);
#pragma unroll
    for (int it = 0; it < RN_IT; ++it) {
      const int n = it * 2048 + tid * 8;
      if (n < N) {
        float g[8], h[8], f[8];
        unpack8(*reinterpret_cast<const u32x4_t*>(norm_w + n), g);
        unpack8(*reinterpret_cast<const u32x4_t*>(norm_b + n), h);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = (y[it][i] - mean) * rstd * g[i] + h[i];
        const u32x4_t o = pack8(f);
        *reinterpret_cast<u32x4_t*>(XN + (size_t)m * ldxn + n) = o;
        unpack8(o, y[it]);                                 // keep the bf16 values for the quantiser below
      }
    }
    if (XN8 != nullptr) {
      __shared__ float redm[4];
      float mx = 0.f;
#pragma unroll
      for (int it = 0; it < RN_IT; ++it)
        if (it * 2048 + tid * 8 < N) {
#pragma unroll
          for (int i = 0; i < 8; ++i) mx = fmaxf(mx, fabsf(y[it][i]));
        }
      mx = wave_max(mx);
      if ((tid & 63) == 0) redm[tid >> 6] = mx;
      __syncthreads();
      mx = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
      const float sc = mx > 0.f ? mx / 448.0f : 1.0f;
      if (tid == 0) xn_scale[m] = sc;
#pragma unroll
      for (int it = 0; it < RN_IT; ++it) {
        const int n = it * 2048 + tid * 8;
        if (n < N) {
          u32x2_t o;
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            int w = 0;
            w = __builtin_amdgcn_cvt_pk_fp8_f32(y[it][4 * hh] / sc, y[it][4 * hh + 1] / sc, w, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(y[it][4 * hh + 2] / sc, y[it][4 * hh + 3] / sc, w, true);
            o[hh] = (unsigned)w;
          }
          *reinterpret_cast<u32x2_t*>(XN8 + (size_t)m * ldxn8 + n) = o;
        }
      }
    }
    return;
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  const float r = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)N + eps);
#pragma unroll
  for (int it = 0; it < RN_IT; ++it) {
    const int n = it * 2048 + tid * 8;
    if (n < N) {
      float g[8], f[8];
      unpack8(*reinterpret_cast<const u32x4_t*>(norm_w + n), g);
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] = g[i] * rbf(y[it][i] * r);
      *reinterpret_cast<u32x4_t*>(XN + (size_t)m * ldxn + n) = pack8(f);
    }
  }
}

extern "C" int svlm_rmsnorm(const void* x, const void* w, void* y, int rows, int cols, float eps, void* stream);
extern "C" int svlm_quant_rows_fp8(const void* x, int ldx, void* q, int ldq, float* scale, int rows, int cols, void* stream);
extern "C" int svlm_layernorm(const void* x, const void* w, const void* b, void* y, int rows, int cols, float eps, void* stream);

static int gemm_impl(const void* A, int lda, const void* W, int ldw, const void* bias, const void* residual, int ldr,
                     void* C, int ldc, int M, int N, int K, int act, void* ws, long long ws_bytes, void* stream,
                     const void* norm_w, const void* norm_b, float eps, void* XN, int ldxn);
int svlm_gemm_reduce_launch(const float* partial, int splits, const void* bias, const void* residual, int ldr, void* C, int ldc, int M, int N,
                            int act, const void* norm_w, const void* norm_b, float eps, void* XN, int ldxn, void* stream, void* XN8 = nullptr,
                            int ldxn8 = 0, float* xn_scale = nullptr);

extern "C" int svlm_gemm_bf16(const void* A, int lda, const void* W, int ldw, const void* bias, const void* residual, int ldr,
                              void* C, int ldc, int M, int N, int K, int act, void* ws, long long ws_bytes, void* stream) {
  return gemm_impl(A, lda, W, ldw, bias, residual, ldr, C, ldc, M, N, K, act, ws, ws_bytes, stream, nullptr, nullptr, 0.f, nullptr, 0);
}

extern "C" int svlm_gemm_bf16_norm(const void* A, int lda, const void* W, int ldw, const void* bias, const void* residual, int ldr,
                                   void* C, int ldc, int M, int N, int K, int act, void* ws, long long ws_bytes,
                                   const void* norm_w, const void* norm_b, float eps, void* XN, int ldxn, void* stream) {
  SVLM_CHECK_ARG(norm_w != nullptr && XN != nullptr && N % 8 == 0 && ldc % 8 == 0 && ldxn % 8 == 0 && ldxn >= N && eps > 0.f,
                 "svlm_gemm_bf16_norm: needs a norm weight, an output with 16-B aligned rows and N %% 8 == 0 (N=%d ldc=%d ldxn=%d)", N, ldc, ldxn);
  return gemm_impl(A, lda, W, ldw, bias, residual, ldr, C, ldc, M, N, K, act, ws, ws_bytes, stream, norm_w, norm_b, eps, XN, ldxn);
}

static int gemm_impl(const void* A, int lda, const void* W, int ldw, const void* bias, const void* residual, int ldr,
                     void* C, int ldc, int M, int N, int K, int act, void* ws, long long ws_bytes, void* stream,
                     const void* norm_w, const void* norm_b, float eps, void* XN, int ldxn) {
  SVLM_CHECK_ARG(M >= 0 && N > 0 && K > 0, "svlm_gemm_bf16: bad shape M=%d N=%d K=%d", M, N, K);
  SVLM_CHECK_ARG(K % 8 == 0 && N % 4 == 0, "svlm_gemm_bf16: K=%d must be a multiple of 8 and N=%d of 4", K, N);
  SVLM_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0 && (!residual || ldr % 4 == 0),
                 "svlm_gemm_bf16: leading dims must keep 16-B row alignment (lda=%d ldw=%d ldc=%d ldr=%d)", lda, ldw, ldc, ldr);
  SVLM_CHECK_ARG(lda >= K && ldw >= K && ldc >= N, "svlm_gemm_bf16: leading dim smaller than row length");
  SVLM_CHECK_ARG(act >= 0 && act <= 4, "svlm_gemm_bf16: unknown activation %d", act);
  const bool swiglu = act == SVLM_ACT_SWIGLU;
  SVLM_CHECK_ARG(!swiglu || (residual == nullptr && norm_w == nullptr && K % GEMM_BK == 0 && N % 8 == 0),
                 "svlm_gemm_bf16: SVLM_ACT_SWIGLU takes W = [gate; up] (bias = [gate; up] too) without residual, K %% 64 == 0 and N %% 8 == 0 (N=%d K=%d)", N, K);
  if (M == 0) return SVLM_OK;
  hipStream_t st = (hipStream_t)stream;
  const int gn = swiglu ? (N + GEMM_BN / 2 - 1) / (GEMM_BN / 2) : (N + GEMM_BN - 1) / GEMM_BN;      // SwiGLU: 64 output columns per tile
  // ---- tile height and K-split from a small cost model.  The path's GEMMs are one or two "rounds" of resident
  // workgroups, so quantisation decides: cost = rounds x K-steps per workgroup x step cost (+ slab traffic).
  // Resident workgroups per CU: LDS 2 x (BM+128) x 128 B -> 48 KB (BM=64): 3, 64 KB (BM=128): 2.
  int best_bm = 64, best_splits = 1;
  double best_cost = 1e30;
  for (int bm_c = 64; bm_c <= 128; bm_c += 64) {
    if (bm_c == 128 && M <= 64) continue;
    const long long tiles_c = (long long)((M + bm_c - 1) / bm_c) * gn;
    const long long cap = 256LL * (bm_c == 64 ? 3 : 2);
    const int max_s = ws != nullptr ? (K >= 1024 ? (K / 512 < 16 ? K / 512 : 16) : 1) : 1;
    for (int sp = 1; sp <= max_s; ++sp) {
      if (sp > 1 && (long long)sp * M * N * 4 > ws_bytes) break;
      const long long wgs = tiles_c * sp;
      const double rounds = (double)((wgs + cap - 1) / cap);
      const double ksteps = (double)((K + sp - 1) / sp + GEMM_BK - 1) / GEMM_BK + 3.0;           // + prologue/epilogue
      const double step_us = bm_c == 64 ? 0.9 : 1.5;                                              // measured per K-step
      const double slab_us = sp > 1 ? 2.0 * sp * M * N * 4.0 / 3e12 * 1e6 + 5.0 : 0.0;            // slab write+read + reduce launch
      const double cost = rounds * ksteps * step_us + slab_us;
      if (cost < best_cost) { best_cost = cost; best_bm = bm_c; best_splits = sp; }
    }
  }
  // ---- tuned plans: shapes of the supported models measured on MI355X (tools/gemm_tune_table.py, cold weights).  The cost
  // model above mis-ranks some of them (it knows nothing of co-resident workgroups sharing a CU's LDS-DMA stream); an entry
  // applies to its exact (N, K) and row bucket ceil(M/64) +- 1.
  struct GemmPlan { int mb, N, K, bm, splits, variant; }; // variant of the 128-row tile: 1 = 8 waves (one workgroup per CU), 2 = 2-stage ring (two per CU)
  static const GemmPlan kTunedPlans[] = {
      {5, 1536, 8960, 64, 4},      // Qwen2-VL-2B down_proj, prefill: 24.6 us vs 28.3 us for the cost model's choice
      {5, 4608, 3584, 64, 2},      // 7B qkv: 26.5 vs 29.9
      {5, 3584, 3584, 64, 3},      // 7B o_proj: 22.9 vs 27.8
      {5, 3584, 18944, 320, 8},    // 7B down_proj: 66.2 (all rows in one 320-row tile) vs 123.6
      {10, 3584, 18944, 128, 3},   // 7B down_proj at two temporal grids per chunk (M ~ 590: 4 frames per second): 145 vs 248 us
      {10, 18944, 3584, 128, 1, 2},// 7B gate/up (SwiGLU pairing) at M ~ 590: 206 vs 253 us on the 64-row tiles
      {5, 2560, 2048, 64, 3},      // Qwen2.5-VL-3B qkv: 16.3 vs 18.1
      {5, 2048, 2048, 64, 3},      // 3B o_proj: 14.4 vs 15.6
      {5, 2048, 11008, 64, 4},     // 3B down_proj: 33.3 vs 43.2
      {16, 3840, 1280, 128, 1, 1}, // ViT qkv (1024 patches): 240 tiles = one 8-wave workgroup per CU, 21.2 vs 26.6 us on 64-row tiles
      {16, 1280, 5120, 128, 3},    // ViT fc2 (1024 patches): 31.3 vs 35.0
      {16, 1280, 1280, 128, 3},    // ViT proj: 16.7 vs 15.4 unsplit, but the split-K reduce then carries the LayerNorm / RMSNorm that follows
                                   // (svlm_gemm_bf16_norm), which costs a 5 us launch of its own otherwise
      {16, 1280, 3424, 128, 3},    // Qwen2.5 ViT down_proj: 28.7 vs 31.2
      {4, 5120, 5120, 128, 3},     // merger mlp.0 (256 merged tokens): 30.1 vs 37.8
      {4, 1536, 5120, 64, 5},      // merger mlp.2 -> 2B: 17.0 vs 21.3
      {4, 3584, 5120, 128, 4},     // merger mlp.2 -> 7B: 25.1 vs 36.1
      {4, 2048, 5120, 64, 4},      // merger mlp.2 -> 3B: 19.7 vs 21.5
  };
  bool plan_w8 = false, plan_ns2 = false;
  if (svlm_env("SVLM_GEMM_NO_TABLE") == nullptr) {
    const int mb = (M + 63) / 64;
    for (const GemmPlan& p : kTunedPlans) {
      if (p.N == N && p.K == K && mb >= p.mb - 1 && mb <= p.mb + 1 && (p.splits == 1 || (ws != nullptr && (long long)p.splits * M * N * 4 <= ws_bytes))) {
        best_bm = p.bm;
        best_splits = p.splits;
        plan_w8 = p.variant == 1;
        plan_ns2 = p.variant == 2;
        break;
      }
    }
  }
  // large-M, long-K GEMMs (the dense prefill's 4096-row passes on the 7B shapes): 128-row tiles on the 2-stage ring, two workgroups
  // per CU -- 610-740 -> 710-820 TFLOP/s against the 64-row tiles (tools/gemm_big.py); K = 1280 (the ViT's batches) gains nothing
  // and 256 x 128 tiles (8 waves, one workgroup per CU) where their grid fills whole rounds of the 256 CUs: 7B down_proj 821 -> 927,
  // gate/up 820 -> 886 TFLOP/s; 7B qkv (576 tiles = 2.25 rounds) stays on the 128-row tiles
  if (M >= 2048 && K >= 2048 && best_splits == 1 && svlm_env("SVLM_GEMM_NO_T128") == nullptr) {
    const long long t256 = (long long)((M + 255) / 256) * gn;
    const long long rounds = (t256 + 255) / 256;
    best_bm = (t256 * 100 >= rounds * 256 * 85) ? 256 : 128;
  }
  if (const char* force = svlm_env("SVLM_GEMM_BM")) {       // tuning aid
    const int fb = atoi(force);
    best_bm = (fb == 192 || fb == 320 || fb == 256) ? fb : (fb == 128 && M > 64 ? 128 : 64);
    plan_w8 = plan_ns2 = false;
    if (const char* fs = svlm_env("SVLM_GEMM_SPLITS")) best_splits = atoi(fs) > 0 ? atoi(fs) : 1; else best_splits = 1;
    if ((long long)best_splits * M * N * 4 > ws_bytes || ws == nullptr || K < 1024) best_splits = 1;
  }
  if (swiglu) {                       // one K pass (the pairing happens in the epilogue), register budget of the 64/128-row tiles
    best_splits = 1;
    if (best_bm > 128 && best_bm != 256) best_bm = 128;
  }
  const bool small = best_bm == 64;
  const int bm = best_bm;
  const int gm = (M + bm - 1) / bm;
  int splits = best_splits;
  int kps = K;
  if (splits > 1) {
    kps = ((K + splits - 1) / splits + GEMM_BK - 1) / GEMM_BK * GEMM_BK;
    splits = (K + kps - 1) / kps;
  }
  float* partial = splits > 1 ? (float*)ws : nullptr;
  const int nblocks = splits > 1 ? gn * gm * splits : 8 * ((gn + 7) / 8) * gm;
  dim3 grid(nblocks, 1, 1);
  // LDS: two stages of (BM + 128) rows x 144 B -- above the 64 KB default for TM = 4, so opt in once
  constexpr int LDS2 = 2 * (64 + GEMM_BN) * GEMM_LD * 2, LDS4 = 2 * (128 + GEMM_BN) * GEMM_LD * 2;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS2);
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS4);
    if (e1 != hipSuccess || e2 != hipSuccess) {
      svlm_set_error("svlm_gemm_bf16: cannot reserve %d B of LDS: %s", LDS4, hipGetErrorString(e1 != hipSuccess ? e1 : e2));
      return SVLM_ELAUNCH;
    }
    attr_done = true;
  }
  const bool dma = (K % GEMM_BK == 0) && svlm_env("SVLM_GEMM_NO_DMA") == nullptr;
  if (!dma && (bm == 192 || bm == 320 || bm == 256 || swiglu)) {
    svlm_set_error("svlm_gemm_bf16: tall tiles and SVLM_ACT_SWIGLU run on the LDS-DMA kernel only (K %% 64 == 0, SVLM_GEMM_NO_DMA unset)");
    return SVLM_EINVAL;
  }
  if (dma) {
    // ring depth: 3 stages (72 KB at BM = 64: two workgroups per CU; 96 KB at BM = 128); 4 measured slower, see the kernel
    constexpr int NS2 = 3, NS4 = 3;
    constexpr int DLDS2 = NS2 * (64 + GEMM_BN) * 128, DLDS4 = NS4 * (128 + GEMM_BN) * 128;
    static bool dma_attr_done = false;
    if (!dma_attr_done) {
      hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds_kernel<2, NS2>), hipFuncAttributeMaxDynamicSharedMemorySize, DLDS2);
      hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds_kernel<4, NS4>), hipFuncAttributeMaxDynamicSharedMemorySize, DLDS4);
      if (e1 != hipSuccess || e2 != hipSuccess) {
        svlm_set_error("svlm_gemm_bf16: cannot reserve %d B of LDS: %s", DLDS4, hipGetErrorString(e1 != hipSuccess ? e1 : e2));
        return SVLM_ELAUNCH;
      }
      dma_attr_done = true;
    }
    const bool tall_unsplit_ns2 = M >= 2048 || plan_ns2 || svlm_env("SVLM_GEMM_T128NS2") != nullptr;
    static const bool w8_128 = svlm_env("SVLM_GEMM_W8") != nullptr;       // tuning aid: force the 8-wave 128 x 128 tile
    if (bm == 128 && splits == 1 && !swiglu && (w8_128 || plan_w8)) {
      constexpr int DL = 3 * (128 + GEMM_BN) * 128;
      static bool w8s_done = false;
      if (!w8s_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds_kernel<2, 3, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, DL);
        w8s_done = true;
      }
      gemm_glds_kernel<2, 3, 4><<<grid, 512, DL, st>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, (const bf16_t*)bias, (const bf16_t*)residual, ldr,
                                                  (bf16_t*)C, ldc, partial, M, N, K, kps, act, gm, gn, splits);
    } else if (bm == 256) {
      constexpr int NS8 = 3, DLDS8 = NS8 * (256 + GEMM_BN) * 128;
      static bool w8_done = false;
      if (!w8_done) {
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds_kernel<4, NS8, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, DLDS8);
        if (e1 != hipSuccess) {
          svlm_set_error("svlm_gemm_bf16: cannot reserve %d B of LDS: %s", DLDS8, hipGetErrorString(e1));
          return SVLM_ELAUNCH;
        }
        w8_done = true;
      }
      gemm_glds_kernel<4, NS8, 4><<<grid, 512, DLDS8, st>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, (const bf16_t*)bias, (const bf16_t*)residual, ldr,
                                                    (bf16_t*)C, ldc, partial, M, N, K, kps, act, gm, gn, splits);
    } else if (bm == 192 || bm == 320) {
      // tall tiles for skinny-M weight-streaming GEMMs (prefill: M ~ 290): 2 x 192 rows, or ALL rows in one 320-row tile so that
      // W is streamed exactly once; 320 rows leave LDS for a 2-stage ring only
      constexpr int DLDS6 = 3 * (192 + GEMM_BN) * 128, DLDS10 = 2 * (320 + GEMM_BN) * 128;
      static bool tall_attr_done = false;
      if (!tall_attr_done) {
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds_kernel<6, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, DLDS6);
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds_kernel<10, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, DLDS10);
        if (e1 != hipSuccess || e2 != hipSuccess) {
          svlm_set_error("svlm_gemm_bf16: cannot reserve %d B of LDS: %s", DLDS10, hipGetErrorString(e1 != hipSuccess ? e1 : e2));
          return SVLM_ELAUNCH;
        }
        tall_attr_done = true;
      }
      if (bm == 192)
        gemm_glds_kernel<6, 3><<<grid, 256, DLDS6, st>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, (const bf16_t*)bias, (const bf16_t*)residual, ldr,
                                                 (bf16_t*)C, ldc, partial, M, N, K, kps, act, gm, gn, splits);
      else
        gemm_glds_kernel<10, 2><<<grid, 256, DLDS10, st>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, (const bf16_t*)bias, (const bf16_t*)residual, ldr,
                                                  (bf16_t*)C, ldc, partial, M, N, K, kps, act, gm, gn, splits);
    } else if (small && splits == 1 && svlm_env("SVLM_GEMM_NS3") == nullptr) {
      // un-split 64-row tiles on a 2-stage ring: 48 KB of LDS, THREE workgroups per CU -- one tile in flight each, the others' MFMAs
      // cover its wait.  Measured against the 3-stage ring (72 KB, two per CU; tools/gemm_shapes.py shapes, MI355X): ViT fc1
      // 31.5 -> 26.6 us, prefill gate/up 36.5 -> 31.2 us, ViT qkv 21.4 -> 20.9 us; split-K shapes are unchanged and keep 3 stages.
      constexpr int DLDS2S = 2 * (64 + GEMM_BN) * 128;
      static bool ns2_done = false;
      if (!ns2_done) {
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds_kernel<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, DLDS2S);
        if (e1 != hipSuccess) {
          svlm_set_error("svlm_gemm_bf16: cannot reserve %d B of LDS: %s", DLDS2S, hipGetErrorString(e1));
          return SVLM_ELAUNCH;
        }
        ns2_done = true;
      }
      gemm_glds_kernel<2, 2><<<grid, 256, DLDS2S, st>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, (const bf16_t*)bias, (const bf16_t*)residual, ldr,
                                               (bf16_t*)C, ldc, partial, M, N, K, kps, act, gm, gn, splits);
    } else if (!small && splits == 1 && tall_unsplit_ns2) {
      // un-split 128-row tiles on a 2-stage ring: 64 KB of LDS, TWO workgroups per CU (large-M GEMMs: the dense prefill's 4096-row
      // passes, the ViT's 8-grid batches)
      constexpr int DLDS4S = 2 * (128 + GEMM_BN) * 128;
      static bool ns2t_done = false;
      if (!ns2t_done) {
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds_kernel<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, DLDS4S);
        if (e1 != hipSuccess) {
          svlm_set_error("svlm_gemm_bf16: cannot reserve %d B of LDS: %s", DLDS4S, hipGetErrorString(e1));
          return SVLM_ELAUNCH;
        }
        ns2t_done = true;
      }
      gemm_glds_kernel<4, 2><<<grid, 256, DLDS4S, st>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, (const bf16_t*)bias, (const bf16_t*)residual, ldr,
                                               (bf16_t*)C, ldc, partial, M, N, K, kps, act, gm, gn, splits);
    } else if (small) {
      gemm_glds_kernel<2, NS2><<<grid, 256, DLDS2, st>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, (const bf16_t*)bias, (const bf16_t*)residual, ldr,
                                               (bf16_t*)C, ldc, partial, M, N, K, kps, act, gm, gn, splits);
    } else {
      gemm_glds_kernel<4, NS4><<<grid, 256, DLDS4, st>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, (const bf16_t*)bias, (const bf16_t*)residual, ldr,
                                               (bf16_t*)C, ldc, partial, M, N, K, kps, act, gm, gn, splits);
    }
  } else if (small) {
    gemm_bf16_kernel<2><<<grid, 256, LDS2, st>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, (const bf16_t*)bias, (const bf16_t*)residual, ldr,
                                             (bf16_t*)C, ldc, partial, M, N, K, kps, act, gm, gn, splits);
  } else {
    gemm_bf16_kernel<4><<<grid, 256, LDS4, st>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, (const bf16_t*)bias, (const bf16_t*)residual, ldr,
                                             (bf16_t*)C, ldc, partial, M, N, K, kps, act, gm, gn, splits);
  }
  int rc = svlm_check_launch("svlm_gemm_bf16");
  if (rc) return rc;
  return svlm_gemm_reduce_launch(partial, splits, bias, residual, ldr, C, ldc, M, N, act, norm_w, norm_b, eps, XN, ldxn, stream);
}

// Tail shared by the bf16 and fp8 GEMMs: split-K reduce (+ epilogue), with the norm of the output row folded in when there is one.
int svlm_gemm_reduce_launch(const float* partial, int splits, const void* bias, const void* residual, int ldr, void* C, int ldc, int M, int N,
                            int act, const void* norm_w, const void* norm_b, float eps, void* XN, int ldxn, void* stream, void* XN8, int ldxn8,
                            float* xn_scale) {
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (splits > 1 && norm_w != nullptr && N <= 2048 * RN_IT) {      // reduce + RMSNorm of the reduced row in one launch
    gemm_splitk_reduce_norm_kernel<<<M, 256, 0, st>>>(partial, splits, (const bf16_t*)bias, (const bf16_t*)residual, ldr, (bf16_t*)C,
                                                                ldc, M, N, act, (const bf16_t*)norm_w, (const bf16_t*)norm_b, eps, (bf16_t*)XN, ldxn,
                                                                norm_b ? (unsigned char*)XN8 : nullptr, ldxn8, xn_scale);
    rc = svlm_check_launch("svlm_gemm_bf16_norm(split-K reduce + norm)");
    if (rc || XN8 == nullptr || norm_b != nullptr) return rc;
    return svlm_quant_rows_fp8(XN, ldxn, XN8, ldxn8, xn_scale, M, N, stream);          // RMSNorm rows: quantised by the stand-alone kernel
  }
  if (splits > 1) {
    const size_t total = (size_t)M * (N / 4);
    int rg = (int)((total + 255) / 256);
    rg = rg > 2048 ? 2048 : rg;
    gemm_splitk_reduce_kernel<<<rg, 256, 0, st>>>(partial, splits, (const bf16_t*)bias, (const bf16_t*)residual, ldr, (bf16_t*)C, ldc, M, N, act);
    rc = svlm_check_launch("svlm_gemm_bf16(split-K reduce)");
    if (rc) return rc;
  }
  if (norm_w == nullptr) return SVLM_OK;
  SVLM_CHECK_ARG(ldc == N && ldxn == N, "svlm_gemm_bf16_norm: the unfused norm needs contiguous rows (ldc=%d ldxn=%d N=%d)", ldc, ldxn, N);
  rc = norm_b ? svlm_layernorm(C, norm_w, norm_b, XN, M, N, eps, stream) : svlm_rmsnorm(C, norm_w, XN, M, N, eps, stream);
  if (rc || XN8 == nullptr) return rc;
  return svlm_quant_rows_fp8(XN, ldxn, XN8, ldxn8, xn_scale, M, N, stream);
}
