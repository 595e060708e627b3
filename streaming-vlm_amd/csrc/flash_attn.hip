// Multi-query flash attention on MFMA (v_mfma_f32_16x16x32_bf16) for the two GEMM-shaped
// attention sites of the path:
//   * LLM prefill, causal, GQA, RoPE-on-load from the slot-mapped KV pool
//     (qwen2/language_forward.py:66-166; T ~ 275 queries x L ~ 2.3k keys per chunk)
//   * ViT block-diagonal attention, one sequence per temporal grid, non-causal, d = 80
//     (qwen2/vision_forward.py:6-34, flash_attn_varlen_func with cu_seqlens = [0, h*w, ...])
// Numerics follow flash-attn: fp32 scores and online softmax, P rounded to bf16 before P.V,
// fp32 accumulation, one final division and rounding.
//
// Prefill runs in two launches per layer (three with key splits): `rope_gather_kernel` applies the shrink-mode post-cache
// M-RoPE (bf16 x*cos + rotate_half(x)*sin, qwen2/language_forward.py:9-64) ONCE to every cached key
// while gathering K and V from their slots into logical order, and rotates the T query rows; the
// attention kernel then streams dense tiles.  (The reference rotates all keys in every layer too,
// and additionally materialises repeat_kv.)
//
// Two attention kernels share the products and the numerics:
//   flash_attn_kernel        (ViT, d = 80 or 128; and the prefill behind SVLM_PREFILL_NO_DMA) -- K / V through buffer descriptors
//                            into a two-slot register ring and padded LDS rows, online softmax with the exact running maximum
//   prefill_attn_dma_kernel  (LLM prefill, d = 128) -- K / V by LDS-DMA into XOR-swizzled unpadded rows, one or two query blocks per
//                            wave, lazy softmax reference, one or two tiles per barrier (further down)
// A workgroup = 4 waves x 16 (x QB) queries of one head.  Both products put the LDS-staged
// operand in the MFMA "A" slot and keep the other in registers:
//   S^T[key][q] = K[key][:] . Q[q][:]      (A = K rows from LDS,   B = Q fragment, loop-invariant)
//   O^T[d][q]   = V^T[d][key] . P^T[key][q] (A = V^T via ds_read_b64_tr_b16 of row-major V in LDS, B = P, straight from the S^T
//                                            accumulators: lane = query column in both products,
//                                            so softmax stats, P and O never leave the lane)
// The k-slot <-> key mapping of the second product is the permutation that makes the S^T
// accumulator registers a valid B fragment; V^T is read with the same permutation.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

#define FA_KT 32        // keys per tile
#define FA_VLD 144      // V LDS row stride (bf16): 288 B = 8 banks (mod 64) per key -> conflict-free ds_read_b64_tr_b16
typedef short v4s_t __attribute__((ext_vector_type(4)));
#define FA_THREADS 256

template <int D, int DP, int NG, int QB = 1>
__global__ __launch_bounds__(256 * NG) __attribute__((amdgpu_waves_per_eu(QB == 2 ? 2 : 3))) void flash_attn_kernel(
    const bf16_t* __restrict__ q, long q_row_stride, long q_head_stride, long q_seq_stride,
    const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, long kv_row_stride, long kv_head_stride, long kv_seq_stride,
    bf16_t* __restrict__ out, long o_row_stride, long o_head_stride, long o_seq_stride,
    int T, int L, int causal_offset, int causal, int Hq, int Hkv, float scale,
    int n_split, float* __restrict__ part_o, float* __restrict__ part_ml) {
  constexpr int KLD = DP + 8;          // K LDS row stride (bf16): 272 B / 208 B, conflict-free ds_read_b128
  constexpr int CPR = D / 8;           // 16-B chunks per row
  constexpr int NKS = DP / 32;         // k-steps of the QK^T product
  constexpr int NDT = D / 16;          // d tiles of the PV product
  // NG wave groups of 4 waves: every group serves the SAME 64 queries and takes one 32-key sub-tile of each staged
  // "super tile" of ST = 32 * NG keys, so that two waves share a SIMD and hide each other's LDS / exp latency (the ViT's 256
  // workgroups are one per CU).  The groups' (m, l, O) are merged through LDS at the end.
  // QB query blocks of 16 rows per wave (long prefills: QB = 2): every K / V^T fragment read from LDS feeds QB MFMAs, so the LDS
  // bytes per flop -- what bounds the one-block kernel, 1 KB per 16-cycle MFMA on each of the four SIMDs = the whole 256 B/clk of
  // the LDS -- fall by QB.
  static_assert(NG == 1 || QB == 1, "wave groups and query blocks are alternatives");
  constexpr int THREADS = 256 * NG, ST = FA_KT * NG, QROWS = 64 * QB;
  constexpr int NCH = (ST * CPR + THREADS - 1) / THREADS;   // chunks staged per thread per super tile
  constexpr int KS_STAGE = ST * KLD, VT_STAGE = ST * FA_VLD;
  extern __shared__ __attribute__((aligned(16))) unsigned char fa_smem[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(fa_smem);
  bf16_t* Vt = Ks + 2 * KS_STAGE;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = (tid >> 6) & 3, grp = NG == 1 ? 0 : __builtin_amdgcn_readfirstlane(tid >> 8);      // grp: wave-uniform
  const int fr = lane & 15, fq = lane >> 4;
  // n_split > 1 (prefill only, one sequence): blockIdx.z is the key split; each split covers an even number of this
  // workgroup's key tiles and leaves un-normalised (O, m, l) partials for flash_combine_kernel
  const int seq = n_split > 1 ? 0 : blockIdx.z, head = blockIdx.y;
  const int sp = n_split > 1 ? blockIdx.z : 0;
  const int kvh = head / (Hq / Hkv);
  const int qbase = (blockIdx.x * 4 + wave) * 16 * QB;
  q += seq * q_seq_stride;
  k += seq * kv_seq_stride + kvh * kv_head_stride;
  v += seq * kv_seq_stride + kvh * kv_head_stride;
  out += seq * o_seq_stride;

  // zero the K pad columns of both stages once (D < DP, ViT d = 80 -> 96)
  if constexpr (DP > D) {
    for (int i = tid; i < 2 * ST * (DP - D); i += THREADS) {
      const int st = i / (ST * (DP - D)), r = i % (ST * (DP - D));
      Ks[st * KS_STAGE + (r / (DP - D)) * KLD + D + r % (DP - D)] = 0;
    }
  }

  // ---- Q fragments (B operand): lane (fr, fq) holds Q[qbase+fr][ks*32 + fq*8 .. +7]
  bf16x8_t qf[QB][NKS];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int tq = qbase + qb * 16 + fr;
    const bf16_t* qr = q + (size_t)(tq < T ? tq : T - 1) * q_row_stride + (size_t)head * q_head_stride;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int d0 = ks * 32 + fq * 8;
      u32x4_t raw = d0 < D ? *reinterpret_cast<const u32x4_t*>(qr + d0) : u32x4_t{0, 0, 0, 0};
      qf[qb][ks] = *reinterpret_cast<bf16x8_t*>(&raw);
    }
  }

  f32x4_t oacc[QB][NDT];
  float m_run[QB], l_run[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) oacc[qb][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    m_run[qb] = -1e30f;
    l_run[qb] = 0.f;
  }
  const float cexp = scale * 1.4426950408889634f;      // scale * log2(e)

  // keys needed by this workgroup
  int kmax = L;
  if (causal) kmax = min(L, min((int)blockIdx.x * QROWS + QROWS - 1, T - 1) + causal_offset + 1);
  const int n_kt_all = (kmax + ST - 1) / ST;          // in super tiles
  const int per = ((n_kt_all + n_split - 1) / n_split + 1) & ~1;      // even: the stage parity below starts at 0
  const int kt_lo = sp * per;
  const int n_kt = min(n_kt_all, kt_lo + per);

  // ---- staging: thread owns chunks idx = tid + i*256 of every tile; two register slots form a ring so that the
  // global loads of tile t+3 are in flight while tile t is computed (one barrier per tile, two iterations to land).
  // K and V come through buffer descriptors: the per-thread byte offset is loop-invariant, the tile's offset is ONE scalar, and
  // the range check of the descriptor returns zeros for rows >= L and for tiles past the end -- no address arithmetic, clamping
  // or zero-fill selects in the loop.
  const unsigned kv_bytes = ((unsigned)(L - 1) * (unsigned)kv_row_stride + D) * 2u;
  const __amdgpu_buffer_rsrc_t k_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(k), 0, kv_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(v), 0, kv_bytes, 0x00020000);
  const unsigned tile_bytes = (unsigned)ST * (unsigned)kv_row_stride * 2u;
  unsigned voff[NCH], lds_k[NCH], lds_v[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int idx = tid + i * THREADS;
    const int row = idx / CPR, c = idx % CPR;
    voff[i] = idx < ST * CPR ? ((unsigned)row * (unsigned)kv_row_stride + c * 8) * 2u : 0xFFFFFFF0u;       // beyond the tile: out of range
    lds_k[i] = row * KLD + c * 8;
    lds_v[i] = row * FA_VLD + c * 8;
  }
  u32x4_t kr0[NCH], vr0[NCH], kr1[NCH], vr1[NCH];
  auto load_tile = [&](int kt, u32x4_t (&kreg)[NCH], u32x4_t (&vreg)[NCH]) {
    const unsigned soff = (unsigned)kt * tile_bytes;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const unsigned off = voff[i] + soff;             // the tile offset rides in the VGPR offset, which the range check is certain to cover
      const auto kk = __builtin_amdgcn_raw_buffer_load_b128(k_rs, off, 0, 0);
      const auto vv = __builtin_amdgcn_raw_buffer_load_b128(v_rs, off, 0, 0);
      kreg[i] = u32x4_t{kk[0], kk[1], kk[2], kk[3]};
      vreg[i] = u32x4_t{vv[0], vv[1], vv[2], vv[3]};
    }
  };
  auto store_tile = [&](int st, int kt, const u32x4_t (&kreg)[NCH], const u32x4_t (&vreg)[NCH]) {
    bf16_t* ks_ = Ks + st * KS_STAGE;
    bf16_t* vt_ = Vt + st * VT_STAGE;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      if ((NCH * THREADS == ST * CPR) || tid + i * THREADS < ST * CPR) {
        *reinterpret_cast<u32x4_t*>(ks_ + lds_k[i]) = kreg[i];
        *reinterpret_cast<u32x4_t*>(vt_ + lds_v[i]) = vreg[i];      // V stays row-major: transposed on read
      }
    }
  };

  // MASK = false is the straight-line body of every tile below the causal diagonal and inside the sequence (almost all of a long
  // prefill); the variant is picked per wave by a scalar branch, so the common body carries no predicate bookkeeping at all
  const int qbase_s = __builtin_amdgcn_readfirstlane(qbase);
  auto compute_t = [&](int kt, int st, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    const bf16_t* ks_ = Ks + st * KS_STAGE + grp * (FA_KT * KLD);        // this wave group's 32 keys of the super tile
    const bf16_t* vt_ = Vt + st * VT_STAGE + grp * (FA_KT * FA_VLD);
    const int j0 = kt * ST + grp * FA_KT;
    // ---- S^T = K . Q^T   (two 16-key sub-tiles; each K fragment serves the wave's QB query blocks)
    f32x4_t sacc[QB][2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) sacc[qb][sub] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(ks_ + (sub * 16 + fr) * KLD + ks * 32 + fq * 8);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) sacc[qb][sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[qb][ks], sacc[qb][sub], 0, 0, 0);
      }
    }
    // ---- online softmax for query column fr; this lane holds keys kt*32 + sub*16 + 4*fq + r.
    // Scores are scaled into the exponent's unit first (t = s * scale*log2 e: a plain VALU product, which also spares the
    // canonicalising v_max hipcc puts in front of an fmaxf on raw MFMA outputs), p = exp2(t - m) is one subtraction + one bare
    // v_exp_f32 per score (exp2f() costs seven instructions for its denormal range), and the row sum stays a per-lane partial
    // until the epilogue (the rescale factor is the same in the four lanes of a query).  m_run is kept in the scaled unit.
    // Masking runs only on tiles that touch the sequence end or the causal diagonal.
    bf16x8_t pb[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const int tq = qbase + qb * 16 + fr;
      float sv[8];
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int r = 0; r < 4; ++r) sv[sub * 4 + r] = sacc[qb][sub][r] * cexp;
      unsigned okbits = 0xFFu;
      if constexpr (MASK) {
        okbits = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int j = j0 + (i >> 2) * 16 + fq * 4 + (i & 3);
          const bool ok = j < L && (!causal || j <= tq + causal_offset);
          sv[i] = ok ? sv[i] : -1e30f;
          okbits |= ok ? (1u << i) : 0u;
        }
      }
      float mx = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
      mx = xor32_max(xor16_max(mx));
      const float m_new = fmaxf(m_run[qb], mx);               // running max of the scaled scores
      float p[8], rs = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        p[i] = __builtin_amdgcn_exp2f(sv[i] - m_new);     // masked scores (-1e30) underflow to exactly 0 ...
      }
      if constexpr (MASK) {                                // ... unless the whole row is masked so far (a key split that
#pragma unroll                                           // starts beyond a query's causal limit): m is still -1e30 there
        for (int i = 0; i < 8; ++i) p[i] = (okbits >> i) & 1u ? p[i] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) rs += p[i];
      // rescale only when some query of the wave saw a new maximum (rare after the first tiles)
      const bool grew = m_new > m_run[qb];
      float alpha = 1.f;
      if (__any(grew)) {
        alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) oacc[qb][dt][r] *= alpha;
      }
      l_run[qb] = l_run[qb] * alpha + rs;                  // this lane's 8 keys of every tile; summed over the query's 4 lanes at the end
      m_run[qb] = m_new;
      u32x4_t pk = pack8(p);
      pb[qb] = *reinterpret_cast<bf16x8_t*>(&pk);
    }
    // ---- O^T += V^T . P^T
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      // ds_read_b64_tr_b16: lane fr of the 16-lane group supplies row (key) 4*fq + fr/4, columns 4*(fr%4)..+3 and
      // receives column (d) dt*16 + fr of the 4 rows -> the V^T fragment without a transposed LDS image
      const bf16_t* vr = vt_ + (fq * 4 + (fr >> 2)) * FA_VLD + dt * 16 + (fr & 3) * 4;
      const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(vr));
      const v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(vr + 16 * FA_VLD));
      const bf16x8_t a = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) oacc[qb][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pb[qb], oacc[qb][dt], 0, 0, 0);
    }
  };
  auto compute = [&](int kt, int st) {
    const int j_hi = __builtin_amdgcn_readfirstlane(kt * ST + grp * FA_KT + FA_KT - 1);
    if (j_hi >= L || (causal && j_hi > qbase_s + causal_offset)) compute_t(kt, st, std::true_type{});       // qbase: smallest query of the wave
    else compute_t(kt, st, std::false_type{});
  };

  load_tile(kt_lo, kr0, vr0);
  load_tile(kt_lo + 1, kr1, vr1);
  __syncthreads();          // pad zeroing visible before the first K store lands next to it
  store_tile(0, kt_lo, kr0, vr0);
  load_tile(kt_lo + 2, kr0, vr0);
  __syncthreads();
  for (int kt = kt_lo; kt < n_kt; kt += 2) {
    // even: stage 0 holds tile kt; slot 1 holds kt+1, slot 0 holds kt+2
    if (kt + 1 < n_kt) store_tile(1, kt + 1, kr1, vr1);
    load_tile(kt + 3, kr1, vr1);
    __builtin_amdgcn_sched_barrier(0);     // keep the prefetch above the MFMAs (hipcc otherwise sinks the loads)
    compute(kt, 0);
    lds_barrier();
    if (kt + 1 >= n_kt) break;
    // odd: stage 1 holds tile kt+1; slot 0 holds kt+2, slot 1 holds kt+3
    if (kt + 2 < n_kt) store_tile(0, kt + 2, kr0, vr0);
    load_tile(kt + 4, kr0, vr0);
    __builtin_amdgcn_sched_barrier(0);
    compute(kt + 1, 1);
    lds_barrier();
  }

#pragma unroll
  for (int qb = 0; qb < QB; ++qb) l_run[qb] = xor32_sum(xor16_sum(l_run[qb]));

  // ---- merge of the wave groups: group g > 0 leaves (m, l, O) in the (now idle) staging LDS, group 0 folds them in
  if constexpr (NG > 1) {
    float* mg = reinterpret_cast<float*>(fa_smem);
    constexpr int MW = NDT * 4 + 2;                    // floats per thread, stored thread-minor (conflict-free)
    const int t256 = tid & 255;
#pragma unroll
    for (int g = 1; g < NG; ++g) {
      if (grp == g) {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) mg[(dt * 4 + r) * 256 + t256] = oacc[0][dt][r];
        mg[(NDT * 4) * 256 + t256] = m_run[0];
        mg[(NDT * 4 + 1) * 256 + t256] = l_run[0];
      }
      __syncthreads();
      if (grp == 0) {
        const float m1 = mg[(NDT * 4) * 256 + t256], l1 = mg[(NDT * 4 + 1) * 256 + t256];
        const float M = fmaxf(m_run[0], m1);
        const float a0 = exp2f(m_run[0] - M), a1 = exp2f(m1 - M);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) oacc[0][dt][r] = oacc[0][dt][r] * a0 + mg[(dt * 4 + r) * 256 + t256] * a1;
        l_run[0] = l_run[0] * a0 + l1 * a1;
        m_run[0] = M;
      }
      if (g + 1 < NG) __syncthreads();
    }
    static_assert(MW * 256 * 4 <= 2 * (KS_STAGE + VT_STAGE) * 2, "merge buffer must fit the staging LDS");
    if (grp != 0) return;
  }

  // ---- epilogue: lane holds O[tq][dt*16 + 4*fq + r]
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int tq = qbase + qb * 16 + fr;
    if (tq >= T) continue;
    if (n_split > 1) {
      const size_t row = ((size_t)sp * T + tq) * Hq + head;
      float* po = part_o + row * D;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) *reinterpret_cast<f32x4_t*>(po + dt * 16 + fq * 4) = oacc[qb][dt];
      if (fq == 0) {
        part_ml[row * 2] = m_run[qb];
        part_ml[row * 2 + 1] = l_run[qb];
      }
      continue;
    }
    const float inv = 1.0f / l_run[qb];
    bf16_t* orow = out + (size_t)tq * o_row_stride + (size_t)head * o_head_stride;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      u32x2_t o;
      o[0] = pack2(oacc[qb][dt][0] * inv, oacc[qb][dt][1] * inv);
      o[1] = pack2(oacc[qb][dt][2] * inv, oacc[qb][dt][3] * inv);
      *reinterpret_cast<u32x2_t*>(orow + dt * 16 + fq * 4) = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// LLM prefill, d = 128: the same products and softmax with K and V staged by LDS-DMA (`buffer_load_dwordx4 ... lds`) instead of
// through a register ring: no staging registers, no ds_write pass, and the descriptor's range check zero-fills rows >= L.
// A DMA piece lands lane-linear (lane i -> base + 16 i: 4 rows of 256 B), so rows cannot be padded; bank conflicts are avoided by
// XOR-swizzling the 16-B chunks of a row through the per-lane SOURCE offset and again on the fragment reads:
//   K (ds_read_b128, lane = (key fr, chunk 4 ks + fq)):                  chunk ^ (key & 15)
//   V (ds_read_b64_tr_b16, lane = (key 4 fq + fr/4, 8-B piece fr % 4)):   chunk ^ ((key & 7) << 1)
// both conflict-free for the lane groups the LDS serves together.  Three 16 KB stages (tile t+2 in flight while tile t is
// computed), counted vmcnt + raw s_barrier, the ring unrolled by three so that every LDS address is a per-lane register plus an
// immediate.  QB query blocks of 16 rows per wave as in flash_attn_kernel.
#define PA_TILE_B (FA_KT * 256)
#define PA_STAGE_B (2 * PA_TILE_B)
template <int QB, bool SUPER = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(QB == 2 ? 2 : 3))) void prefill_attn_dma_kernel(
    const bf16_t* __restrict__ q, long q_row_stride, long q_head_stride, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    long kv_head_stride, bf16_t* __restrict__ out, long o_row_stride, long o_head_stride, int T, int L, int causal_offset, int Hq,
    int Hkv, float scale, int n_split, float* __restrict__ part_o, float* __restrict__ part_ml) {
  constexpr int D = 128, NKS = 4, NDT = 8, QROWS = 64 * QB;
  extern __shared__ __attribute__((aligned(16))) unsigned char fa_smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int head = blockIdx.y, sp = n_split > 1 ? blockIdx.z : 0;
  const int kvh = head / (Hq / Hkv);
  const int qbase = (blockIdx.x * 4 + wave) * 16 * QB;
  k += kvh * kv_head_stride;
  v += kvh * kv_head_stride;

  bf16x8_t qf[QB][NKS];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int tq = qbase + qb * 16 + fr;
    const bf16_t* qr = q + (size_t)(tq < T ? tq : T - 1) * q_row_stride + (size_t)head * q_head_stride;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      u32x4_t raw = *reinterpret_cast<const u32x4_t*>(qr + ks * 32 + fq * 8);
      qf[qb][ks] = *reinterpret_cast<bf16x8_t*>(&raw);
    }
  }
  f32x4_t oacc[QB][NDT];
  float m_run[QB], l_run[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) oacc[qb][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    m_run[qb] = -1e30f;
    l_run[qb] = 0.f;
  }
  const float cexp = scale * 1.4426950408889634f;

  const int kmax = min(L, min((int)blockIdx.x * QROWS + QROWS - 1, T - 1) + causal_offset + 1);
  const int n_kt_all = (kmax + FA_KT - 1) / FA_KT;
  const int per = ((n_kt_all + n_split - 1) / n_split + 1) & ~1;
  const int kt_lo = sp * per;
  const int n_kt = min(n_kt_all, kt_lo + per);

  // ---- DMA: wave w fills rows 8 w .. 8 w + 7 of the K and of the V tile, two 1-KiB pieces each
  const __amdgpu_buffer_rsrc_t k_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(k), 0, L * 256, 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(v), 0, L * 256, 0x00020000);
  // piece 1 holds the rows 4 below piece 0: its swizzle differs in one bit (K: chunk ^ 4, V: chunk ^ 8), so one source offset per
  // operand is kept and the other derived (two query blocks per wave leave no register to spare)
  unsigned src_k0, src_v0;
  {
    const int r = wave * 8 + (lane >> 4), pch = lane & 15;
    src_k0 = r * 256 + ((pch ^ (r & 15)) * 16);
    src_v0 = r * 256 + ((pch ^ ((r & 7) << 1)) * 16);
  }
  auto issue = [&](int kt, int stage) {
    const unsigned tb = (unsigned)kt * PA_TILE_B;          // rides in the VGPR offset, which the range check is certain to cover
    unsigned char* sk = fa_smem + stage * PA_STAGE_B + wave * 2048;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rs, (__attribute__((address_space(3))) void*)(sk), 16, src_k0 + tb, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rs, (__attribute__((address_space(3))) void*)(sk + PA_TILE_B), 16, src_v0 + tb, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rs, (__attribute__((address_space(3))) void*)(sk + 1024), 16, ((src_k0 + 1024) ^ 64u) + tb, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rs, (__attribute__((address_space(3))) void*)(sk + PA_TILE_B + 1024), 16, ((src_v0 + 1024) ^ 128u) + tb, 0, 0, 0);
  };

  // ---- per-lane LDS read offsets (stage and sub-tile offsets are immediates)
  // K chunk of k-step ks: (4 ks + fq) ^ fr = (ks << 2) ^ (fq ^ fr) -> byte offset k_base ^ (ks << 6); V likewise v_base ^ (dt << 5)
  unsigned k_base, v_base;
  {
    k_base = fr * 256 + ((fq ^ fr) * 16);
    const int key = fq * 4 + (fr >> 2);
    const int xk = ((key & 7) << 1) ^ ((fr & 3) >> 1);
    v_base = PA_TILE_B + key * 256 + xk * 16 + (fr & 1) * 8;
  }
  const int qbase_s = __builtin_amdgcn_readfirstlane(qbase);

  // Online softmax with a LAZY reference: p = exp2(s*c - m_ref) is exact for ANY reference that keeps it in range, so the running
  // maximum is not tracked tile by tile.  The fast tail costs one FMA, one bare v_exp_f32, one add and half a convert per score --
  // no max tree, no cross-lane exchange, no pass over the accumulators -- and only checks that no lane's partial row sum left the
  // safe range (a score more than ~12 binades above the reference, or the very first tile, whose reference is -1e30: exp2 gives
  // +inf and the test fires before p is used).  The slow tail then takes the exact maximum as the new reference and rescales.
  // The two tails are whole variants picked by a scalar branch (a conditional rescale inside one body makes hipcc merge the
  // accumulator tuples of both paths through copies: 256 VGPRs plus scratch at two query blocks per wave, against 165).
  constexpr float LAZY_SUM_MAX = 4096.f;
  auto compute_t = [&](int kt, auto stage_c, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    constexpr int SB = decltype(stage_c)::value * PA_STAGE_B;
    const unsigned char* st = fa_smem + SB;
    const int j0 = kt * FA_KT;
    f32x4_t sacc[QB][2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) sacc[qb][sub] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(st + (k_base ^ (ks << 6)) + sub * 4096);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) sacc[qb][sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[qb][ks], sacc[qb][sub], 0, 0, 0);
      }
    }
    // per query block: fast attempt, exact redo when the reference has to move (only scalars-per-lane change inside that branch);
    // the accumulators are touched by whole variants of the P.V stage only
    bf16x8_t pb[QB];
    float alpha[QB];
    bool moved = false;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float p[8], rs = 0.f;
      bool big = MASK;
      if constexpr (!MASK) {
        const float mref = -m_run[qb];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            p[sub * 4 + r] = __builtin_amdgcn_exp2f(fmaf(sacc[qb][sub][r], cexp, mref));
            rs += p[sub * 4 + r];
          }
        big = __any(!(rs <= LAZY_SUM_MAX));
      }
      alpha[qb] = 1.f;
      if (big) {                   // exact maximum, new reference
        const int tq = qbase + qb * 16 + fr;
        float sv[8];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int r = 0; r < 4; ++r) sv[sub * 4 + r] = sacc[qb][sub][r] * cexp;
        unsigned okbits = 0xFFu;
        if constexpr (MASK) {
          okbits = 0;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int j = j0 + (i >> 2) * 16 + fq * 4 + (i & 3);
            const bool ok = j < L && j <= tq + causal_offset;
            sv[i] = ok ? sv[i] : -1e30f;
            okbits |= ok ? (1u << i) : 0u;
          }
        }
        float mx = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
        mx = xor32_max(xor16_max(mx));
        const float m_new = fmaxf(m_run[qb], mx);
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = __builtin_amdgcn_exp2f(sv[i] - m_new);     // masked scores (-1e30) underflow to 0 ...
        if constexpr (MASK) {                                // ... unless the whole row is masked so far (m still -1e30)
#pragma unroll
          for (int i = 0; i < 8; ++i) p[i] = (okbits >> i) & 1u ? p[i] : 0.f;
        }
        rs = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) rs += p[i];
        alpha[qb] = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
        m_run[qb] = m_new;
        moved = true;
      }
      l_run[qb] = l_run[qb] * alpha[qb] + rs;               // this lane's 8 keys of every tile; summed over the query's 4 lanes at the end
      u32x4_t pk = pack8(p);
      pb[qb] = *reinterpret_cast<bf16x8_t*>(&pk);
    }
    auto pv = [&](auto rescale_tag) {
      if constexpr (decltype(rescale_tag)::value) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) oacc[qb][dt][r] *= alpha[qb];
      }
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(st + (v_base ^ (dt << 5))));
        const v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(st + (v_base ^ (dt << 5)) + 4096));
        const bf16x8_t a = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) oacc[qb][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pb[qb], oacc[qb][dt], 0, 0, 0);
      }
    };
    if (moved) pv(std::true_type{});
    else pv(std::false_type{});
  };
  auto step = [&](int kt, auto stage_c) {
    constexpr int S = decltype(stage_c)::value;
    // tile kt has landed once at most the NEWER tile's four pieces of this wave are outstanding; behind the barrier everyone is
    // past compute(kt - 1), whose stage (S + 2) % 3 may be refilled
    if (kt + 1 < n_kt) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (kt + 2 < n_kt) issue(kt + 2, (S + 2) % 3);
    const int j_hi = kt * FA_KT + FA_KT - 1;
    if (j_hi >= L || j_hi > qbase_s + causal_offset) compute_t(kt, stage_c, std::true_type{});
    else compute_t(kt, stage_c, std::false_type{});
  };

  // SUPER: two tiles per barrier on a 2-deep ring of 64-key super tiles (four 16 KB images): half the barriers, waits and loop
  // control per key; the next super tile's eight pieces have a whole super tile of arithmetic to land
  auto run = [&](int kt, auto image_c) {
    const int j_hi = kt * FA_KT + FA_KT - 1;
    if (j_hi >= L || j_hi > qbase_s + causal_offset) compute_t(kt, image_c, std::true_type{});
    else compute_t(kt, image_c, std::false_type{});
  };
  auto sstep = [&](int kt, auto ss_c) {
    constexpr int SS = decltype(ss_c)::value;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (kt + 2 < n_kt) issue(kt + 2, 2 * (1 - SS));
    if (kt + 3 < n_kt) issue(kt + 3, 2 * (1 - SS) + 1);
    run(kt, std::integral_constant<int, 2 * SS>{});
    if (kt + 1 < n_kt) run(kt + 1, std::integral_constant<int, 2 * SS + 1>{});
  };
  if (kt_lo < n_kt) issue(kt_lo, 0);
  if (kt_lo + 1 < n_kt) issue(kt_lo + 1, 1);
  if constexpr (SUPER) {
    for (int kt = kt_lo; kt < n_kt;) {
      sstep(kt, std::integral_constant<int, 0>{});
      kt += 2;
      if (kt >= n_kt) break;
      sstep(kt, std::integral_constant<int, 1>{});
      kt += 2;
    }
  } else {
    for (int kt = kt_lo; kt < n_kt;) {
      step(kt, std::integral_constant<int, 0>{});
      if (++kt >= n_kt) break;
      step(kt, std::integral_constant<int, 1>{});
      if (++kt >= n_kt) break;
      step(kt, std::integral_constant<int, 2>{});
      ++kt;
    }
  }

#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const float l = xor32_sum(xor16_sum(l_run[qb]));
    const int tq = qbase + qb * 16 + fr;
    if (tq >= T) continue;
    if (n_split > 1) {
      const size_t row = ((size_t)sp * T + tq) * Hq + head;
      float* po = part_o + row * D;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) *reinterpret_cast<f32x4_t*>(po + dt * 16 + fq * 4) = oacc[qb][dt];
      if (fq == 0) {
        part_ml[row * 2] = m_run[qb];
        part_ml[row * 2 + 1] = l;
      }
      continue;
    }
    const float inv = 1.0f / l;
    bf16_t* orow = out + (size_t)tq * o_row_stride + (size_t)head * o_head_stride;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      u32x2_t o;
      o[0] = pack2(oacc[qb][dt][0] * inv, oacc[qb][dt][1] * inv);
      o[1] = pack2(oacc[qb][dt][2] * inv, oacc[qb][dt][3] * inv);
      *reinterpret_cast<u32x2_t*>(orow + dt * 16 + fq * 4) = o;
    }
  }
}

template <int D, int DP, int NG>
static int fa_smem_bytes() {
  return 2 * (FA_KT * NG * (DP + 8) + FA_KT * NG * FA_VLD) * 2;
}
// Sets the dynamic-LDS limit once per instantiation (> 64 KB for two wave groups at d = 128).
template <int D, int DP, int NG, int QB = 1>
static void fa_prepare() {
  static bool done = false;
  if (!done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_attn_kernel<D, DP, NG, QB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        fa_smem_bytes<D, DP, NG>());
    done = true;
  }
}
static int fa_groups(const char* env, int dflt) {
  const char* e = svlm_env(env);
  const int v = e ? atoi(e) : dflt;
  return v == 2 ? 2 : 1;
}

// Merge of the key splits: out[t][h][:] = sum_s w_s O_s / sum_s w_s l_s,  w_s = exp2(m_s - max_s m_s), m_s = max score * scale*log2 e.
// One thread per (t, h, 4 consecutive d).
__global__ __launch_bounds__(256) void flash_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                            bf16_t* __restrict__ out, long o_row_stride, int T, int Hq, int n_split,
                                                            float scale) {
  constexpr int D = 128;
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= (long)T * Hq * (D / 4)) return;
  const int c = (int)(i % (D / 4));
  const long th = i / (D / 4);                       // t * Hq + h
  float m[8], l[8];
  f32x4_t o[8];
  float M = -1e30f;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    if (s < n_split) {
      const size_t row = (size_t)s * T * Hq + th;
      m[s] = part_ml[row * 2];
      l[s] = part_ml[row * 2 + 1];
      o[s] = *reinterpret_cast<const f32x4_t*>(part_o + row * D + c * 4);
      M = fmaxf(M, m[s]);
    }
  }
  float Ls = 0.f;
  f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    if (s < n_split) {
      const float w = exp2f(m[s] - M);               // m comes in the exponent's unit; an empty split has l = 0 and O = 0
      Ls += w * l[s];
      acc += o[s] * w;
    }
  }
  const float inv = 1.0f / Ls;
  const int t = (int)(th / Hq), h = (int)(th % Hq);
  u32x2_t r;
  r[0] = pack2(acc[0] * inv, acc[1] * inv);
  r[1] = pack2(acc[2] * inv, acc[3] * inv);
  *reinterpret_cast<u32x2_t*>(out + (size_t)t * o_row_stride + h * D + c * 4) = r;
}

// Key splits of a prefill: ~450 workgroups in all (measured, tools/prefill_attn_splits.py: a 290-row chunk is 60 query tiles on the
// 12-head 2B model -> 8 splits, 83 -> 35 us; 140 tiles on the 28-head 7B model -> 3 splits, 151 -> 86 us), at least four 32-key
// tiles per split, at most 8.
static inline int prefill_splits(int T, int L, int Hq) {
  const int base = ((T + 63) / 64) * Hq;
  static const int force = svlm_env("SVLM_PREFILL_SPLITS") ? atoi(svlm_env("SVLM_PREFILL_SPLITS")) : 0;      // tuning aid
  if (force > 0) return force > 8 ? 8 : force;
  int ns = base > 0 ? (480 + base / 2) / base : 1;      // ~480 workgroups: the 2B chunk (60 query tiles) takes 8 splits (28.0 us per layer, 29.6 with 7)
  const int by_len = L / (4 * FA_KT);
  ns = ns < by_len ? ns : by_len;
  return ns < 1 ? 1 : (ns > 8 ? 8 : ns);
}

// K: gather + rotate; V: gather; Q: rotate.  D = 128; one thread per 16-B chunk.
//   k_rot, v_lin: (Hkv, L, 128) in logical order;  q_rot: (T, Hq*128)
__global__ __launch_bounds__(256) void rope_gather_kernel(const bf16_t* __restrict__ q, int q_stride,
                                                          const bf16_t* __restrict__ k_new, const bf16_t* __restrict__ v_new, int kv_new_stride,
                                                          bf16_t* __restrict__ k_planes, bf16_t* __restrict__ v_planes,
                                                          const int* __restrict__ slot_of, const bf16_t* __restrict__ rope_cs,
                                                          bf16_t* __restrict__ q_rot, bf16_t* __restrict__ k_rot,
                                                          bf16_t* __restrict__ v_lin, int T, int L, int Hq, int Hkv, int n_slots,
                                                          bf16_t* __restrict__ k_keep, bf16_t* __restrict__ v_keep, int lin_rows,
                                                          int* __restrict__ lin_len_dev) {
  // k_keep / v_keep (optional): the layer's LINEAR PLANES, which outlive the prefill -- the decode steps of the chunk stream the
  // rotated keys from there instead of rotating the pool rows again (layout and contract: svlm_decode_attn_lin, decode_attn.hip)
  constexpr int D = 128, CPR = 16;
  const long nk = (long)Hkv * L * CPR, nq = (long)T * Hq * CPR;
  const long total = 2 * nk + nq;
  if (lin_len_dev != nullptr && blockIdx.x == 0 && threadIdx.x == 0) { lin_len_dev[0] = L; lin_len_dev[1] = 1; }   // {rows rotated, appended rows follow}
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    if (i >= nk && i < 2 * nk) {                      // V: plain gather (new rows: taken from the projection output and APPENDED)
      const long t = i - nk;
      const int c = (int)(t % CPR), j = (int)((t / CPR) % L), h = (int)(t / ((long)CPR * L));
      bf16_t* slot_p = v_planes + ((size_t)h * n_slots + slot_of[j]) * D + c * 8;
      u32x4_t v;
      if (v_new != nullptr && j >= L - T) {
        v = *reinterpret_cast<const u32x4_t*>(v_new + (size_t)(j - (L - T)) * kv_new_stride + h * D + c * 8);
        *reinterpret_cast<u32x4_t*>(slot_p) = v;     // StreamingCache.update (streaming_cache.py:72-73)
      } else {
        v = *reinterpret_cast<const u32x4_t*>(slot_p);
      }
      *reinterpret_cast<u32x4_t*>(v_lin + ((size_t)h * L + j) * D + c * 8) = v;
      if (v_keep != nullptr) *reinterpret_cast<u32x4_t*>(v_keep + ((size_t)h * lin_rows + j) * D + c * 8) = v;
      continue;
    }
    const bf16_t* src;
    bf16_t* dst;
    bf16_t* keep = nullptr;
    int c, pos;
    if (i < nk) {
      c = (int)(i % CPR);
      const int j = (int)((i / CPR) % L), h = (int)(i / ((long)CPR * L));
      bf16_t* slot_p = k_planes + ((size_t)h * n_slots + slot_of[j]) * D;
      src = slot_p;
      if (k_new != nullptr && j >= L - T) {           // new row: un-rotated K goes to its pool slot, the rotated copy to k_rot
        src = k_new + (size_t)(j - (L - T)) * kv_new_stride + h * D;
        *reinterpret_cast<u32x4_t*>(slot_p + c * 8) = *reinterpret_cast<const u32x4_t*>(src + c * 8);
      }
      dst = k_rot + ((size_t)h * L + j) * D;
      if (k_keep != nullptr)        // tile j / 16 of the kv head, chunk c of key j % 16 at [c >> 2][(c & 3) * 16 + j % 16][8]
        keep = k_keep + ((size_t)h * (lin_rows >> 4) + (j >> 4)) * 2048 + (c >> 2) * 512 + ((c & 3) * 16 + (j & 15)) * 8;
      pos = j;
    } else {
      const long t = i - 2 * nk;
      c = (int)(t % CPR);
      const int h = (int)((t / CPR) % Hq), r = (int)(t / ((long)CPR * Hq));
      src = q + (size_t)r * q_stride + h * D;
      dst = q_rot + ((size_t)r * Hq + h) * D;
      pos = L - T + r;                                // right-aligned query positions (language_forward.py:44-53)
    }
    const bf16_t* csr = rope_cs + (size_t)pos * D;
    float x[8], xp[8], cc[8], sn[8], o[8];
    unpack8(*reinterpret_cast<const u32x4_t*>(src + c * 8), x);
    unpack8(*reinterpret_cast<const u32x4_t*>(src + (c ^ 8) * 8), xp);
    unpack8(*reinterpret_cast<const u32x4_t*>(csr + (c & 7) * 8), cc);
    unpack8(*reinterpret_cast<const u32x4_t*>(csr + 64 + (c & 7) * 8), sn);
    const bool upper = c >= 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float rot = upper ? xp[e] : -xp[e];
      o[e] = rbf(rbf(x[e] * cc[e]) + rbf(rot * sn[e]));
    }
    const u32x4_t ov = pack8(o);
    *reinterpret_cast<u32x4_t*>(dst + c * 8) = ov;
    if (keep != nullptr) *reinterpret_cast<u32x4_t*>(keep) = ov;
  }
}

// workspace: q_rot (T*Hq*128) | k_rot (Hkv*L*128) | v_lin (Hkv*L*128)  bf16 | split partials O (ns*T*Hq*128) | m,l (ns*T*Hq*2) fp32
extern "C" long long svlm_prefill_attn_ws_bytes(int T, int L, int Hq, int Hkv) {
  if (T < 0 || L < 0 || Hq <= 0 || Hkv <= 0) return SVLM_EINVAL;
  const int ns = prefill_splits(T, L, Hq);
  return ((long long)T * Hq + 2LL * Hkv * L) * 128 * 2 + (ns > 1 ? (long long)ns * T * Hq * (128 + 2) * 4 : 0);
}

// LLM prefill: q (T, Hq*128) un-rotated rows, pool planes of one layer, out (T, Hq*128).
// k_lin / v_lin / lin_len_dev (optional, together): the layer's linear planes of lin_rows rows (svlm_decode_attn_lin); the launch that
// rotates and gathers the keys for this prefill also leaves them there and sets lin_len_dev[0..1] = {L, 1}.
extern "C" int svlm_prefill_attn_ropeload_lin(const void* q, int q_stride, const void* k_new, const void* v_new, int kv_new_stride,
                                              void* k_planes, void* v_planes, const int* slot_of,
                                              const void* rope_cs, void* out, int o_stride, int T, int L, int Hq, int Hkv, int D,
                                              int n_slots, float scale, void* ws, long long ws_bytes, void* k_lin, void* v_lin_keep,
                                              int lin_rows, int* lin_len_dev, void* stream) {
  SVLM_CHECK_ARG((k_lin == nullptr) == (v_lin_keep == nullptr) && (k_lin == nullptr) == (lin_len_dev == nullptr),
                 "svlm_prefill_attn_ropeload_lin: k_lin, v_lin and lin_len_dev come together");
  SVLM_CHECK_ARG(k_lin == nullptr || (lin_rows > 0 && lin_rows % 16 == 0 && lin_rows >= L),
                 "svlm_prefill_attn_ropeload_lin: lin_rows=%d must be a multiple of 16 and >= L=%d", lin_rows, L);
  SVLM_CHECK_ARG((k_new == nullptr) == (v_new == nullptr) && (k_new == nullptr || (kv_new_stride % 8 == 0 && kv_new_stride >= Hkv * D)),
                 "svlm_prefill_attn_ropeload: k_new / v_new come as a pair with a 16-B aligned row stride >= Hkv*D (stride %d)", kv_new_stride);
  SVLM_CHECK_ARG(D == 128, "svlm_prefill_attn_ropeload: head_dim %d unsupported (128 only)", D);
  SVLM_CHECK_ARG(Hq > 0 && Hkv > 0 && Hq % Hkv == 0, "svlm_prefill_attn_ropeload: Hq=%d Hkv=%d", Hq, Hkv);
  SVLM_CHECK_ARG(T >= 0 && L >= T && n_slots > 0, "svlm_prefill_attn_ropeload: need 0 <= T=%d <= L=%d", T, L);
  SVLM_CHECK_ARG(q_stride % 8 == 0 && o_stride % 4 == 0, "svlm_prefill_attn_ropeload: strides must keep 16-B alignment");
  SVLM_CHECK_ARG(ws != nullptr && ws_bytes >= svlm_prefill_attn_ws_bytes(T, L, Hq, Hkv), "svlm_prefill_attn_ropeload: workspace too small (%lld B)", ws_bytes);
  SVLM_CHECK_ARG((long long)(L + 256) * 256 < (1LL << 31), "svlm_prefill_attn_ropeload: L=%d exceeds the 32-bit buffer offsets of the attention kernel", L);
  if (T == 0) return SVLM_OK;
  hipStream_t st = (hipStream_t)stream;
  bf16_t* q_rot = (bf16_t*)ws;
  bf16_t* k_rot = q_rot + (size_t)T * Hq * 128;
  bf16_t* v_lin = k_rot + (size_t)Hkv * L * 128;
  const long total = (2L * Hkv * L + (long)T * Hq) * 16;
  int g = (int)((total + 255) / 256);
  g = g > 4096 ? 4096 : g;
  rope_gather_kernel<<<g, 256, 0, st>>>((const bf16_t*)q, q_stride, (const bf16_t*)k_new, (const bf16_t*)v_new, kv_new_stride,
                                       (bf16_t*)k_planes, (bf16_t*)v_planes, slot_of,
                                       (const bf16_t*)rope_cs, q_rot, k_rot, v_lin, T, L, Hq, Hkv, n_slots, (bf16_t*)k_lin, (bf16_t*)v_lin_keep,
                                       lin_rows, lin_len_dev);
  int rc = svlm_check_launch("svlm_prefill_attn_ropeload(rope_gather)");
  if (rc) return rc;
  const int ns = prefill_splits(T, L, Hq);
  float* part_o = (float*)(v_lin + (size_t)Hkv * L * 128);
  float* part_ml = part_o + (size_t)ns * T * Hq * 128;
  static const int ng = fa_groups("SVLM_PREFILL_FA_GROUPS", 1);
  // long prefills (the dense-frame forward of configs[4]: 4096-row passes over up to 83k keys) run two query blocks per wave;
  // a streaming chunk (T ~ 290) keeps the 64-row tiles, whose key splits fill the chip
  static const int qb_env = svlm_env("SVLM_PREFILL_QB") ? atoi(svlm_env("SVLM_PREFILL_QB")) : 0;
  const int qb = qb_env > 0 ? qb_env : (T >= 1024 && ns == 1 ? 2 : 1);
  static const bool use_dma = svlm_env("SVLM_PREFILL_NO_DMA") == nullptr;
  if (use_dma && ng != 2) {
    dim3 gridd((T + 64 * qb - 1) / (64 * qb), Hq, ns);
    static const bool super_tiles = svlm_env("SVLM_PREFILL_NO_SUPER") == nullptr;      // 64-key super tiles: 818 -> 869 TFLOP/s at 4096 x 83k
    if (qb == 2 && super_tiles)
      prefill_attn_dma_kernel<2, true><<<gridd, 256, 4 * PA_STAGE_B, st>>>(q_rot, (long)Hq * 128, 128, k_rot, v_lin, (long)L * 128, (bf16_t*)out, o_stride, 128,
                                                                         T, L, L - T, Hq, Hkv, scale, ns, part_o, part_ml);
    else if (qb == 2)
      prefill_attn_dma_kernel<2><<<gridd, 256, 3 * PA_STAGE_B, st>>>(q_rot, (long)Hq * 128, 128, k_rot, v_lin, (long)L * 128, (bf16_t*)out, o_stride, 128,
                                                                   T, L, L - T, Hq, Hkv, scale, ns, part_o, part_ml);
    else
      prefill_attn_dma_kernel<1><<<gridd, 256, 3 * PA_STAGE_B, st>>>(q_rot, (long)Hq * 128, 128, k_rot, v_lin, (long)L * 128, (bf16_t*)out, o_stride, 128,
                                                                   T, L, L - T, Hq, Hkv, scale, ns, part_o, part_ml);
    rc = svlm_check_launch("svlm_prefill_attn_ropeload");
    if (rc || ns == 1) return rc;
    const long n_thrd = (long)T * Hq * 32;
    flash_combine_kernel<<<(int)((n_thrd + 255) / 256), 256, 0, st>>>(part_o, part_ml, (bf16_t*)out, o_stride, T, Hq, ns, scale);
    return svlm_check_launch("svlm_prefill_attn_ropeload(combine)");
  }
  dim3 grid((T + 63) / 64, Hq, ns);
  if (ng == 2) {
    fa_prepare<128, 128, 2>();
    flash_attn_kernel<128, 128, 2><<<grid, 512, fa_smem_bytes<128, 128, 2>(), st>>>(
        q_rot, (long)Hq * 128, 128, 0, k_rot, v_lin, 128, (long)L * 128, 0, (bf16_t*)out, o_stride, 128, 0, T, L, L - T, 1, Hq, Hkv,
        scale, ns, part_o, part_ml);
  } else {
    fa_prepare<128, 128, 1>();
    flash_attn_kernel<128, 128, 1><<<grid, 256, fa_smem_bytes<128, 128, 1>(), st>>>(
        q_rot, (long)Hq * 128, 128, 0, k_rot, v_lin, 128, (long)L * 128, 0, (bf16_t*)out, o_stride, 128, 0, T, L, L - T, 1, Hq, Hkv,
        scale, ns, part_o, part_ml);
  }
  rc = svlm_check_launch("svlm_prefill_attn_ropeload");
  if (rc || ns == 1) return rc;
  const long n_thr = (long)T * Hq * 32;
  flash_combine_kernel<<<(int)((n_thr + 255) / 256), 256, 0, st>>>(part_o, part_ml, (bf16_t*)out, o_stride, T, Hq, ns, scale);
  return svlm_check_launch("svlm_prefill_attn_ropeload(combine)");
}

extern "C" int svlm_prefill_attn_ropeload(const void* q, int q_stride, const void* k_new, const void* v_new, int kv_new_stride,
                                          void* k_planes, void* v_planes, const int* slot_of,
                                          const void* rope_cs, void* out, int o_stride, int T, int L, int Hq, int Hkv, int D,
                                          int n_slots, float scale, void* ws, long long ws_bytes, void* stream) {
  return svlm_prefill_attn_ropeload_lin(q, q_stride, k_new, v_new, kv_new_stride, k_planes, v_planes, slot_of, rope_cs, out, o_stride, T, L, Hq,
                                        Hkv, D, n_slots, scale, ws, ws_bytes, nullptr, nullptr, 0, nullptr, stream);
}

// ViT: qkv (N, 3, H, d) fused buffer already rotated by svlm_vit_rope; n_seq sequences of seq_len rows.
extern "C" int svlm_vit_attn(const void* qkv, void* out, int n_seq, int seq_len, int H, int d, float scale, void* stream) {
  SVLM_CHECK_ARG(d == 80 || d == 128, "svlm_vit_attn: head_dim %d unsupported (80 or 128)", d);
  SVLM_CHECK_ARG(n_seq >= 0 && seq_len > 0 && H > 0, "svlm_vit_attn: bad shape n_seq=%d seq_len=%d H=%d", n_seq, seq_len, H);
  if (n_seq == 0) return SVLM_OK;
  SVLM_CHECK_ARG((long long)(seq_len + 256) * 3 * H * d * 2 < (1LL << 31), "svlm_vit_attn: seq_len=%d exceeds the 32-bit buffer offsets of the attention kernel", seq_len);
  const bf16_t* base = (const bf16_t*)qkv;
  const long row = 3L * H * d;
  dim3 grid((seq_len + 63) / 64, H, n_seq);
  // two wave groups per workgroup (2 waves per SIMD share every staged K/V tile): 33 -> 24 us for the 1024-token frame; the
  // 64-token Qwen2.5 windows hold a single 64-key super tile and keep one group
  static const int ng_env = fa_groups("SVLM_VIT_FA_GROUPS", 0);
  const int ng = svlm_env("SVLM_VIT_FA_GROUPS") ? ng_env : (seq_len >= 256 ? 2 : 1);
#define SVLM_VIT_FA(D_, DP_, NG_)                                                                                           \
  do {                                                                                                                      \
    fa_prepare<D_, DP_, NG_>();                                                                                             \
    flash_attn_kernel<D_, DP_, NG_><<<grid, 256 * NG_, fa_smem_bytes<D_, DP_, NG_>(), (hipStream_t)stream>>>(               \
        base, row, d, row * seq_len, base + (long)H * d, base + 2L * H * d, row, d, row * seq_len, (bf16_t*)out,            \
        (long)H * d, d, (long)H * d * seq_len, seq_len, seq_len, 0, 0, H, H, scale, 1, nullptr, nullptr);                   \
  } while (0)
  if (d == 80) {
    if (ng == 2) SVLM_VIT_FA(80, 96, 2); else SVLM_VIT_FA(80, 96, 1);
  } else {
    if (ng == 2) SVLM_VIT_FA(128, 128, 2); else SVLM_VIT_FA(128, 128, 1);
  }
#undef SVLM_VIT_FA
  return svlm_check_launch("svlm_vit_attn");
}
