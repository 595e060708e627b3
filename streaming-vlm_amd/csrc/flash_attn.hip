// Multi-query flash attention on MFMA (v_mfma_f32_16x16x32_bf16) for the two GEMM-shaped
// attention sites of the path:
//   * LLM prefill over the slot-mapped KV pool with RoPE-on-load, causal, GQA
//     (qwen2/language_forward.py:66-166; T ~ 275 queries x L ~ 2.3k keys per chunk)
//   * ViT block-diagonal attention, one sequence per temporal grid, non-causal, d = 80
//     (qwen2/vision_forward.py:6-34, flash_attn_varlen_func with cu_seqlens = [0, h*w, ...])
// Numerics follow flash-attn: fp32 scores and online softmax, P rounded to bf16 before P.V,
// fp32 accumulation, one final division and rounding.
//
// Both products put the LDS-staged operand in the MFMA "A" slot and keep the other in registers:
//   S^T[key][q] = K[key][:] . Q[q][:]      (A = K rows from LDS,   B = Q fragment, loop-invariant)
//   O^T[d][q]   = V^T[d][key] . P^T[key][q] (A = V^T rows from LDS, B = P, straight from the S^T
//                                            accumulators: lane = query column in both products,
//                                            so softmax stats, P and O never leave the lane)
// The k-slot <-> key mapping of the second product is the permutation that makes the S^T
// accumulator registers a valid B fragment; V^T is read with the same permutation.
#include "common.h"

#define FA_KT 32        // keys per tile
#define FA_VLD 40       // V^T LDS row stride (bf16): 80 B, conflict-free ds_read_b64 over 16 rows x 2 groups

template <int D, int DP, bool ROPE, bool WAVE_IS_HEAD>
__global__ __launch_bounds__(512) void flash_attn_kernel(
    const bf16_t* __restrict__ q, long q_row_stride, long q_head_stride, long q_seq_stride,
    const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, long kv_row_stride, long kv_head_stride, long kv_seq_stride,
    const int* __restrict__ slot_of, const bf16_t* __restrict__ rope_cs,
    bf16_t* __restrict__ out, long o_row_stride, long o_head_stride, long o_seq_stride,
    int T, int L, int causal_offset, int causal, int Hq, int Hkv, float scale) {
  constexpr int KLD = DP + 8;          // K LDS row stride (bf16): 272 B / 208 B, conflict-free ds_read_b128
  constexpr int CPR = D / 8;           // 16-B chunks per row
  constexpr int NKS = DP / 32;         // k-steps of the QK^T product
  constexpr int NDT = D / 16;          // d tiles of the PV product
  __shared__ __attribute__((aligned(16))) bf16_t Ks[FA_KT * KLD];
  __shared__ __attribute__((aligned(16))) bf16_t Vt[D * FA_VLD];

  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int G = Hq / Hkv;
  const int seq = blockIdx.z;
  int head, kvh, qbase;
  if (WAVE_IS_HEAD) {
    kvh = blockIdx.y;
    head = kvh * G + wave;
    qbase = blockIdx.x * 16;
  } else {
    head = blockIdx.y;
    kvh = head / G;
    qbase = (blockIdx.x * nw + wave) * 16;
  }
  q += seq * q_seq_stride;
  k += seq * kv_seq_stride + kvh * kv_head_stride;
  v += seq * kv_seq_stride + kvh * kv_head_stride;
  out += seq * o_seq_stride;

  // zero the K pad columns once (D < DP, ViT d=80 -> 96)
  if constexpr (DP > D) {
    for (int i = tid; i < FA_KT * (DP - D); i += nthr) Ks[(i / (DP - D)) * KLD + D + i % (DP - D)] = 0;
  }

  // ---- Q fragments (B operand): lane (fr, fq) holds Q[qbase+fr][ks*32 + fq*8 .. +7]
  const int tq = qbase + fr;
  const int tq_c = tq < T ? tq : T - 1;
  bf16x8_t qf[NKS];
  {
    const bf16_t* qr = q + (size_t)tq_c * q_row_stride + (size_t)head * q_head_stride;
    u32x4_t raw[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int d0 = ks * 32 + fq * 8;
      raw[ks] = d0 < D ? *reinterpret_cast<const u32x4_t*>(qr + d0) : u32x4_t{0, 0, 0, 0};
    }
    if constexpr (ROPE) {
      // D == 128: partner of d is d ^ 64 -> k-step ks ^ 2, same lane
      const bf16_t* csr = rope_cs + (size_t)(tq_c + causal_offset) * D;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const int d0 = ks * 32 + fq * 8;
        float x[8], xp[8], c[8], sn[8], o[8];
        unpack8(raw[ks], x);
        unpack8(raw[ks ^ 2], xp);
        unpack8(*reinterpret_cast<const u32x4_t*>(csr + (d0 & 63)), c);
        unpack8(*reinterpret_cast<const u32x4_t*>(csr + 64 + (d0 & 63)), sn);
        const bool upper = d0 >= 64;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float rot = upper ? xp[i] : -xp[i];
          o[i] = rbf(rbf(x[i] * c[i]) + rbf(rot * sn[i]));
        }
        u32x4_t pk = pack8(o);
        qf[ks] = *reinterpret_cast<bf16x8_t*>(&pk);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) qf[ks] = *reinterpret_cast<bf16x8_t*>(&raw[ks]);
    }
  }

  f32x4_t oacc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) oacc[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m_run = -1e30f, l_run = 0.f;

  // keys needed by this workgroup
  int kmax = L;
  if (causal) {
    const int q_last = WAVE_IS_HEAD ? qbase + 15 : (blockIdx.x * nw + nw) * 16 - 1;
    const int lim = min(q_last, T - 1) + causal_offset + 1;
    kmax = min(L, lim);
  }
  const int n_kt = (kmax + FA_KT - 1) / FA_KT;

  for (int kt = 0; kt < n_kt; ++kt) {
    __syncthreads();  // previous tile fully consumed
    // ---- stage K (rotated) and V^T
    for (int idx = tid; idx < FA_KT * CPR; idx += nthr) {
      const int row = idx / CPR, c = idx % CPR;
      const int j = kt * FA_KT + row;
      u32x4_t kv4 = u32x4_t{0, 0, 0, 0}, vv4 = u32x4_t{0, 0, 0, 0};
      if (j < L) {
        const size_t roff = (size_t)(slot_of ? slot_of[j] : j) * kv_row_stride;
        kv4 = *reinterpret_cast<const u32x4_t*>(k + roff + c * 8);
        vv4 = *reinterpret_cast<const u32x4_t*>(v + roff + c * 8);
        if constexpr (ROPE) {
          const u32x4_t kp4 = *reinterpret_cast<const u32x4_t*>(k + roff + (c ^ 8) * 8);
          const bf16_t* csr = rope_cs + (size_t)j * D;
          float x[8], xp[8], cc[8], sn[8], o[8];
          unpack8(kv4, x);
          unpack8(kp4, xp);
          unpack8(*reinterpret_cast<const u32x4_t*>(csr + (c & 7) * 8), cc);
          unpack8(*reinterpret_cast<const u32x4_t*>(csr + 64 + (c & 7) * 8), sn);
          const bool upper = c >= 8;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float rot = upper ? xp[i] : -xp[i];
            o[i] = rbf(rbf(x[i] * cc[i]) + rbf(rot * sn[i]));
          }
          kv4 = pack8(o);
        }
      }
      *reinterpret_cast<u32x4_t*>(Ks + row * KLD + c * 8) = kv4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        Vt[(c * 8 + 2 * i) * FA_VLD + row] = (bf16_t)(vv4[i] & 0xFFFFu);
        Vt[(c * 8 + 2 * i + 1) * FA_VLD + row] = (bf16_t)(vv4[i] >> 16);
      }
    }
    __syncthreads();

    // ---- S^T = K . Q^T   (two 16-key sub-tiles)
    f32x4_t sacc[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      sacc[sub] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(Ks + (sub * 16 + fr) * KLD + ks * 32 + fq * 8);
        sacc[sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[ks], sacc[sub], 0, 0, 0);
      }
    }
    // ---- online softmax for query column fr; this lane holds keys kt*32 + sub*16 + 4*fq + r
    float sc[8];
    float mx = -1e30f;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = kt * FA_KT + sub * 16 + fq * 4 + r;
        const bool ok = j < L && (!causal || j <= tq + causal_offset);
        const float s_ = ok ? sacc[sub][r] * scale : -1e30f;
        sc[sub * 4 + r] = s_;
        mx = fmaxf(mx, s_);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - m_new);
    float p[8], rs = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      p[i] = sc[i] > -1e29f ? __expf(sc[i] - m_new) : 0.f;
      rs += p[i];
    }
    rs += __shfl_xor(rs, 16, 64);
    rs += __shfl_xor(rs, 32, 64);
    l_run = l_run * alpha + rs;
    m_run = m_new;
    u32x4_t pk = pack8(p);
    const bf16x8_t pb = *reinterpret_cast<bf16x8_t*>(&pk);
    // ---- O^T += V^T . P^T
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const bf16_t* vr = Vt + (dt * 16 + fr) * FA_VLD + fq * 4;
      u32x4_t a4;
      const u32x2_t lo = *reinterpret_cast<const u32x2_t*>(vr);
      const u32x2_t hi = *reinterpret_cast<const u32x2_t*>(vr + 16);
      a4[0] = lo[0]; a4[1] = lo[1]; a4[2] = hi[0]; a4[3] = hi[1];
      const bf16x8_t a = *reinterpret_cast<bf16x8_t*>(&a4);
#pragma unroll
      for (int r = 0; r < 4; ++r) oacc[dt][r] *= alpha;
      oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pb, oacc[dt], 0, 0, 0);
    }
  }

  // ---- epilogue: lane holds O[tq][dt*16 + 4*fq + r]
  if (tq < T && (!WAVE_IS_HEAD || wave < G)) {
    const float inv = 1.0f / l_run;
    bf16_t* orow = out + (size_t)tq * o_row_stride + (size_t)head * o_head_stride;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      u32x2_t o;
      o[0] = pack2(oacc[dt][0] * inv, oacc[dt][1] * inv);
      o[1] = pack2(oacc[dt][2] * inv, oacc[dt][3] * inv);
      *reinterpret_cast<u32x2_t*>(orow + dt * 16 + fq * 4) = o;
    }
  }
}

// LLM prefill: q (T, Hq*128) rows, pool planes of one layer, out (T, Hq*128).
extern "C" int svlm_prefill_attn_ropeload(const void* q, int q_stride, const void* k_planes, const void* v_planes, const int* slot_of,
                                          const void* rope_cs, void* out, int o_stride, int T, int L, int Hq, int Hkv, int D,
                                          int n_slots, float scale, void* stream) {
  SVLM_CHECK_ARG(D == 128, "svlm_prefill_attn_ropeload: head_dim %d unsupported (128 only)", D);
  SVLM_CHECK_ARG(Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && Hq / Hkv <= 8, "svlm_prefill_attn_ropeload: Hq=%d Hkv=%d (group must be <= 8)", Hq, Hkv);
  SVLM_CHECK_ARG(T >= 0 && L >= T && n_slots > 0, "svlm_prefill_attn_ropeload: need 0 <= T=%d <= L=%d", T, L);
  SVLM_CHECK_ARG(q_stride % 8 == 0 && o_stride % 4 == 0, "svlm_prefill_attn_ropeload: strides must keep 16-B alignment");
  if (T == 0) return SVLM_OK;
  const int G = Hq / Hkv;
  dim3 grid((T + 15) / 16, Hkv, 1);
  flash_attn_kernel<128, 128, true, true><<<grid, 64 * G, 0, (hipStream_t)stream>>>(
      (const bf16_t*)q, q_stride, D, 0, (const bf16_t*)k_planes, (const bf16_t*)v_planes, D, (long)n_slots * D, 0, slot_of,
      (const bf16_t*)rope_cs, (bf16_t*)out, o_stride, D, 0, T, L, L - T, 1, Hq, Hkv, scale);
  return svlm_check_launch("svlm_prefill_attn_ropeload");
}

// ViT: qkv (N, 3, H, d) fused buffer already rotated by svlm_vit_rope; n_seq sequences of seq_len rows.
extern "C" int svlm_vit_attn(const void* qkv, void* out, int n_seq, int seq_len, int H, int d, float scale, void* stream) {
  SVLM_CHECK_ARG(d == 80 || d == 128, "svlm_vit_attn: head_dim %d unsupported (80 or 128)", d);
  SVLM_CHECK_ARG(n_seq >= 0 && seq_len > 0 && H > 0, "svlm_vit_attn: bad shape n_seq=%d seq_len=%d H=%d", n_seq, seq_len, H);
  if (n_seq == 0) return SVLM_OK;
  const bf16_t* base = (const bf16_t*)qkv;
  const long row = 3L * H * d;
  dim3 grid((seq_len + 63) / 64, H, n_seq);
  if (d == 80) {
    flash_attn_kernel<80, 96, false, false><<<grid, 256, 0, (hipStream_t)stream>>>(
        base, row, d, row * seq_len, base + (long)H * d, base + 2L * H * d, row, d, row * seq_len, nullptr, nullptr, (bf16_t*)out,
        (long)H * d, d, (long)H * d * seq_len, seq_len, seq_len, 0, 0, H, H, scale);
  } else {
    flash_attn_kernel<128, 128, false, false><<<grid, 256, 0, (hipStream_t)stream>>>(
        base, row, d, row * seq_len, base + (long)H * d, base + 2L * H * d, row, d, row * seq_len, nullptr, nullptr, (bf16_t*)out,
        (long)H * d, d, (long)H * d * seq_len, seq_len, seq_len, 0, 0, H, H, scale);
  }
  return svlm_check_launch("svlm_vit_attn");
}
