// Single-query (decode-step) GQA attention over the slot-mapped KV pool with RoPE-ON-LOAD.
//
// Reference semantics (qwen2/language_forward.py:66-166, shrink mode): keys are cached
// un-rotated (:95-97); on every forward M-RoPE is applied to the query (right-aligned last
// position, :44-53) and to ALL cached keys (:55-63) in the activation dtype
// (x*cos -> bf16, rotate_half(x)*sin -> bf16, sum -> bf16), then causal flash attention
// (:148) with fp32 scores/softmax, P rounded to bf16 before P.V, fp32 accumulation.
// The reference materialises repeat_kv (:107-108) and the rotated K; here one workgroup
// streams each K/V row ONCE for all G query heads of its kv head and rotates in registers.
//
// HBM/latency-bound (2*L*Hkv*D*2 B of K/V per layer-step, ~2 MB at the BASELINE window).
// Split-KV (flash-decoding): grid = (n_splits, Hkv); a workgroup owns up to 64 consecutive
// logical keys = 4 steps of 16 rows (16 lanes x 16 B per 256-B row, 4 rows per wave-load).
// ALL of a workgroup's loads (slot indices, then K, V, cos, sin rows of every step) are issued
// before the first use, so a workgroup costs two dependent memory latencies, not eight.
// Wave shuffles reduce the dots (xor 1,2,4,8) and merge the online-softmax states (xor 16,32);
// the four waves merge through LDS; a second small kernel merges the splits with the splits
// spread over waves and the loads unrolled.
#include "common.h"

#define DA_D 128
#define DA_GMAX 8
#define DA_MAX_STEPS 4     // chunk <= 64

__device__ __forceinline__ void rope8(const float (&x)[8], const float (&xp)[8], const float (&c)[8], const float (&s)[8],
                                      bool upper, float (&out)[8]) {
  // out = bf16(bf16(x*cos) + bf16(rotate_half(x)*sin)); rotate_half = (-x2, x1): lower half pairs with -partner
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float rot = upper ? xp[i] : -xp[i];
    out[i] = rbf(rbf(x[i] * c[i]) + rbf(rot * s[i]));
  }
}

template <int G>
__global__ __launch_bounds__(256) void decode_attn_split_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k_planes, const bf16_t* __restrict__ v_planes,
    const int* __restrict__ slot_of, const bf16_t* __restrict__ rope_cs, const int* __restrict__ len_dev, int len_add,
    float* __restrict__ ws_m, float* __restrict__ ws_l, float* __restrict__ ws_acc, int Hq, int Hkv, int n_slots,
    int chunk, float scale, int max_len) {
  const int start = blockIdx.x * chunk;
  const int kvh = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane >> 4, s = lane & 15;
  const bool upper = s >= 8;
  const int fc = (s & 7) * 8;  // frequency chunk
  const bf16_t* kp = k_planes + (size_t)kvh * n_slots * DA_D + s * 8;
  const bf16_t* vp = v_planes + (size_t)kvh * n_slots * DA_D + s * 8;

  // ---- issue every load of this workgroup up front.  The slot lookups do NOT wait for the length: rows past the
  // end read stale-but-valid entries of slot_of (always < n_slots; the table is max_len long and max_len is a
  // multiple of the chunk) and are masked later, so the length load and the slot loads overlap.
  int rows[DA_MAX_STEPS], slots[DA_MAX_STEPS];
#pragma unroll
  for (int it = 0; it < DA_MAX_STEPS; ++it) {
    rows[it] = min(start + it * 16 + wave * 4 + grp, max_len - 1);
    slots[it] = slot_of[rows[it]];
  }
  const int L = (len_dev ? *len_dev : 0) + len_add;
  if (start >= L) return;
  const int end = min(start + chunk, L);
  u32x4_t qraw[G];
#pragma unroll
  for (int g = 0; g < G; ++g) qraw[g] = *reinterpret_cast<const u32x4_t*>(q + (size_t)(kvh * G + g) * DA_D + s * 8);
  const bf16_t* csq = rope_cs + (size_t)(L - 1) * DA_D;
  const u32x4_t qc = *reinterpret_cast<const u32x4_t*>(csq + fc);
  const u32x4_t qs = *reinterpret_cast<const u32x4_t*>(csq + 64 + fc);
  u32x4_t kraw[DA_MAX_STEPS], vraw[DA_MAX_STEPS], craw[DA_MAX_STEPS], sraw[DA_MAX_STEPS];
  const int n_steps = (end - start + 15) >> 4;      // workgroup-uniform
#pragma unroll
  for (int it = 0; it < DA_MAX_STEPS; ++it) {
    if (it >= n_steps) break;
    kraw[it] = *reinterpret_cast<const u32x4_t*>(kp + (size_t)slots[it] * DA_D);
    vraw[it] = *reinterpret_cast<const u32x4_t*>(vp + (size_t)slots[it] * DA_D);
    const bf16_t* csr = rope_cs + (size_t)rows[it] * DA_D;
    craw[it] = *reinterpret_cast<const u32x4_t*>(csr + fc);
    sraw[it] = *reinterpret_cast<const u32x4_t*>(csr + 64 + fc);
  }

  // ---- query heads of this kv head, rotated at position L-1 (the row just appended)
  float qf[G][8];
  {
    float c[8], sn[8];
    unpack8(qc, c);
    unpack8(qs, sn);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      u32x4_t rp;
#pragma unroll
      for (int i = 0; i < 4; ++i) rp[i] = __shfl_xor(qraw[g][i], 8, 64);
      float x[8], xp[8];
      unpack8(qraw[g], x);
      unpack8(rp, xp);
      rope8(x, xp, c, sn, upper, qf[g]);
    }
  }

  float st_m[G], st_l[G], st_acc[G][8];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    st_m[g] = -1e30f;
    st_l[g] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) st_acc[g][i] = 0.f;
  }

#pragma unroll
  for (int it = 0; it < DA_MAX_STEPS; ++it) {
    if (it >= n_steps) break;
    const bool valid = start + it * 16 + wave * 4 + grp < end;
    u32x4_t kpr;
#pragma unroll
    for (int i = 0; i < 4; ++i) kpr[i] = __shfl_xor(kraw[it][i], 8, 64);
    float kx[8], kxp[8], c[8], sn[8], kr[8], vf[8];
    unpack8(kraw[it], kx);
    unpack8(kpr, kxp);
    unpack8(craw[it], c);
    unpack8(sraw[it], sn);
    unpack8(vraw[it], vf);
    rope8(kx, kxp, c, sn, upper, kr);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float d = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) d = fmaf(qf[g][i], kr[i], d);
      d += __shfl_xor(d, 1, 64);
      d += __shfl_xor(d, 2, 64);
      d += __shfl_xor(d, 4, 64);
      d += __shfl_xor(d, 8, 64);
      const float sc = valid ? d * scale : -1e30f;
      const float mn = fmaxf(st_m[g], sc);
      const float alpha = __expf(st_m[g] - mn);
      const float p = valid ? __expf(sc - mn) : 0.f;
      const float pb = rbf(p);
      st_l[g] = st_l[g] * alpha + p;
#pragma unroll
      for (int i = 0; i < 8; ++i) st_acc[g][i] = st_acc[g][i] * alpha + pb * vf[i];
      st_m[g] = mn;
    }
  }

  // ---- merge the 4 row groups of the wave (lanes s, s+16, s+32, s+48 hold the same d-chunk)
#pragma unroll
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
      const float mo = __shfl_xor(st_m[g], off, 64);
      const float lo = __shfl_xor(st_l[g], off, 64);
      const float mn = fmaxf(st_m[g], mo);
      const float a = __expf(st_m[g] - mn), b = __expf(mo - mn);
      st_l[g] = st_l[g] * a + lo * b;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float ao = __shfl_xor(st_acc[g][i], off, 64);
        st_acc[g][i] = st_acc[g][i] * a + ao * b;
      }
      st_m[g] = mn;
    }
  }

  // ---- merge the 4 waves through LDS
  __shared__ float sm_m[4][DA_GMAX], sm_l[4][DA_GMAX];
  __shared__ __attribute__((aligned(16))) float sm_acc[4][G][DA_D];
  if (grp == 0) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (s == 0) { sm_m[wave][g] = st_m[g]; sm_l[wave][g] = st_l[g]; }
      *reinterpret_cast<f32x4_t*>(&sm_acc[wave][g][s * 8]) = f32x4_t{st_acc[g][0], st_acc[g][1], st_acc[g][2], st_acc[g][3]};
      *reinterpret_cast<f32x4_t*>(&sm_acc[wave][g][s * 8 + 4]) = f32x4_t{st_acc[g][4], st_acc[g][5], st_acc[g][6], st_acc[g][7]};
    }
  }
  __syncthreads();
  const size_t part = (size_t)blockIdx.x * Hq;
  for (int idx = threadIdx.x; idx < G * DA_D; idx += 256) {
    const int g = idx / DA_D, d = idx % DA_D;
    float mn = sm_m[0][g];
#pragma unroll
    for (int w = 1; w < 4; ++w) mn = fmaxf(mn, sm_m[w][g]);
    float l = 0.f, a = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float e = __expf(sm_m[w][g] - mn);
      l += sm_l[w][g] * e;
      a += sm_acc[w][g][d] * e;
    }
    const int hq = kvh * G + g;
    ws_acc[(part + hq) * DA_D + d] = a;
    if (d == 0) { ws_m[part + hq] = mn; ws_l[part + hq] = l; }
  }
}

// One workgroup per q head; wave w merges splits w, w+4, ... with 4 loads in flight per lane, then the
// waves merge through LDS.  Lane owns d = 2*lane, 2*lane+1.
__global__ __launch_bounds__(256) void decode_attn_combine_kernel(const float* __restrict__ ws_m, const float* __restrict__ ws_l,
                                                                  const float* __restrict__ ws_acc, const int* __restrict__ len_dev,
                                                                  int len_add, bf16_t* __restrict__ out, int Hq, int chunk) {
  const int L = (len_dev ? *len_dev : 0) + len_add;
  const int ns = (L + chunk - 1) / chunk;
  const int hq = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // first batch of this wave's partials is requested before anything depends on the global max
  float m_[4], l_[4];
  float2 v_[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i0 = min(wave + 4 * u, ns - 1);
    const size_t p = (size_t)i0 * Hq + hq;
    m_[u] = ws_m[p];
    l_[u] = ws_l[p];
    v_[u] = *reinterpret_cast<const float2*>(ws_acc + p * DA_D + 2 * lane);
  }
  // global max over splits (every wave computes it redundantly: ns floats, L2-resident)
  float mx = -1e30f;
  for (int i = lane; i < ns; i += 64) mx = fmaxf(mx, ws_m[(size_t)i * Hq + hq]);
  mx = wave_max(mx);
  float l = 0.f, a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (wave + 4 * u < ns) {
      const float e = __expf(m_[u] - mx);
      l += l_[u] * e;
      a0 += v_[u].x * e;
      a1 += v_[u].y * e;
    }
  }
  int i = wave + 16;
  for (; i + 12 < ns; i += 16) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t p = (size_t)(i + 4 * u) * Hq + hq;
      m_[u] = ws_m[p];
      l_[u] = ws_l[p];
      v_[u] = *reinterpret_cast<const float2*>(ws_acc + p * DA_D + 2 * lane);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float e = __expf(m_[u] - mx);
      l += l_[u] * e;
      a0 += v_[u].x * e;
      a1 += v_[u].y * e;
    }
  }
  for (; i < ns; i += 4) {
    const size_t p = (size_t)i * Hq + hq;
    const float e = __expf(ws_m[p] - mx);
    const float2 v = *reinterpret_cast<const float2*>(ws_acc + p * DA_D + 2 * lane);
    l += ws_l[p] * e;
    a0 += v.x * e;
    a1 += v.y * e;
  }
  __shared__ float sl[4];
  __shared__ float sa[4][DA_D];
  if (lane == 0) sl[wave] = l;
  sa[wave][2 * lane] = a0;
  sa[wave][2 * lane + 1] = a1;
  __syncthreads();
  if (threadIdx.x < DA_D) {
    const int d = threadIdx.x;
    const float lt = sl[0] + sl[1] + sl[2] + sl[3];
    const float at = sa[0][d] + sa[1][d] + sa[2][d] + sa[3][d];
    out[(size_t)hq * DA_D + d] = f2bf(at / lt);
  }
}

// ws layout: [max_splits*Hq] m | [max_splits*Hq] l | [max_splits*Hq*128] acc   (floats)
extern "C" long long svlm_decode_attn_ws_bytes(int Hq, int max_len, int chunk) {
  if (Hq <= 0 || max_len <= 0 || chunk <= 0) return SVLM_EINVAL;
  const long long ns = (max_len + chunk - 1) / chunk;
  return ns * Hq * (2 + DA_D) * (long long)sizeof(float);
}

template <int G>
static void launch_split(dim3 grid, hipStream_t st, const bf16_t* q, const bf16_t* kp, const bf16_t* vp, const int* slot_of,
                         const bf16_t* cs, const int* len_dev, int len_add, float* ws_m, float* ws_l, float* ws_acc, int Hq, int Hkv,
                         int n_slots, int chunk, float scale, int max_len) {
  decode_attn_split_kernel<G><<<grid, 256, 0, st>>>(q, kp, vp, slot_of, cs, len_dev, len_add, ws_m, ws_l, ws_acc, Hq, Hkv, n_slots, chunk, scale, max_len);
}

extern "C" int svlm_decode_attn_ropeload(const void* q, const void* k_planes, const void* v_planes, const int* slot_of,
                                         const void* rope_cs, const int* len_dev, int len_add, void* out, void* ws,
                                         int Hq, int Hkv, int D, int n_slots, int max_len, int chunk, float scale, void* stream) {
  SVLM_CHECK_ARG(D == DA_D, "svlm_decode_attn_ropeload: head_dim %d unsupported (128 only)", D);
  SVLM_CHECK_ARG(Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && Hq / Hkv <= DA_GMAX, "svlm_decode_attn_ropeload: Hq=%d Hkv=%d (group must be <= %d)", Hq, Hkv, DA_GMAX);
  SVLM_CHECK_ARG(chunk > 0 && chunk % 16 == 0 && chunk <= 16 * DA_MAX_STEPS && max_len > 0 && n_slots > 0,
                 "svlm_decode_attn_ropeload: chunk=%d must be a multiple of 16 in [16, %d]", chunk, 16 * DA_MAX_STEPS);
  SVLM_CHECK_ARG(len_dev != nullptr || (len_add > 0 && len_add <= max_len), "svlm_decode_attn_ropeload: length %d outside (0, %d]", len_add, max_len);
  const int ns = (max_len + chunk - 1) / chunk;
  float* ws_m = (float*)ws;
  float* ws_l = ws_m + (size_t)ns * Hq;
  float* ws_acc = ws_l + (size_t)ns * Hq;
  dim3 grid(ns, Hkv);
  hipStream_t st = (hipStream_t)stream;
  const bf16_t *qq = (const bf16_t*)q, *kp = (const bf16_t*)k_planes, *vp = (const bf16_t*)v_planes, *cs = (const bf16_t*)rope_cs;
#define SVLM_DA_CASE(GG) \
  case GG: launch_split<GG>(grid, st, qq, kp, vp, slot_of, cs, len_dev, len_add, ws_m, ws_l, ws_acc, Hq, Hkv, n_slots, chunk, scale, max_len); break;
  switch (Hq / Hkv) {
    SVLM_DA_CASE(1) SVLM_DA_CASE(2) SVLM_DA_CASE(3) SVLM_DA_CASE(4) SVLM_DA_CASE(5) SVLM_DA_CASE(6) SVLM_DA_CASE(7) SVLM_DA_CASE(8)
  }
#undef SVLM_DA_CASE
  int rc = svlm_check_launch("svlm_decode_attn_ropeload(split)");
  if (rc) return rc;
  decode_attn_combine_kernel<<<Hq, 256, 0, st>>>(ws_m, ws_l, ws_acc, len_dev, len_add, (bf16_t*)out, Hq, chunk);
  return svlm_check_launch("svlm_decode_attn_ropeload(combine)");
}
