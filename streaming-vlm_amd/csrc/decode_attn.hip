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
// HBM-bound (2*L*Hkv*D*2 B of K/V per layer-step).  Split-KV (flash-decoding): grid =
// (n_splits, Hkv); a workgroup owns `chunk` consecutive logical keys; 16 lanes hold one
// 256-B row (16 B per lane), 4 rows per wave-load, 16 rows per workgroup step; wave shuffles
// reduce the dots (xor 1,2,4,8) and merge the online-softmax states (xor 16,32); the four
// waves merge through LDS; a second tiny kernel merges the splits.
#include "common.h"

#define DA_D 128
#define DA_GMAX 8

struct DaState {  // per-lane online-softmax state for one q head
  float m, l;
  float acc[8];
};

__device__ __forceinline__ void rope8(const float (&x)[8], const float (&xp)[8], const float (&c)[8], const float (&s)[8],
                                      bool upper, float (&out)[8]) {
  // out = bf16(bf16(x*cos) + bf16(rotate_half(x)*sin)); rotate_half = (-x2, x1): lower half pairs with -partner
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float rot = upper ? xp[i] : -xp[i];
    out[i] = rbf(rbf(x[i] * c[i]) + rbf(rot * s[i]));
  }
}

__global__ __launch_bounds__(256) void decode_attn_split_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k_planes, const bf16_t* __restrict__ v_planes,
    const int* __restrict__ slot_of, const bf16_t* __restrict__ rope_cs, const int* __restrict__ len_dev, int len_add,
    float* __restrict__ ws_m, float* __restrict__ ws_l, float* __restrict__ ws_acc, int Hq, int Hkv, int n_slots,
    int chunk, float scale) {
  const int L = (len_dev ? *len_dev : 0) + len_add;
  const int start = blockIdx.x * chunk;
  if (start >= L) return;
  const int kvh = blockIdx.y;
  const int G = Hq / Hkv;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane >> 4, s = lane & 15;
  const bool upper = s >= 8;
  const int fc = (s & 7) * 8;  // frequency chunk

  // ---- query heads of this kv head, rotated at position L-1 (the row just appended)
  float qf[DA_GMAX][8];
  {
    const bf16_t* csr = rope_cs + (size_t)(L - 1) * DA_D;
    float c[8], sn[8];
    unpack8(*reinterpret_cast<const u32x4_t*>(csr + fc), c);
    unpack8(*reinterpret_cast<const u32x4_t*>(csr + 64 + fc), sn);
#pragma unroll
    for (int g = 0; g < DA_GMAX; ++g) {
      if (g < G) {
        u32x4_t raw = *reinterpret_cast<const u32x4_t*>(q + (size_t)(kvh * G + g) * DA_D + s * 8);
        u32x4_t rp;
#pragma unroll
        for (int i = 0; i < 4; ++i) rp[i] = __shfl_xor(raw[i], 8, 64);
        float x[8], xp[8];
        unpack8(raw, x);
        unpack8(rp, xp);
        rope8(x, xp, c, sn, upper, qf[g]);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) qf[g][i] = 0.f;
      }
    }
  }

  DaState st[DA_GMAX];
#pragma unroll
  for (int g = 0; g < DA_GMAX; ++g) {
    st[g].m = -1e30f;
    st[g].l = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) st[g].acc[i] = 0.f;
  }

  const bf16_t* kp = k_planes + (size_t)kvh * n_slots * DA_D;
  const bf16_t* vp = v_planes + (size_t)kvh * n_slots * DA_D;
  const int end = min(start + chunk, L);
  for (int r0 = start; r0 < end; r0 += 16) {
    const int r = r0 + wave * 4 + grp;
    const bool valid = r < end;
    const int rr = valid ? r : end - 1;
    const int slot = slot_of[rr];
    const u32x4_t kraw = *reinterpret_cast<const u32x4_t*>(kp + (size_t)slot * DA_D + s * 8);
    const u32x4_t vraw = *reinterpret_cast<const u32x4_t*>(vp + (size_t)slot * DA_D + s * 8);
    const bf16_t* csr = rope_cs + (size_t)rr * DA_D;
    const u32x4_t craw = *reinterpret_cast<const u32x4_t*>(csr + fc);
    const u32x4_t sraw = *reinterpret_cast<const u32x4_t*>(csr + 64 + fc);
    u32x4_t kpr;
#pragma unroll
    for (int i = 0; i < 4; ++i) kpr[i] = __shfl_xor(kraw[i], 8, 64);
    float kx[8], kxp[8], c[8], sn[8], kr[8], vf[8];
    unpack8(kraw, kx);
    unpack8(kpr, kxp);
    unpack8(craw, c);
    unpack8(sraw, sn);
    unpack8(vraw, vf);
    rope8(kx, kxp, c, sn, upper, kr);
#pragma unroll
    for (int g = 0; g < DA_GMAX; ++g) {
      if (g < G) {
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) d = fmaf(qf[g][i], kr[i], d);
        d += __shfl_xor(d, 1, 64);
        d += __shfl_xor(d, 2, 64);
        d += __shfl_xor(d, 4, 64);
        d += __shfl_xor(d, 8, 64);
        const float sc = valid ? d * scale : -1e30f;
        const float mn = fmaxf(st[g].m, sc);
        const float alpha = __expf(st[g].m - mn);
        const float p = valid ? __expf(sc - mn) : 0.f;
        const float pb = rbf(p);
        st[g].l = st[g].l * alpha + p;
#pragma unroll
        for (int i = 0; i < 8; ++i) st[g].acc[i] = st[g].acc[i] * alpha + pb * vf[i];
        st[g].m = mn;
      }
    }
  }

  // ---- merge the 4 row groups of the wave (lanes s, s+16, s+32, s+48 hold the same d-chunk)
#pragma unroll
  for (int g = 0; g < DA_GMAX; ++g) {
    if (g < G) {
#pragma unroll
      for (int off = 16; off <= 32; off <<= 1) {
        const float mo = __shfl_xor(st[g].m, off, 64);
        const float lo = __shfl_xor(st[g].l, off, 64);
        const float mn = fmaxf(st[g].m, mo);
        const float a = __expf(st[g].m - mn), b = __expf(mo - mn);
        st[g].l = st[g].l * a + lo * b;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float ao = __shfl_xor(st[g].acc[i], off, 64);
          st[g].acc[i] = st[g].acc[i] * a + ao * b;
        }
        st[g].m = mn;
      }
    }
  }

  // ---- merge the 4 waves through LDS
  __shared__ float sm_m[4][DA_GMAX], sm_l[4][DA_GMAX];
  __shared__ float sm_acc[4][DA_GMAX][DA_D];
  if (grp == 0) {
#pragma unroll
    for (int g = 0; g < DA_GMAX; ++g) {
      if (g < G) {
        if (s == 0) { sm_m[wave][g] = st[g].m; sm_l[wave][g] = st[g].l; }
#pragma unroll
        for (int i = 0; i < 8; ++i) sm_acc[wave][g][s * 8 + i] = st[g].acc[i];
      }
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

__global__ __launch_bounds__(128) void decode_attn_combine_kernel(const float* __restrict__ ws_m, const float* __restrict__ ws_l,
                                                                  const float* __restrict__ ws_acc, const int* __restrict__ len_dev,
                                                                  int len_add, bf16_t* __restrict__ out, int Hq, int chunk) {
  const int L = (len_dev ? *len_dev : 0) + len_add;
  const int ns = (L + chunk - 1) / chunk;
  const int hq = blockIdx.x, d = threadIdx.x;
  float mn = -1e30f;
  for (int i = 0; i < ns; ++i) mn = fmaxf(mn, ws_m[(size_t)i * Hq + hq]);
  float l = 0.f, a = 0.f;
  for (int i = 0; i < ns; ++i) {
    const float e = __expf(ws_m[(size_t)i * Hq + hq] - mn);
    l += ws_l[(size_t)i * Hq + hq] * e;
    a += ws_acc[((size_t)i * Hq + hq) * DA_D + d] * e;
  }
  out[(size_t)hq * DA_D + d] = f2bf(a / l);
}

// ws layout: [max_splits*Hq] m | [max_splits*Hq] l | [max_splits*Hq*128] acc   (floats)
extern "C" long long svlm_decode_attn_ws_bytes(int Hq, int max_len, int chunk) {
  if (Hq <= 0 || max_len <= 0 || chunk <= 0) return SVLM_EINVAL;
  const long long ns = (max_len + chunk - 1) / chunk;
  return ns * Hq * (2 + DA_D) * (long long)sizeof(float);
}

extern "C" int svlm_decode_attn_ropeload(const void* q, const void* k_planes, const void* v_planes, const int* slot_of,
                                         const void* rope_cs, const int* len_dev, int len_add, void* out, void* ws,
                                         int Hq, int Hkv, int D, int n_slots, int max_len, int chunk, float scale, void* stream) {
  SVLM_CHECK_ARG(D == DA_D, "svlm_decode_attn_ropeload: head_dim %d unsupported (128 only)", D);
  SVLM_CHECK_ARG(Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && Hq / Hkv <= DA_GMAX, "svlm_decode_attn_ropeload: Hq=%d Hkv=%d (group must be <= %d)", Hq, Hkv, DA_GMAX);
  SVLM_CHECK_ARG(chunk > 0 && chunk % 16 == 0 && max_len > 0 && n_slots > 0, "svlm_decode_attn_ropeload: chunk=%d must be a positive multiple of 16", chunk);
  SVLM_CHECK_ARG(len_dev != nullptr || (len_add > 0 && len_add <= max_len), "svlm_decode_attn_ropeload: length %d outside (0, %d]", len_add, max_len);
  const int ns = (max_len + chunk - 1) / chunk;
  float* ws_m = (float*)ws;
  float* ws_l = ws_m + (size_t)ns * Hq;
  float* ws_acc = ws_l + (size_t)ns * Hq;
  dim3 grid(ns, Hkv);
  decode_attn_split_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)q, (const bf16_t*)k_planes, (const bf16_t*)v_planes, slot_of,
                                                                 (const bf16_t*)rope_cs, len_dev, len_add, ws_m, ws_l, ws_acc, Hq, Hkv, n_slots, chunk, scale);
  int rc = svlm_check_launch("svlm_decode_attn_ropeload(split)");
  if (rc) return rc;
  decode_attn_combine_kernel<<<Hq, DA_D, 0, (hipStream_t)stream>>>(ws_m, ws_l, ws_acc, len_dev, len_add, (bf16_t*)out, Hq, chunk);
  return svlm_check_launch("svlm_decode_attn_ropeload(combine)");
}
