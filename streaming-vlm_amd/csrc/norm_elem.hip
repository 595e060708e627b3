// RMSNorm / LayerNorm / elementwise / row-gather kernels.  HBM-bound: one wave per row for
// the norms (16-B vector loads, wave shuffle reductions), grid-stride elementwise ops.
#include "common.h"

// ---------------------------------------------------------------- RMSNorm
// Qwen2RMSNorm (reference call sites qwen2/language_forward.py:183,200,315):
//   var = mean(float(x)^2); y = w * bf16(float(x) * rsqrt(var + eps))     (two roundings)
// One wave per row; cols % 8 == 0.
__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                      bf16_t* __restrict__ y, int rows, int cols, float eps) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const bf16_t* xr = x + (size_t)row * cols;
  float ss = 0.f;
  for (int c = lane * 8; c < cols; c += 512) {
    u32x4_t v = *reinterpret_cast<const u32x4_t*>(xr + c);
    float f[8];
    unpack8(v, f);
#pragma unroll
    for (int i = 0; i < 8; ++i) ss += f[i] * f[i];
  }
  ss = wave_sum(ss);
  const float r = rsqrtf(ss / (float)cols + eps);
  bf16_t* yr = y + (size_t)row * cols;
  for (int c = lane * 8; c < cols; c += 512) {
    u32x4_t v = *reinterpret_cast<const u32x4_t*>(xr + c);
    u32x4_t wv = *reinterpret_cast<const u32x4_t*>(w + c);
    float f[8], g[8];
    unpack8(v, f);
    unpack8(wv, g);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = g[i] * rbf(f[i] * r);
    *reinterpret_cast<u32x4_t*>(yr + c) = pack8(f);
  }
}

// ---------------------------------------------------------------- LayerNorm
// torch LayerNorm on bf16 (ViT norm1/norm2/ln_q, qwen2/vision_forward.py:43-49): fp32 mean/var,
// y = bf16((x - mean) * rstd * w + b)   (one rounding).  One wave per row.
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                        const bf16_t* __restrict__ b, bf16_t* __restrict__ y,
                                                        int rows, int cols, float eps) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const bf16_t* xr = x + (size_t)row * cols;
  float s = 0.f;
  for (int c = lane * 8; c < cols; c += 512) {
    u32x4_t v = *reinterpret_cast<const u32x4_t*>(xr + c);
    float f[8];
    unpack8(v, f);
#pragma unroll
    for (int i = 0; i < 8; ++i) s += f[i];
  }
  const float mean = wave_sum(s) / (float)cols;
  float ss = 0.f;
  for (int c = lane * 8; c < cols; c += 512) {
    u32x4_t v = *reinterpret_cast<const u32x4_t*>(xr + c);
    float f[8];
    unpack8(v, f);
#pragma unroll
    for (int i = 0; i < 8; ++i) { float d = f[i] - mean; ss += d * d; }
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)cols + eps);
  bf16_t* yr = y + (size_t)row * cols;
  for (int c = lane * 8; c < cols; c += 512) {
    u32x4_t v = *reinterpret_cast<const u32x4_t*>(xr + c);
    u32x4_t wv = *reinterpret_cast<const u32x4_t*>(w + c);
    u32x4_t bv = *reinterpret_cast<const u32x4_t*>(b + c);
    float f[8], g[8], h[8];
    unpack8(v, f);
    unpack8(wv, g);
    unpack8(bv, h);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (f[i] - mean) * rstd * g[i] + h[i];
    *reinterpret_cast<u32x4_t*>(yr + c) = pack8(f);
  }
}

// ---------------------------------------------------------------- elementwise
// y = bf16(a + b)   (residual add)
__global__ void add_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ y, size_t n8) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    u32x4_t va = reinterpret_cast<const u32x4_t*>(a)[i], vb = reinterpret_cast<const u32x4_t*>(b)[i];
    float fa[8], fb[8];
    unpack8(va, fa);
    unpack8(vb, fb);
#pragma unroll
    for (int k = 0; k < 8; ++k) fa[k] += fb[k];
    reinterpret_cast<u32x4_t*>(y)[i] = pack8(fa);
  }
}

// Qwen2MLP gate/up epilogue (qwen2/language_forward.py:201): h = bf16(bf16(silu(g)) * u),
// gu rows are [gate(I) | up(I)] (the fused gate_up GEMM output).
__global__ void silu_mul_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ h, int rows, int inter) {
  const int per_row = inter / 8;
  const size_t total = (size_t)rows * per_row;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / per_row;
    const int c = (int)(i % per_row) * 8;
    u32x4_t vg = *reinterpret_cast<const u32x4_t*>(gu + r * 2 * inter + c);
    u32x4_t vu = *reinterpret_cast<const u32x4_t*>(gu + r * 2 * inter + inter + c);
    float g[8], u[8];
    unpack8(vg, g);
    unpack8(vu, u);
#pragma unroll
    for (int k = 0; k < 8; ++k) g[k] = apply_act(g[k], SVLM_ACT_SILU) * u[k];
    *reinterpret_cast<u32x4_t*>(h + r * inter + c) = pack8(g);
  }
}

// out[t] = idx[t] >= 0 ? table[idx[t]] : alt[-1 - idx[t]]   (embed_tokens gather + masked_scatter of the
// vision rows in one pass; qwen2/model_forward.py:34,62-69).  idx may live behind `idx_base[*idx_off]`.
__global__ void gather_rows_kernel(const bf16_t* __restrict__ table, const bf16_t* __restrict__ alt,
                                   const int* __restrict__ idx, const int* __restrict__ idx_off,
                                   bf16_t* __restrict__ out, int rows, int cols) {
  const int per_row = cols / 8;
  const int off = idx_off ? *idx_off : 0;
  const size_t total = (size_t)rows * per_row;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / per_row), c = (int)(i % per_row) * 8;
    const int id = idx[off + r];
    const bf16_t* src = id >= 0 ? table + (size_t)id * cols : alt + (size_t)(-1 - id) * cols;
    *reinterpret_cast<u32x4_t*>(out + (size_t)r * cols + c) = *reinterpret_cast<const u32x4_t*>(src + c);
  }
}

// ---------------------------------------------------------------- ViT 2-D rope, in place on the fused qkv buffer
// apply_rotary_pos_emb_vision (transformers modeling_qwen2_vl.py:225-236; call qwen2/vision_forward.py:27):
//   fp32:  x*cos + rotate_half(x)*sin, one rounding.  qkv is (N, 3, H, d); cos/sin are fp32 (N, d/2)
//   (the table's two halves are identical: emb = cat(freqs, freqs)).
__global__ void vit_rope_kernel(bf16_t* __restrict__ qkv, const float* __restrict__ cosT, const float* __restrict__ sinT,
                                int N, int H, int d) {
  const int half = d / 2;
  const size_t total = (size_t)N * 2 * H * half;     // q and k only
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % half);
    size_t t = i / half;
    const int h = (int)(t % H);
    t /= H;
    const int which = (int)(t % 2);
    const int n = (int)(t / 2);
    bf16_t* p = qkv + (((size_t)n * 3 + which) * H + h) * d;
    const float c = cosT[(size_t)n * half + j], s = sinT[(size_t)n * half + j];
    const float x1 = bf2f(p[j]), x2 = bf2f(p[j + half]);
    p[j] = f2bf(x1 * c - x2 * s);
    p[j + half] = f2bf(x2 * c + x1 * s);
  }
}

// ---------------------------------------------------------------- Infinity-Cache warm-up
// Streams a byte range through plain 16-B loads and throws the data away: the lines stay in the 256 MiB
// Infinity Cache.  Launched on a side stream for the NEXT layer's weights while the current layer's
// latency-bound kernels (QKV, attention, o_proj) leave HBM idle; purely a performance hint (no result,
// nothing depends on it).
__global__ __launch_bounds__(256) void prefetch_kernel(const u32x4_t* __restrict__ p, long n16) {
  const long stride = (long)gridDim.x * 256;
  long i = blockIdx.x * 256L + threadIdx.x;
  for (; i + 7 * stride < n16; i += 8 * stride) {
    u32x4_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[i + u * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) asm volatile("" ::"v"(v[u]));
  }
  for (; i < n16; i += stride) {
    u32x4_t v = p[i];
    asm volatile("" ::"v"(v));
  }
}

// ================================================================ host launchers
static inline int grid_for(size_t n, int block) {
  size_t g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

extern "C" int svlm_rmsnorm(const void* x, const void* w, void* y, int rows, int cols, float eps, void* stream) {
  SVLM_CHECK_ARG(rows >= 0 && cols > 0 && cols % 8 == 0, "svlm_rmsnorm: cols=%d must be a positive multiple of 8", cols);
  if (rows == 0) return SVLM_OK;
  rmsnorm_kernel<<<(rows + 3) / 4, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, rows, cols, eps);
  return svlm_check_launch("svlm_rmsnorm");
}

extern "C" int svlm_layernorm(const void* x, const void* w, const void* b, void* y, int rows, int cols, float eps, void* stream) {
  SVLM_CHECK_ARG(rows >= 0 && cols > 0 && cols % 8 == 0, "svlm_layernorm: cols=%d must be a positive multiple of 8", cols);
  if (rows == 0) return SVLM_OK;
  layernorm_kernel<<<(rows + 3) / 4, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, rows, cols, eps);
  return svlm_check_launch("svlm_layernorm");
}

extern "C" int svlm_add(const void* a, const void* b, void* y, long long n, void* stream) {
  SVLM_CHECK_ARG(n >= 0 && n % 8 == 0, "svlm_add: n=%lld must be a multiple of 8", n);
  if (n == 0) return SVLM_OK;
  add_kernel<<<grid_for(n / 8, 256), 256, 0, (hipStream_t)stream>>>((const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y, (size_t)n / 8);
  return svlm_check_launch("svlm_add");
}

extern "C" int svlm_silu_mul(const void* gu, void* h, int rows, int inter, void* stream) {
  SVLM_CHECK_ARG(rows >= 0 && inter > 0 && inter % 8 == 0, "svlm_silu_mul: inter=%d must be a positive multiple of 8", inter);
  if (rows == 0) return SVLM_OK;
  silu_mul_kernel<<<grid_for((size_t)rows * inter / 8, 256), 256, 0, (hipStream_t)stream>>>((const bf16_t*)gu, (bf16_t*)h, rows, inter);
  return svlm_check_launch("svlm_silu_mul");
}

extern "C" int svlm_gather_rows(const void* table, const void* alt, const int* idx, const int* idx_off, void* out,
                                int rows, int cols, void* stream) {
  SVLM_CHECK_ARG(rows >= 0 && cols > 0 && cols % 8 == 0, "svlm_gather_rows: cols=%d must be a positive multiple of 8", cols);
  if (rows == 0) return SVLM_OK;
  gather_rows_kernel<<<grid_for((size_t)rows * cols / 8, 256), 256, 0, (hipStream_t)stream>>>(
      (const bf16_t*)table, (const bf16_t*)alt, idx, idx_off, (bf16_t*)out, rows, cols);
  return svlm_check_launch("svlm_gather_rows");
}

extern "C" int svlm_vit_rope(void* qkv, const float* cosT, const float* sinT, int N, int H, int d, void* stream) {
  SVLM_CHECK_ARG(N >= 0 && H > 0 && d > 0 && d % 2 == 0, "svlm_vit_rope: bad shape N=%d H=%d d=%d", N, H, d);
  if (N == 0) return SVLM_OK;
  vit_rope_kernel<<<grid_for((size_t)N * 2 * H * (d / 2), 256), 256, 0, (hipStream_t)stream>>>((bf16_t*)qkv, cosT, sinT, N, H, d);
  return svlm_check_launch("svlm_vit_rope");
}

extern "C" int svlm_prefetch(const void* ptr, long long bytes, int n_wgs, void* stream) {
  SVLM_CHECK_ARG(ptr != nullptr && bytes >= 0 && n_wgs > 0 && ((uintptr_t)ptr & 15) == 0, "svlm_prefetch: bad range");
  if (bytes < 16) return SVLM_OK;
  prefetch_kernel<<<n_wgs, 256, 0, (hipStream_t)stream>>>((const u32x4_t*)ptr, bytes / 16);
  return svlm_check_launch("svlm_prefetch");
}

// ---------------------------------------------------------------- frame ingest
// uint8 frames (T, 3, H, W) -> bf16 patches (gt*gh*gw, 3*TP*P*P): rescale 1/255, per-channel normalise and the
// merge-block-major patch order of the Qwen2-VL video processor (the 2x2 patches that the merger fuses are consecutive
// rows); a trailing odd frame is repeated to fill its temporal patch.  One workgroup per output row.
__global__ __launch_bounds__(256) void patchify_u8_kernel(const unsigned char* __restrict__ frames, bf16_t* __restrict__ out,
                                                          int T, int H, int W, int P, int TP, int MG, float m0, float m1, float m2,
                                                          float s0, float s1, float s2) {
  const int gh = H / P, gw = W / P;
  const int row = blockIdx.x;
  // row = ((t * gh/MG + hb) * gw/MG + wb) * MG*MG + mh*MG + mw
  const int mm = row % (MG * MG), blk = row / (MG * MG);
  const int wb = blk % (gw / MG), hb = (blk / (gw / MG)) % (gh / MG), t = blk / ((gw / MG) * (gh / MG));
  const int ph0 = (hb * MG + mm / MG) * P, pw0 = (wb * MG + mm % MG) * P;
  const int cols = 3 * TP * P * P;
  for (int col = threadIdx.x; col < cols; col += 256) {
    const int pw = col % P, ph = (col / P) % P, tt = (col / (P * P)) % TP, c = col / (P * P * TP);
    const int f = min(t * TP + tt, T - 1);
    const float x = (float)frames[(((size_t)f * 3 + c) * H + ph0 + ph) * W + pw0 + pw] / 255.0f;
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    out[(size_t)row * cols + col] = f2bf((x - mean) / sd);
  }
}

extern "C" int svlm_patchify_u8(const void* frames, void* out, int T, int H, int W, int patch, int temporal, int merge,
                                float m0, float m1, float m2, float s0, float s1, float s2, void* stream) {
  SVLM_CHECK_ARG(T > 0 && patch > 0 && temporal > 0 && merge > 0, "svlm_patchify_u8: bad T=%d patch=%d temporal=%d merge=%d", T, patch, temporal, merge);
  SVLM_CHECK_ARG(H > 0 && W > 0 && H % (patch * merge) == 0 && W % (patch * merge) == 0,
                 "svlm_patchify_u8: frame %dx%d is not a multiple of %d", H, W, patch * merge);
  SVLM_CHECK_ARG(s0 != 0.f && s1 != 0.f && s2 != 0.f, "svlm_patchify_u8: zero std");
  const int gt = (T + temporal - 1) / temporal;
  const long long rows = (long long)gt * (H / patch) * (W / patch);
  SVLM_CHECK_ARG(rows < (1LL << 31), "svlm_patchify_u8: too many patches");
  patchify_u8_kernel<<<(int)rows, 256, 0, (hipStream_t)stream>>>((const unsigned char*)frames, (bf16_t*)out, T, H, W, patch, temporal, merge,
                                                                 m0, m1, m2, s0, s1, s2);
  return svlm_check_launch("svlm_patchify_u8");
}
