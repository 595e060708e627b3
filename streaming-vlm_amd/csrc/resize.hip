// Frame ingest, first half: antialiased bicubic resize of uint8 frames.
//
// The reference resizes every decoded clip on the host before the processor sees it
// (livecc_utils/src/livecc_utils/video_process_patch.py:134-153: torchvision's v1 tensor resize, BICUBIC, antialias=True, on the
// uint8 (T, C, H, W) clip from decord).  v1 casts to float32 and runs torch's CPU antialias kernel
// (aten/src/ATen/native/cpu/UpSampleKernel.cpp), then clamps, rounds half to even and casts back.  The arithmetic here is that
// kernel's, bit for bit -- its formulas (cubic_convolution1 / 2 of ATen/native/UpSample.h:400-407), the double intermediates its
// source's literals imply, and the FMA contraction of its build, all established against the installed torch (oracle/_c/resize_ref.c
// tells how): weights = fma-contracted cubics normalised in fp32; every tap sum is t = src0 * w0, then t = fma(src_j, w_j, t) in tap
// order; width pass first (uint8 -> fp32), height pass second (fp32 -> uint8).  Both passes are plain HBM streams: one thread per
// output value, taps read through L1/L2.
#include "common.h"

#include <math.h>

#pragma clang fp contract(off)      // every fused multiply-add below is an explicit fmaf

static float resize_cubic(float x) {
  const float A = -0.5f;
  x = fabsf(x);
  if (x < 1.f) return fmaf(fmaf(A + 2.f, x, -(A + 3.f)) * x, x, 1.f);
  if (x < 2.f) return fmaf(fmaf(fmaf(A, x, -(5.f * A)), x, 8.f * A), x, -(4.f * A));
  return 0.f;
}

// Host arithmetic only (no GPU): tap tables of ONE axis.  Returns K, the row stride of `wt` (taps per output, incl. zero
// padding), after filling xmin[out_size], xsize[out_size] and wt[out_size * K] when they are non-NULL (call once with NULL
// pointers to size the buffers); negative on bad arguments.  (HelperInterpBase::_compute_indices_min_size_weights_aa with
// scalar_t = float: the expressions with a double literal are evaluated in double and rounded where the source stores a float.)
extern "C" int svlm_resize_aa_tables(int in_size, int out_size, int* xmin, int* xsize, float* wt, int wt_stride) {
  if (in_size <= 0 || out_size <= 0) return SVLM_EINVAL;
  const float scale = (float)in_size / (float)out_size;
  const float support = scale >= 1.f ? (float)((4 * 0.5) * (double)scale) : (float)(4 * 0.5);
  const float invscale = scale >= 1.f ? (float)(1.0 / (double)scale) : 1.0f;
  const int K = (int)ceilf(support) * 2 + 1;
  if (!xmin && !xsize && !wt) return K;
  if (!xmin || !xsize || !wt || wt_stride < K) return SVLM_EINVAL;
  for (int i = 0; i < out_size; ++i) {
    const float center = (float)((double)scale * ((double)i + 0.5));
    long long lo = (long long)((double)(center - support) + 0.5);
    lo = lo > 0 ? lo : 0;
    long long n = (long long)((double)(center + support) + 0.5);
    n = (n < in_size ? n : in_size) - lo;
    n = n < 0 ? 0 : (n > K ? K : n);
    float tot = 0.f;
    float* w = wt + (size_t)i * wt_stride;
    for (int j = 0; j < n; ++j) {
      const float d = (float)(j + lo) - center;
      w[j] = resize_cubic((float)(((double)d + 0.5) * (double)invscale));
      tot = tot + w[j];
    }
    if (tot != 0.f)
      for (int j = 0; j < n; ++j) w[j] = w[j] / tot;
    for (int j = (int)n; j < wt_stride; ++j) w[j] = 0.f;
    xmin[i] = (int)lo;
    xsize[i] = (int)n;
  }
  return K;
}

// width pass: src uint8 (rows = planes*H, W) -> tmp fp32 (rows, w); blockIdx.y = row (folded over 65535), blockIdx.x = 256 outputs
__global__ __launch_bounds__(256) void resize_aa_width_kernel(const unsigned char* __restrict__ src, float* __restrict__ tmp,
                                                              const int* __restrict__ xmin, const int* __restrict__ xsize,
                                                              const float* __restrict__ wt, int K, long long rows, int W, int w) {
  const int xo = blockIdx.x * 256 + threadIdx.x;
  if (xo >= w) return;
  const int n = xsize[xo];
  const int x0 = xmin[xo];
  const float* wr = wt + (size_t)xo * K;
  for (long long row = blockIdx.y; row < rows; row += gridDim.y) {
    const unsigned char* s = src + row * W + x0;
    float acc = (float)s[0] * wr[0];
    for (int j = 1; j < n; ++j) acc = fmaf((float)s[j], wr[j], acc);
    tmp[row * w + xo] = n > 0 ? acc : 0.f;
  }
}

// height pass: tmp fp32 (planes, H, w) -> dst uint8 (planes, h, w), clamp + round half to even; blockIdx.y = plane*h + yo
__global__ __launch_bounds__(256) void resize_aa_height_kernel(const float* __restrict__ tmp, unsigned char* __restrict__ dst,
                                                               const int* __restrict__ ymin, const int* __restrict__ ysize,
                                                               const float* __restrict__ wt, int K, long long planes, int H, int h,
                                                               int w) {
  const int xo = blockIdx.x * 256 + threadIdx.x;
  if (xo >= w) return;
  for (long long r = blockIdx.y; r < planes * h; r += gridDim.y) {
    const int yo = (int)(r % h);
    const long long p = r / h;
    const float* s = tmp + (p * H + ymin[yo]) * w + xo;
    const float* wr = wt + (size_t)yo * K;
    const int n = ysize[yo];
    float acc = s[0] * wr[0];
    for (int j = 1; j < n; ++j) acc = fmaf(s[(size_t)j * w], wr[j], acc);
    acc = n > 0 ? acc : 0.f;
    acc = fminf(fmaxf(acc, 0.f), 255.f);
    dst[r * w + xo] = (unsigned char)rintf(acc);
  }
}

extern "C" long long svlm_resize_ws_bytes(int planes, int H, int w) {
  if (planes <= 0 || H <= 0 || w <= 0) return SVLM_EINVAL;
  return (long long)planes * H * w * (long long)sizeof(float);
}

extern "C" int svlm_resize_bicubic_aa_u8(const void* src, void* dst, int planes, int H, int W, int h, int w, const int* xmin,
                                         const int* xsize, const float* wt_x, int Kx, const int* ymin, const int* ysize,
                                         const float* wt_y, int Ky, void* ws, long long ws_bytes, void* stream) {
  SVLM_CHECK_ARG(planes > 0 && H > 0 && W > 0 && h > 0 && w > 0, "svlm_resize_bicubic_aa_u8: bad shape planes=%d %dx%d -> %dx%d", planes, H,
                 W, h, w);
  SVLM_CHECK_ARG(Kx > 0 && Ky > 0 && xmin && xsize && wt_x && ymin && ysize && wt_y, "svlm_resize_bicubic_aa_u8: missing tap tables (Kx=%d Ky=%d)", Kx, Ky);
  SVLM_CHECK_ARG(ws && ws_bytes >= svlm_resize_ws_bytes(planes, H, w), "svlm_resize_bicubic_aa_u8: workspace %lld < %lld bytes", ws_bytes,
                 svlm_resize_ws_bytes(planes, H, w));
  hipStream_t st = (hipStream_t)stream;
  const long long rows = (long long)planes * H;
  const long long n1 = rows * w, n2 = (long long)planes * h * w;
  SVLM_CHECK_ARG(n1 < (1LL << 39) && n2 < (1LL << 39), "svlm_resize_bicubic_aa_u8: clip too large (%lld values)", n1 > n2 ? n1 : n2);
  const unsigned gx = (unsigned)((w + 255) / 256);
  resize_aa_width_kernel<<<dim3(gx, (unsigned)(rows < 65535 ? rows : 65535)), 256, 0, st>>>((const unsigned char*)src, (float*)ws, xmin, xsize, wt_x, Kx, rows, W, w);
  int rc = svlm_check_launch("svlm_resize_bicubic_aa_u8(width)");
  if (rc) return rc;
  const long long r2 = (long long)planes * h;
  resize_aa_height_kernel<<<dim3(gx, (unsigned)(r2 < 65535 ? r2 : 65535)), 256, 0, st>>>((const float*)ws, (unsigned char*)dst, ymin, ysize, wt_y, Ky, planes, H, h, w);
  return svlm_check_launch("svlm_resize_bicubic_aa_u8(height)");
}
