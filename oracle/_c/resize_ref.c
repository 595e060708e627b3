/* TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): plain-C restatement of the frame resize in front of the path.
 *
 * The reference resizes every decoded uint8 clip ON THE HOST with torchvision's v1 tensor resize, bicubic with antialiasing
 * (livecc_utils/src/livecc_utils/video_process_patch.py:145-152: transforms.functional.resize(video, [h, w], BICUBIC, antialias=True)):
 * v1 casts a uint8 tensor to float32, runs torch's CPU kernel and rounds back.  torchvision is absent here; what it calls is
 * torch.nn.functional.interpolate(x.float(), size, mode="bicubic", antialias=True, align_corners=False), i.e. ATen's CPU kernel
 * (aten/src/ATen/native/cpu/UpSampleKernel.cpp: HelperInterpCubic::aa_filter, _compute_indices_min_size_weights_aa,
 * basic_loop_aa_horizontal / _vertical; the polynomials are cubic_convolution1 / 2 of ATen/native/UpSample.h:400-407).
 *
 * The arithmetic below is that kernel's, found EMPIRICALLY against the installed torch (the .cpp is not shipped, only its headers):
 * one-hot rows through F.interpolate give the tap weights bit for bit, full images the accumulation order.  What matched:
 *   - the C promotion rules of the published source: `scale * (i + 0.5)`, `1.0 / scale`, `(j + xmin - center + 0.5) * invscale`
 *     pass through double because of their double literals, and are rounded to float where the source assigns a scalar_t;
 *   - the compiler's FMA contraction (the kernels are built with -mfma): cubic_convolution1 = fma(fma(A+2, x, -(A+3)) * x, x, 1),
 *     cubic_convolution2 = fma(fma(fma(A, x, -5A), x, 8A), x, -4A), and every tap sum is t = src0 * w0; t = fma(src_j, w_j, t);
 *   - width pass first, then height, each in fp32.
 * tests/test_resize.py holds this file to EXACT equality with F.interpolate (fp32 bits and the rounded uint8) on every case,
 * against the installed torch and against a committed fixture of its outputs (tests/golden/resize_torch_cpu.npz).
 */
#include <math.h>
#include <stdint.h>

static float aa_cubic(float x) {
  const float A = -0.5f;
  x = fabsf(x);
  if (x < 1.0f) return fmaf(fmaf(A + 2.0f, x, -(A + 3.0f)) * x, x, 1.0f);
  if (x < 2.0f) return fmaf(fmaf(fmaf(A, x, -(5.0f * A)), x, 8.0f * A), x, -(4.0f * A));
  return 0.0f;
}

/* Tap tables of one axis.  Returns K (taps per output incl. zero padding); fills the tables when they are non-NULL. */
int svlm_ref_aa_tables(int in_size, int out_size, int* xmin, int* xsize, float* wt, int wt_stride) {
  if (in_size <= 0 || out_size <= 0) return -22;
  const float scale = (float)in_size / (float)out_size;                       /* area_pixel_compute_scale<float> */
  const float support = scale >= 1.0f ? (float)((4 * 0.5) * (double)scale) : (float)(4 * 0.5);
  const float invscale = scale >= 1.0f ? (float)(1.0 / (double)scale) : 1.0f;
  const int K = (int)ceilf(support) * 2 + 1;
  if (!xmin && !xsize && !wt) return K;
  if (!xmin || !xsize || !wt || wt_stride < K) return -22;
  for (int i = 0; i < out_size; ++i) {
    const float center = (float)((double)scale * ((double)i + 0.5));
    int64_t lo = (int64_t)((double)(center - support) + 0.5);
    if (lo < 0) lo = 0;
    int64_t n = (int64_t)((double)(center + support) + 0.5);
    if (n > in_size) n = in_size;
    n -= lo;
    if (n < 0) n = 0;
    if (n > K) n = K;
    float* w = wt + (long)i * wt_stride;
    float tot = 0.0f;
    for (int j = 0; j < n; ++j) {
      const float d = (float)(j + lo) - center;
      w[j] = aa_cubic((float)(((double)d + 0.5) * (double)invscale));
      tot += w[j];
    }
    if (tot != 0.0f)
      for (int j = 0; j < n; ++j) w[j] /= tot;
    for (int j = (int)n; j < wt_stride; ++j) w[j] = 0.0f;
    xmin[i] = (int)lo;
    xsize[i] = (int)n;
  }
  return K;
}

/* out[r][i] = sum_j in[r][xmin[i] + j] * wt[i][j] for `rows` rows of `n_in` values (stride n_in), fp32, FMA chain in tap order */
void svlm_ref_resize_rows(const float* in, long rows, int n_in, float* out, int n_out, const int* xmin, const int* xsize, const float* wt, int K) {
  for (long r = 0; r < rows; ++r) {
    const float* s = in + r * n_in;
    float* o = out + r * n_out;
    for (int i = 0; i < n_out; ++i) {
      const float* w = wt + (long)i * K;
      const float* p = s + xmin[i];
      float t = 0.0f;
      if (xsize[i] > 0) {
        t = p[0] * w[0];
        for (int j = 1; j < xsize[i]; ++j) t = fmaf(p[j], w[j], t);
      }
      o[i] = t;
    }
  }
}
