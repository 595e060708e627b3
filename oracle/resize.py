"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the frame resize in front of the path.

The reference resizes every decoded uint8 clip with torchvision's tensor resize, bicubic with antialiasing, before the HF
processor sees it (livecc_utils/src/livecc_utils/video_process_patch.py:134-153: `smart_resize` to a multiple of 28 inside the
pixel budget, then `transforms.functional.resize(video, [h, w], BICUBIC, antialias=True).float()`).  torchvision is absent
here; its tensor path is `torch.nn.functional.interpolate(x.float(), size, mode="bicubic", antialias=True,
align_corners=False)` followed by clamp(0, 255), round-half-even and the cast back to uint8.  The separable filter below
restates torch's published helper formulas (`torch/include/ATen/native/hip/UpSample.cuh`, namespace upsample_antialias:
`_compute_weights_span`, `_compute_weights`, `BicubicFilterFunctor`, `interpolate_aa_single_dim`), in fp32, width first then
height.

Pinning: tests/test_resize.py runs `F.interpolate` of the installed torch on CPU beside this file.  torch's CPU kernel sums the
taps in an association that its headers do not publish, so the fp32 intermediate differs in the last bits and, after rounding
to uint8, about 2 pixels in 100 000 differ by ONE level (never more); the test bounds exactly that.  The HIP kernel
(svlm_resize_bicubic_aa_u8) follows THIS file's operation order and is compared bit for bit.
"""
from __future__ import annotations

import math

import numpy as np

f32 = np.float32


def smart_resize(height: int, width: int, factor: int = 28, min_pixels: int = 100 * 28 * 28, max_pixels: int = 768 * 28 * 28):
    """qwen_vl_utils.vision_process.smart_resize (pinned 0.0.11 by the reference's infer_requirements.txt:92; the same
    arithmetic as transformers' qwen2_vl smart_resize): both sides multiples of `factor`, area inside the budget, aspect kept."""
    if max(height, width) / min(height, width) > 200:
        raise ValueError(f"absolute aspect ratio must be smaller than 200, got {max(height, width) / min(height, width)}")
    h_bar = max(factor, round(height / factor) * factor)
    w_bar = max(factor, round(width / factor) * factor)
    if h_bar * w_bar > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        h_bar = math.floor(height / beta / factor) * factor
        w_bar = math.floor(width / beta / factor) * factor
    elif h_bar * w_bar < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h_bar = math.ceil(height * beta / factor) * factor
        w_bar = math.ceil(width * beta / factor) * factor
    return h_bar, w_bar


def _cubic(x):
    a = f32(-0.5)
    x = abs(x)
    if x < 1:
        return ((a + f32(2)) * x - (a + f32(3))) * x * x + f32(1)
    if x < 2:
        return (((x - f32(5)) * x + f32(8)) * x - f32(4)) * a
    return f32(0)


def aa_tables(in_size: int, out_size: int):
    """(xmin[out], xsize[out], weights[out][K]) of one axis; every operation in fp32, in this order."""
    scale = f32(in_size) / f32(out_size)
    support = f32(2.0) * scale if scale >= 1 else f32(2.0)
    invscale = f32(1.0) / scale if scale >= 1 else f32(1.0)
    K = int(math.ceil(float(support))) * 2 + 1
    xmin = np.zeros(out_size, np.int32)
    xsize = np.zeros(out_size, np.int32)
    wt = np.zeros((out_size, K), f32)
    for i in range(out_size):
        center = scale * (f32(i) + f32(0.5))
        lo = max(int(center - support + f32(0.5)), 0)
        n = min(int(center + support + f32(0.5)), in_size) - lo
        n = min(max(n, 0), K)
        ws, tot = [], f32(0)
        for j in range(n):
            w = _cubic((f32(j + lo) - center + f32(0.5)) * invscale)
            ws.append(w)
            tot = tot + w
        for j in range(n):
            wt[i, j] = ws[j] / tot if tot != 0 else ws[j]
        xmin[i], xsize[i] = lo, n
    return xmin, xsize, wt


def _resize_last_axis(x: np.ndarray, out_size: int) -> np.ndarray:
    xmin, xsize, wt = aa_tables(x.shape[-1], out_size)
    out = np.zeros(x.shape[:-1] + (out_size,), f32)
    for i in range(out_size):
        acc = x[..., xmin[i]] * wt[i, 0]                         # fp32 product, then fp32 sums in tap order (no fused multiply-add)
        for j in range(1, xsize[i]):
            acc = acc + x[..., xmin[i] + j] * wt[i, j]
        out[..., i] = acc
    return out


def resize_bicubic_aa_f32(frames: np.ndarray, h: int, w: int) -> np.ndarray:
    """(..., H, W) any real dtype -> (..., h, w) fp32, un-rounded."""
    x = np.ascontiguousarray(frames, dtype=f32)
    y = _resize_last_axis(x, w)                                                  # width first
    y = _resize_last_axis(np.ascontiguousarray(np.swapaxes(y, -1, -2)), h)       # then height
    return np.ascontiguousarray(np.swapaxes(y, -1, -2))


def resize_bicubic_aa_u8(frames: np.ndarray, h: int, w: int) -> np.ndarray:
    """uint8 (..., H, W) -> uint8 (..., h, w): clamp, round half to even, cast (torchvision's _cast_squeeze_out)."""
    y = resize_bicubic_aa_f32(frames, h, w)
    return np.rint(np.clip(y, 0, 255)).astype(np.uint8)


def spatial_resize_video(video: np.ndarray, nframes: int | None = None, *, video_max_pixels: int = 768 * 28 * 28,
                         video_total_pixels: int = 4 * 24576 * 28 * 28, video_min_pixels: int = 100 * 28 * 28,
                         frame_factor: int = 2, image_factor: int = 28) -> np.ndarray:
    """_spatial_resize_video (video_process_patch.py:134-153) on a uint8 (T, C, H, W) clip; returns uint8 (the reference's
    trailing .float() is exact).  Budgets: the module's own environment defaults (:11-15) on top of qwen_vl_utils' constants."""
    T, _, H, W = video.shape
    n = nframes or T
    max_pixels = max(min(video_max_pixels, video_total_pixels / n * frame_factor), int(video_min_pixels * 1.05))
    h, w = smart_resize(H, W, factor=image_factor, min_pixels=video_min_pixels, max_pixels=max_pixels)
    return resize_bicubic_aa_u8(video, h, w)
