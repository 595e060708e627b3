"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the frame resize in front of the path.

The reference resizes every decoded uint8 clip with torchvision's v1 tensor resize, bicubic with antialiasing, before the HF
processor sees it (livecc_utils/src/livecc_utils/video_process_patch.py:134-153: `smart_resize` to a multiple of 28 inside the
pixel budget, then `transforms.functional.resize(video, [h, w], BICUBIC, antialias=True).float()`).  v1 `resize` casts a uint8
tensor to float32 and calls `torch.nn.functional.interpolate(x, size, mode="bicubic", antialias=True, align_corners=False)` -- the
clip comes from decord on the host, so that is ATen's CPU kernel -- then clamps, rounds half to even and casts back.  (torch's
native uint8 kernel, which transforms.v2 would take, is a different fixed-point filter: up to 12 grey levels away.)

The arithmetic lives in oracle/_c/resize_ref.c (plain C, compiled with gcc on first use into oracle/_build/): the float kernel's
formulas with the C promotion rules of its source and the FMA contraction of its build, both established empirically against the
installed torch (see that file's header).  Pinning: tests/test_resize.py requires EXACT equality with F.interpolate -- fp32 bits and
rounded uint8 -- on every case, against the installed torch and a committed fixture of its outputs.  The HIP kernel
(svlm_resize_bicubic_aa_u8) and the product's host tap-table function are compared with this file bit for bit.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess

import numpy as np

f32 = np.float32
_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "_c", "resize_ref.c")
_LIB = os.path.join(_HERE, "_build", "libresize_ref.so")
_lib = None


def build_c(force: bool = False) -> str:
    """gcc -O2 -ffp-contract=off (every fma in the source is an explicit fmaf) -> oracle/_build/libresize_ref.so"""
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(_SRC):
        os.makedirs(os.path.dirname(_LIB), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", _LIB, _SRC, "-lm"])
    return _LIB


def _c():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(build_c())
        ip, fp = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_float)
        lib.svlm_ref_aa_tables.restype = ctypes.c_int
        lib.svlm_ref_aa_tables.argtypes = [ctypes.c_int, ctypes.c_int, ip, ip, fp, ctypes.c_int]
        lib.svlm_ref_resize_rows.restype = None
        lib.svlm_ref_resize_rows.argtypes = [fp, ctypes.c_long, ctypes.c_int, fp, ctypes.c_int, ip, ip, fp, ctypes.c_int]
        _lib = lib
    return _lib


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


def aa_tables(in_size: int, out_size: int):
    """(xmin[out], xsize[out], weights[out][K]) of one axis, as torch's CPU kernel computes them."""
    lib = _c()
    K = lib.svlm_ref_aa_tables(in_size, out_size, None, None, None, 0)
    if K <= 0:
        raise ValueError(f"aa_tables({in_size}, {out_size})")
    xmin, xsize, wt = np.zeros(out_size, np.int32), np.zeros(out_size, np.int32), np.zeros((out_size, K), f32)
    ip, fp = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_float)
    assert lib.svlm_ref_aa_tables(in_size, out_size, xmin.ctypes.data_as(ip), xsize.ctypes.data_as(ip), wt.ctypes.data_as(fp), K) == K
    return xmin, xsize, wt


def _resize_last_axis(x: np.ndarray, out_size: int) -> np.ndarray:
    lib = _c()
    x = np.ascontiguousarray(x, dtype=f32)
    xmin, xsize, wt = aa_tables(x.shape[-1], out_size)
    out = np.zeros(x.shape[:-1] + (out_size,), f32)
    ip, fp = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_float)
    lib.svlm_ref_resize_rows(x.ctypes.data_as(fp), int(np.prod(x.shape[:-1], dtype=np.int64)), x.shape[-1], out.ctypes.data_as(fp), out_size,
                             xmin.ctypes.data_as(ip), xsize.ctypes.data_as(ip), wt.ctypes.data_as(fp), wt.shape[1])
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
