"""Frame ingest in front of the ViT: the reference's per-chunk `_spatial_resize_video`
(livecc_utils/src/livecc_utils/video_process_patch.py:134-153, called at src/streaming_vlm/inference/inference.py:342) with the
pixels moved by a HIP kernel (svlm_resize_bicubic_aa_u8) instead of torchvision on the host.

The target size is host integer arithmetic (`smart_resize`: qwen_vl_utils.vision_process, pinned 0.0.11 by the reference;
both sides multiples of 28, area inside the pixel budget).  The budget constants are the ones the reference module sets up at
import time (:11-15: VIDEO_MAX_PIXELS / VIDEO_MIN_PIXELS / FPS_MAX_FRAMES environment overrides on top of qwen_vl_utils'
defaults).  Frames whose size already satisfies the budget -- every BASELINE stream: 448x448 and 224x224 -- pass through
untouched, without a launch.
"""
from __future__ import annotations

import math
import os

import torch

IMAGE_FACTOR = 28
FRAME_FACTOR = 2
# qwen_vl_utils.vision_process reads VIDEO_MAX_PIXELS from the environment for its TOTAL budget; the reference presets that
# variable to 4 * 24576 * 28 * 28 (video_process_patch.py:12) and lowers the per-frame minimum to 100 * 28 * 28 (:14)
VIDEO_TOTAL_PIXELS = int(float(os.environ.get("VIDEO_MAX_PIXELS", 4 * 24576 * 28 * 28)))
VIDEO_MIN_PIXELS = int(os.environ.get("VIDEO_MIN_PIXELS", 100 * 28 * 28))
VIDEO_MAX_PIXELS = 768 * 28 * 28


def smart_resize(height: int, width: int, factor: int = IMAGE_FACTOR, min_pixels: int = VIDEO_MIN_PIXELS,
                 max_pixels: int = VIDEO_MAX_PIXELS):
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


def resized_shape(height: int, width: int, nframes: int):
    """Target (h, w) of a clip of `nframes` frames (video_process_patch.py:139-146)."""
    max_pixels = max(min(VIDEO_MAX_PIXELS, VIDEO_TOTAL_PIXELS / nframes * FRAME_FACTOR), int(VIDEO_MIN_PIXELS * 1.05))
    return smart_resize(height, width, factor=IMAGE_FACTOR, min_pixels=VIDEO_MIN_PIXELS, max_pixels=max_pixels)


def spatial_resize_video(video: torch.Tensor, ops, device="cuda", nframes: int | None = None) -> torch.Tensor:
    """uint8 (T, C, H, W) clip -> uint8 (T, C, h, w).  A clip that needs resizing is resized ON THE DEVICE (it is uploaded
    first when it lives on the host, and stays there: the device processor patchifies it in place); there is no host path."""
    if video.dim() != 4 or video.dtype != torch.uint8:
        raise ValueError(f"spatial_resize_video: uint8 (T, C, H, W) clip expected, got {video.dtype} {tuple(video.shape)}")
    T, _, H, W = video.shape
    h, w = resized_shape(H, W, nframes or T)
    if (h, w) == (H, W):
        return video
    if ops is None:
        raise RuntimeError(f"frames of {H}x{W} need resizing to {h}x{w}: the resize runs on the GPU (svlm_resize_bicubic_aa_u8)")
    return ops.resize_u8(video.to(device, non_blocking=True).contiguous(), h, w)
