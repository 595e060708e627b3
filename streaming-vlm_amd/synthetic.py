"""Synthetic, tokenizer-free inputs for the streaming loop (no checkpoint, tokenizer or video
exists offline -- SURVEY 8c/8d).

* frames: frame t of stream s = uint8 U[0,255] (3, S, S) from ``torch.Generator().manual_seed(1234 + 1000*s + t)``
* preprocessing: /255, CLIP mean/std, duplicate the last frame up to the temporal patch, patchify
  into ``(grid_t*grid_h*grid_w, C*T*P*P)`` rows in the merge-block-major order the HF Qwen2-VL
  processor emits (transformers ``image_processing_qwen2_vl.py`` ``_preprocess``), which the
  merger's ``view(-1, 4*embed)`` relies on (qwen2/vision_forward.py:80)
* ``SyntheticProcessor``: the subset of the ``AutoProcessor`` surface ``streaming_inference`` uses
  (``apply_chat_template``, ``__call__(text=, videos=)``, ``batch_decode``) over a deterministic
  stand-in tokenizer whose chat-template token ids are the reference's hard-coded ones
  (src/streaming_vlm/utils/get_qwen_range.py:2-13).
"""
from __future__ import annotations

import re
import zlib
from typing import List, Optional

import numpy as np
import torch

from .config import IM_END, IM_START, LF, VIDEO_PAD, VISION_END, VISION_START

OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


# ----------------------------------------------------------------------------- frames
def synthetic_frame(stream: int, t: int, size=448) -> torch.Tensor:
    """`size`: side of a square frame, or (height, width)."""
    g = torch.Generator()
    g.manual_seed(1234 + 1000 * stream + t)
    h, w = (size, size) if isinstance(size, int) else (int(size[0]), int(size[1]))
    return torch.randint(0, 256, (3, h, w), generator=g, dtype=torch.uint8)


class SyntheticVideo:
    """Stand-in for the decord reader: frames are addressed by index at `fps` frames per second."""

    # True = "decoded at the file's native size": the driver then applies the reference's per-chunk _spatial_resize_video
    # (inference.py:342) through the HIP resize; False = frames already at processor size (every BASELINE stream)
    spatial_resize = False

    def __init__(self, size=448, fps: float = 1.0, stream: int = 0, spatial_resize: bool = False):
        self.size, self.fps, self.stream = size, fps, stream
        self.spatial_resize = spatial_resize

    def chunk(self, start_s: float, duration_s: float) -> torch.Tensor:
        n = max(1, int(round(duration_s * self.fps)))
        first = int(round(start_s * self.fps))
        return torch.stack([synthetic_frame(self.stream, first + k, self.size) for k in range(n)])

    @classmethod
    def from_path(cls, path: str) -> Optional["SyntheticVideo"]:
        """``synthetic://448x448@1fps?stream=3``; ``synthetic-raw://1280x720@2fps`` = native-size frames that still need the
        reference's smart resize."""
        m = re.fullmatch(r"synthetic(-raw)?://(\d+)x(\d+)@([\d.]+)fps(?:\?stream=(\d+))?", path or "")
        if not m:
            return None
        w, h = int(m.group(2)), int(m.group(3))          # WxH, like a resolution string
        return cls(w if w == h else (h, w), float(m.group(4)), int(m.group(5) or 0), spatial_resize=bool(m.group(1)))


def patchify(frames: torch.Tensor, patch: int = 14, temporal: int = 2, merge: int = 2, device=None):
    """uint8 (T, 3, H, W) -> (pixel_values float32 (N, 3*temporal*patch*patch), grid_thw [[t, h, w]])."""
    if device is not None:
        frames = frames.to(device)
    T, C, H, W = frames.shape
    if H % (patch * merge) or W % (patch * merge):
        raise ValueError(f"frame {H}x{W} is not a multiple of {patch * merge}")
    x = frames.float() / 255.0
    mean = torch.tensor(OPENAI_CLIP_MEAN, device=x.device).view(1, 3, 1, 1)
    std = torch.tensor(OPENAI_CLIP_STD, device=x.device).view(1, 3, 1, 1)
    x = (x - mean) / std
    if T % temporal:
        x = torch.cat([x, x[-1:].repeat(temporal - T % temporal, 1, 1, 1)], 0)
    gt, gh, gw = x.shape[0] // temporal, H // patch, W // patch
    x = x.reshape(gt, temporal, C, gh // merge, merge, patch, gw // merge, merge, patch)
    x = x.permute(0, 3, 6, 4, 7, 2, 1, 5, 8)
    return x.reshape(gt * gh * gw, C * temporal * patch * patch).contiguous(), [[gt, gh, gw]]


# ----------------------------------------------------------------------------- stand-in tokenizer
_SPECIAL = {"<|im_start|>": IM_START, "<|im_end|>": IM_END, "<|vision_start|>": VISION_START,
            "<|vision_end|>": VISION_END, "<|video_pad|>": VIDEO_PAD, "<|endoftext|>": 151643}
_KNOWN = {"system": 8948, "user": 872, "assistant": 77091, "previous": 19702, " text": 1467, "Time": 1462,
          "You": 2610, " are": 525, " a": 264, " helpful": 10950, " assistant": 17847, " ...": 2503, "\n": LF}
_PIECE = re.compile(r"<\|[a-z_]+\|>|\n| \.\.\.| ?[A-Za-z]+|\d|[^\sA-Za-z\d]| +")


class SyntheticTokenizer:
    """Deterministic text -> ids: chat-template pieces get the real Qwen ids, single ASCII symbols and
    digits their byte-level ids (ord - 33), every other word a stable hash into [20000, 140000)."""

    def __init__(self):
        self._inv = {v: k for k, v in {**_SPECIAL, **_KNOWN}.items()}

    def encode(self, text: str) -> List[int]:
        out = []
        for piece in _PIECE.findall(text):
            if piece in _SPECIAL:
                tid = _SPECIAL[piece]
            elif piece in _KNOWN:
                tid = _KNOWN[piece]
            elif len(piece) == 1 and 33 <= ord(piece) <= 126:
                tid = ord(piece) - 33
            elif piece.strip() == "":
                tid = 220 if len(piece) == 1 else 256 + min(len(piece), 30)
            else:
                tid = 20000 + zlib.crc32(piece.encode()) % 120000
            self._inv.setdefault(tid, piece)
            out.append(tid)
        return out

    def decode(self, ids, skip_special_tokens: bool = True) -> str:
        special = set(_SPECIAL.values())
        s = []
        for t in ids:
            t = int(t)
            if skip_special_tokens and t in special:
                continue
            s.append(self._inv.get(t, f"<{t}>"))
        return "".join(s)


class _Batch(dict):
    def to(self, device):
        return _Batch({k: (v.to(device) if hasattr(v, "to") else v) for k, v in self.items()})


class SyntheticProcessor:
    """Minimal AutoProcessor look-alike (Qwen2-VL chat template, no system-prompt customisation)."""

    SYSTEM = "<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n"        # 58 characters = SYSTEM_PROMPT_OFFSET

    def __init__(self, patch: int = 14, temporal: int = 2, merge: int = 2):
        self.tokenizer = SyntheticTokenizer()
        self.patch, self.temporal, self.merge = patch, temporal, merge

    def apply_chat_template(self, conversation, tokenize: bool = False, add_generation_prompt: bool = True) -> str:
        assert not tokenize
        text = self.SYSTEM
        for turn in conversation:
            content = turn["content"]
            if isinstance(content, str):
                body = content
            else:
                body = ""
                for item in content:
                    if item["type"] == "text":
                        body += item["text"]
                    elif item["type"] == "video":
                        body += "<|vision_start|><|video_pad|><|vision_end|>"
            text += f"<|im_start|>{turn['role']}\n{body}<|im_end|>\n"
        if add_generation_prompt:
            text += "<|im_start|>assistant\n"
        return text

    def __call__(self, text=None, videos=None, padding=True, return_tensors="pt", **_):
        texts = [text] if isinstance(text, str) else list(text)
        assert len(texts) == 1, "batch size 1 (the streaming loop never batches)"
        out = _Batch()
        t = texts[0]
        if videos is not None:
            pix, grid = patchify(videos, self.patch, self.temporal, self.merge)
            n_tok = grid[0][0] * grid[0][1] * grid[0][2] // (self.merge ** 2)
            assert t.count("<|video_pad|>") == 1, "one video per call"
            t = t.replace("<|video_pad|>", "<|video_pad|>" * n_tok)
            out["pixel_values_videos"] = pix
            out["video_grid_thw"] = torch.tensor(grid, dtype=torch.long)
        ids = self.tokenizer.encode(t)
        if return_tensors == "pt" or videos is not None:
            out["input_ids"] = torch.tensor([ids], dtype=torch.long)
            out["attention_mask"] = torch.ones_like(out["input_ids"])
        else:
            out["input_ids"] = [ids]
        return out

    def batch_decode(self, ids, skip_special_tokens: bool = True):
        rows = ids.tolist() if hasattr(ids, "tolist") else ids
        return [self.tokenizer.decode(r, skip_special_tokens) for r in rows]


# ----------------------------------------------------------------------------- HBM-resident stream (benchmarks)
class ResidentChunk:
    """Handle to one chunk whose patches already live on the device."""

    def __init__(self, pixel_values, grid):
        self.pixel_values, self.grid = pixel_values, grid


class ResidentVideo:
    """Pre-patchified synthetic stream: `chunk()` hands out device-resident patch tensors so that a timed
    region starts with its inputs in HBM (bench.py)."""

    def __init__(self, n_chunks: int, size: int, fps: float, stream: int, device, chunk_duration: float = 1.0,
                 patch: int = 14, temporal: int = 2, merge: int = 2, period: int = 0):
        """`period` > 0 keeps only that many distinct chunks resident and cycles through them (hour-long streams:
        3600 chunks of 448x448 would otherwise hold 8.7 GB of patches and minutes of host patchify)."""
        src = SyntheticVideo(size, fps, stream)
        self.chunk_duration = chunk_duration
        self.frames_per_chunk = max(1, int(round(chunk_duration * fps)))
        self.chunks = []
        for i in range(min(n_chunks, period) if period > 0 else n_chunks):
            pix, grid = patchify(src.chunk(i * chunk_duration, chunk_duration), patch, temporal, merge)
            self.chunks.append(ResidentChunk(pix.to(device), grid))

    def chunk(self, start_s: float, duration_s: float) -> ResidentChunk:
        return self.chunks[int(round(start_s / self.chunk_duration)) % len(self.chunks)]


class PinnedVideo:
    """Synthetic uint8 frames generated up front into page-locked host memory: what a decoder thread would hand over.
    `chunk()` returns host tensors; the device copy and the patchify belong to the processor (DeviceFrameProcessor)."""

    def __init__(self, n_chunks: int, size: int, fps: float, stream: int, chunk_duration: float = 1.0, period: int = 0):
        src = SyntheticVideo(size, fps, stream)
        self.chunk_duration = chunk_duration
        self.frames_per_chunk = max(1, int(round(chunk_duration * fps)))
        n = min(n_chunks, period) if period > 0 else n_chunks
        self.chunks = []
        for i in range(n):
            f = src.chunk(i * chunk_duration, chunk_duration).contiguous()
            self.chunks.append(f.pin_memory() if torch.cuda.is_available() else f)

    def chunk(self, start_s: float, duration_s: float) -> torch.Tensor:
        return self.chunks[int(round(start_s / self.chunk_duration)) % len(self.chunks)]


class DeviceFrameProcessor(SyntheticProcessor):
    """Processor whose pixel path runs on the GPU: uint8 frames cross PCIe (0.6 MB per 448x448 frame instead of 2.4 MB
    of fp32/bf16 patches), rescale + normalise + merge-block-major patchify are one HIP kernel (svlm_patchify_u8)."""

    def __init__(self, ops, device="cuda", **kw):
        super().__init__(**kw)
        self.ops, self.device = ops, torch.device(device)

    def __call__(self, text=None, videos=None, padding=True, return_tensors="pt", **kw):
        if videos is None:
            return super().__call__(text=text, videos=None, padding=padding, return_tensors=return_tensors, **kw)
        frames = videos.to(self.device, non_blocking=True)          # (a clip the driver resized is on the device already)
        pix, grid = self.ops.patchify_u8(frames.contiguous(), self.patch, self.temporal, self.merge)
        t = (text if isinstance(text, str) else text[0])
        assert t.count("<|video_pad|>") == 1, "one video per call"
        t = t.replace("<|video_pad|>", "<|video_pad|>" * (grid[0][0] * grid[0][1] * grid[0][2] // self.merge ** 2))
        ids = torch.tensor([self.tokenizer.encode(t)], dtype=torch.long)
        return _Batch(input_ids=ids, attention_mask=torch.ones_like(ids), pixel_values_videos=pix,
                      video_grid_thw=torch.tensor(grid, dtype=torch.long))


class ResidentProcessor(SyntheticProcessor):
    """SyntheticProcessor that accepts `ResidentChunk` handles for `videos=`."""

    def __call__(self, text=None, videos=None, padding=True, return_tensors="pt", **kw):
        if not isinstance(videos, ResidentChunk):
            return super().__call__(text=text, videos=videos, padding=padding, return_tensors=return_tensors, **kw)
        t = (text if isinstance(text, str) else text[0])
        g = videos.grid[0]
        t = t.replace("<|video_pad|>", "<|video_pad|>" * (g[0] * g[1] * g[2] // self.merge ** 2))
        ids = torch.tensor([self.tokenizer.encode(t)], dtype=torch.long)
        return _Batch(input_ids=ids, attention_mask=torch.ones_like(ids), pixel_values_videos=videos.pixel_values,
                      video_grid_thw=torch.tensor(videos.grid, dtype=torch.long))
