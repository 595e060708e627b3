"""Chat-template span finder -- host-side counterpart of the reference's
``src/streaming_vlm/utils/get_qwen_range.py`` (same function name, arguments and closed-interval
result; token ids from its ``TOKEN_IDS`` table, :2-13).

Implementation differs from the reference's Python double loop: start/end pattern hits are found
with vectorised numpy comparisons and the left-to-right pairing walks the (few dozen) hits with
``searchsorted``, so a 6k-token history costs microseconds instead of a full Python scan.
"""
from __future__ import annotations

import numpy as np

from .config import IM_END, IM_START, LF, VIDEO_PAD, VISION_END, VISION_START

SYSTEM_PROMPT_OFFSET = 58
TOKEN_IDS = {
    "<|im_start|>": IM_START, "<|im_end|>": IM_END, "user": 872, "assistant": 77091,
    "<|vision_start|>": VISION_START, "<|vision_end|>": VISION_END, "<|video_pad|>": VIDEO_PAD,
    "\n": LF, "previous text": [19702, 1467], "Time": 1462,
}

_PATTERNS = {
    "user": ([IM_START, 872], [IM_END]),
    "previous text": ([IM_START, 19702, 1467, LF], [IM_END]),
    "user_text": ([1462], [VISION_START]),
    "assistant": ([IM_START, 77091], [IM_END]),
    "vision": ([VISION_START], [VISION_END]),
}


def _as_array(input_ids) -> np.ndarray:
    if hasattr(input_ids, "detach"):            # torch tensor (the reference passes a (1, L) tensor)
        input_ids = input_ids.detach().flatten().cpu().numpy()
    return np.asarray(input_ids).reshape(-1)


def _hits(ids: np.ndarray, pat) -> np.ndarray:
    n, k = ids.shape[0], len(pat)
    if n < k:
        return np.zeros(0, dtype=np.int64)
    m = ids[: n - k + 1] == pat[0]
    for j in range(1, k):
        m &= ids[j: n - k + 1 + j] == pat[j]
    return np.flatnonzero(m)


def all_ranges(input_ids, label: str, contain_lf: bool = True):
    assert label in _PATTERNS, label
    ids = _as_array(input_ids)
    sp, ep = _PATTERNS[label]
    starts, ends = _hits(ids, sp), _hits(ids, ep)
    ls, le, n = len(sp), len(ep), ids.shape[0]
    out = []
    cur = 0
    while True:
        a = np.searchsorted(starts, cur)
        if a >= starts.shape[0]:
            break
        s = int(starts[a])
        b = np.searchsorted(ends, s + ls)
        if b >= ends.shape[0]:
            break                                  # unterminated segment ends the scan
        e = int(ends[b]) + le - 1
        if contain_lf and e + 1 < n and ids[e + 1] == LF:
            e += 1
        out.append((s, e))
        cur = int(ends[b]) + le
    if label == "user_text":
        out = [(s, e - 1) for s, e in out]
    return out


def get_qwen_range(input_ids, label: str, index: int, contain_lf: bool = True):
    """(start, end) of the index-th `label` segment, closed interval; IndexError if absent."""
    return all_ranges(input_ids, label, contain_lf)[index]
