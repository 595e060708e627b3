"""M-RoPE position-id builder (oracle; test infrastructure only).

Restates ``src/streaming_vlm/inference/qwen2/pos_emb.py:4-154`` for the case
the streaming loop uses (shrink mode, ``qwen2/model_forward.py:119-126``):
batch 1, attention mask all ones, video tokens only, grids consumed in order
of appearance.  Output is ``(3, L)`` int64: text runs get ``arange`` on all
three axes; each vision span gets (t, h, w) grid ids offset by the running
start; the next text run starts at ``max + 1`` (pos_emb.py:113-133).
"""
from __future__ import annotations

import numpy as np

from .qwen_range import VIDEO_PAD, VISION_START


def get_rope_index(ids, video_grid_thw, spatial_merge_size: int = 2,
                   video_token_id: int = VIDEO_PAD, vision_start_token_id: int = VISION_START):
    ids = [int(t) for t in ids]
    n = len(ids)
    # number of video segments = vision_start tokens followed by a video token  (pos_emb.py:74-77)
    n_vid = 0
    for i, t in enumerate(ids):
        if t == vision_start_token_id and i + 1 < n and ids[i + 1] == video_token_id:
            n_vid += 1
    pos = []
    st = 0
    nxt = 0          # st_idx of the reference: max of everything emitted so far + 1
    vi = 0
    for _ in range(n_vid):                                # :82
        ed = ids.index(video_token_id, st)                # :88
        t, h, w = (int(x) for x in video_grid_thw[vi])    # :101-106
        vi += 1
        gt, gh, gw = t, h // spatial_merge_size, w // spatial_merge_size   # :110-114
        text_len = ed - st
        if text_len > 0:
            a = np.arange(text_len, dtype=np.int64) + nxt
            pos.append(np.stack([a, a, a]))
        base = text_len + nxt                             # :123
        ti = np.repeat(np.arange(gt), gh * gw)
        hi = np.tile(np.repeat(np.arange(gh), gw), gt)
        wi = np.tile(np.arange(gw), gt * gh)
        pos.append(np.stack([ti, hi, wi]).astype(np.int64) + base)
        nxt = int(pos[-1].max()) + 1
        st = ed + gt * gh * gw                            # :124
    if st < n:                                            # :126-129
        a = np.arange(n - st, dtype=np.int64) + nxt
        pos.append(np.stack([a, a, a]))
    out = np.concatenate(pos, axis=1) if pos else np.zeros((3, 0), np.int64)
    # A token-count cut may leave fewer listed positions than tokens only if a
    # vision span is truncated at the END of ids; the reference would raise on
    # the shape mismatch (pos_emb.py:132).  Mirror that.
    if out.shape[1] != n:
        raise ValueError(f"rope index length {out.shape[1]} != sequence length {n}")
    return out


def get_rope_index_2_5(ids, video_grid_thw, spatial_merge_size: int = 2, video_token_id: int = VIDEO_PAD,
                       vision_start_token_id: int = VISION_START, second_per_grid_t: float = 1.0,
                       tokens_per_second: float = 2.0):
    """Qwen2.5 variant, restating ``src/streaming_vlm/inference/qwen2_5/pos_emb.py:72-160``: float32 positions, the
    temporal index of a vision span is ``arange(t) * second_per_grid_t * tokens_per_second`` (:121-125, with
    second_per_grid_t = 2 / FPS pinned at :107-108), the other axes and the text runs as in Qwen2-VL."""
    ids = [int(t) for t in ids]
    n = len(ids)
    n_vid = sum(1 for i, t in enumerate(ids) if t == vision_start_token_id and i + 1 < n and ids[i + 1] == video_token_id)
    pos, st, vi = [], 0, 0
    spg = np.float32(second_per_grid_t)
    for _ in range(n_vid):
        ed = ids.index(video_token_id, st)
        t, h, w = (int(x) for x in video_grid_thw[vi])
        vi += 1
        gt, gh, gw = t, h // spatial_merge_size, w // spatial_merge_size
        text_len = ed - st
        st_idx = (pos[-1].max() + np.float32(1)) if pos else np.float32(0)                         # :116
        a = np.arange(text_len, dtype=np.float32) + st_idx                                         # :117-119
        pos.append(np.stack([a, a, a]))
        ti = (np.repeat(np.arange(gt), gh * gw).astype(np.float32) * spg) * np.float32(tokens_per_second)   # :122-126
        hi = np.tile(np.repeat(np.arange(gh), gw), gt).astype(np.float32)
        wi = np.tile(np.arange(gw), gt * gh).astype(np.float32)
        pos.append(np.stack([ti, hi, wi]) + np.float32(text_len) + st_idx)                         # :131-133
        st = ed + gt * gh * gw
    if st < n:
        st_idx = (pos[-1].max() + np.float32(1)) if pos else np.float32(0)
        a = np.arange(n - st, dtype=np.float32) + st_idx
        pos.append(np.stack([a, a, a]))
    out = np.concatenate([p for p in pos if p.shape[1]], axis=1).astype(np.float32) if pos else np.zeros((3, 0), np.float32)
    if out.shape[1] != n:
        raise ValueError(f"rope index length {out.shape[1]} != sequence length {n}")
    return out


def get_1d_rope_index(n: int):
    """``all_text`` positions (qwen2_5/model_forward.py:6-28 with a full mask): 0..L-1 on all three axes."""
    a = np.arange(n, dtype=np.int64)
    return np.stack([a, a, a])
