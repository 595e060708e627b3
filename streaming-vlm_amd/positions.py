"""M-RoPE position ids for shrink mode -- host-side counterpart of the reference's
``get_rope_index`` (src/streaming_vlm/inference/qwen2/pos_emb.py:4-154) for the case the
streaming loop exercises (batch 1, full attention mask, video segments only;
qwen2/model_forward.py:119-126).  Returns int32 (3, L).

The reference walks the id list in Python with ``.tolist()/.index()/.item()`` on EVERY forward,
including each decode step; here vision segments are located with numpy once per chunk, the
table is filled with slice assignments, and decode steps extend it by ``+1`` (text tokens).
"""
from __future__ import annotations

import numpy as np


def rope_index_qwen2(ids, video_grid_thw, spatial_merge_size: int, video_token_id: int, vision_start_token_id: int):
    ids = np.asarray(ids).reshape(-1)
    n = ids.shape[0]
    pos = np.empty((3, n), dtype=np.int32)
    is_start = np.flatnonzero(ids[:-1] == vision_start_token_id) if n > 1 else np.zeros(0, np.int64)
    n_vid = int(np.count_nonzero(ids[is_start + 1] == video_token_id)) if is_start.size else 0
    pads = np.flatnonzero(ids == video_token_id)
    st = 0
    nxt = 0
    for vi in range(n_vid):
        a = np.searchsorted(pads, st)
        if a >= pads.shape[0]:
            raise ValueError("video segment without <|video_pad|> tokens")
        ed = int(pads[a])
        t, h, w = (int(x) for x in video_grid_thw[vi])
        gh, gw = h // spatial_merge_size, w // spatial_merge_size
        nv = t * gh * gw
        if ed + nv > n:
            raise ValueError(f"vision span of {nv} tokens at {ed} exceeds sequence length {n}")
        text_len = ed - st
        if text_len:
            pos[:, st:ed] = np.arange(nxt, nxt + text_len, dtype=np.int32)
        base = nxt + text_len
        k = np.arange(nv, dtype=np.int32)
        pos[0, ed:ed + nv] = base + k // (gh * gw)
        pos[1, ed:ed + nv] = base + (k // gw) % gh
        pos[2, ed:ed + nv] = base + k % gw
        nxt = base + max(t, gh, gw)
        st = ed + nv
    if st < n:
        pos[:, st:] = np.arange(nxt, nxt + (n - st), dtype=np.int32)
        nxt += n - st
    return pos, nxt       # nxt = position of the next (text) token


def rope_index_qwen2_5(ids, video_grid_thw, spatial_merge_size: int, video_token_id: int, vision_start_token_id: int,
                       second_per_grid_t: float, tokens_per_second: float):
    """Qwen2.5-VL counterpart (reference qwen2_5/pos_emb.py:6-160): the temporal index of vision tokens advances by
    `second_per_grid_t * tokens_per_second` per temporal grid step (the reference pins second_per_grid_t = 2 / FPS,
    :107-108), so positions are float32 -- same operation order as the reference: ((t * spg) * tps) + text_len + start.
    Returns float32 (3, L) and the position of the next text token."""
    ids = np.asarray(ids).reshape(-1)
    n = ids.shape[0]
    pos = np.empty((3, n), dtype=np.float32)
    is_start = np.flatnonzero(ids[:-1] == vision_start_token_id) if n > 1 else np.zeros(0, np.int64)
    n_vid = int(np.count_nonzero(ids[is_start + 1] == video_token_id)) if is_start.size else 0
    pads = np.flatnonzero(ids == video_token_id)
    spg, tps = np.float32(second_per_grid_t), np.float32(tokens_per_second)
    st = 0
    nxt = np.float32(0)
    for vi in range(n_vid):
        a = np.searchsorted(pads, st)
        if a >= pads.shape[0]:
            raise ValueError("video segment without <|video_pad|> tokens")
        ed = int(pads[a])
        t, h, w = (int(x) for x in video_grid_thw[vi])
        gh, gw = h // spatial_merge_size, w // spatial_merge_size
        nv = t * gh * gw
        if ed + nv > n:
            raise ValueError(f"vision span of {nv} tokens at {ed} exceeds sequence length {n}")
        text_len = ed - st
        if text_len:
            pos[:, st:ed] = np.arange(text_len, dtype=np.float32) + nxt
        k = np.arange(nv)
        tl = np.float32(text_len)
        pos[0, ed:ed + nv] = ((k // (gh * gw)).astype(np.float32) * spg) * tps + tl + nxt
        pos[1, ed:ed + nv] = ((k // gw) % gh).astype(np.float32) + tl + nxt
        pos[2, ed:ed + nv] = (k % gw).astype(np.float32) + tl + nxt
        nxt = np.float32(pos[:, ed:ed + nv].max() + np.float32(1))
        st = ed + nv
    if st < n:
        pos[:, st:] = np.arange(n - st, dtype=np.float32) + nxt
        nxt = np.float32(pos[0, n - 1] + np.float32(1))
    return pos, float(nxt)


def rope_index_1d(n: int):
    """`all_text`: plain 0..L-1 on all three axes (reference qwen2_5/model_forward.py:6-28, get_1d_rope_index)."""
    p = np.arange(n, dtype=np.int32)
    return np.stack([p, p, p]), n
