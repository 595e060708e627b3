"""KV / token-id eviction primitives and policies (oracle; test infrastructure only).

Restates ``src/streaming_vlm/inference/inference.py``:
  * ``prune_id_and_kv_cache``  :50-61   delete CLOSED interval [s, e] from ids and every layer's K, V
  * ``resort_id_and_kv``       :100-108 move rows [src_s, src_e] to directly after row ``dst``
  * ``process_past_kv``        :87-172  the structural policy (text rounds / vision window / previous-text sink+window)
and defines the BASELINE ``sink=S, window=W`` token-count policy on the same
primitive (no reference code path has it).

The KV cache is the reference's representation: a list of per-layer ``[K, V]``
tensors of shape ``(1, Hkv, L, D)`` (``generate/streaming_cache.py:72-73``).
Every structural edit is also appended to ``trace`` so that "identical eviction
indices" can be asserted against the HIP engine.
"""
from __future__ import annotations

import torch

from . import qwen_range as qr


class ListKV:
    """list-of-tensors KV store grown by ``torch.cat`` (streaming_cache.py:30-74)."""

    def __init__(self, n_layers: int):
        self.key_cache = [None] * n_layers
        self.value_cache = [None] * n_layers

    def get_seq_length(self) -> int:
        k = self.key_cache[0]
        return 0 if k is None else k.shape[2]

    def update(self, k, v, layer_idx):
        if self.key_cache[layer_idx] is None:
            self.key_cache[layer_idx] = k
            self.value_cache[layer_idx] = v
        else:
            self.key_cache[layer_idx] = torch.cat([self.key_cache[layer_idx], k], dim=-2)
            self.value_cache[layer_idx] = torch.cat([self.value_cache[layer_idx], v], dim=-2)
        return self.key_cache[layer_idx], self.value_cache[layer_idx]

    def __iter__(self):
        return iter(zip(self.key_cache, self.value_cache))

    def __len__(self):
        return len(self.key_cache)


def prune_id_and_kv_cache(ids, kv, start, end, trace=None):
    """inference.py:50-61 -- keep [0, start) U (end, L).  `ids` is a python list."""
    ids = ids[:start] + ids[end + 1:]
    if kv is not None:
        for i, (k, v) in enumerate(kv):
            L = k.shape[2]
            keep = torch.tensor(list(range(start)) + list(range(end + 1, L)), dtype=torch.long)
            kv.key_cache[i] = torch.index_select(k, 2, keep)
            kv.value_cache[i] = torch.index_select(v, 2, keep)
    if trace is not None:
        trace.append(("prune", int(start), int(end)))
    return ids, kv


def resort_id_and_kv(ids, kv, src_s, src_e, dst, trace=None):
    """inference.py:100-108 -- move [src_s, src_e] to after `dst`."""
    assert dst < src_s <= src_e
    ids = ids[:dst + 1] + ids[src_s:src_e + 1] + ids[dst + 1:src_s] + ids[src_e + 1:]
    if kv is not None:
        for i, (k, v) in enumerate(kv):
            kv.key_cache[i] = torch.cat([k[:, :, :dst + 1], k[:, :, src_s:src_e + 1],
                                         k[:, :, dst + 1:src_s], k[:, :, src_e + 1:]], dim=2)
            kv.value_cache[i] = torch.cat([v[:, :, :dst + 1], v[:, :, src_s:src_e + 1],
                                           v[:, :, dst + 1:src_s], v[:, :, src_e + 1:]], dim=2)
    if trace is not None:
        trace.append(("move", int(src_s), int(src_e), int(dst)))
    return ids, kv


def process_past_kv(kv, i, ids, text_round, visual_round, text_sink, text_sliding_window,
                    assistant_start_bias=3, assistant_end_bias=2, trace=None):
    """Index math of inference.py:87-172 (conversation-history strings omitted:
    they only feed prompt text, not indices).  `ids` = prev_generated_ids as a list."""
    if i >= text_round:                                                        # :91
        a_s, a_e = qr.get_qwen_range(ids, "assistant", 0)                      # :110
        p_s, p_e = qr.get_qwen_range(ids, "previous text", 0, contain_lf=False)  # :112
        if ids[a_e] == qr.LF:                                                  # :114-119
            src_s, src_e = a_s + assistant_start_bias, a_e - assistant_end_bias - 1
        else:
            src_s, src_e = a_s + assistant_start_bias, a_e - assistant_end_bias
        if src_s <= src_e:                                                     # :120-121
            ids, kv = resort_id_and_kv(ids, kv, src_s, src_e, p_e - 1, trace)
        if visual_round > text_round:                                          # :130-134
            u_s, u_e = qr.get_qwen_range(ids, "user_text", -text_round, contain_lf=False)
            ids, kv = prune_id_and_kv_cache(ids, kv, u_s, u_e, trace)
        a_s, a_e = qr.get_qwen_range(ids, "assistant", -text_round)            # :136
        ids, kv = prune_id_and_kv_cache(ids, kv, a_s, a_e, trace)              # :139
    if i >= visual_round:                                                      # :141
        if visual_round < text_round:                                          # :144-154
            v_s, v_e = qr.get_qwen_range(ids, "vision", 0)
            ids, kv = prune_id_and_kv_cache(ids, kv, v_s, v_e, trace)
    if i >= max(visual_round, text_round):                                     # :156-160
        u_s, u_e = qr.get_qwen_range(ids, "user", 0)
        ids, kv = prune_id_and_kv_cache(ids, kv, u_s, u_e, trace)
    if i > 0:                                                                  # :162-170
        if text_sink is not None or text_sliding_window is not None:
            p_s, p_e = qr.get_qwen_range(ids, "previous text", 0)
            cut_s = p_s + text_sink + 4 if text_sink is not None else p_s
            cut_e = p_e - text_sliding_window - 1 if text_sliding_window is not None else p_e
            if cut_s <= cut_e:
                ids, kv = prune_id_and_kv_cache(ids, kv, cut_s, cut_e, trace)
    return kv, ids


def snap_cut_end(ids, end):
    """A token-count cut must not split a <|vision_start|>..<|vision_end|> span,
    otherwise get_rope_index would mis-assign grids.  If
    token `end` lies inside a vision span (or on its <|vision_start|>), the cut
    is extended forward to that span's <|vision_end|>, so the retained window
    never exceeds W.  Returns the snapped end."""
    n = len(ids)
    # inside a span <=> scanning left from `end` we meet VISION_START/VIDEO_PAD run before anything else
    j = end
    if ids[j] == qr.VISION_END:
        return end
    if ids[j] not in (qr.VISION_START, qr.VIDEO_PAD):
        # `end` is a text token; but is the NEXT token an orphan-able video pad? no: pads follow start only
        return end
    while j < n and ids[j] != qr.VISION_END:
        j += 1
    if j >= n:
        raise ValueError("unterminated vision span")
    return j


def sink_window_policy(kv, ids, kv_len, sink, window, trace=None):
    """BASELINE `sink=S, window=W`: while L_kv > S+W: prune(ids, kv, S, L_kv-W-1),
    cut end snapped with `snap_cut_end`."""
    while kv_len > sink + window:
        end = snap_cut_end(ids, kv_len - window - 1)
        ids, kv = prune_id_and_kv_cache(ids, kv, sink, end, trace)
        kv_len -= end - sink + 1
    return kv, ids, kv_len
