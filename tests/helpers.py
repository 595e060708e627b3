"""Shared test helpers: config bridging between the product and the oracle, chunk sources."""
from __future__ import annotations

import torch

from oracle import generate as og
from oracle import model as om

import streaming_vlm_amd as S
from streaming_vlm_amd import config as C

_V_KEYS = ["depth", "embed_dim", "num_heads", "mlp_hidden", "patch_size", "temporal_patch_size", "spatial_merge_size", "in_channels",
           "arch", "window_size", "fullatt_block_indexes", "out_hidden", "tokens_per_second"]
_T_KEYS = ["hidden_size", "num_layers", "num_heads", "num_kv_heads", "head_dim", "intermediate_size", "vocab_size", "rms_eps",
           "rope_theta", "mrope_section", "tie_word_embeddings"]


def oracle_cfg(cfg: C.ModelConfig) -> om.ModelCfg:
    return om.ModelCfg(om.VisionCfg(**{k: getattr(cfg.vision, k) for k in _V_KEYS}),
                       om.TextCfg(**{k: getattr(cfg.text, k) for k in _T_KEYS}),
                       video_token_id=cfg.video_token_id, vision_start_token_id=cfg.vision_start_token_id)


def chunk_source(proc, video, previous_text="", query="Commentate on this match", chunk_duration=1, skip_first_chunk=0):
    """Per-chunk (ids, pixel_values, grid) exactly as streaming_inference builds them (inference.py:351-395)."""
    def src(i, piling=False):
        start = (i + skip_first_chunk) * chunk_duration
        prompt = f"Time={start:.1f}-{start + chunk_duration:.1f}s"
        frames = video.chunk(start, chunk_duration)
        if i == 0:
            conv = [{"role": "previous text", "content": previous_text},
                    {"role": "user", "content": [{"type": "text", "text": prompt}, {"type": "video", "video": ""},
                                                 {"type": "text", "text": query}]}]
            text = proc.apply_chat_template(conv, tokenize=False, add_generation_prompt=not piling)
        else:
            conv = [{"role": "user", "content": [{"type": "text", "text": prompt}, {"type": "video", "video": ""}]}]
            text = "\n" + proc.apply_chat_template(conv, tokenize=False, add_generation_prompt=not piling)[S.SYSTEM_PROMPT_OFFSET:]
        out = proc(text=[text], videos=frames, return_tensors="pt")
        return out["input_ids"][0].tolist(), out["pixel_values_videos"], out["video_grid_thw"].tolist()
    return src


def run_oracle_stream(cfg, sd, n_chunks, size=56, fps=1.0, policy="sink_window", sink=4, window=64, max_new=8, suppress_eos=True,
                      previous_text="hello world", keep_logits=False, force_tokens=None, dense_prefill_chunks=0, **policy_kw):
    proc = S.SyntheticProcessor()
    video = S.SyntheticVideo(size, fps, 0)
    scfg = og.StreamCfg(policy=policy, sink=sink, window=window, max_new_tokens=max_new, suppress_eos=suppress_eos, **policy_kw)
    return og.streaming_loop(sd, oracle_cfg(cfg), scfg, n_chunks, chunk_source(proc, video, previous_text), keep_logits=keep_logits,
                             force_tokens=force_tokens, dense_prefill_chunks=dense_prefill_chunks)


def decisive_offset(size=56, max_new=8, all_text=False):
    """Copy distance of `weights.decisive_state_dict` for a stream geometry: far enough back that no decode step of a chunk
    (the first chunk carries 4 query tokens more) looks at the chunk's own vision span, whose rows all share one temporal
    position (M-RoPE) and hold ViT features, not token embeddings."""
    h, w = (size, size) if isinstance(size, int) else (int(size[0]), int(size[1]))
    gh, gw = h // 28, w // 28
    return (gh * gw if all_text else max(gh, gw)) + max_new + 12


def decisive_weights(cfg, size=56, max_new=8, all_text=False, seed=0, **_):
    from streaming_vlm_amd.weights import decisive_state_dict
    return decisive_state_dict(cfg, seed, "cpu", offset=decisive_offset(size, max_new, all_text))


def greedy_margins(ref, suppress_eos=True, penalty=1.05):
    """Top-2 margin of the score that decides each greedy token of an oracle stream: AFTER the repetition penalty and the EOS
    suppression (streaming_generate_qwen.py:75-99)."""
    out = []
    for ids, gen, logits in zip(ref["ids"], ref["generated"], ref["logits"]):
        n_prompt = len(ids) - len(gen) - (1 if len(ids) and ids[-1] == 151645 and (not gen or gen[-1] != 151645) else 0)
        for j, b in enumerate(logits):
            sc = og.repetition_penalty(b.clone(), ids[:n_prompt + j], penalty)
            if suppress_eos:
                sc[[151645, 151643]] = float("-inf")
            top2 = torch.topk(sc, 2).values
            out.append(float(top2[0] - top2[1]))
    return out


def run_engine_stream(model, n_chunks, size=56, fps=1.0, policy="sink_window", sink=4, window=64, max_new=8, suppress_eos=True,
                      previous_text="hello world", **kw):
    trace, counts, ids_log = [], [], []
    size_s = f"{size[1]}x{size[0]}" if isinstance(size, (tuple, list)) else f"{size}x{size}"      # (h, w) -> "WxH"
    base = "Qwen2_5" if model._svlm_engine.cfg.family == "qwen2_5" else "Qwen2"
    res = S.streaming_inference(model=model, processor=S.SyntheticProcessor(), video_path=f"synthetic://{size_s}@{fps:g}fps",
                                model_base=base, duration=n_chunks, previous_text=previous_text, kv_policy=policy, sink=sink,
                                window=window, do_sample=False, max_new_tokens=max_new, suppress_eos=suppress_eos, quiet=True,
                                trace=trace, token_counts=counts, ids_log=ids_log, **kw)
    return res, trace, counts, ids_log
