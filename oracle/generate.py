"""Generation loop + per-chunk streaming driver (oracle; test infrastructure only).

Restates
  * ``generate/streaming_generate_qwen.py:8-127`` (``_sample``: prefill + decode loop,
    fp32 last-row logits :73, logits processors :75, argmax/multinomial :95-99, EOS :101-109)
  * ``generate/prepare_generation.py:31-35,131-134`` (only the un-cached suffix is fed;
    pixels dropped on decode steps; position_ids=None -> recomputed in forward)
  * ``qwen2/model_forward.py:119-126`` + ``qwen2/language_forward.py:323-325`` (shrink mode:
    rope index recomputed from the FULL ids every forward; ids padded by one 0 per step)
  * ``inference.py:309-517`` (the per-chunk loop) on tokenizer-free synthetic inputs.
HF's RepetitionPenaltyLogitsProcessor (transformers ``generation/logits_process.py``) is
restated in ``repetition_penalty``.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional

import numpy as np
import torch

from . import kv_policy, qwen_range as qr
from .model import ModelCfg, model_forward
from .rope_index import get_1d_rope_index, get_rope_index, get_rope_index_2_5


def repetition_penalty(logits_f32: torch.Tensor, ids, penalty: float):
    """score<0 -> score*penalty else score/penalty for every token id present in `ids`."""
    if penalty == 1.0:
        return logits_f32
    idx = torch.unique(torch.as_tensor(ids, dtype=torch.long))
    sc = logits_f32[idx]
    logits_f32 = logits_f32.clone()
    logits_f32[idx] = torch.where(sc < 0, sc * penalty, sc / penalty)
    return logits_f32


def warp_scores(scores: torch.Tensor, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0) -> torch.Tensor:
    """The warpers HF's generate() appends to the processor list under do_sample=True, in its order (transformers
    generation/utils.py _get_logits_processor; generation/logits_process.py): TemperatureLogitsWarper (scores / T),
    TopKLogitsWarper (everything below the k-th largest score -> -inf; ties at the threshold survive), TopPLogitsWarper
    (sort ascending, softmax, cumulative sum; tokens whose cumulative probability is <= 1 - top_p -> -inf, the largest always
    survives).  The reference reaches them through `logits_processor(input_ids, next_token_logits)`,
    generate/streaming_generate_qwen.py:75."""
    sc = scores.clone()
    if temperature != 1.0:
        sc = sc / temperature
    if top_k and top_k > 0:
        k = min(int(top_k), sc.numel())
        sc = sc.masked_fill(sc < torch.topk(sc, k).values[-1], float("-inf"))
    if top_p < 1.0:
        sorted_sc, sorted_idx = torch.sort(sc, descending=False)
        cum = sorted_sc.softmax(dim=-1).cumsum(dim=-1)
        remove = cum <= (1 - top_p)
        remove[-1:] = False
        sc = sc.masked_fill(torch.zeros_like(remove).scatter(0, sorted_idx, remove), float("-inf"))
    return sc


@dataclass
class GenOut:
    sequences: List[int]
    past_key_values: object
    logits: List[torch.Tensor] = field(default_factory=list)   # fp32 raw last-row logits per step
    own: List[int] = field(default_factory=list)               # the token each step chose itself (differs from the sequence under force_tokens)
    n_generated: int = 0


def generate(w, cfg: ModelCfg, ids: List[int], kv, video_grid_thw, pixel_values=None, grid_thw=None,
             max_new_tokens=20, rep_penalty=1.05, eos_ids=(151645, 151643), suppress_eos=False,
             do_sample=False, temperature=1.0, top_k: int = 0, top_p: float = 1.0, generator: Optional[torch.Generator] = None,
             keep_logits=False, all_text=False, second_per_grid_t=1.0, pos_mode="shrink", sargs: Optional[dict] = None,
             force_tokens: Optional[List[int]] = None) -> GenOut:
    """One ``model.generate(**inputs, past_key_values=kv, streaming_args=...)`` call.  `sargs` is the part of StreamingArgs
    that outlives a call: {"last_cache_position": int} (streaming_args.py:9, used by pos_mode="append").
    `force_tokens` (test aid, not in the reference): teacher forcing -- each step's own choice is recorded in `.own` and the
    given token is appended instead, so two arithmetic variants can be compared forward by forward on one history."""
    if sargs is None:
        sargs = {"last_cache_position": -1}

    def index(seq, grids):
        if all_text:                                                        # qwen2_5/model_forward.py:99
            return get_1d_rope_index(len(seq))
        if cfg.vision.arch == "qwen2_5":                                    # qwen2_5/model_forward.py:101-110
            return get_rope_index_2_5(seq, grids, cfg.vision.spatial_merge_size, cfg.video_token_id,
                                      cfg.vision_start_token_id, second_per_grid_t, cfg.vision.tokens_per_second)
        return get_rope_index(seq, grids, cfg.vision.spatial_merge_size, cfg.video_token_id, cfg.vision_start_token_id)

    ids = list(ids)
    sa_ids = list(ids)                       # streaming_args.input_ids (padded with 0 per forward)
    out_logits = []
    own = []
    n_new = 0
    first = True
    if force_tokens is not None:
        max_new_tokens = len(force_tokens)
    while True:
        kv_len = kv.get_seq_length()
        new_ids = ids[kv_len:]                                              # prepare_generation.py:31-35
        if pos_mode == "shrink":                                            # model_forward.py:119-126
            pos3 = index(sa_ids, video_grid_thw)
        elif kv_len == 0:                                                   # append, branch 1 (:77-90)
            pos3 = index(new_ids, grid_thw if grid_thw is not None else video_grid_thw)
        elif len(new_ids) != 1:                                             # append, chunk prefill (:91-99)
            pos3 = index(new_ids, grid_thw if grid_thw is not None else []) + (sargs["last_cache_position"] + 1)
        else:                                                               # append, decode (:100-114)
            import numpy as _np
            pos3 = _np.full((3, 1), sargs["last_cache_position"] + 1, dtype=_np.float32 if cfg.vision.arch == "qwen2_5" else _np.int64)
        if pos_mode == "append":
            sargs["last_cache_position"] = pos3[0, -1].item()               # :117
        has_vid = cfg.video_token_id in new_ids
        logits = model_forward(w, cfg, new_ids, kv, pos3,
                               pixel_values if (first and has_vid) else None,
                               grid_thw if (first and has_vid) else None, pos_mode=pos_mode)
        first = False
        sa_ids = sa_ids + [0]                                               # language_forward.py:323-325
        raw = logits[-1].float()                                            # streaming_generate_qwen.py:73
        if keep_logits:
            out_logits.append(raw.clone())
        sc = repetition_penalty(raw, ids, rep_penalty)                      # :75
        if suppress_eos:
            sc = sc.clone()
            sc[list(eos_ids)] = float("-inf")
        if do_sample:                                                       # :92-97
            probs = torch.softmax(warp_scores(sc, temperature, top_k, top_p), dim=-1)
            nxt = int(torch.multinomial(probs, 1, generator=generator))
        else:
            nxt = int(torch.argmax(sc))                                     # :99
        own.append(nxt)
        if force_tokens is not None:
            nxt = int(force_tokens[n_new])
        ids.append(nxt)
        n_new += 1
        if nxt in eos_ids or n_new >= max_new_tokens:                       # :101-109
            break
    return GenOut(ids, kv, out_logits, own, n_new)


@dataclass
class StreamCfg:
    """Knobs of streaming_inference (inference.py:181-207) that affect indices."""
    policy: str = "structural"            # "structural" (reference) | "sink_window" (BASELINE) | "none"
    window_size: int = 16
    text_round: int = 16
    text_sink: Optional[int] = None
    text_sliding_window: Optional[int] = None
    sink: int = 4
    window: int = 2048
    max_new_tokens: int = 20              # MAX_TOKEN_PER_DURATION, inference.py:45
    repetition_penalty: float = 1.05
    suppress_eos: bool = False
    assistant_start_bias: int = 3         # len(tok("<|im_start|>assistant\n"))  inference.py:228
    assistant_end_bias: int = 2           # len(tok(" ...<|im_end|>"))           inference.py:229
    all_text: bool = False                # StreamingArgs.all_text: 1-D rope (qwen2_5/model_forward.py:99)
    second_per_grid_t: float = 1.0        # 2 / FPS (qwen2_5/pos_emb.py:107-108)
    pos_mode: str = "shrink"              # StreamingArgs.pos_mode (streaming_args.py:2-6)


def streaming_loop(w, cfg: ModelCfg, scfg: StreamCfg, n_chunks: int,
                   chunk_source: Callable[[int], tuple], keep_logits=False, force_tokens: Optional[List[List[int]]] = None,
                   dense_prefill_chunks: int = 0):
    """The loop of inference.py:309-517 on synthetic inputs.

    chunk_source(i) -> (chunk_ids: list[int], pixel_values (N,1176), grid_thw [[t,h,w]])
    Returns dict(ids_per_chunk, new_tokens_per_chunk, kv_len_per_chunk, trace, logits).
    `dense_prefill_chunks` (build-defined, BASELINE configs[4]): the first N chunks' user turns pile up unanswered and are
    forwarded in ONE generate() call with the N-th -- all their frames in one forward, as the recompute path of
    inference.py:423-438 feeds every retained chunk's frames; chunk_source(i, piling=True) omits the assistant header.
    """
    kv = kv_policy.ListKV(cfg.text.num_layers)
    prev_ids: Optional[List[int]] = None
    sargs = {"last_cache_position": -1}                                     # StreamingArgs lives for the whole stream (inference.py:213)
    grids: List[List[int]] = []
    trace, new_tokens, kv_lens, all_logits, ids_hist, own_hist, generated = [], [], [], [], [], [], []
    for i in range(n_chunks):
        chunk_trace = []
        # ---- process_past_kv (inference.py:319)
        if prev_ids is not None:
            if scfg.policy == "structural":
                kv, prev_ids = kv_policy.process_past_kv(
                    kv, i, prev_ids, scfg.text_round, scfg.window_size, scfg.text_sink,
                    scfg.text_sliding_window, scfg.assistant_start_bias, scfg.assistant_end_bias, chunk_trace)
            elif scfg.policy == "sink_window":
                kv, prev_ids, _ = kv_policy.sink_window_policy(
                    kv, prev_ids, kv.get_seq_length(), scfg.sink, scfg.window, chunk_trace)
        trace.append(chunk_trace)
        if dense_prefill_chunks and i < dense_prefill_chunks:
            chunk_ids, pix, grid = chunk_source(i, piling=i < dense_prefill_chunks - 1)
            if i == 0:
                pile_ids, pile_pix, pile_grid = list(chunk_ids), [pix], [list(g) for g in grid]
            else:
                pile_ids, pile_pix, pile_grid = pile_ids + list(chunk_ids)[1:], pile_pix + [pix], pile_grid + [list(g) for g in grid]
            if i < dense_prefill_chunks - 1:
                continue
            chunk_ids, pix, grid = pile_ids, torch.cat(pile_pix, 0), pile_grid
        else:
            chunk_ids, pix, grid = chunk_source(i)
        # ---- ids = cat(prev, new[skip dup "\n"])  (inference.py:397-406)
        if prev_ids is None:
            ids = list(chunk_ids)
        elif prev_ids[-1] != qr.LF:
            ids = prev_ids + list(chunk_ids)
        else:
            ids = prev_ids + list(chunk_ids)[1:]
        grids = grids + [list(g) for g in grid]                             # :411-416 (never pruned)
        cur_len = len(ids)
        out = generate(w, cfg, ids, kv, grids, pix, grid, scfg.max_new_tokens, scfg.repetition_penalty,
                       suppress_eos=scfg.suppress_eos, keep_logits=keep_logits, all_text=scfg.all_text,
                       second_per_grid_t=scfg.second_per_grid_t, pos_mode=scfg.pos_mode, sargs=sargs,
                       force_tokens=None if force_tokens is None else force_tokens[len(new_tokens)])
        gen = out.sequences
        own_hist.append(out.own)
        generated.append(gen[cur_len:])                                     # without the appended <|im_end|>
        if gen[-1] != qr.IM_END:                                            # :457-459
            gen = gen + [qr.IM_END]
        new_tokens.append(gen[cur_len:])
        kv_lens.append(kv.get_seq_length())
        all_logits.append(out.logits)
        prev_ids = list(gen)                                                # :482
        ids_hist.append(list(gen))
    return dict(ids=ids_hist, new_tokens=new_tokens, kv_len=kv_lens, trace=trace, logits=all_logits, kv=kv, own=own_hist,
                generated=generated)
