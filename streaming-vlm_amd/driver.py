"""Per-chunk streaming driver on the HIP engine -- keeps the call surface of the reference's
``src/streaming_vlm/inference/inference.py`` (``streaming_inference`` :181-522 with the same keyword
names, ``process_past_kv`` :87-172, ``prune_id_and_kv_cache`` :50-61, ``contiguous_id_and_kv`` :63-68,
``load_model_and_processor`` :70-85) so the reference's harnesses (eval/efficiency/efficiency_test.py:74,
eval/livesports3kcc/distributed_generate_streaming.py:100) can call it as a drop-in.

What changes underneath: ids live on the host (no ``.tolist()`` device syncs), the KV cache is a
``KVPool`` whose prune/move are slot-table edits, and ``model.generate`` runs the HIP kernels.
Build-defined additions are keyword-only: ``kv_policy="sink_window"`` with ``sink``/``window``
(BASELINE's token-count policy), ``do_sample``, ``suppress_eos``, ``trace``.
"""
from __future__ import annotations

import json
import os
import sys
import time
from contextlib import contextmanager
from typing import List, Optional

import torch

from .config import IM_END, VIDEO_PAD, VISION_END, VISION_START
from .spans import SYSTEM_PROMPT_OFFSET, TOKEN_IDS, get_qwen_range
from .kv_pool import KVPool
from .model import StreamingArgs, StreamingQwen2VL, convert_qwen2_5_to_streaming, convert_qwen2_to_streaming
from .synthetic import SyntheticProcessor, SyntheticVideo

TOTAL_VIDEO_DURATION = 6000
DEFAULT_CHUNK_DURATION = 1
DEFAULT_WINDOW_SIZE = 16
DEFAULT_TEXT_ROUND = 16
DEFAULT_TEXT_SINK = 512
DEFAULT_TEXT_SLIDING_WINDOW = 512
DEFAULT_TEMPERATURE = 0.9
DEFAULT_REPETITION_PENALTY = 1.05
MAX_TOKEN_PER_DURATION = 20
FPS = float(os.environ.get("QWENVL_FPS", "2.0"))       # the reference reads qwen_vl_utils.FPS (env-driven)
GRID_ROWS_KEPT = 4096          # rows of streaming_args.video_grid_thw kept on an unbounded stream: the first and the last 2048 (see the loop)


# ----------------------------------------------------------------------------- WebVTT output (reference: utils/vtt_utils.py:5-16)
@contextmanager
def open_vtt(path):
    """Append to `path`, writing the WEBVTT header first when the file is new."""
    new = not os.path.exists(path)
    with open(path, "w" if new else "a", encoding="utf-8") as f:
        if new:
            f.write("WEBVTT\n\n")
        yield f


def sec2ts(sec: float) -> str:
    whole = int(sec)
    ms = int((sec - whole) * 1000)
    return f"{whole // 3600:02d}:{whole % 3600 // 60:02d}:{whole % 60:02d}.{ms:03d}"


# ----------------------------------------------------------------------------- KV / id edits
def prune_id_and_kv_cache(input_ids, past_key_values, start_index, end_index, trace=None):
    """Delete the CLOSED interval [start_index, end_index] from the ids and from every layer's K, V."""
    input_ids = torch.cat([input_ids[:, :start_index], input_ids[:, end_index + 1:]], dim=1)
    if past_key_values is not None:
        past_key_values.release_reserved()
        past_key_values.prune(int(start_index), int(end_index))
    if trace is not None:
        trace.append(("prune", int(start_index), int(end_index)))
    return input_ids, past_key_values


def resort_id_and_kv(input_ids, past_key_values, src_start_idx, src_end_idx, dst_idx, trace=None):
    """Move [src_start_idx, src_end_idx] to directly after dst_idx (assistant text -> previous-text block)."""
    assert dst_idx < src_start_idx <= src_end_idx
    input_ids = torch.cat([input_ids[:, :dst_idx + 1], input_ids[:, src_start_idx:src_end_idx + 1],
                           input_ids[:, dst_idx + 1:src_start_idx], input_ids[:, src_end_idx + 1:]], dim=1)
    if past_key_values is not None:
        past_key_values.release_reserved()
        past_key_values.move(int(src_start_idx), int(src_end_idx), int(dst_idx))
    if trace is not None:
        trace.append(("move", int(src_start_idx), int(src_end_idx), int(dst_idx)))
    return input_ids, past_key_values


def contiguous_id_and_kv(input_ids, past_key_values):
    """The reference re-materialises every layer (.contiguous()); the pool needs nothing unless it is
    fragmented enough to defragment in place."""
    if past_key_values is not None and past_key_values.fragmentation() > 0.5:
        past_key_values.defragment()
    return input_ids.contiguous(), past_key_values


def snap_cut_end(ids_row, end: int) -> int:
    """Extend a token-count cut forward to <|vision_end|> when it would split a vision span."""
    t = int(ids_row[end])
    if t not in (VISION_START, VIDEO_PAD):
        return end
    # one vectorised search (a per-token int(tensor[j]) walk costs ~1 us per video token on the chunk's critical path)
    hit = (ids_row[end:] == VISION_END).nonzero()
    if hit.numel() == 0:
        raise ValueError("unterminated vision span")
    return end + int(hit[0])


def sink_window_evict(past_key_values, input_ids, sink: int, window: int, trace=None):
    """BASELINE policy: while L_kv > sink + window: prune(ids, kv, sink, L_kv - window - 1)."""
    kv_len = past_key_values.get_seq_length()
    while kv_len > sink + window:
        end = snap_cut_end(input_ids[0], kv_len - window - 1)
        input_ids, past_key_values = prune_id_and_kv_cache(input_ids, past_key_values, sink, end, trace)
        kv_len = past_key_values.get_seq_length()
    return past_key_values, input_ids


def process_past_kv(past_key_values, i, text_round, visual_round, full_conversation_history, prev_generated_ids,
                    assistant_start_bias, assistant_end_bias, recent_video_window_clips, recent_pixel_values_videos,
                    text_sink, text_sliding_window, trace=None):
    """Structural eviction at the start of round i (reference inference.py:87-172)."""
    ids, kv = prev_generated_ids, past_key_values
    hist = full_conversation_history
    if i >= text_round:
        # oldest retained assistant turn: its text migrates into the "previous text" block
        assert hist[0]["role"] == "previous text"
        assert hist[-2 * text_round]["role"] == "user" and hist[-(2 * text_round - 1)]["role"] == "assistant"
        hist[0]["content"] += hist[-(2 * text_round - 1)]["content"][:-4]
        a_s, a_e = get_qwen_range(ids, "assistant", 0)
        _, p_e = get_qwen_range(ids, "previous text", 0, contain_lf=False)
        src_s = a_s + assistant_start_bias
        src_e = a_e - assistant_end_bias - (1 if int(ids[0, a_e]) == TOKEN_IDS["\n"] else 0)
        if src_s <= src_e:
            ids, kv = resort_id_and_kv(ids, kv, src_s, src_e, p_e - 1, trace)
        user_turn = hist[-2 * text_round]["content"]
        for k, item in enumerate(user_turn):
            if item["type"] == "text":
                del user_turn[k]
                break
        del hist[-(2 * text_round - 1)]
        if visual_round > text_round:
            u_s, u_e = get_qwen_range(ids, "user_text", -text_round, contain_lf=False)
            ids, kv = prune_id_and_kv_cache(ids, kv, u_s, u_e, trace)
        a_s, a_e = get_qwen_range(ids, "assistant", -text_round)
        ids, kv = prune_id_and_kv_cache(ids, kv, a_s, a_e, trace)
    if i >= visual_round:
        recent_video_window_clips.pop(0)
        recent_pixel_values_videos.pop(0)
        if visual_round < text_round:
            turn = hist[-2 * visual_round]
            turn["content"] = [item for item in turn["content"] if item["type"] != "video"]
            v_s, v_e = get_qwen_range(ids, "vision", 0)
            ids, kv = prune_id_and_kv_cache(ids, kv, v_s, v_e, trace)
    if i >= max(visual_round, text_round):
        del hist[1]
        u_s, u_e = get_qwen_range(ids, "user", 0)
        ids, kv = prune_id_and_kv_cache(ids, kv, u_s, u_e, trace)
    if i > 0:
        if text_sink is not None or text_sliding_window is not None:
            p_s, p_e = get_qwen_range(ids, "previous text", 0)
            cut_s = p_s + text_sink + 4 if text_sink is not None else p_s
            cut_e = p_e - text_sliding_window - 1 if text_sliding_window is not None else p_e
            if cut_s <= cut_e:
                ids, kv = prune_id_and_kv_cache(ids, kv, cut_s, cut_e, trace)
        ids, kv = contiguous_id_and_kv(ids, kv)
    return kv, ids, recent_video_window_clips, recent_pixel_values_videos


def _history_edits(i, text_round, visual_round, hist, recent_video_window_clips, recent_pixel_values_videos):
    """The part of process_past_kv (inference.py:87-172) that edits the conversation history and the retained clips, without the
    index arithmetic: what is left for the host when the eviction indices come from the device (`apply_eviction_plan`)."""
    if i >= text_round:
        assert hist[0]["role"] == "previous text"
        assert hist[-2 * text_round]["role"] == "user" and hist[-(2 * text_round - 1)]["role"] == "assistant"
        hist[0]["content"] += hist[-(2 * text_round - 1)]["content"][:-4]
        user_turn = hist[-2 * text_round]["content"]
        for k, item in enumerate(user_turn):
            if item["type"] == "text":
                del user_turn[k]
                break
        del hist[-(2 * text_round - 1)]
    if i >= visual_round:
        recent_video_window_clips.pop(0)
        recent_pixel_values_videos.pop(0)
        if visual_round < text_round:
            turn = hist[-2 * visual_round]
            turn["content"] = [item for item in turn["content"] if item["type"] != "video"]
    if i >= max(visual_round, text_round):
        del hist[1]


def apply_eviction_plan(ids, kv, ops, trace=None):
    """Apply the op list of `svlm_evict_plan` (device-side span finder + policy) to the host ids and the KV pool: the
    same prune / move calls, with the indices the device computed."""
    for op in ops:
        if op[0] == "prune":
            ids, kv = prune_id_and_kv_cache(ids, kv, op[1], op[2], trace)
        else:
            ids, kv = resort_id_and_kv(ids, kv, op[1], op[2], op[3], trace)
    return ids, kv


# ----------------------------------------------------------------------------- loading
def required_max_len(tokens_per_chunk, max_new_tokens=MAX_TOKEN_PER_DURATION, kv_policy="structural", window_size=DEFAULT_WINDOW_SIZE,
                     text_round=DEFAULT_TEXT_ROUND, text_sink=None, text_sliding_window=None, sink=4, window=2048, num_chunks=TOTAL_VIDEO_DURATION,
                     previous_text_tokens=0, query_tokens=8) -> int:
    """Largest logical sequence (cached rows + un-cached suffix + generated tokens) a stream with these settings reaches: the
    `max_len` its engine needs (rope table, slot table, KV pool).  The reference grows torch tensors as it goes; here HBM is
    sized once, for the whole stream, from the eviction policy."""
    chunk = int(tokens_per_chunk) + 24 + int(max_new_tokens) + 2        # user header + vision span + assistant turn
    head = 16 + query_tokens                                            # system prompt, first chunk's query
    if kv_policy == "sink_window":
        return sink + window + 2 * chunk + head
    grown = previous_text_tokens + num_chunks * (max_new_tokens + 1)     # what the previous-text block can have absorbed by the end
    if kv_policy == "none":
        return head + previous_text_tokens + (num_chunks + 1) * chunk
    pt = grown if (text_sink is None or text_sliding_window is None) else min(grown, text_sink + text_sliding_window + 12)
    rounds_v, rounds_t = min(window_size, num_chunks), min(text_round, num_chunks)
    return head + pt + 8 + rounds_v * (int(tokens_per_chunk) + 24) + rounds_t * (max_new_tokens + 6) + chunk + max_new_tokens


def load_model_and_processor(model_path, model_base="Qwen2_5", max_len=None, max_new_tokens=None):
    """HF checkpoint -> converted model + AutoProcessor (needs local weights; nothing is downloaded).
    ``random:<2b|7b|tiny>[:seed]`` builds random-init weights of the real shapes with the synthetic processor.
    `max_len` / `max_new_tokens` size the engine (see `required_max_len`; `streaming_inference` derives them from its own
    arguments when it loads the model itself)."""
    ekw = {}
    if max_len is not None:
        ekw["max_len"] = int(max_len)
    if max_new_tokens is not None:
        ekw["max_new_tokens"] = int(max_new_tokens)
    if model_path.startswith("random:"):
        from . import config as C
        from .weights import random_state_dict
        parts = model_path.split(":")
        cfg = {"2b": C.qwen2_vl_2b, "7b": C.qwen2_vl_7b, "tiny": C.tiny, "2.5-3b": C.qwen2_5_vl_3b, "2.5-7b": C.qwen2_5_vl_7b,
               "tiny-2.5": C.tiny_2_5}[parts[1]]()
        seed = int(parts[2]) if len(parts) > 2 else 0
        return StreamingQwen2VL(cfg, random_state_dict(cfg, seed, "cuda"), "cuda", **ekw), SyntheticProcessor()
    from transformers import AutoProcessor
    if model_base == "Qwen2_5":                        # reference inference.py:72-78
        from transformers import Qwen2_5_VLForConditionalGeneration
        model = Qwen2_5_VLForConditionalGeneration.from_pretrained(model_path, torch_dtype="auto", device_map="cuda")
        return convert_qwen2_5_to_streaming(model, **ekw), AutoProcessor.from_pretrained(model_path, use_fast=False)
    if model_base != "Qwen2":
        raise ValueError(f"model_base must be 'Qwen2' or 'Qwen2_5', not {model_base!r}")
    from transformers import Qwen2VLForConditionalGeneration
    model = Qwen2VLForConditionalGeneration.from_pretrained(model_path, torch_dtype="auto", device_map="cuda")
    return convert_qwen2_to_streaming(model, **ekw), AutoProcessor.from_pretrained(model_path, use_fast=False)


def printq(*args, quiet=False, **kwargs):
    if not quiet:
        print(*args, **kwargs)


# ----------------------------------------------------------------------------- main loop
def _fetch_chunk(video, start_s, duration_s, model, device):
    """Frames of one chunk; a source of native-size frames (`spatial_resize = True`) goes through the reference's
    _spatial_resize_video (inference.py:342), resized on the device by svlm_resize_bicubic_aa_u8."""
    frames = video.chunk(start_s, duration_s)
    if getattr(video, "spatial_resize", False) and torch.is_tensor(frames):
        from .ingest import spatial_resize_video
        eng = getattr(model, "_svlm_engine", None)
        frames = spatial_resize_video(frames, getattr(eng, "ops", None), device)
    return frames


def streaming_inference(model_path="", video_path="", output_dir=None, model_base="Qwen2_5", model=None, processor=None,
                        window_size=DEFAULT_WINDOW_SIZE, chunk_duration=DEFAULT_CHUNK_DURATION, text_round=DEFAULT_TEXT_ROUND,
                        previous_text="", test_data_json=None, test_data_idx=None, pos_mode="shrink", all_text=False,
                        skip_first_chunk=0, recompute=False, gt_json=None, gt_idx=None, text_sink=None,
                        text_sliding_window=None, temperature=DEFAULT_TEMPERATURE, duration=TOTAL_VIDEO_DURATION,
                        query="Commentate on this match", repetition_penalty=DEFAULT_REPETITION_PENALTY, quiet=False,
                        emit_json=False, time_test=False, *, kv_policy="structural", sink=4, window=2048, do_sample=True,
                        max_new_tokens=MAX_TOKEN_PER_DURATION, suppress_eos=False, trace: Optional[List] = None,
                        token_counts: Optional[List] = None, ids_log: Optional[List] = None, video=None,
                        generator=None, keep_logits=False, chunk_callback=None, vision_lookahead=True, force_tokens=None,
                        max_len=None, top_k=None, top_p=None, dense_prefill_chunks=0, device_policy=False):
    # The reference synchronises the device around every section to print per-section times.  Nobody reads them when the
    # loop is quiet and not under time_test, and each of the dozen syncs per chunk is host time the GPU spends idle.
    timed_sections = time_test or not quiet

    def _sync():
        if timed_sections and torch.cuda.is_available():
            torch.cuda.synchronize()

    if window_size % chunk_duration:
        raise AssertionError("window_size must be divisible by chunk_duration")
    if test_data_json is not None:
        raise NotImplementedError("LMMDataset-driven input (training data pipeline) is outside the hot path")
    if kv_policy not in ("structural", "sink_window", "none"):
        raise ValueError(f"unknown kv_policy {kv_policy!r}")

    streaming_args = StreamingArgs(pos_mode=pos_mode, all_text=all_text)
    if video is None:
        video = SyntheticVideo.from_path(video_path)
    if video is None:
        raise FileNotFoundError(
            f"{video_path!r}: real video decode (decord + bicubic resize) is host I/O outside the hot path; "
            f"pass video=<object with .chunk(start_s, duration_s) -> uint8 (T,3,H,W)> or a synthetic://WxH@Ffps path")
    num_chunks = int((duration + chunk_duration - 1) // chunk_duration)

    def engine_len(proc):
        """`max_len` for an engine built HERE: the policy's bound with the token count of the stream's first chunk."""
        if max_len is not None:
            return int(max_len)
        from .ingest import resized_shape
        f0 = video.chunk(skip_first_chunk * chunk_duration, chunk_duration)
        h, w = f0.shape[2], f0.shape[3]
        if getattr(video, "spatial_resize", False):
            h, w = resized_shape(h, w, f0.shape[0])
        n_tok = ((f0.shape[0] + 1) // 2) * (h // 28) * (w // 28)
        n_prev = len(proc(text=previous_text)["input_ids"][0]) if previous_text else 0
        # the loop below evicts with visual_round = window_size (the reference's own quirk, inference.py:320: rounds, not seconds), so
        # window_size rounds are retained whatever chunk_duration is
        return required_max_len(n_tok, max_new_tokens, kv_policy, window_size, text_round, text_sink, text_sliding_window,
                                sink, window, num_chunks, n_prev)

    if model is None or processor is None:
        if model_path.startswith("random:") or processor is not None:
            model, processor = load_model_and_processor(model_path, model_base, engine_len(processor or SyntheticProcessor()), max_new_tokens)
        else:           # the checkpoint's own processor is only known once it is loaded
            from transformers import AutoProcessor
            processor = AutoProcessor.from_pretrained(model_path, use_fast=False)
            model, _ = load_model_and_processor(model_path, model_base, engine_len(processor), max_new_tokens)
    elif getattr(model, "_svlm_engine", None) is None:
        if model_base not in ("Qwen2", "Qwen2_5"):
            raise ValueError(f"model_base must be 'Qwen2' or 'Qwen2_5', not {model_base!r}")
        model = (convert_qwen2_5_to_streaming if model_base == "Qwen2_5" else convert_qwen2_to_streaming)(
            model, max_len=engine_len(processor), max_new_tokens=max_new_tokens)
    device = model.device

    assistant_start_bias = len(processor(text="<|im_start|>assistant\n")["input_ids"][0])
    assistant_end_bias = len(processor(text=" ...<|im_end|>")["input_ids"][0])

    gt_dict = None
    if gt_json is not None:
        with open(gt_json, "r") as f:
            for ln, line in enumerate(f):
                if ln == gt_idx:
                    gt_dict = json.loads(line)
                    break

    if output_dir is not None:
        if os.path.exists(output_dir):
            os.remove(output_dir)
        with open_vtt(output_dir):
            pass
        printq(f"Subtitles will be written to: {output_dir}", quiet=quiet)

    past_key_values = None
    full_conversation_history = []
    prev_generated_ids = None
    recent_video_window_clips, recent_pixel_values_videos = [], []
    responses, time_results = [], []
    printq(f"num_chunks: {num_chunks}", quiet=quiet)

    # BASELINE configs[4] ("dense-frame prefill then live decode"): the first `dense_prefill_chunks` chunks are not answered one by
    # one; their user turns (Time=..s + frames) pile up and go through ONE generate() call with the last of them -- every retained
    # chunk's frames in one forward, the input shape of the reference's recompute path (inference.py:423-438) and of LiveCC's
    # multi-frame opening turn (baselines/livecc/demo/infer.py:24-32) -- after which the stream continues chunk by chunk.
    dense = int(dense_prefill_chunks)
    if dense and (kv_policy == "structural" or recompute or gt_json is not None):
        raise ValueError("dense_prefill_chunks works with kv_policy='sink_window' / 'none' on a plain stream")
    dense_ids, dense_pix, dense_grids = [], [], []

    lookahead = None          # (frames, device patches, grid) of the next chunk, fetched one chunk early
    for i in range(num_chunks):
        if chunk_callback is not None:
            chunk_callback(i)
        _sync()
        loop_start = time.perf_counter()
        section_time = {k: 0.0 for k in ["PKV", "CHECK", "VIDEO", "INPUT", "GEN", "POST"]}
        start_time = (i + skip_first_chunk) * chunk_duration
        chunk_trace = [] if trace is not None else None

        # ---- evict
        _sync(); _t = time.perf_counter()
        if prev_generated_ids is not None and device_policy and kv_policy in ("structural", "sink_window"):
            # f-1: spans and eviction indices from the device (svlm_evict_plan over the device copy of the ids); the host applies
            # the op list to its id tensor and to the KV pool's slot table, and keeps the conversation-history strings
            ops_dev = model._svlm_engine.ops
            if kv_policy == "structural":
                plan, _ = ops_dev.evict_plan(prev_generated_ids[0].tolist(), "structural", i, text_round, window_size, text_sink, text_sliding_window,
                                             assistant_start_bias, assistant_end_bias, device=device)
                _history_edits(i, text_round, window_size, full_conversation_history, recent_video_window_clips, recent_pixel_values_videos)
            else:
                plan, _ = ops_dev.evict_plan(prev_generated_ids[0].tolist(), "sink_window", sink=sink, window=window,
                                             kv_len=past_key_values.get_seq_length(), device=device)
                if len(recent_video_window_clips) >= window_size:
                    recent_video_window_clips.pop(0)
                    recent_pixel_values_videos.pop(0)
            prev_generated_ids, past_key_values = apply_eviction_plan(prev_generated_ids, past_key_values, plan, chunk_trace)
            if kv_policy == "structural" and i > 0:
                prev_generated_ids, past_key_values = contiguous_id_and_kv(prev_generated_ids, past_key_values)
        elif prev_generated_ids is not None:
            if kv_policy == "structural":
                past_key_values, prev_generated_ids, recent_video_window_clips, recent_pixel_values_videos = process_past_kv(
                    past_key_values, i, text_round=text_round, visual_round=window_size,
                    full_conversation_history=full_conversation_history, prev_generated_ids=prev_generated_ids,
                    assistant_start_bias=assistant_start_bias, assistant_end_bias=assistant_end_bias,
                    recent_video_window_clips=recent_video_window_clips,
                    recent_pixel_values_videos=recent_pixel_values_videos, text_sink=text_sink,
                    text_sliding_window=text_sliding_window, trace=chunk_trace)
            elif kv_policy == "sink_window":
                past_key_values, prev_generated_ids = sink_window_evict(past_key_values, prev_generated_ids, sink, window, chunk_trace)
                if len(recent_video_window_clips) >= window_size:
                    recent_video_window_clips.pop(0)
                    recent_pixel_values_videos.pop(0)
        if trace is not None:
            trace.append(chunk_trace)
        _sync(); section_time["PKV"] += time.perf_counter() - _t

        # ---- frames of this chunk
        _sync(); _t = time.perf_counter()
        ahead, lookahead = lookahead, None
        try:
            current_video_chunk = ahead[0] if ahead is not None else _fetch_chunk(video, start_time, chunk_duration, model, device)
        except Exception as e:                          # the reference breaks the loop on a decode failure (:343-345)
            print(f"Error in streaming_inference: {e}")
            break
        recent_video_window_clips.append(current_video_chunk)
        _sync(); section_time["VIDEO"] += time.perf_counter() - _t

        # ---- prompt + patches
        _sync(); _t = time.perf_counter()
        prompt = f"Time={start_time:.1f}-{start_time + chunk_duration:.1f}s"
        piling = i < dense - 1                # a dense-prefill turn that is not answered yet: no assistant header behind it
        if i == 0:
            user_content = [{"type": "text", "text": prompt}, {"type": "video", "video": video_path},
                            {"type": "text", "text": query}]
            full_conversation_history = [{"role": "previous text", "content": previous_text},
                                         {"role": "user", "content": user_content}]
            text = processor.apply_chat_template(full_conversation_history, tokenize=False, add_generation_prompt=not piling)
        else:
            user_content = [{"type": "text", "text": prompt},
                            {"type": "video", "video": video_path, "start": start_time, "duration": chunk_duration}]
            full_conversation_history.append({"role": "user", "content": user_content})
            text = processor.apply_chat_template([{"role": "user", "content": user_content}], tokenize=False,
                                                 add_generation_prompt=not piling)
            text = "\n" + text[SYSTEM_PROMPT_OFFSET:]
        inputs = processor(text=[text], videos=recent_video_window_clips[-1], padding=True, return_tensors="pt")
        new_ids = inputs["input_ids"].cpu()
        if dense and i < dense:
            # the turns end with "<|im_end|>\n": drop the leading "\n" of every later one (the rule of inference.py:402-405)
            dense_ids.append(new_ids if i == 0 else new_ids[:, 1:])
            dense_pix.append(inputs["pixel_values_videos"].to(device))
            dense_grids.append(inputs["video_grid_thw"])
            if piling:
                section_time["INPUT"] += time.perf_counter() - _t
                continue
            new_ids = torch.cat(dense_ids, dim=1)
            inputs["pixel_values_videos"] = torch.cat(dense_pix, dim=0)
            inputs["video_grid_thw"] = torch.cat(dense_grids, dim=0)
            dense_ids, dense_pix, dense_grids = [], [], []
            # the piled chunks' clips are only ever looked at again by the recompute path, through the last window_size of them
            del recent_video_window_clips[:max(0, len(recent_video_window_clips) - window_size)]
        if prev_generated_ids is not None:
            # the history ends with <|im_end|> (keep the new "\n") or already with "\n" (drop the duplicate)
            if int(prev_generated_ids[0, -1]) != TOKEN_IDS["\n"]:
                new_ids = torch.cat([prev_generated_ids, new_ids], dim=1)
            else:
                new_ids = torch.cat([prev_generated_ids, new_ids[:, 1:]], dim=1)
        inputs["input_ids"] = new_ids
        inputs["attention_mask"] = torch.ones_like(new_ids)
        # the look-ahead handed these patches to the ViT already (same frames, same tensor object)
        inputs["pixel_values_videos"] = ahead[1] if ahead is not None else inputs["pixel_values_videos"].to(device)
        if vision_lookahead and not recompute and i + 1 < num_chunks:
            # frames do not depend on the generated text: fetch the next chunk's now so its ViT pass can run underneath
            # this chunk's decode steps (engine.vision_prefetch)
            try:
                nf = _fetch_chunk(video, start_time + chunk_duration, chunk_duration, model, device)
                nin = processor(text=["<|vision_start|><|video_pad|><|vision_end|>"], videos=nf, padding=True, return_tensors="pt")
                lookahead = (nf, nin["pixel_values_videos"].to(device), nin["video_grid_thw"])
            except Exception:
                lookahead = None          # the regular path reports the failure when its turn comes
        recent_pixel_values_videos.append(inputs["pixel_values_videos"])
        streaming_args.input_ids = new_ids
        if streaming_args.video_grid_thw is None:
            streaming_args.video_grid_thw = inputs["video_grid_thw"]
        else:
            g = streaming_args.video_grid_thw
            if g.shape[0] >= GRID_ROWS_KEPT:
                # The reference appends one row per chunk for ever (inference.py:415) and reads this tensor in two ways only: from row 0,
                # one row per surviving vision span (qwen2/pos_emb.py:85-108), and -- when it recomputes -- the rows of the last
                # retained chunks.  Both ends are kept, the middle goes: on an unbounded stream the tensor stays small (a 4-hour
                # stream crossed torch's 32768-element threshold for multi-threaded CPU copies at chunk 10923, and on a 256-core host
                # behind a 16-CPU quota the thread pool that one torch.cat woke up throttled the whole process: 24.4 -> 46 ms per chunk).
                g = torch.cat([g[:GRID_ROWS_KEPT // 2], g[-(GRID_ROWS_KEPT // 2 - 1):]], dim=0)
            streaming_args.video_grid_thw = torch.cat([g, inputs["video_grid_thw"]], dim=0)
        current_input_len = new_ids.shape[1]
        _sync(); section_time["INPUT"] += time.perf_counter() - _t

        # ---- generate
        _sync(); _t = time.perf_counter()
        gen_kw = dict(max_new_tokens=max_new_tokens, use_cache=True, return_dict_in_generate=True, do_sample=do_sample,
                      repetition_penalty=repetition_penalty, streaming_args=streaming_args, pad_token_id=IM_END,
                      temperature=temperature, suppress_eos=suppress_eos, generator=generator, keep_logits=keep_logits)
        if lookahead is not None:
            gen_kw["next_vision"] = (lookahead[1], lookahead[2])
        if top_k is not None:                 # None: the model's generation_config / HF defaults (model.py)
            gen_kw["top_k"] = top_k
        if top_p is not None:
            gen_kw["top_p"] = top_p
        if force_tokens is not None:          # parity tests: per-chunk teacher forcing (engine.generate)
            gen_kw["force_tokens"] = force_tokens[len(responses)]          # one entry per ANSWERED chunk
        if recompute:
            if past_key_values is not None:
                past_key_values.release_reserved()
                past_key_values.truncate(0)
            n_keep = len(recent_pixel_values_videos)
            outputs = model.generate(input_ids=new_ids, attention_mask=inputs["attention_mask"],
                                     pixel_values_videos=torch.cat(recent_pixel_values_videos, dim=0),
                                     video_grid_thw=streaming_args.video_grid_thw[-n_keep:],
                                     past_key_values=past_key_values, **gen_kw)
        else:
            outputs = model.generate(input_ids=new_ids, attention_mask=inputs["attention_mask"],
                                     pixel_values_videos=inputs["pixel_values_videos"],
                                     video_grid_thw=inputs["video_grid_thw"], past_key_values=past_key_values, **gen_kw)
        _sync(); section_time["GEN"] += time.perf_counter() - _t

        # ---- post
        _sync(); _t = time.perf_counter()
        generated_ids = outputs.sequences.cpu()
        n_decoded = generated_ids.shape[1] - current_input_len
        if int(generated_ids[0, -1]) != IM_END:
            generated_ids = torch.cat([generated_ids, torch.tensor([[IM_END]])], dim=1)
        newly_generated_ids = generated_ids[:, current_input_len:]
        response = processor.batch_decode(newly_generated_ids, skip_special_tokens=True)[0]
        responses.append({"response": response[:-4], "start_time": start_time, "end_time": start_time + chunk_duration})
        if token_counts is not None:
            token_counts.append(int(n_decoded))
        if ids_log is not None:
            ids_log.append({"ids": generated_ids[0].tolist(), "new": newly_generated_ids[0].tolist(),
                            "kv_len": outputs.past_key_values.get_seq_length(), "logits": getattr(outputs, "logits", None),
                            "own": getattr(outputs, "own", None)})
        time_key = prompt
        past_key_values = outputs.past_key_values
        hms = lambda s: time.strftime("%H:%M:%S", time.gmtime(int(s)))
        printq(f"Time={hms(start_time)}-{hms(start_time + chunk_duration)}: \033[1m\033[34m{response}\033[0m",
               f"past_key_values: {past_key_values.get_seq_length() if past_key_values is not None else 0}", flush=True, quiet=quiet)
        if emit_json:
            sys.stdout.write(json.dumps({"type": "segment", "start": float(start_time), "end": float(start_time + chunk_duration),
                                         "text": response[:-4]}, ensure_ascii=False) + "\n")
            sys.stdout.flush()
        prev_generated_ids = generated_ids.clone()
        if gt_dict is not None and gt_dict[time_key]["phrase"] != response:
            # teacher forcing: drop the rows of the wrong answer, splice the ground-truth tokens (reference :483-487)
            printq(f"Decoded text [{response}] is incorrect. Use ground truth [{gt_dict[time_key]['phrase']}] instead", quiet=quiet)
            if past_key_values.get_seq_length() > current_input_len:
                past_key_values.release_reserved()
                past_key_values.prune(current_input_len, past_key_values.get_seq_length() - 1)
            response = gt_dict[time_key]["phrase"]
            gt_ids = processor(text=[response + "<|im_end|>\n"])["input_ids"]
            prev_generated_ids = torch.cat([new_ids, torch.as_tensor(gt_ids, dtype=torch.long).reshape(1, -1)], dim=1)
        full_conversation_history.append({"role": "assistant", "content": response})
        _sync(); section_time["POST"] += time.perf_counter() - _t

        _sync()
        loop_total = time.perf_counter() - loop_start
        printq(f"[Loop {i}] total={loop_total:.3f}s | " + " | ".join(f"{k}={section_time[k]:.3f}s" for k in section_time),
               flush=True, quiet=quiet)
        if time_test:
            time_results.append(section_time)
        if output_dir is not None:
            with open_vtt(output_dir) as vf:
                vf.write(f"{sec2ts(start_time)} --> {sec2ts(start_time + chunk_duration)}\n Infer Time: {loop_total:.3f}s\n {response}\n\n")
    if output_dir is not None:
        printq(f"\nSubtitles saved to: {output_dir}\n", quiet=quiet)
    if time_test:
        return time_results
    return responses


# ----------------------------------------------------------------------------- CLI (reference: inference.py:524-561)
def _cli(argv=None):
    """`python -m streaming_vlm_amd.driver`: the reference's command line with its flag names and defaults.
    `--model_path random:<2b|7b|2.5-3b|2.5-7b|tiny|tiny-2.5>[:seed]` and `--video_path synthetic://WxH@Ffps` run without
    any checkpoint or video file; build-defined switches come after the reference's."""
    import argparse
    ap = argparse.ArgumentParser(prog="streaming_vlm_amd.driver")
    ap.add_argument("--pos_mode", type=str, default="shrink", choices=["append", "shrink"])
    ap.add_argument("--all_text", action="store_true", default=False)
    ap.add_argument("--model_path", type=str, default="mit-han-lab/StreamingVLM")
    ap.add_argument("--model_base", type=str, choices=["Qwen2_5", "Qwen2"], default="Qwen2_5")
    ap.add_argument("--video_path", type=str, default="synthetic://448x448@1fps")
    ap.add_argument("--window_size", type=int, default=DEFAULT_WINDOW_SIZE)
    ap.add_argument("--chunk_duration", type=int, default=DEFAULT_CHUNK_DURATION)
    ap.add_argument("--text_round", type=int, default=DEFAULT_TEXT_ROUND)
    ap.add_argument("--previous_text", type=str, default="")
    ap.add_argument("--skip_first_chunk", type=int, default=0)
    ap.add_argument("--recompute", action="store_true")
    ap.add_argument("--temperature", type=float, default=DEFAULT_TEMPERATURE)
    ap.add_argument("--text_sink", type=int, default=DEFAULT_TEXT_SINK)
    ap.add_argument("--text_sliding_window", type=int, default=DEFAULT_TEXT_SLIDING_WINDOW)
    ap.add_argument("--output_dir", type=str)
    ap.add_argument("--emit_json", action="store_true", help="one JSON line per chunk on stdout")
    ap.add_argument("--test_data_json", type=str, default=None)
    ap.add_argument("--test_data_idx", type=int, default=None)
    ap.add_argument("--gt_json", type=str, default=None)
    ap.add_argument("--gt_idx", type=int, default=0)
    # build-defined
    ap.add_argument("--duration", type=int, default=TOTAL_VIDEO_DURATION, help="seconds of video to process")
    ap.add_argument("--kv_policy", default="structural", choices=["structural", "sink_window", "none"])
    ap.add_argument("--sink", type=int, default=4)
    ap.add_argument("--window", type=int, default=2048)
    ap.add_argument("--greedy", action="store_true", help="do_sample=False")
    ap.add_argument("--max_len", type=int, default=None, help="engine capacity in tokens (default: derived from the eviction policy)")
    ap.add_argument("--max_new_tokens", type=int, default=MAX_TOKEN_PER_DURATION)
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args(argv)
    kw = dict(vars(a))
    kw["do_sample"] = not kw.pop("greedy")
    if kw["output_dir"] is None:
        os.makedirs("output", exist_ok=True)
        kw["output_dir"] = (f"output/{a.model_path.replace('/', '_').replace(':', '_')}_viswin{a.window_size}_txtwin{a.text_round}"
                            f"_prvsink{a.text_sink}_prvwin{a.text_sliding_window}_tprt{a.temperature}.vtt")
    return streaming_inference(**kw)


if __name__ == "__main__":
    _cli()
