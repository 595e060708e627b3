"""``convert_qwen2_to_streaming`` -- same entry point as the reference's
``src/streaming_vlm/inference/qwen2/patch_model.py:18-34``.

The reference rebinds eleven HF methods so that the stock modules call flash-attn; here the model's
weights are handed to the HIP engine once and ``model.generate`` is rebound to it, keeping the call
signature the streaming loop uses (inference.py:440-451) and the fields it reads back
(``.sequences``, ``.past_key_values``; inference.py:456,466).
"""
from __future__ import annotations

from types import MethodType, SimpleNamespace
from typing import Optional

import torch

from .config import ModelConfig, from_hf_config
from .engine import SvlmEngine


class StreamingArgs:
    def __init__(self, pos_mode: str, all_text: bool = False):
        if pos_mode not in ("append", "shrink"):
            raise AssertionError("pos_mode must be in ['append', 'shrink']")
        self.pos_mode = pos_mode          # "shrink": positions stay contiguous after eviction; "append": grow forever
        self.all_text = all_text
        self.input_ids = None             # full (pruned) ids of the stream; shrink mode derives positions from them
        self.video_grid_thw = None        # cumulative, one row per chunk (never pruned, inference.py:415)
        self.second_per_grid_ts = None
        self.last_cache_position = -1


def _grid_list(g):
    if g is None:
        return []
    return [[int(v) for v in row] for row in (g.tolist() if hasattr(g, "tolist") else g)]


def streaming_generate(self, input_ids=None, attention_mask=None, pixel_values_videos=None, video_grid_thw=None,
                       past_key_values=None, max_new_tokens: int = 20, use_cache: bool = True,
                       return_dict_in_generate: bool = True, do_sample: bool = True, repetition_penalty: float = 1.0,
                       streaming_args: Optional[StreamingArgs] = None, pad_token_id=None, temperature: float = 1.0,
                       second_per_grid_ts=None, suppress_eos: bool = False, keep_logits: bool = False, generator=None, next_vision=None,
                       force_tokens=None, top_k=None, top_p=None, pixel_values=None, image_grid_thw=None, **unknown):
    """Greedy / sampling generation on the HIP engine (reference: streaming_generate + _sample,
    generate/streaming_generate_qwen.py:130-278, 8-127).  Keyword arguments this path does not implement are REFUSED, not dropped:
    image inputs (qwen2/model_forward.py:36-50 embeds `pixel_values` / `image_grid_thw`; the streaming loop only ever sends
    videos, inference.py:440-451) raise NotImplementedError, anything else a TypeError naming it."""
    if pixel_values is not None or image_grid_thw is not None:
        raise NotImplementedError("image inputs (pixel_values / image_grid_thw) are not supported by the streaming path: it takes "
                                  "video frames only (pixel_values_videos / video_grid_thw), as the reference's streaming loop sends them")
    if unknown:
        raise TypeError(f"streaming generate() got unexpected keyword argument(s): {sorted(unknown)}")
    eng: SvlmEngine = self._svlm_engine
    if streaming_args is None:
        raise ValueError("streaming_args is required (reference: every forward reads streaming_args.pos_mode)")
    if input_ids.shape[0] != 1:
        raise ValueError("the streaming loop is batch-1")
    ids = input_ids[0].tolist()
    # The reference never prunes streaming_args.video_grid_thw (one more row per chunk, inference.py:415) and indexes it
    # from row 0 in order of appearance of the surviving vision spans (qwen2/pos_emb.py:85-108).  Same rows here, but only
    # as many as there are spans: the conversion stays O(window) on an hour-long stream instead of O(chunks so far).
    g_all = streaming_args.video_grid_thw if streaming_args.video_grid_thw is not None else video_grid_thw
    if g_all is not None and len(g_all) > 64:
        g_all = g_all[:ids.count(eng.cfg.vision_start_token_id)]
    grids_all = _grid_list(g_all)
    out = eng.generate(ids, past_key_values, grids_all, pixel_values_videos, _grid_list(video_grid_thw), max_new_tokens,
                       repetition_penalty, do_sample, temperature, suppress_eos, keep_logits, generator, next_vision,
                       all_text=bool(streaming_args.all_text), pos_mode=streaming_args.pos_mode,
                       last_cache_position=streaming_args.last_cache_position, force_tokens=force_tokens, top_k=top_k, top_p=top_p)
    streaming_args.last_cache_position = eng.last_position          # qwen2/model_forward.py:117
    # the reference pads streaming_args.input_ids by one per forward (qwen2/language_forward.py:323-325)
    if streaming_args.input_ids is not None:
        streaming_args.input_ids = torch.nn.functional.pad(streaming_args.input_ids, (0, out.n_new), "constant", 0)
    seq = torch.tensor([out.sequences], dtype=torch.long, device=input_ids.device)
    if not return_dict_in_generate:
        return seq
    return SimpleNamespace(sequences=seq, past_key_values=out.past_key_values, logits=out.logits, n_new=out.n_new, own=out.own)


class StreamingQwen2VL:
    """Stand-alone model object (no transformers dependency): config + weights + ``generate``."""

    def __init__(self, cfg: ModelConfig, state_dict, device="cuda", ops=None, **engine_kw):
        self.config = cfg
        self.device = torch.device(device)
        self._svlm_engine = SvlmEngine(cfg, state_dict, device, ops=ops, **engine_kw)
        self.generate = MethodType(streaming_generate, self)

    def new_cache(self):
        return self._svlm_engine.new_cache()


def convert_qwen2_5_to_streaming(model, ops=None, **engine_kw):
    """Same entry point as the reference's qwen2_5/patch_model.py:18-38 for `Qwen2_5_VLForConditionalGeneration`
    (the family of the released StreamingVLM checkpoint): windowed RMSNorm/SwiGLU vision tower, float temporal M-RoPE."""
    model = convert_qwen2_to_streaming(model, ops=ops, **engine_kw)
    if model._svlm_engine.cfg.family != "qwen2_5":
        raise ValueError("convert_qwen2_5_to_streaming needs a Qwen2.5-VL model (use convert_qwen2_to_streaming)")
    return model


def convert_qwen2_to_streaming(model, ops=None, keep_hf_weights: bool = False, **engine_kw):
    """Accepts an HF ``Qwen2VLForConditionalGeneration`` (weights are copied into the engine and
    ``generate`` is rebound) or an already converted model (returned unchanged).
    The engine holds its own (fused) copies of the weights; unless `keep_hf_weights`, the module's parameters are then re-pointed
    at empty tensors so the originals do not stay in HBM beside them (2 x 15 GB on a 7B): tensors the engine took over without a
    copy stay alive through the engine's own reference.  The module's HF forward is not usable afterwards -- like the reference's
    conversion (11 re-bound methods, patch_model.py:18-34), the object exists to be driven through `generate(streaming_args=...)`."""
    if getattr(model, "_svlm_engine", None) is not None:
        return model
    cfg = from_hf_config(model.config)
    device = next(model.parameters()).device
    sd = {k: v for k, v in model.state_dict().items()}
    if cfg.text.tie_word_embeddings:
        sd.pop("lm_head.weight", None)
    eng = model._svlm_engine = SvlmEngine(cfg, sd, device, ops=ops, **engine_kw)
    # HF's generate() merges the checkpoint's generation_config into every call: its top_k / top_p become logits warpers under
    # do_sample=True (stock Qwen2-VL checkpoints ship top_k = 1, i.e. effectively greedy) and its eos_token_id ends a turn
    gc = getattr(model, "generation_config", None)
    if gc is not None:
        if getattr(gc, "top_k", None) is not None:
            eng.default_top_k = int(gc.top_k)
        if getattr(gc, "top_p", None) is not None:
            eng.default_top_p = float(gc.top_p)
        eos = getattr(gc, "eos_token_id", None)
        if eos is not None:
            eos = tuple(int(e) for e in (eos if isinstance(eos, (list, tuple)) else [eos]))
            cfg.eos_token_ids = eos
            eng.eos_dev = torch.tensor(list(eos), dtype=torch.int32, device=eng.device)
    model.generate = MethodType(streaming_generate, model)
    if not keep_hf_weights:
        del sd
        for prm in list(model.parameters()) + list(model.buffers()):
            prm.data = torch.empty(0, dtype=prm.dtype, device=prm.device)
    return model
