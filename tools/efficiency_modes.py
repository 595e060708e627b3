#!/usr/bin/env python
"""Counterpart of the reference's eval/efficiency/efficiency_test.py (modes a-d of the paper's efficiency figure) on the
synthetic stream: per-chunk GEN time and GEN time per decoded token as the video gets longer.

  (a) full attention           window_size = text_round = 100000 (nothing is ever evicted; KV grows ~280 tokens per chunk)
  (b) sliding window, no overlap   window_size = text_round = 100
  (c) sliding window with overlap  defaults + recompute=True (every chunk re-encodes the retained frames, no KV reuse)
  (d) StreamingVLM             defaults + text_sink/text_sliding_window 512/512

The reference harness unpacks `(time_results, token_decoded_nums)` from streaming_inference(time_test=True) although the
driver returns one list; here the token counts come through the `token_counts=` list."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import streaming_vlm_amd as S
from streaming_vlm_amd import config as C, driver as D
from streaming_vlm_amd.synthetic import ResidentProcessor, ResidentVideo
from streaming_vlm_amd.weights import random_state_dict

MODES = {
    "a": dict(window_size=100000, text_round=100000, text_sink=None, text_sliding_window=None, recompute=False),
    "b": dict(window_size=100, text_round=100, text_sink=None, text_sliding_window=None, recompute=False),
    "c": dict(window_size=D.DEFAULT_WINDOW_SIZE, text_round=D.DEFAULT_TEXT_ROUND, text_sink=None, text_sliding_window=None, recompute=True),
    "d": dict(window_size=D.DEFAULT_WINDOW_SIZE, text_round=D.DEFAULT_TEXT_ROUND, text_sink=D.DEFAULT_TEXT_SINK,
              text_sliding_window=D.DEFAULT_TEXT_SLIDING_WINDOW, recompute=False),
}
ap = argparse.ArgumentParser()
ap.add_argument("--model", default="2b", choices=["2b", "7b", "2.5-3b", "2.5-7b"])
ap.add_argument("--modes", default="abcd")
ap.add_argument("--chunks", type=int, default=64)
ap.add_argument("--size", type=int, default=448)
ap.add_argument("--out", default=None)
args = ap.parse_args()
cfg = {"2b": C.qwen2_vl_2b, "7b": C.qwen2_vl_7b, "2.5-3b": C.qwen2_5_vl_3b, "2.5-7b": C.qwen2_5_vl_7b}[args.model]()
sd = random_state_dict(cfg, 0, "cuda")
tok_chunk = (args.size // 28) ** 2 + 24 + D.MAX_TOKEN_PER_DURATION
report = {}
for mode in args.modes:
    kw = MODES[mode]
    keep = min(args.chunks, kw["window_size"])
    # recompute re-forwards every retained chunk; sizes follow from the policy
    max_len = 64 + (keep + 2) * tok_chunk + 1100
    model = S.StreamingQwen2VL(cfg, sd, "cuda", max_len=max_len, max_new_tokens=D.MAX_TOKEN_PER_DURATION)
    video = ResidentVideo(args.chunks + 1, args.size, 1.0, 0, "cuda", period=32)
    counts = []
    t0 = time.perf_counter()
    times = S.streaming_inference(model=model, processor=ResidentProcessor(), video=video, duration=args.chunks, previous_text="",
                                  model_base="Qwen2_5" if cfg.family == "qwen2_5" else "Qwen2", do_sample=False, suppress_eos=True,
                                  quiet=True, time_test=True, token_counts=counts, **kw)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    per_tok = [t["GEN"] / max(1, n) for t, n in zip(times, counts)]
    k = max(1, len(per_tok) // 8)
    report[mode] = {"config": {k_: v for k_, v in kw.items()}, "chunks": len(times), "wall_s": round(wall, 3),
                    "gen_ms_per_token_first": round(1e3 * sum(per_tok[1:1 + k]) / k, 3),
                    "gen_ms_per_token_last": round(1e3 * sum(per_tok[-k:]) / k, 3),
                    "gen_ms_per_chunk_last": round(1e3 * sum(t["GEN"] for t in times[-k:]) / k, 3),
                    "kv_len_last": model._svlm_engine._last_cache.get_seq_length()}
    print(mode, json.dumps(report[mode]), flush=True)
    del model
    torch.cuda.empty_cache()
if args.out:
    with open(args.out, "w") as f:
        json.dump({"model": cfg.name, "size": args.size, "report": report}, f, indent=1)
