#!/usr/bin/env python
"""Counterpart of the reference's eval/efficiency/efficiency_test.py (modes a-d of the paper's efficiency figure) on the
synthetic stream: per-chunk GEN time and GEN time per decoded token as the video gets longer.

  (a) full attention               window_size = text_round = 100000 (nothing is ever evicted; KV grows ~300 tokens per chunk)
  (b) sliding window, no overlap   window_size = text_round = 100 (the saw-tooth of assets/efficiency.png needs > 100 chunks)
  (c) sliding window with overlap  defaults + recompute=True (every chunk re-encodes the retained frames, no KV reuse)
  (d) StreamingVLM                 defaults + text_sink/text_sliding_window 512/512

Each mode produces the reference's JSON document -- {"meta", "per_chunk", "summary"} with the field names of
efficiency_test.py:87-136 -- written to output/efficiency/<auto name>.json like the reference does; `--out` adds one file
with all modes.  The reference harness unpacks `(time_results, token_decoded_nums)` from streaming_inference(time_test=True)
although the driver returns one list (inference.py:520-521); here the token counts come through the
`token_counts=` list."""
import argparse, json, os, sys, time
from datetime import datetime

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import streaming_vlm_amd as S  # noqa: E402
from streaming_vlm_amd import driver as D  # noqa: E402

MODES = {
    "a": dict(window_size=100000, chunk_duration=D.DEFAULT_CHUNK_DURATION, text_round=100000, text_sink=None, text_sliding_window=None, recompute=False),
    "b": dict(window_size=100, chunk_duration=D.DEFAULT_CHUNK_DURATION, text_round=100, text_sink=None, text_sliding_window=None, recompute=False),
    "c": dict(window_size=D.DEFAULT_WINDOW_SIZE, chunk_duration=D.DEFAULT_CHUNK_DURATION, text_round=D.DEFAULT_TEXT_ROUND, text_sink=None,
              text_sliding_window=None, recompute=True),
    "d": dict(window_size=D.DEFAULT_WINDOW_SIZE, chunk_duration=D.DEFAULT_CHUNK_DURATION, text_round=D.DEFAULT_TEXT_ROUND,
              text_sink=D.DEFAULT_TEXT_SINK, text_sliding_window=D.DEFAULT_TEXT_SLIDING_WINDOW, recompute=False),
}
MODE_NAMES = {"a": "baseline_a", "b": "baseline_b", "c": "baseline_c", "d": "streaming"}


def mode_max_len(mode_cfg, chunks, tokens_per_chunk, max_new=D.MAX_TOKEN_PER_DURATION):
    """Engine capacity of a mode: its policy's bound (recompute re-forwards every retained chunk from an empty cache)."""
    return D.required_max_len(tokens_per_chunk, max_new, "structural", mode_cfg["window_size"], mode_cfg["text_round"], mode_cfg["text_sink"],
                              mode_cfg["text_sliding_window"], num_chunks=chunks) + 64


def efficiency_payload(mode, model, processor, video, chunks, model_path="", model_base="Qwen2", video_path="", pos_mode="shrink",
                       all_text=False, temperature=D.DEFAULT_TEMPERATURE, previous_text="", **overrides):
    """One run of efficiency_test.py's loop body for `mode`: returns the document it saves (same keys, same arithmetic)."""
    cfg = dict(MODES[mode], **overrides)
    counts = []
    t0 = time.perf_counter()
    time_results = S.streaming_inference(model=model, processor=processor, video=video, video_path=video_path, duration=chunks,
                                         previous_text=previous_text, model_base=model_base, pos_mode=pos_mode, all_text=all_text,
                                         temperature=temperature, do_sample=False, suppress_eos=True, quiet=True, time_test=True,
                                         token_counts=counts, **cfg)
    wall = time.perf_counter() - t0
    chunk_dur = cfg["chunk_duration"]
    records = []
    for i, sec_time in enumerate(time_results):
        gen_t = float(sec_time.get("GEN", 0.0))
        dec = int(counts[i]) if i < len(counts) else 0
        records.append({"chunk_index": i, "time_start_sec": i * chunk_dur, "video_len_sec": (i + 1) * chunk_dur, "gen_time_sec": gen_t,
                        "decoded_tokens": dec, "gen_time_per_token": (gen_t / dec) if dec > 0 else None})
    meta = {"timestamp": datetime.now().strftime("%Y%m%d-%H%M%S"), "model_path": model_path, "model_base": model_base, "video_path": video_path,
            "pos_mode": pos_mode, "all_text": all_text, "skip_first_chunk": 0, "temperature": temperature, "mode": MODE_NAMES[mode],
            "window_size": cfg["window_size"], "chunk_duration": cfg["chunk_duration"], "text_round": cfg["text_round"],
            "text_sink": cfg["text_sink"], "text_sliding_window": cfg["text_sliding_window"], "recompute": cfg["recompute"],
            "duration_tested_sec": chunks * chunk_dur}
    with_tok = [r for r in records if r["gen_time_per_token"] is not None]
    return {"meta": meta, "per_chunk": records,
            "summary": {"num_chunks": len(records),
                        "avg_gen_time_sec": float(sum(r["gen_time_sec"] for r in records) / max(len(records), 1)),
                        "avg_gen_time_per_token": float(sum(r["gen_time_per_token"] for r in with_tok) / max(len(with_tok), 1))},
            # build-side additions (not in the reference's document)
            "svlm": {"wall_s": round(wall, 3), "kv_len_last": model._svlm_engine._last_cache.get_seq_length()}}


def save_payload(payload, out_dir=os.path.join("output", "efficiency")):
    """File name as efficiency_test.py:118 builds it."""
    safe = lambda s: str(s).replace("/", "_").replace("\\", "_").replace(" ", "_")
    m = payload["meta"]
    os.makedirs(out_dir, exist_ok=True)
    name = (f"{safe(m['mode'])}__{safe(m['model_base'])}__{safe(m['model_path'])}__{safe(os.environ.get('QWENVL_FPS', '2.0'))}___"
            f"{safe(os.path.basename(m['video_path']))}__s{m['skip_first_chunk']}__w{m['window_size']}__c{m['chunk_duration']}__t{m['text_round']}__"
            f"{m['timestamp']}.json")
    path = os.path.join(out_dir, name)
    with open(path, "w", encoding="utf-8") as f:
        json.dump(payload, f, ensure_ascii=False, indent=2)
    return path


def main(argv=None):
    import torch
    from streaming_vlm_amd import config as C
    from streaming_vlm_amd.synthetic import ResidentProcessor, ResidentVideo
    from streaming_vlm_amd.weights import random_state_dict
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="2b", choices=["2b", "7b", "2.5-3b", "2.5-7b", "tiny"])
    ap.add_argument("--modes", default="abcd")
    ap.add_argument("--chunks", type=int, default=64)
    ap.add_argument("--size", type=int, default=448)
    ap.add_argument("--out", default=None, help="one JSON with every mode's document (the per-mode files go to output/efficiency/)")
    args = ap.parse_args(argv)
    # the host side of the loop is single-threaded Python; torch's CPU thread pool (one thread per core) must not wake up for the id
    # tensors of mode (a) -- beyond 32768 elements every torch.cat would, and behind a CPU quota that throttles the whole process
    torch.set_num_threads(min(torch.get_num_threads(), 8))
    cfg = {"2b": C.qwen2_vl_2b, "7b": C.qwen2_vl_7b, "2.5-3b": C.qwen2_5_vl_3b, "2.5-7b": C.qwen2_5_vl_7b, "tiny": C.tiny}[args.model]()
    sd = random_state_dict(cfg, 0, "cuda")
    tok = (args.size // 28) ** 2
    report = {}
    for mode in args.modes:
        model = S.StreamingQwen2VL(cfg, sd, "cuda", max_len=mode_max_len(MODES[mode], args.chunks, tok), max_new_tokens=D.MAX_TOKEN_PER_DURATION)
        video = ResidentVideo(args.chunks + 1, args.size, 1.0, 0, "cuda", period=32)
        p = efficiency_payload(mode, model, ResidentProcessor(), video, args.chunks, model_path=f"random:{args.model}",
                               model_base="Qwen2_5" if cfg.family == "qwen2_5" else "Qwen2", video_path=f"synthetic://{args.size}x{args.size}@1fps")
        torch.cuda.synchronize()
        path = save_payload(p)
        per_tok = [r["gen_time_per_token"] for r in p["per_chunk"] if r["gen_time_per_token"] is not None]
        k = max(1, len(per_tok) // 8)
        p["svlm"].update(gen_ms_per_token_first=round(1e3 * sum(per_tok[1:1 + k]) / k, 3), gen_ms_per_token_last=round(1e3 * sum(per_tok[-k:]) / k, 3),
                         gen_ms_per_token_max=round(1e3 * max(per_tok[1:]), 3), gen_ms_per_token_min=round(1e3 * min(per_tok[1:]), 3))
        report[mode] = p
        print(mode, json.dumps({"summary": p["summary"], "svlm": p["svlm"], "file": path}), flush=True)
        del model
        torch.cuda.empty_cache()
    if args.out:
        with open(args.out, "w") as f:
            json.dump({"model": cfg.name, "size": args.size, "report": report}, f, indent=1)


if __name__ == "__main__":
    main()
