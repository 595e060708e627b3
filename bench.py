#!/usr/bin/env python
"""Headline benchmark: sustained frames/s (+ decode tokens/s) of the streaming hot path per MI355X.

One "step" = one chunk of the stream = evict (sink/window) -> ViT on the chunk's frame -> merger ->
LLM prefill of the chunk's ~275 new tokens -> 20 greedy decode tokens, through
``streaming_inference`` with its inputs already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model 2b|7b|tiny]

N > 1: one independent stream per rank, RCCL barrier only around the timed region (streams never
interact -> "weak" scaling).  Launched by the driver with torch.distributed.run (RANK / WORLD_SIZE / MASTER_* in
the environment), or -- when `--gpus N` is given WITHOUT such an environment -- by bench.py itself: it starts N
children (one per device, rendezvous on 127.0.0.1) before anything touches a GPU and forwards rank 0's line.
A WORLD_SIZE that disagrees with `--gpus` is refused.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TFLOPS = 2500.0


_T0 = time.perf_counter()


def log(msg):
    """progress on stderr (stdout carries only the one JSON line)"""
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def self_launch(n: int) -> int:
    """`--gpus N` without a launcher: N children of this same command line, RANK = LOCAL_RANK = 0..N-1, rendezvous on localhost
    (the pattern of eval/livesports3kcc/distributed_generate_streaming.py:127-143: one process per device).  Nothing here touches
    a GPU.  Rank 0's stdout is this process's stdout; the exit code is the worst child's."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for pr in procs:
        rc = max(rc, abs(pr.wait()))
    return rc


# BASELINE.json configs by number (0 is the CPU plumbing case of the parity tests, not a bench line)
CONFIGS = {1: dict(model="2b", fps=1.0, window=2048), 2: dict(model="7b", fps=2.0, window=4096), 3: dict(model="7b", fps=2.0, window=4096),
           4: dict(model="7b", fps=2.0, window=4096, scenario="dense_prefill")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="2b", choices=["2b", "7b", "tiny", "2.5-3b", "2.5-7b"])
    ap.add_argument("--size", type=int, default=448)
    ap.add_argument("--fps", type=float, default=1.0)
    ap.add_argument("--sink", type=int, default=4)
    ap.add_argument("--window", type=int, default=2048)
    ap.add_argument("--new-tokens", type=int, default=20)
    ap.add_argument("--cpu-chunks", type=int, default=3, help="chunks of the same stream timed on the host cores (rank 0, N=1)")
    ap.add_argument("--ingest", default="resident", choices=["resident", "host"],
                    help="resident: patches already in HBM when the timed region starts (the contract's `value`); host: uint8 frames in "
                         "pinned host memory, H2D + GPU patchify inside the timed region (the PCIe-inclusive rate, DESIGN.md section 6)")
    ap.add_argument("--sampling", default="greedy", choices=["greedy", "temperature", "hf-default"],
                    help="token choice: greedy (the contract's line, BASELINE configs); temperature = the reference's call (do_sample, T 0.9, "
                         "Gumbel-max in the captured graph); hf-default = the same with HF's default top_k 50 (the filter kernel)")
    ap.add_argument("--scenario", default="stream", choices=["stream", "dense_prefill"],
                    help="stream: the contract's line (BASELINE configs[1]/[2]); dense_prefill: configs[4] -- `--prefill-chunks` chunks of frames "
                         "go through ONE forward (ViT in 8-frame-grid passes, LLM prefill in 4096-row passes), then the KV cache is compacted "
                         "to sink+window and the stream continues live; reports prefill frames/s beside the steady rate")
    ap.add_argument("--prefill-chunks", type=int, default=300, help="dense_prefill: chunks piled into the opening forward (300 = 5 min)")
    ap.add_argument("--vit-fp8", action="store_true", help="vision tower Linears on the fp8 MFMA path (svlm_gemm_fp8; configs[4])")
    ap.add_argument("--decode-tail", action="store_true", help="decode step on the persistent layer-tail kernel (svlm_dec_tail) instead of the per-op launches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extra-values", action="store_true", help="skip the strict-causal / PCIe-inclusive / EOS-polling passes behind the contract's run")
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS), default=None,
                    help="BASELINE.json configs[n]: 1 = the default line (2B, 1 fps, window 2048); 2 = 7B, 2 fps, window 4096; 3 = config 2 as "
                         "N independent streams (use with --gpus 8); 4 = config 2's model opened by the 5-minute dense prefill")
    ap.add_argument("--dry-run", action="store_true", help="rendezvous + aggregation only (no model, no GPU work): what the CPU test of the "
                                                          "self-launcher runs; prints the ranks the collective saw")
    args = ap.parse_args()
    if args.config is not None:
        for k, v in CONFIGS[args.config].items():
            setattr(args, k, v)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))            # before anything touches a GPU
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={env_world}; refusing to report a line for the wrong job size")
    if args.dry_run:
        from streaming_vlm_amd import multi_stream as MS
        dist, rank, world, local_rank = MS.init_distributed("gloo" if not torch.cuda.is_available() else os.environ.get("SVLM_DIST_BACKEND", "nccl"))
        MS.fence(dist)
        agg = MS.aggregate(1.0, 1.0, 1.0, dist, "cpu" if (dist is None or dist.get_backend() == "gloo") else torch.device("cuda", local_rank))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": args.gpus, "ranks_seen": agg["world"], "per_gpu_frames_per_sec": agg["per_rank_frames_per_sec"]}), flush=True)
        return

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    import streaming_vlm_amd as S
    from streaming_vlm_amd import multi_stream as MS
    # nccl == RCCL over xGMI on ROCm.  SVLM_DIST_BACKEND=gloo is the rehearsal mode for a box with fewer GPUs than ranks
    # (ranks then share cards; RCCL needs one device per rank): same barriers, same aggregation, host-side collectives.
    backend = os.environ.get("SVLM_DIST_BACKEND", "nccl")
    dist, rank, world, local_rank = MS.init_distributed(backend)
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from streaming_vlm_amd import config as C
    from streaming_vlm_amd.synthetic import DeviceFrameProcessor, PinnedVideo, ResidentProcessor, ResidentVideo
    from streaming_vlm_amd.weights import random_state_dict

    cfg = {"2b": C.qwen2_vl_2b, "7b": C.qwen2_vl_7b, "tiny": C.tiny, "2.5-3b": C.qwen2_5_vl_3b, "2.5-7b": C.qwen2_5_vl_7b}[args.model]()
    tok_per_frame = (args.size // 28) ** 2
    chunk_tokens = tok_per_frame + 24 + args.new_tokens
    max_len = args.sink + args.window + 2 * chunk_tokens + 64
    # The timed region must see the steady state whatever --warmup says: `fill` untimed chunks bring the KV cache up to the
    # window first (a chunk adds >= tok_per_frame + 19 + new_tokens rows), then come the W warmup chunks, the K timed ones and one
    # more untimed chunk, so that each timed chunk carries exactly one look-ahead ViT pass (the first timed chunk's frames were
    # encoded under the last warmup chunk; the last timed chunk encodes the trailing chunk's).
    fill = max(0, -(-(args.sink + args.window) // (tok_per_frame + 19 + args.new_tokens)) + 1 - args.warmup)
    dense = args.prefill_chunks if args.scenario == "dense_prefill" else 0
    if dense:          # the opening forward holds every piled chunk; the window is full (and over-full) right behind it
        fill = dense
        max_len = max(max_len, 64 + dense * (tok_per_frame + 24) + 2 * chunk_tokens)
    first_timed = fill + args.warmup
    n_chunks = first_timed + args.steps + 1
    log(f"rank {rank}/{world}: building {cfg.name} random weights on {dev}")
    sd = random_state_dict(cfg, 0, dev)
    model = S.StreamingQwen2VL(cfg, sd, dev, max_len=max_len, max_new_tokens=args.new_tokens, vit_fp8=args.vit_fp8, decode_tail=args.decode_tail)
    if dense:
        model._svlm_engine.section_events = []
    log("engine ready; staging the synthetic stream in HBM")
    # inputs resident in HBM before timing; long runs cycle through 128 distinct chunks
    period = 128 if n_chunks > 256 else 0
    if args.ingest == "host":
        video = PinnedVideo(n_chunks + 1, args.size, args.fps, rank, period=period)
        proc = DeviceFrameProcessor(model._svlm_engine.ops, dev)
    else:
        video = ResidentVideo(n_chunks + 1, args.size, args.fps, rank, dev, period=period)
        proc = ResidentProcessor()
    frames_per_chunk = video.frames_per_chunk

    t = {}
    counts = []
    kvlog = []          # KV rows after each answered chunk, read from the cache object at the next chunk's start.  (NOT the driver's
                        # `ids_log`: that keeps every chunk's full id list -- 31 M Python ints over a 4-hour stream, whose garbage-collector
                        # passes alone took the chunk from 24.5 to 35.5 ms by chunk 14400)
    kv_steady = [args.sink + args.window]

    def fence():
        MS.fence(dist, dev)

    stamps = []

    # the host side of the loop is a few hundred microseconds of single-threaded Python per chunk; torch's CPU thread pool (one thread per
    # core: 256 on these boxes, behind a 16-CPU quota) has nothing to do here and must not wake up for a stray CPU tensor op
    torch.set_num_threads(min(torch.get_num_threads(), 8))
    probe = int(os.environ.get("SVLM_BENCH_PROBE", "0"))          # diagnostics: every N chunks, host-side growth indicators on stderr
    if os.environ.get("SVLM_BENCH_NOGC") == "1":
        import gc
        gc.disable()

    prof_at = [int(v) for v in os.environ.get("SVLM_BENCH_CPROFILE", "").split(",") if v]          # diagnostics: cProfile of 300 chunks from each of these
    prof = {}

    def on_chunk(i):
        stamps.append(time.perf_counter())
        if i in prof_at:
            import cProfile
            prof["p"] = cProfile.Profile()
            prof["p"].enable()
        if (i - 300) in prof_at and "p" in prof:
            import pstats, io
            prof["p"].disable()
            buf = io.StringIO()
            pstats.Stats(prof.pop("p"), stream=buf).sort_stats("tottime").print_stats(18)
            log(f"cProfile of chunks {i - 300}..{i}:\n" + buf.getvalue())
        if probe and i and i % probe == 0:
            import gc
            c0 = time.perf_counter()
            sum(range(300000))                     # fixed host-only work: does the HOST get slower, whatever the stream does?
            cal = 1e3 * (time.perf_counter() - c0)
            log(f"chunk {i}: {1e3 * (stamps[-1] - stamps[-1 - probe]) / probe:.3f} ms/chunk over the last {probe}; host calibration loop {cal:.2f} ms; "
                f"cuda allocated {torch.cuda.memory_allocated() >> 20} MiB reserved {torch.cuda.memory_reserved() >> 20} MiB; "
                f"python objects {len(gc.get_objects())}; gc counts {gc.get_count()}")
        pool = getattr(model._svlm_engine, "_last_cache", None)
        if pool is not None:
            kvlog.append({"kv_len": int(pool.length)})
        if i == 0:
            log(f"stream started ({fill} chunks to fill the KV window + {args.warmup} warmup chunks)")
            if dense:
                fence()
                t["dense0"] = time.perf_counter()
        if dense and i == dense:
            fence()
            t["dense1"] = time.perf_counter()
            t["dense_kv"] = kvlog[-1]["kv_len"]
            log(f"dense prefill of {dense} chunks done in {t['dense1'] - t['dense0']:.2f} s (KV {t['dense_kv']} rows)")
        if dense and i == dense + 1:
            fence()
            t["compact1"] = time.perf_counter()
        if i == first_timed:
            fence()
            kv_now = kvlog[-1]["kv_len"] if kvlog else 0
            if kv_now < args.sink + args.window - chunk_tokens:
                raise SystemExit(f"KV cache holds {kv_now} rows at t0: the window (sink {args.sink} + {args.window}) is not full")
            t["kv_at_t0"] = kv_now
            log(f"timed region starts (KV {kv_now} rows)")
            t["t0"] = time.perf_counter()
        if i == first_timed + args.steps:
            fence()
            t["t1"] = time.perf_counter()

    S.streaming_inference(model=model, processor=proc, video=video, model_base="Qwen2_5" if cfg.family == "qwen2_5" else "Qwen2",
                          duration=n_chunks, previous_text="",
                          kv_policy="sink_window", sink=args.sink, window=args.window, do_sample=args.sampling != "greedy",
                          temperature=0.9, top_k={"greedy": None, "temperature": 0, "hf-default": 50}[args.sampling], top_p=1.0,
                          max_new_tokens=args.new_tokens, suppress_eos=True, quiet=True, token_counts=counts, chunk_callback=on_chunk,
                          dense_prefill_chunks=dense)
    fence()
    kvlog.append({"kv_len": int(model._svlm_engine._last_cache.length)})
    kv_steady[0] = kvlog[-1]["kv_len"]
    kv_max = max(e["kv_len"] for e in kvlog)
    pool = getattr(model._svlm_engine, "_last_cache", None)
    cache_stats = dict(kv_len_max=kv_max, kv_len_at_t0=t["kv_at_t0"], **(pool.stats if pool is not None else {}))
    elapsed = t["t1"] - t["t0"]
    log(f"timed region done: {elapsed:.3f} s for {args.steps} chunks")
    frames = args.steps * frames_per_chunk
    if dense:          # piled chunks produce no ids_log / token_counts entries: the first answered chunk is the last piled one
        counts = [0] * (dense - 1) + counts
    tokens = sum(counts[first_timed:first_timed + args.steps])
    agg = MS.aggregate(frames, tokens, elapsed, dist, dev if backend == "nccl" else "cpu")
    t_max, fps_total, tps_total, per_gpu_fps = agg["t_max"], agg["frames_per_sec"], agg["tokens_per_sec"], agg["per_rank_frames_per_sec"]

    out = {
        "metric": "frames_per_sec", "value": round(fps_total, 3), "unit": "frames/s",
        "decode_tokens_per_sec": round(tps_total, 2), "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * t_max / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic" if args.ingest == "resident" else "synthetic uint8 frames from pinned host memory (PCIe-inclusive)",
        "config": {"workload": f"{cfg.name} bf16, {args.size}x{args.size} @{args.fps:g}fps synthetic stream, KV sink={args.sink} "
                               f"window={args.window}, {args.new_tokens} {'greedy' if args.sampling == 'greedy' else 'sampled (' + args.sampling + ')'} tokens/chunk, one stream per GPU",
                   "frames_per_chunk": frames_per_chunk, "new_tokens_per_chunk": args.new_tokens, "kv_len_steady": kv_steady[0],
                   "kv_fill_chunks": fill, "vit_passes_in_timed_region": args.steps, "parallelism": f"streams{world}",
                   "decode_step": "persistent layer tail (svlm_dec_tail), 3 launches per layer" if args.decode_tail else "per-op launches, 6 per layer",
                   "kv_layout": "slot-mapped pool of un-rotated keys + linear planes of the chunk's rotated keys" if model._svlm_engine.linear_planes else "slot-mapped pool of un-rotated keys, rotated on load"},
        "per_gpu_frames_per_sec": [round(v, 3) for v in per_gpu_fps],
        "per_gpu_min": round(min(per_gpu_fps), 3), "per_gpu_max": round(max(per_gpu_fps), 3),
        "per_gpu_stdev": round((sum((v - sum(per_gpu_fps) / len(per_gpu_fps)) ** 2 for v in per_gpu_fps) / len(per_gpu_fps)) ** 0.5, 4),
        "ranks_seen": agg["world"],          # ranks the collective (RCCL) gathered from: must equal n_gpus
        # wall time of a chunk per generated token, the figure eval/efficiency/efficiency_test.py:87-99 reports
        "chunk_ms_per_token": round(1e3 * t_max / max(1, tokens), 4),
        "kv_pool": {k: int(v) for k, v in cache_stats.items()},
    }
    if dense:
        ev = {}
        for name, a, b in model._svlm_engine.section_events[:2]:          # the opening forward's ViT and LLM-prefill phases
            ev[name] = a.elapsed_time(b) / 1e3
        fr = dense * frames_per_chunk
        out["scenario"] = "dense_prefill"
        out["config"]["workload"] += f"; opened by a dense prefill of {fr} frames ({dense} chunks in one forward)" + (", fp8 ViT" if args.vit_fp8 else "")
        out["dense_prefill"] = {"frames": fr, "chunks": dense, "seconds": round(t["dense1"] - t["dense0"], 3),
                                "frames_per_sec": round(fr / (t["dense1"] - t["dense0"]), 2), "vit_seconds": round(ev.get("vit", 0.0), 3),
                                "vit_frames_per_sec": round(fr / ev["vit"], 1) if ev.get("vit") else None,
                                "llm_prefill_seconds": round(ev.get("prefill", 0.0), 3), "prompt_rows": t["dense_kv"] - args.new_tokens + 1,
                                "kv_rows_after": t["dense_kv"],
                                # first live chunk: sink/window eviction compacts the cache (slot-table edit + page recycling), then a normal chunk
                                "first_live_chunk_ms": round(1e3 * (t["compact1"] - t["dense1"]), 2), "kv_rows_live": kv_steady[0],
                                "vit_fp8": bool(args.vit_fp8)}
    if args.steps >= 400:      # drift over a long stream: mean chunk time of the first / last 100 timed chunks
        d = [1e3 * (b - a) for a, b in zip(stamps[first_timed:first_timed + args.steps], stamps[first_timed + 1:first_timed + args.steps + 1])]
        out["ms_per_step_first100"], out["ms_per_step_last100"] = round(sum(d[:100]) / 100, 3), round(sum(d[-100:]) / 100, 3)

    # ---- the same stream under the three conditions the contract's `value` excludes (N = 1 only; each its own timed pass of K chunks
    # on the same engine): strict-causal (no look-ahead ViT: chunk i+1's frames are not touched before chunk i is answered),
    # PCIe-inclusive (uint8 frames from pinned host memory, H2D + patchify inside the timed region), and with EOS live (tokens polled
    # every 4th step, the path every real captioning run takes; random weights almost never emit EOS, so the token count is the same)
    if world == 1 and not dense and not args.no_extra_values:
        def extra_pass(lookahead=True, ingest=args.ingest, suppress_eos=True):
            n2 = fill + 3 + args.steps + 1
            first2 = fill + 3
            if ingest == "host":
                v2, p2 = PinnedVideo(n2 + 1, args.size, args.fps, rank, period=128 if n2 > 256 else 0), DeviceFrameProcessor(model._svlm_engine.ops, dev)
            else:
                v2, p2 = ResidentVideo(n2 + 1, args.size, args.fps, rank, dev, period=128 if n2 > 256 else 0), ResidentProcessor()
            tt, cnt = {}, []

            def cb(i):
                if i == first2 or i == first2 + args.steps:
                    torch.cuda.synchronize()
                    tt[i] = time.perf_counter()
            S.streaming_inference(model=model, processor=p2, video=v2, model_base="Qwen2_5" if cfg.family == "qwen2_5" else "Qwen2",
                                  duration=n2, previous_text="", kv_policy="sink_window", sink=args.sink, window=args.window,
                                  do_sample=args.sampling != "greedy", temperature=0.9,
                                  top_k={"greedy": None, "temperature": 0, "hf-default": 50}[args.sampling], top_p=1.0,
                                  max_new_tokens=args.new_tokens, suppress_eos=suppress_eos, quiet=True, token_counts=cnt, chunk_callback=cb,
                                  vision_lookahead=lookahead)
            torch.cuda.synchronize()
            dt = tt[first2 + args.steps] - tt[first2]
            return round(args.steps * v2.frames_per_chunk / dt, 3), round(sum(cnt[first2:first2 + args.steps]) / dt, 2)
        log("extra passes: strict-causal, PCIe-inclusive, EOS polling")
        out["value_no_lookahead"], out["decode_tokens_per_sec_no_lookahead"] = extra_pass(lookahead=False)
        if args.ingest != "host":
            out["value_ingest_host"], _ = extra_pass(ingest="host")
        out["value_eos_polling"], out["decode_tokens_per_sec_eos_polling"] = extra_pass(suppress_eos=False)
    if rank == 0 and not args.no_roofline:
        log("roofline pass (eager launches bracketed by HIP events)")
        out.update(roofline_pass(model, args, kv_steady[0]))
    if rank == 0 and dense:
        out.update(dense_prefill_rooflines(model, out["dense_prefill"]["prompt_rows"]))
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not dense:
        log("cpu baseline (oracle on host cores)")
        out["cpu_baseline"] = cpu_baseline(cfg, sd, args)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def dense_prefill_rooflines(model, prompt_rows):
    """configs[4]: the two MFMA kernels that carry the opening forward, timed live at its LAST pass (PREFILL_ROWS query rows over
    the whole prompt): causal attention of one layer and the four projection GEMMs of one layer, HIP events on the launching stream."""
    import math
    eng = model._svlm_engine
    o, w, tc = eng.ops, eng.w, eng.cfg.text
    dev = eng.device
    H, I, D, Hq, Hkv, qd, kd = tc.hidden_size, tc.intermediate_size, tc.head_dim, tc.num_heads, tc.num_kv_heads, eng.qd, eng.kd
    T = min(eng.PREFILL_ROWS, prompt_rows)
    L = min(prompt_rows, eng.max_len)
    c = eng.new_cache()
    c.reserve(L)
    c.commit(L)
    c.sync_device()
    c.pool[0].normal_()                  # random K / V rows: zero-filled operands read high (lower toggle rate, higher clock)
    bf = torch.bfloat16
    q = torch.randn((T, qd), device=dev).to(bf)
    att = torch.empty((T, qd), dtype=bf, device=dev)
    x = torch.randn((T, H), device=dev).to(bf)
    qkv = torch.empty((T, qd + 2 * kd), dtype=bf, device=dev)
    hm = torch.empty((T, I), dtype=bf, device=dev)
    l0 = w.layers[0]

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        best = None
        for _ in range(reps):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            fn()
            e.record()
            torch.cuda.synchronize()
            best = s.elapsed_time(e) if best is None else min(best, s.elapsed_time(e))
        return best
    ms_a = timed(lambda: o.prefill_attn(q, c.pool, 0, c.slot_of_dev, eng.rope_cs, att, T, L, Hq, 1.0 / math.sqrt(D)))
    fl_a = 4.0 * T * (L - T / 2.0) * Hq * D

    def gemms():
        o.gemm(x, l0["qkv_w"], bias=l0["qkv_b"], out=qkv)
        o.gemm(att, l0["o_w"], residual=x, out=x)
        o.gemm(x, l0["gu_w"], out=hm, act=eng_act_swiglu)
        o.gemm(hm, l0["down_w"], residual=x, out=x)
    from streaming_vlm_amd._lib import ACT_SWIGLU as eng_act_swiglu
    ms_g = timed(gemms)
    fl_g = 2.0 * T * H * ((qd + 2 * kd) + qd + 2 * I + I)
    mk = lambda kern, ms, fl, extra: dict({"kernel": kern, "bound": "mfma", "achieved": round(fl / ms / 1e9, 1), "peak": 2500.0, "unit": "TFLOP/s",
                                           "frac": round(fl / ms / 1e9 / 2500.0, 4), "avg_launch_us": round(ms * 1e3, 1), "traffic": None}, **extra)
    del c
    return {"roofline_prefill_attn": mk("rope_gather_kernel+prefill_attn_dma_kernel", ms_a, fl_a, {"query_rows": T, "keys": L, "q_heads": Hq, "kv_heads": Hkv}),
            "roofline_prefill_gemm": mk("gemm_glds_kernel (qkv, o_proj, gate/up + SwiGLU, down_proj of one layer)", ms_g, fl_g, {"rows": T})}


def roofline_pass(model, args, kv_len):
    """Per-kernel durations measured live with HIP events on the launching stream, on the model's real operands.

    Each kernel symbol is launched back-to-back over ALL layers (every launch streams different weights / a
    different layer's KV, as in the real step) from a captured HIP graph between ONE event pair, after a 512 MiB
    write that evicts L2 and the 256 MiB Infinity Cache; avg launch = elapsed / launches (includes the ~1 us
    kernel-to-kernel boundary, like the real decode graph).  (Bracketing every single launch with its own
    event pair adds ~7 us of event latency per kernel -- more than most of these kernels take.)
    Algorithmic bytes / flops per launch are the formulas of DESIGN.md 'Kernels and rooflines'."""
    import math
    eng = model._svlm_engine
    o, w, cfg = eng.ops, eng.w, eng.cfg
    tc, vc = cfg.text, cfg.vision
    dev = eng.device
    H, I, V, D = tc.hidden_size, tc.intermediate_size, tc.vocab_size, tc.head_dim
    Hq, Hkv, qd, kd, NL = tc.num_heads, tc.num_kv_heads, eng.qd, eng.kd, tc.num_layers
    L = int(kv_len)
    T = (args.size // 28) ** 2 + 19                      # prefill rows of a steady-state chunk
    n_dec = args.new_tokens - 1
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    c = eng.new_cache()
    c.reserve(L + 1)
    c.commit(L)
    c.sync_device()
    c.pool.normal_()                     # random rows: zero-filled operands read high (lower toggle rate -> higher clock)
    lin = c.lin_args()
    if lin is not None:                  # the decode step in the MIDDLE of a chunk: the prefill's rotated copy covers all but the rows decoded since
        c.lin.normal_()
        c.lin_len_dev.copy_(torch.tensor([max(L - (args.new_tokens - 1) // 2, 0), 1], dtype=torch.int32))
    pos = torch.arange(L + 1, dtype=torch.int32, device=dev).repeat(3, 1)
    eng.pos3_dev[:, :L + 1].copy_(pos)
    o.mrope_table(eng.pos3_dev, eng.inv_freq, eng.rope_cs, 0, L + 1, tc.mrope_section)
    eng.state.copy_(torch.tensor([L, 0], dtype=torch.int32))
    kv_dev = eng.state[0:1]
    scale = 1.0 / math.sqrt(D)
    results = []

    def timed(name, launches, per_chunk, nbytes, flops, fn, reps=5):
        fn()                                     # warm: lazy workspaces are allocated outside the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()               # replay from a HIP graph: Python cannot launch 5-us kernels back-to-back
        with torch.cuda.graph(g):
            fn()
        tms = []
        for _ in range(reps):
            flush.fill_(1)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            g.replay()
            e.record()
            torch.cuda.synchronize()
            tms.append(s.elapsed_time(e))
        us = 1e3 * (sum(tms) / len(tms)) / launches           # MEAN of the cold replays (the fastest one is kept beside it)
        r = {"kernel": name, "avg_us": round(us, 2), "min_us": round(1e3 * min(tms) / launches, 2), "launches_per_chunk": per_chunk,
             "ms_per_chunk": round(us * per_chunk / 1e3, 3)}
        if nbytes:
            r["bytes_per_launch"] = int(nbytes)
            r["GBps"] = round(nbytes / us / 1e3, 1)
        if flops:
            r["TFLOPs"] = round(flops / us / 1e6, 2)
        results.append(r)

    lw = w.layers
    # ---- decode-step kernels (T = 1)
    timed("dec_qkv_kernel", NL, NL * n_dec, 2 * ((qd + 2 * kd) * H + 2 * H + 2 * (qd + 2 * kd)), 2.0 * (qd + 2 * kd) * H,
          lambda: [o.dec_qkv(eng.d_x, l["ln1"], tc.rms_eps, l["qkv_w"], l["qkv_b"], eng.d_qkv, c.pool, i, c.slot_of_dev, qd, kd, len_dev=kv_dev)
                   for i, l in enumerate(lw)])
    attn_bytes = 2 * (L + 1) * Hkv * D * 2 + (L + 1) * 3 * 4 + 2 * Hq * D * 2 + 2 * Hkv * D * 2
    timed("decode_attn_split_kernel+decode_attn_combine_kernel", NL, NL * n_dec, attn_bytes, 4.0 * (L + 1) * Hq * D,
          lambda: [o.decode_attn(eng.d_qkv[:qd], c.pool, i, c.slot_of_dev, eng.rope_cs, eng.d_attn, eng.d_ws, Hq, eng._attn_len,
                                 eng.decode_chunk, scale, length=1, len_dev=kv_dev, lin=lin) for i in range(NL)])
    timed("gemv_bf16_kernel(o_proj)", NL, NL * n_dec, 2 * (H * qd + qd + 2 * H), 2.0 * H * qd,
          lambda: [o.gemv(eng.d_attn, l["o_w"], residual=eng.d_x, out=eng.d_x) for l in lw])
    timed("dec_gate_up_kernel", NL, NL * n_dec, 2 * (2 * I * H + 2 * H + I), 4.0 * I * H,
          lambda: [o.dec_gate_up(eng.d_x, l["ln2"], tc.rms_eps, l["gu_w"], eng.d_h) for l in lw])
    timed("gemv_bf16_ksplit_kernel(down_proj)", NL, NL * n_dec, 2 * (H * I + I + 2 * H), 2.0 * H * I,
          lambda: [o.gemv(eng.d_h, l["down_w"], residual=eng.d_x, out=eng.d_x) for l in lw])
    timed("dec_lm_head_kernel", 1, n_dec, 2 * (V * H + 2 * H) + 5 * V, 2.0 * V * H,
          lambda: o.dec_lm_head(eng.d_x, w.final_norm, tc.rms_eps, w.lm_head, eng.logits, eng.seen, 1.05, None, eng.d_sws))
    # ---- prefill (T rows) and ViT (N patches) MFMA kernels
    x = torch.randn((T, H), device=dev).to(torch.bfloat16)
    qkv = torch.empty((T, qd + 2 * kd), dtype=torch.bfloat16, device=dev)
    att = torch.randn((T, qd), device=dev).to(torch.bfloat16)
    gu = torch.empty((T, 2 * I), dtype=torch.bfloat16, device=dev)
    hm = torch.randn((T, I), device=dev).to(torch.bfloat16)
    gflops = lambda M, N, K: 2.0 * M * N * K
    gbytes = lambda M, N, K: 2 * (M * K + N * K + M * N)

    def llm_gemms():
        for l in lw:
            o.gemm(x, l["qkv_w"], bias=l["qkv_b"], out=qkv)
            o.gemm(att, l["o_w"], residual=x, out=x)
            o.gemm(x, l["gu_w"], out=gu)
            o.gemm(hm, l["down_w"], residual=x, out=x)
    fl = gflops(T, qd + 2 * kd, H) + gflops(T, H, qd) + gflops(T, 2 * I, H) + gflops(T, H, I)
    by = gbytes(T, qd + 2 * kd, H) + gbytes(T, H, qd) + gbytes(T, 2 * I, H) + gbytes(T, H, I)
    timed("gemm_bf16_kernel(prefill: qkv,o,gate_up,down)", 4 * NL, 4 * NL, by / 4, fl / 4, llm_gemms)
    c2 = eng.new_cache()
    Lp = min(L + T, eng.max_len)
    c2.reserve(Lp)
    c2.commit(Lp)
    c2.sync_device()
    c2.pool.normal_()
    timed("rope_gather_kernel+prefill_attn_dma_kernel+flash_combine_kernel(prefill)", NL, NL, 2 * Lp * Hkv * D * 2 * 2 + 2 * T * Hq * D * 2,
          4.0 * T * (Lp - T / 2.0) * Hq * D,
          lambda: [o.prefill_attn(qkv[:, :qd], c2.pool, i, c2.slot_of_dev, eng.rope_cs, att, T, Lp, Hq, scale) for i in range(NL)])
    N = (args.size // 14) ** 2
    E, Hh, dv, F = vc.embed_dim, vc.num_heads, vc.head_dim, vc.mlp_hidden
    xv = torch.randn((N, E), device=dev).to(torch.bfloat16)
    qv = torch.randn((N, 3 * E), device=dev).to(torch.bfloat16)
    av = torch.randn((N, E), device=dev).to(torch.bfloat16)
    if vc.arch == "qwen2_5":           # RMSNorm / SwiGLU tower: gate|up fused (padded intermediate), windowed attention
        Fp = vc.mlp_padded
        guv = torch.randn((N, 2 * Fp), device=dev).to(torch.bfloat16)
        hv = torch.randn((N, Fp), device=dev).to(torch.bfloat16)

        def vit_gemms():
            for b in w.vit:
                o.gemm(xv, b["qkv_w"], bias=b["qkv_b"], out=qv)
                o.gemm(av, b["proj_w"], bias=b["proj_b"], residual=xv, out=xv)
                o.gemm(xv, b["gu_w"], bias=b["gu_b"], out=guv)
                o.gemm(hv, b["down_w"], bias=b["down_b"], residual=xv, out=xv)
        fl = gflops(N, 3 * E, E) + gflops(N, E, E) + 3 * gflops(N, Fp, E)
        by = gbytes(N, 3 * E, E) + gbytes(N, E, E) + 3 * gbytes(N, Fp, E)
        timed("gemm_bf16_kernel(vit2.5: qkv,proj,gate_up,down)", 4 * vc.depth, 4 * vc.depth, by / 4, fl / 4, vit_gemms)
        plan = eng._vit_windows([[1, args.size // 14, args.size // 14]])
        n_full = len(vc.fullatt_block_indexes)

        def vit_attn():
            for bi in range(vc.depth):
                for row, n, ln in (plan["full_runs"] if bi in vc.fullatt_block_indexes else plan["win_runs"]):
                    o.vit_attn(qv[row:row + n * ln], n, ln, Hh, dv, 1.0 / math.sqrt(dv), out=av[row:row + n * ln])
        wl = plan["win_runs"][0][2]
        timed(f"flash_attn_kernel<{dv},{96 if dv == 80 else dv}>(vit2.5: {vc.depth - n_full} windowed + {n_full} full blocks)", vc.depth, vc.depth,
              2 * N * 4 * Hh * dv, 4.0 * N * Hh * dv * (n_full * N + (vc.depth - n_full) * wl) / vc.depth, vit_attn)
    else:
        fv = torch.randn((N, F), device=dev).to(torch.bfloat16)

        def vit_gemms():
            for b in w.vit:
                o.gemm(xv, b["qkv_w"], bias=b["qkv_b"], out=qv)
                o.gemm(av, b["proj_w"], bias=b["proj_b"], residual=xv, out=xv)
                o.gemm(xv, b["fc1_w"], bias=b["fc1_b"], out=fv, act=1)
                o.gemm(fv, b["fc2_w"], bias=b["fc2_b"], residual=xv, out=xv)
        fl = gflops(N, 3 * E, E) + gflops(N, E, E) + 2 * gflops(N, F, E)
        by = gbytes(N, 3 * E, E) + gbytes(N, E, E) + 2 * gbytes(N, F, E)
        timed("gemm_bf16_kernel(vit: qkv,proj,fc1,fc2)", 4 * vc.depth, 4 * vc.depth, by / 4, fl / 4, vit_gemms)
        timed(f"flash_attn_kernel<{dv},{96 if dv == 80 else dv}>(vit)", vc.depth, vc.depth, 2 * N * 4 * Hh * dv, 4.0 * N * N * Hh * dv,
              lambda: [o.vit_attn(qv, 1, N, Hh, dv, 1.0 / math.sqrt(dv), out=av) for _ in range(vc.depth)])

    results.sort(key=lambda r: -r["ms_per_chunk"])
    # dominant = the single kernel SYMBOL with the most time per chunk (pairs timed together are listed, not ranked)
    dom = next(r for r in results if "+" not in r["kernel"])
    if "flash" in dom["kernel"] or dom["kernel"].startswith("gemm"):
        roof = {"kernel": dom["kernel"], "bound": "mfma", "achieved": dom.get("TFLOPs"), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(dom.get("TFLOPs", 0.0) / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None}
    else:
        roof = {"kernel": dom["kernel"], "bound": "hbm", "achieved": dom.get("GBps"), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(dom.get("GBps", 0.0) / HBM_PEAK_GBS, 4), "traffic": None}
    roof.update(avg_launch_us=dom["avg_us"], algorithmic_bytes_per_launch=dom.get("bytes_per_launch"), launches_per_chunk=dom["launches_per_chunk"])
    # HBM traffic per launch from the committed rocprofv3 --pmc summary of the same kernel ON THE SAME MODEL's shapes
    # (profiles/pmc_traffic.json, keyed by model name, then kernel symbol); null when no such pass has been recorded
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f)
        roof["traffic"] = pmc.get(cfg.name, {}).get(dom["kernel"].split("(")[0], {}).get("hbm_bytes_per_launch")
    except Exception:
        pass
    da = next(r for r in results if r["kernel"].startswith("decode_attn"))
    extra = {"roofline": roof, "kernels": results, "kv_len_timed": L,
             "roofline_decode_attn": {"bound": "hbm", "achieved": da["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": round(da["GBps"] / HBM_PEAK_GBS, 4), "avg_launch_us": da["avg_us"],
                                      "algorithmic_bytes_per_launch": da["bytes_per_launch"], "kv_len": L + 1}}
    # long-window points (mode (a) "full attention" regime), where the kernel is bandwidth-bound: this model's head geometry and the
    # 7B's (28 query / 4 kv heads) at 32k and 131k keys; split + combine, 8 or more different cold pools (>= 640 MB) replayed from one graph
    def long_point(hq, hkv, Lbig):
        cap = Lbig + 64
        slot = torch.arange(cap, dtype=torch.int32, device=dev)
        rope = torch.randn((cap, D), device=dev).to(torch.bfloat16)
        ch = type(eng).pick_decode_chunk(cap, hkv, eng.linear_planes)
        ws = o.decode_attn_ws(hq, cap, ch, dev)
        out = torch.empty(hq * D, dtype=torch.bfloat16, device=dev)
        qq = torch.randn(hq * D, device=dev).to(torch.bfloat16)
        n_pools = max(8, -(-(640 << 20) // (2 * hkv * cap * D * 2)))     # >= 640 MB of distinct K/V per replay: 2.5x the Infinity Cache
        pools = [(torch.randn((1, 2, hkv, cap, D), device=dev) * 0.5).to(torch.bfloat16) for _ in range(n_pools)]
        # as the engine runs it: the cache's linear planes hold the prefill's rotated keys / values of all rows but the ones decoded since
        lin_rows = -(-cap // 16) * 16
        lins = [((torch.randn((1, 2, hkv, lin_rows, D), device=dev) * 0.5).to(torch.bfloat16),
                 torch.tensor([Lbig - 16, 1], dtype=torch.int32, device=dev)) if eng.linear_planes else None for _ in range(n_pools)]
        fn = lambda: [o.decode_attn(qq, p, 0, slot, rope, out, ws, hq, cap, ch, scale, length=Lbig, lin=ln) for p, ln in zip(pools, lins)]
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        ts = []
        for _ in range(5):
            flush.fill_(1)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3):                   # every replay reads >= 640 MB of distinct K/V (2.5x the 256 MB Infinity Cache), so each one is cold;
                g.replay()                       # three per event pair keep the ~25 us between the events and the graph's first kernel out of a 200 us measurement
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / (3 * len(pools)))
        best = sum(ts) / len(ts)                 # mean over the cold replays
        nb = 2 * Lbig * hkv * D * 2 + Lbig * 3 * 4
        return {"kv_len": Lbig, "q_heads": hq, "kv_heads": hkv, "keys_per_workgroup": ch, "avg_launch_us": round(best * 1e3, 2),
                "achieved": round(nb / best / 1e6, 1), "unit": "GB/s", "peak": HBM_PEAK_GBS, "frac": round(nb / best / 1e6 / HBM_PEAK_GBS, 4),
                "algorithmic_bytes_per_launch": nb, "linear_planes": bool(eng.linear_planes)}
    try:
        extra["roofline_decode_attn_32k"] = long_point(Hq, Hkv, 32768)
        pts = [long_point(Hq, Hkv, 131072)]
        if (Hq, Hkv) != (28, 4):
            pts += [long_point(28, 4, 32768), long_point(28, 4, 131072)]
        extra["roofline_decode_attn_long"] = pts
    except Exception as ex:          # measurement extra only
        extra["roofline_decode_attn_32k"] = {"error": str(ex)}
    return extra


def cpu_baseline(cfg, sd, args):
    """The reference has no runnable CPU path: time the CPU oracle (eager-PyTorch restatement of the
    reference algorithm: torch.cat KV, index_select eviction, RoPE of all cached keys every step) on this
    box's host cores, on a bounded sample of the SAME workload IN ITS STEADY STATE: the stream opens with a
    previous-text block long enough that the KV window is full after the (untimed) first chunk, and the `--cpu-chunks`
    chunks behind it are timed, each evicting, encoding one frame, prefilling and decoding against a full window."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(avail, int(os.environ.get("SVLM_CPU_THREADS", "16"))))     # a 1-GPU box owns a 16-core share
    torch.set_num_threads(cores)
    sd_cpu = {k: v.cpu() for k, v in sd.items()}
    n = max(1, args.cpu_chunks)
    tok_per_frame = (args.size // 28) ** 2
    words = max(0, args.sink + args.window - tok_per_frame - 40)                  # one token per word in the stand-in tokenizer
    alphabet = "abcdefghijklmnopqrstuvwxyz"
    prev = " ".join(alphabet[i % 26] + alphabet[(i // 26) % 26] + alphabet[(i // 676) % 26] for i in range(words))
    stamps = []
    import streaming_vlm_amd as S
    from oracle import generate as og
    proc, video = S.SyntheticProcessor(), S.SyntheticVideo(args.size, args.fps, 0)
    src = H.chunk_source(proc, video, prev)

    def timed_src(i):
        stamps.append(time.perf_counter())
        return src(i)
    scfg = og.StreamCfg(policy="sink_window", sink=args.sink, window=args.window, max_new_tokens=args.new_tokens, suppress_eos=True)
    out = og.streaming_loop(sd_cpu, H.oracle_cfg(cfg), scfg, n + 1, timed_src)
    dt = time.perf_counter() - stamps[1]
    fpc = max(1, int(round(args.fps)))
    return {"value": round(n * fpc / dt, 4), "unit": "frames/s", "cores": torch.get_num_threads(),
            "kind": "port", "decode_tokens_per_sec": round(n * args.new_tokens / dt, 3),
            "sample": f"{n} steady-state chunks of the same workload (KV {out['kv_len'][0]} rows after an untimed first chunk that opens with a "
                      f"{words}-token previous-text block; KV {min(out['kv_len'][1:])}-{max(out['kv_len'][1:])} rows in the timed chunks), bf16 weights, "
                      f"torch {torch.__version__} eager, {dt:.1f} s wall for the timed chunks ({stamps[1] - stamps[0]:.1f} s for the fill chunk)"}


if __name__ == "__main__":
    main()
