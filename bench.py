#!/usr/bin/env python
"""Headline benchmark: sustained frames/s (+ decode tokens/s) of the streaming hot path per MI355X.

One "step" = one chunk of the stream = evict (sink/window) -> ViT on the chunk's frame -> merger ->
LLM prefill of the chunk's ~275 new tokens -> 20 greedy decode tokens, through
``streaming_inference`` with its inputs already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model 2b|7b|tiny]

N > 1 is launched by the driver with torch.distributed.run: one independent stream per rank, RCCL
barrier only around the timed region (streams never interact -> "weak" scaling).
Prints ONE JSON line on rank 0.
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


class TimedOps:
    """Wraps HipOps: every launch is bracketed by HIP events on the launching stream and tagged with its
    algorithmic bytes / flops (formulas in DESIGN.md 'Kernels and rooflines')."""

    def __init__(self, ops):
        self._ops = ops
        self.records = []
        self.name = "hip-timed"

    def decode_attn_ws(self, *a, **k):
        return self._ops.decode_attn_ws(*a, **k)

    def _run(self, kernel, nbytes, flops, fn, *a, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = fn(*a, **k)
        e.record()
        self.records.append((kernel, nbytes, flops, s, e))
        return r

    def gemm(self, A, W, bias=None, residual=None, out=None, act=0):
        M, K = A.shape
        N = W.shape[0]
        nb = 2 * (M * K + N * K + M * N * (2 if residual is not None else 1))
        return self._run("gemm_bf16_kernel", nb, 2.0 * M * N * K, self._ops.gemm, A, W, bias, residual, out, act)

    def gemv(self, x, W, bias=None, residual=None, out=None, out_f32=None, act=0):
        N, K = W.shape
        nb = 2 * (N * K + K + N * (2 if residual is not None else 1)) + (4 * N if out_f32 is not None else 0)
        return self._run("gemv_bf16_kernel", nb, 2.0 * N * K, self._ops.gemv, x, W, bias, residual, out, out_f32, act)

    def dec_qkv(self, x, ln_w, eps, W, bias, q_out, pool, layer, slot_of, qd, kd, length=0, len_dev=None):
        N, K = W.shape
        return self._run("dec_qkv_kernel", 2 * (N * K + 2 * K + 2 * N), 2.0 * N * K, self._ops.dec_qkv, x, ln_w, eps, W, bias, q_out,
                         pool, layer, slot_of, qd, kd, length, len_dev)

    def dec_gate_up(self, x, ln_w, eps, W, h):
        N, K = W.shape
        return self._run("dec_gate_up_kernel", 2 * (N * K + 2 * K + N // 2), 2.0 * N * K, self._ops.dec_gate_up, x, ln_w, eps, W, h)

    def dec_lm_head(self, x, ln_w, eps, W, logits, seen, penalty, suppress, ws):
        N, K = W.shape
        return self._run("dec_lm_head_kernel", 2 * (N * K + 2 * K) + 5 * N, 2.0 * N * K, self._ops.dec_lm_head, x, ln_w, eps, W, logits,
                         seen, penalty, suppress, ws)

    def sampling_ws(self, *a, **k):
        return self._ops.sampling_ws(*a, **k)

    def decode_attn(self, q, pool, layer, slot_of, rope_cs, out, ws, Hq, max_len, chunk, scale, length=0, len_dev=None):
        _, _, Hkv, _, D = pool.shape
        L = (int(len_dev[0]) if len_dev is not None else 0) + length     # host read: instrumented pass only
        nb = 2 * L * Hkv * D * 2 + L * 3 * 4 + 2 * Hq * D * 2 + 2 * Hkv * D * 2
        return self._run("decode_attn_split_kernel", nb, 4.0 * L * Hq * D, self._ops.decode_attn, q, pool, layer, slot_of, rope_cs,
                         out, ws, Hq, max_len, chunk, scale, length, len_dev)

    def prefill_attn(self, q, pool, layer, slot_of, rope_cs, out, T, L, Hq, scale):
        _, _, Hkv, _, D = pool.shape
        nb = 2 * L * Hkv * D * 2 + 2 * T * Hq * D * 2
        return self._run("flash_attn_kernel(prefill)", nb, 4.0 * T * (L - T / 2.0) * Hq * D, self._ops.prefill_attn, q, pool, layer,
                         slot_of, rope_cs, out, T, L, Hq, scale)

    def vit_attn(self, qkv, n_seq, seq_len, H, d, scale, out=None):
        N = qkv.shape[0]
        return self._run("flash_attn_kernel(vit)", 2 * N * 4 * H * d, 4.0 * n_seq * seq_len * seq_len * H * d, self._ops.vit_attn,
                         qkv, n_seq, seq_len, H, d, scale, out)

    def __getattr__(self, name):
        fn = getattr(self._ops, name)

        def wrapped(*a, **k):
            return self._run(name, 0, 0.0, fn, *a, **k)
        return wrapped

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for kernel, nb, fl, s, e in self.records:
            a = agg.setdefault(kernel, dict(launches=0, ms=0.0, bytes=0, flops=0.0))
            a["launches"] += 1
            a["ms"] += s.elapsed_time(e)
            a["bytes"] += nb
            a["flops"] += fl
        return agg


_T0 = time.perf_counter()


def log(msg):
    """progress on stderr (stdout carries only the one JSON line)"""
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="2b", choices=["2b", "7b", "tiny"])
    ap.add_argument("--size", type=int, default=448)
    ap.add_argument("--fps", type=float, default=1.0)
    ap.add_argument("--sink", type=int, default=4)
    ap.add_argument("--window", type=int, default=2048)
    ap.add_argument("--new-tokens", type=int, default=20)
    ap.add_argument("--cpu-chunks", type=int, default=3, help="chunks of the same stream timed on the host cores (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    import streaming_vlm_amd as S
    from streaming_vlm_amd import multi_stream as MS
    dist, rank, world, local_rank = MS.init_distributed("nccl")          # nccl == RCCL over xGMI on ROCm
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from streaming_vlm_amd import config as C
    from streaming_vlm_amd.synthetic import ResidentProcessor, ResidentVideo
    from streaming_vlm_amd.weights import random_state_dict

    cfg = {"2b": C.qwen2_vl_2b, "7b": C.qwen2_vl_7b, "tiny": C.tiny}[args.model]()
    n_chunks = args.warmup + args.steps
    tok_per_frame = (args.size // 28) ** 2
    chunk_tokens = tok_per_frame + 24 + args.new_tokens
    max_len = args.sink + args.window + 2 * chunk_tokens + 64
    log(f"rank {rank}/{world}: building {cfg.name} random weights on {dev}")
    sd = random_state_dict(cfg, 0, dev)
    model = S.StreamingQwen2VL(cfg, sd, dev, max_len=max_len, max_new_tokens=args.new_tokens)
    log("engine ready; staging the synthetic stream in HBM")
    video = ResidentVideo(n_chunks + 1, args.size, args.fps, rank, dev)          # inputs resident in HBM before timing
    proc = ResidentProcessor()
    frames_per_chunk = video.frames_per_chunk

    t = {}
    counts = []

    def fence():
        MS.fence(dist, dev)

    def on_chunk(i):
        if i == 0:
            log("stream started (warmup)")
        if i == args.warmup:
            fence()
            log("timed region starts")
            t["t0"] = time.perf_counter()

    S.streaming_inference(model=model, processor=proc, video=video, model_base="Qwen2", duration=n_chunks, previous_text="",
                          kv_policy="sink_window", sink=args.sink, window=args.window, do_sample=False,
                          max_new_tokens=args.new_tokens, suppress_eos=True, quiet=True, token_counts=counts, chunk_callback=on_chunk)
    fence()
    elapsed = time.perf_counter() - t["t0"]
    log(f"timed region done: {elapsed:.3f} s for {args.steps} chunks")
    frames = args.steps * frames_per_chunk
    tokens = sum(counts[args.warmup:])
    agg = MS.aggregate(frames, tokens, elapsed, dist, dev)
    t_max, fps_total, tps_total, per_gpu_fps = agg["t_max"], agg["frames_per_sec"], agg["tokens_per_sec"], agg["per_rank_frames_per_sec"]

    out = {
        "metric": "frames_per_sec", "value": round(fps_total, 3), "unit": "frames/s",
        "decode_tokens_per_sec": round(tps_total, 2), "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * t_max / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"{cfg.name} bf16, {args.size}x{args.size} @{args.fps:g}fps synthetic stream, KV sink={args.sink} "
                               f"window={args.window}, {args.new_tokens} greedy tokens/chunk, one stream per GPU",
                   "frames_per_chunk": frames_per_chunk, "new_tokens_per_chunk": args.new_tokens, "kv_len_steady": None,
                   "parallelism": f"streams{world}"},
        "per_gpu_frames_per_sec": [round(v, 3) for v in per_gpu_fps],
    }

    if rank == 0 and not args.no_roofline:
        log("roofline pass (eager launches bracketed by HIP events)")
        out.update(roofline_pass(model, proc, video, args, n_chunks))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log("cpu baseline (oracle on host cores)")
        out["cpu_baseline"] = cpu_baseline(cfg, sd, args)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def roofline_pass(model, proc, video, args, n_chunks):
    """Re-run the last chunk's generate with eager launches bracketed by HIP events (same operands, weights
    cold in HBM: 3 GB of weights per token never fit the 256 MiB Infinity Cache)."""
    import streaming_vlm_amd as S
    eng = model._svlm_engine
    timed = TimedOps(eng.ops)
    real_ops, real_graph = eng.ops, eng.use_graph
    counts = []
    # a fresh short stream at steady-state length: warm the cache with untimed chunks, then time 2 chunks eagerly
    fill = max(2, (args.sink + args.window) // ((args.size // 28) ** 2 + 40) + 2)
    fill = min(fill, n_chunks - 2)

    def on_chunk(i):
        if i == fill:
            torch.cuda.synchronize()
            eng.ops, eng.use_graph = timed, False
            eng._graph = None
    try:
        kvlog = []
        S.streaming_inference(model=model, processor=proc, video=video, model_base="Qwen2", duration=fill + 2, previous_text="",
                              kv_policy="sink_window", sink=args.sink, window=args.window, do_sample=False,
                              max_new_tokens=args.new_tokens, suppress_eos=True, quiet=True, token_counts=counts,
                              chunk_callback=on_chunk, ids_log=kvlog)
    finally:
        eng.ops, eng.use_graph = real_ops, real_graph
        eng._graph = None
    agg = timed.summary()
    kernels = []
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
        if a["launches"] == 0:
            continue
        us = 1e3 * a["ms"] / a["launches"]
        k = {"kernel": name, "launches": a["launches"], "total_ms": round(a["ms"], 3), "avg_us": round(us, 2)}
        if a["bytes"]:
            k["GBps"] = round(a["bytes"] / (a["ms"] * 1e-3) / 1e9, 1)
        if a["flops"]:
            k["TFLOPs"] = round(a["flops"] / (a["ms"] * 1e-3) / 1e12, 2)
        kernels.append(k)
    dom = kernels[0]
    mfma_bound = dom["kernel"].startswith("gemm") or dom["kernel"].startswith("flash")
    if mfma_bound:
        ach = dom.get("TFLOPs", 0.0)
        roof = {"kernel": dom["kernel"], "bound": "mfma", "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None}
    else:
        ach = dom.get("GBps", 0.0)
        roof = {"kernel": dom["kernel"], "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None}
    roof["avg_launch_us"] = dom["avg_us"]
    roof["launches"] = dom["launches"]
    da = next((k for k in kernels if k["kernel"].startswith("decode_attn")), None)
    extra = {"roofline": roof, "kernels": kernels[:12], "kv_len_timed": kvlog[-1]["kv_len"] if kvlog else None}
    if da is not None:
        extra["roofline_decode_attn"] = {"bound": "hbm", "achieved": da.get("GBps"), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": round(da.get("GBps", 0.0) / HBM_PEAK_GBS, 4), "avg_launch_us": da["avg_us"]}
    return extra


def cpu_baseline(cfg, sd, args):
    """The reference has no runnable CPU path: time the CPU oracle (eager-PyTorch restatement of the
    reference algorithm: torch.cat KV, index_select eviction, RoPE of all cached keys every step) on this
    box's host cores, on a bounded sample of the SAME stream."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(avail, int(os.environ.get("SVLM_CPU_THREADS", "16"))))     # a 1-GPU box owns a 16-core share
    torch.set_num_threads(cores)
    sd_cpu = {k: v.cpu() for k, v in sd.items()}
    n = max(1, args.cpu_chunks)
    t0 = time.perf_counter()
    H.run_oracle_stream(cfg, sd_cpu, n, size=args.size, fps=args.fps, policy="sink_window", sink=args.sink, window=args.window,
                        max_new=args.new_tokens, suppress_eos=True, previous_text="")
    dt = time.perf_counter() - t0
    return {"value": round(n * max(1, int(round(args.fps))) / dt, 4), "unit": "frames/s", "cores": torch.get_num_threads(),
            "kind": "port", "decode_tokens_per_sec": round(n * args.new_tokens / dt, 3),
            "sample": f"first {n} chunks of the same synthetic stream (KV below the window), bf16 weights, torch {torch.__version__} eager, "
                      f"{dt:.1f} s wall"}


if __name__ == "__main__":
    main()
