#!/usr/bin/env python
"""Measure (tile rows, split-K) for every GEMM shape of the supported models on this GPU and print the tuned-plan table that
streaming-vlm_amd/csrc/gemm.hip embeds (kTunedPlans).  8-28 different weight matrices per timing (cold weights), graph replay.
Keeps a plan only when it beats the cost model's own choice by more than 4 %."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streaming_vlm_amd.ops import HipOps

o = HipOps()
bf = torch.bfloat16
r = lambda *s: (torch.randn(*s, device="cuda") * 0.05).to(bf)
MP, MV = 290, 1024          # prefill rows of a 448x448 chunk (bucket ceil(M/64) = 5), ViT patches of a 448x448 grid (bucket 16)
only = os.environ.get("GROUPS")
shapes = {
    "2b": [(MP, 2048, 1536), (MP, 1536, 1536), (MP, 17920, 1536), (MP, 1536, 8960)],
    "7b": [(MP, 4608, 3584), (MP, 3584, 3584), (MP, 37888, 3584), (MP, 3584, 18944)],
    "2.5-3b": [(MP, 2560, 2048), (MP, 2048, 2048), (MP, 22016, 2048), (MP, 2048, 11008)],
    "vit": [(MV, 1280, 1176), (MV, 3840, 1280), (MV, 1280, 1280), (MV, 5120, 1280), (MV, 1280, 5120)],
    "vit2.5": [(MV, 6848, 1280), (MV, 1280, 3424)],
    "merger": [(256, 5120, 5120), (256, 1536, 5120), (256, 3584, 5120), (256, 2048, 5120)],
}


def timeit(fn, n_w):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        s.record()
        for _ in range(3):
            g.replay()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / (3 * n_w))
    return best


plans = []
for group, lst in shapes.items():
    if only and group not in only.split(","):
        continue
    for M, N, K in lst:
        n_w = max(4, min(28, int(1.2e9 // (N * K * 2))))          # > 1 GB of distinct weights: colder than the Infinity Cache
        Ws = [r(N, K) for _ in range(n_w)]
        A, C, res = r(M, K), torch.empty(M, N, dtype=bf, device="cuda"), r(M, N)
        fn = lambda: [o.gemm(A, W, residual=res, out=C) for W in Ws]
        for k in ("SVLM_GEMM_BM", "SVLM_GEMM_SPLITS", "SVLM_GEMM_NO_TABLE"):
            os.environ.pop(k, None)
        os.environ["SVLM_GEMM_NO_TABLE"] = "1"
        base = timeit(fn, n_w)
        cands = []
        for bm in (64, 128, 192, 320):
            for sp in (1, 2, 3, 4, 5, 6, 8):
                if sp > 1 and K < 1024:
                    continue
                if bm > 128 and (K % 64 or M <= 128):
                    continue
                os.environ["SVLM_GEMM_BM"], os.environ["SVLM_GEMM_SPLITS"] = str(bm), str(sp)
                cands.append((timeit(fn, n_w), bm, sp))
        os.environ.pop("SVLM_GEMM_BM", None); os.environ.pop("SVLM_GEMM_SPLITS", None)
        cands.sort()
        t, bm, sp = cands[0]
        keep = t < 0.96 * base
        print(f"{group:8s} M={M:5d} N={N:6d} K={K:6d}: model {base:7.2f} us, best {t:7.2f} us (bm{bm}, s{sp}){'  <- table' if keep else ''}   top: "
              + " ".join(f"{t_:.1f}(bm{b_},s{s_})" for t_, b_, s_ in cands[:5]), flush=True)
        if keep:
            plans.append({"group": group, "mb": (M + 63) // 64, "N": N, "K": K, "bm": bm, "splits": sp, "us": round(t, 2), "model_us": round(base, 2)})
        del Ws
print("\n// ---- paste into gemm.hip (kTunedPlans)")
for p in plans:
    print(f"    {{{p['mb']}, {p['N']}, {p['K']}, {p['bm']}, {p['splits']}}},   // {p['group']}: {p['us']} us vs {p['model_us']} us for the cost model's choice")
json.dump(plans, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "gemm_plans_mi355x.json"), "w"), indent=1)
