#!/usr/bin/env python
"""Sweep (tile rows, split-K) for every GEMM shape of a chunk against the cost model's own choice (SVLM_GEMM_BM / SVLM_GEMM_SPLITS
are read per call).  28 different weight matrices per timing, replayed from a graph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streaming_vlm_amd.ops import HipOps

o = HipOps()
bf = torch.bfloat16
r = lambda *s: (torch.randn(*s, device="cuda") * 0.05).to(bf)
Mp = int(os.environ.get("M_PREFILL", 290))
if os.environ.get("SHAPES") == "7b":
    shapes = [("7b qkv", Mp, 4608, 3584), ("7b o", Mp, 3584, 3584), ("7b gate_up", Mp, 37888, 3584), ("7b down", Mp, 3584, 18944)]
else:
  shapes = [("prefill qkv", Mp, 2048, 1536), ("prefill o", Mp, 1536, 1536), ("prefill gate_up", Mp, 17920, 1536),
          ("prefill down", Mp, 1536, 8960), ("vit qkv", 1024, 3840, 1280), ("vit proj", 1024, 1280, 1280),
          ("vit fc1", 1024, 5120, 1280), ("vit fc2", 1024, 1280, 5120)]
NW = 8 if os.environ.get("SHAPES") == "7b" else 28


def timeit(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(4):
        g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (4 * NW)


for name, M, N, K in shapes:
    Ws = [r(N, K) for _ in range(NW)]
    A, C = r(M, K), torch.empty(M, N, dtype=bf, device="cuda")
    res = r(M, N)
    fn = lambda: [o.gemm(A, W, residual=res, out=C) for W in Ws]
    os.environ.pop("SVLM_GEMM_BM", None); os.environ.pop("SVLM_GEMM_SPLITS", None)
    base = timeit(fn)
    out = []
    for bm in (64, 128):
        for sp in (1, 2, 3, 4, 6, 8, 12):
            if sp > 1 and K < 1024:
                continue
            os.environ["SVLM_GEMM_BM"], os.environ["SVLM_GEMM_SPLITS"] = str(bm), str(sp)
            out.append((timeit(fn), bm, sp))
    os.environ.pop("SVLM_GEMM_BM", None); os.environ.pop("SVLM_GEMM_SPLITS", None)
    out.sort()
    print(f"{name:16s} M={M} N={N} K={K}: model's choice {base:7.2f} us ({2 * M * N * K / base / 1e6:5.0f} TF/s, W {N * K * 2 / base / 1e3:5.0f} GB/s) | best " + ", ".join(f"{t:6.2f}us(bm{b},s{s})" for t, b, s in out[:4]), flush=True)
