#!/usr/bin/env python
"""Time svlm_gemm_bf16 on every GEMM shape of one chunk (BASELINE configs[1]), each shape over 28 DIFFERENT weight
matrices replayed from one HIP graph (cold weights, no launch gaps).  Prints us / TFLOP/s / weight GB/s per shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streaming_vlm_amd.ops import HipOps
from streaming_vlm_amd._lib import ACT_NONE

o = HipOps()
bf = torch.bfloat16
r = lambda *s: (torch.randn(*s, device="cuda") * 0.05).to(bf)
Mp = int(os.environ.get("M_PREFILL", 290))
shapes = [("prefill qkv", Mp, 2048, 1536, True, False), ("prefill o", Mp, 1536, 1536, False, True),
          ("prefill gate_up", Mp, 17920, 1536, False, False), ("prefill down", Mp, 1536, 8960, False, True),
          ("vit qkv", 1024, 3840, 1280, True, False), ("vit proj", 1024, 1280, 1280, True, True),
          ("vit fc1", 1024, 5120, 1280, True, False), ("vit fc2", 1024, 1280, 5120, True, True),
          ("merger 0", 256, 5120, 5120, True, False), ("merger 2", 256, 1536, 5120, True, False)]
only = sys.argv[1] if len(sys.argv) > 1 else None
tot = 0.0
for name, M, N, K, has_bias, has_res in shapes:
    if only and only not in name:
        continue
    nW = 28
    # PAD_W / PAD_A (elements): leading-dimension padding, to see whether power-of-two-ish row strides cost L2 channel conflicts
    pw, pa = int(os.environ.get("PAD_W", 0)), int(os.environ.get("PAD_A", 0))
    Ws = [r(N, K + pw)[:, :K] for _ in range(nW)]
    A, C = r(M, K + pa)[:, :K], torch.empty(M, N, dtype=bf, device="cuda")
    bias = r(N) if has_bias else None
    res = r(M, N) if has_res else None
    fn = lambda: [o.gemm(A, W, bias=bias, residual=res, out=C) for W in Ws]
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        g.replay()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / (5 * nW)
    print(f"{name:16s} M={M:5d} N={N:6d} K={K:5d}  {us:8.2f} us  {2*M*N*K/us/1e6:7.1f} TF/s  weights {N*K*2/us/1e3:7.1f} GB/s")
