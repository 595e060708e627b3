#!/usr/bin/env python
"""svlm_gemm_bf16 at the large-M shapes of configs[4]: the dense prefill's 4096-row LLM passes (Qwen2-VL-7B) and the ViT's
8-grid batches (M = 8192).  Each shape over several different weight matrices replayed from one graph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streaming_vlm_amd.ops import HipOps
from streaming_vlm_amd._lib import ACT_NONE, ACT_SWIGLU

o = HipOps()
bf = torch.bfloat16
r = lambda *s: (torch.randn(*s, device="cuda") * 0.05).to(bf)
M7 = int(os.environ.get("GB_M", 4096))       # rows of the 7B shapes (590 = the streaming chunk of configs[2])
shapes = [("7b qkv", M7, 4608, 3584, ACT_NONE), ("7b o", M7, 3584, 3584, ACT_NONE), ("7b gate_up", M7, 18944, 3584, ACT_SWIGLU),
          ("7b down", M7, 3584, 18944, ACT_NONE), ("vit qkv", 8192, 3840, 1280, ACT_NONE), ("vit proj", 8192, 1280, 1280, ACT_NONE),
          ("vit fc1", 8192, 5120, 1280, ACT_NONE), ("vit fc2", 8192, 1280, 5120, ACT_NONE)]
only = sys.argv[1] if len(sys.argv) > 1 else None
for name, M, N, K, act in shapes:
    if only and only not in name:
        continue
    nW = 4 if M >= 2048 else 12
    Ws = [r(2 * N if act == ACT_SWIGLU else N, K) for _ in range(nW)]
    A, C = r(M, K), torch.empty(M, N, dtype=bf, device="cuda")
    fn = lambda: [o.gemm(A, W, out=C, act=act) for W in Ws]
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
    fl = 2.0 * M * (2 * N if act == ACT_SWIGLU else N) * K
    print(f"{name:12s} M={M:5d} N={N:6d} K={K:5d}  {us:8.1f} us  {fl/us/1e6:7.1f} TF/s", flush=True)
