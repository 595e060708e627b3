#!/usr/bin/env python
"""bf16 vs fp8 GEMM on the ViT shapes (one frame: M = 1024; dense-prefill pass: M = 8192), graph-replayed, cold operands."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streaming_vlm_amd.ops import HipOps
o = HipOps()
bf = torch.bfloat16
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(bf)
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
def timed(fn, n):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    best = 1e9
    for _ in range(3):
        flush.fill_(1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / n)
    return best
for M in (1024, 8192):
    for name, N, K in (("qkv", 3840, 1280), ("proj", 1280, 1280), ("fc1", 5120, 1280), ("fc2", 1280, 5120)):
        A, W = r(M, K), r(N, K)
        C = torch.empty(M, N, dtype=bf, device="cuda")
        a8, sa = o.quant_rows_fp8(A); w8, sw = o.quant_rows_fp8(W)
        q8, s8 = torch.empty_like(a8), torch.empty_like(sa)
        t16 = timed(lambda: o.gemm(A, W, out=C), 8)
        t8 = timed(lambda: o.gemm_fp8(a8, sa, w8, sw, out=C), 8)
        tq = timed(lambda: o.quant_rows_fp8(A, q8, s8), 8)
        fl = 2.0 * M * N * K
        print(f"M={M:5d} {name:4s} bf16 {t16:7.2f} us ({fl / t16 / 1e6:6.0f} TF/s) | fp8 {t8:7.2f} us ({fl / t8 / 1e6:6.0f} TF/s) | quant {tq:6.2f} us", flush=True)
