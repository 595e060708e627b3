#!/usr/bin/env python
"""Decode attention (split + combine) time vs keys per workgroup, for the bounded-window and long-cache regimes.
8 different KV pools per timing (cold K/V), replayed from one graph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streaming_vlm_amd.ops import HipOps

o = HipOps()
bf = torch.bfloat16
D = 128
cases = [(12, 2, 2150), (28, 4, 4600), (12, 2, 8192), (12, 2, 32768), (28, 4, 32768)]
chunks = [32, 48, 64, 128, 192, 256, 384, 512]
LIN = "lin" in sys.argv[1:]   # with the cache's linear planes valid up to L - 20 (what a decode step late in a chunk sees)
argv = [a for a in sys.argv[1:] if a != "lin"]
if len(argv) > 1:            # one case, one chunk: for rocprofv3 --kernel-trace --stats
    cases = [tuple(int(v) for v in argv[0].split(","))]
    chunks = [int(v) for v in argv[1].split(",")]
for Hq, Hkv, L in cases:
    cap = L + 64
    NP = 8
    pools = [(torch.randn((1, 2, Hkv, cap, D), device="cuda") * 0.5).to(bf) for _ in range(NP)]
    slot = torch.arange(cap, dtype=torch.int32, device="cuda")
    rope = (torch.randn((cap, D), device="cuda")).to(bf)
    q = torch.randn(Hq * D, device="cuda").to(bf)
    out = torch.empty(Hq * D, dtype=bf, device="cuda")
    lin_rows = -(-cap // 16) * 16
    lins = [((torch.randn((1, 2, Hkv, lin_rows, D), device="cuda") * 0.5).to(bf), torch.tensor([L - 20, 1], dtype=torch.int32, device="cuda"))
            for _ in range(NP)] if LIN else [None] * NP
    nb = 2 * L * Hkv * D * 2 + L * 3 * 4
    line = []
    for ch in chunks:
        ws = o.decode_attn_ws(Hq, cap, ch, "cuda")
        fn = lambda: [o.decode_attn(q, p, 0, slot, rope, out, ws, Hq, cap, ch, D ** -0.5, length=L, lin=ln) for p, ln in zip(pools, lins)]
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
        us = s.elapsed_time(e) * 1e3 / (5 * NP)
        line.append(f"{ch}:{us:6.2f}us({nb / us / 1e3:5.0f}GB/s)")
    print(f"Hq{Hq} Hkv{Hkv} L{L}{' lin' if LIN else ''}: " + "  ".join(line), flush=True)
    del pools
