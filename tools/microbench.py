#!/usr/bin/env python
"""Launch one kernel of the path repeatedly on BASELINE shapes (for rocprofv3 --pmc / --kernel-trace runs).
    python tools/microbench.py vit_attn|prefill_attn|gemm_fc2|gemm_qkv|decode_attn|gemv_down|resize|gu_cold|gu_prefetched|pf_only [reps]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from streaming_vlm_amd.ops import HipOps  # noqa: E402

which = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
o = HipOps()
dev = "cuda"
bf = torch.bfloat16
r = lambda *s: torch.randn(*s, device=dev).to(bf)
if which == "vit_attn":
    qkv, out = r(1024, 3840), torch.empty(1024, 1280, dtype=bf, device=dev)
    fn = lambda: o.vit_attn(qkv, 1, 1024, 16, 80, 80 ** -0.5, out=out)
elif which == "prefill_attn":
    Hq, Hkv, T, L = 12, 2, 275, 2400
    pool = r(1, 2, Hkv, 2560, 128)
    slot = torch.arange(2560, dtype=torch.int32, device=dev)
    rope = r(2560, 128)
    q, out = r(T, Hq * 128), torch.empty(T, Hq * 128, dtype=bf, device=dev)
    fn = lambda: o.prefill_attn(q, pool, 0, slot, rope, out, T, L, Hq, 128 ** -0.5)
elif which.startswith("gemm"):
    M, N, K = {"gemm_fc2": (1024, 1280, 5120), "gemm_qkv": (1024, 3840, 1280), "gemm_fc1": (1024, 5120, 1280),
               "gemm_gu": (275, 17920, 1536), "gemm_down": (275, 1536, 8960)}[which]
    A, W, C = r(M, K), r(N, K), torch.empty(M, N, dtype=bf, device=dev)
    fn = lambda: o.gemm(A, W, out=C)
elif which == "decode_attn":
    Hq, Hkv, L = 12, 2, 2350
    pool = r(1, 2, Hkv, 2560, 128)
    slot = torch.arange(2560, dtype=torch.int32, device=dev)
    rope = r(2560, 128)
    q, out = r(Hq * 128), torch.empty(Hq * 128, dtype=bf, device=dev)
    ch = int(os.environ.get("MB_CHUNK", 48))          # keys per workgroup: the engine's choice for bounded windows
    ws = o.decode_attn_ws(Hq, 2560, ch, dev)
    # MB_LIN=0: without the cache's linear planes (every row rotated from the pool); default: planes valid up to the rows decoded since the prefill
    lin = (r(1, 2, Hkv, 2560, 128), torch.tensor([L - 10, 1], dtype=torch.int32, device=dev)) if os.environ.get("MB_LIN", "1") != "0" else None
    fn = lambda: o.decode_attn(q, pool, 0, slot, rope, out, ws, Hq, 2560, ch, 128 ** -0.5, length=L, lin=lin)
elif which == "decode_attn_long":
    # long-cache streaming kernel: 8 cold pools at MB_L keys (default 32768) with MB_HEADS = "Hq,Hkv" (default the 7B's 28,4)
    Hq, Hkv = (int(v) for v in os.environ.get("MB_HEADS", "28,4").split(","))
    L = int(os.environ.get("MB_L", 32768))
    cap = L + 64
    from streaming_vlm_amd.engine import SvlmEngine
    use_lin = os.environ.get("MB_LIN", "1") != "0"
    ch = SvlmEngine.pick_decode_chunk(cap, Hkv, use_lin)
    pools = [r(1, 2, Hkv, cap, 128) for _ in range(8)]
    lin_rows = -(-cap // 16) * 16
    lins = [(r(1, 2, Hkv, lin_rows, 128), torch.tensor([L - 16, 1], dtype=torch.int32, device=dev)) if use_lin else None for _ in range(8)]
    slot = torch.arange(cap, dtype=torch.int32, device=dev)
    rope = r(cap, 128)
    q, out = r(Hq * 128), torch.empty(Hq * 128, dtype=bf, device=dev)
    ws = o.decode_attn_ws(Hq, cap, ch, dev)
    fn = lambda: [o.decode_attn(q, p, 0, slot, rope, out, ws, Hq, cap, ch, 128 ** -0.5, length=L, lin=ln) for p, ln in zip(pools, lins)]
elif which == "dec_gate_up":
    # 28 DIFFERENT weight matrices (1.5 GB > Infinity Cache), like the 28 layers of a decode step; MB_MODEL=7b: the 7B's shapes
    H, I = (3584, 18944) if os.environ.get("MB_MODEL") == "7b" else (1536, 8960)
    Ws = [r(2 * I, H) for _ in range(28)]
    x, lnw, h = r(H), r(H), torch.empty(I, dtype=bf, device=dev)
    fn = lambda: [o.dec_gate_up(x, lnw, 1e-6, W, h) for W in Ws]
elif which in ("gu_cold", "gu_prefetched", "pf_only"):
    # does an Infinity-Cache-resident weight matrix stream faster?  28 different (2I, H) matrices (1.5 GB, far beyond the cache),
    # graph-replayed: gate/up alone, prefetch kernel + gate/up, prefetch kernel alone
    H, I = 1536, 8960
    Ws = [r(2 * I, H) for _ in range(28)]
    x, lnw, h = r(H), r(H), torch.empty(I, dtype=bf, device=dev)
    nwg = int(os.environ.get("MB_PF_WGS", 1024))
    def layers():
        for W in Ws:
            if which != "gu_cold":
                o.prefetch(W, nwg)
            if which != "pf_only":
                o.dec_gate_up(x, lnw, 1e-6, W, h)
    layers(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        layers()
    fn = g.replay
elif which == "resize":
    # one chunk of a 720p source at 2 fps: (2, 3, 720, 1280) uint8 -> (2, 3, 560, 1008), the reference's smart-resized size
    src = torch.randint(0, 256, (2, 3, 720, 1280), dtype=torch.uint8, device=dev)
    dst = torch.empty((2, 3, 560, 1008), dtype=torch.uint8, device=dev)
    fn = lambda: o.resize_u8(src, 560, 1008, out=dst)
elif which == "gemv_down":
    x, W, y = r(8960), r(1536, 8960), torch.zeros(1536, dtype=bf, device=dev)
    fn = lambda: o.gemv(x, W, residual=y, out=y)
else:
    raise SystemExit(f"unknown kernel {which}")
for _ in range(3):
    fn()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps):
    fn()
e.record()
torch.cuda.synchronize()
print("done", which, reps, "avg_us", round(1e3 * s.elapsed_time(e) / reps, 2), "env", {k: v for k, v in os.environ.items() if k.startswith("SVLM_")})
