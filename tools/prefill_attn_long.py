#!/usr/bin/env python
"""Prefill attention at the dense-frame forward's shapes (configs[4]: 4096-row passes of a 7B-geometry layer over a growing
cache): time per call and TFLOP/s (4 * T * (L - T/2) * Hq * 128 flop, causal).  SVLM_PREFILL_QB=1|2 picks the query blocks per wave."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streaming_vlm_amd.ops import HipOps

o = HipOps()
Hq, Hkv = (int(x) for x in os.environ.get("MB_HEADS", "28,4").split(","))
T = int(os.environ.get("MB_T", 4096))
bf = torch.bfloat16
for L in [int(x) for x in os.environ.get("MB_LS", "4096,16384,32768,82944").split(",")]:
    cap = L
    pool = torch.randn(1, 2, Hkv, cap, 128, device="cuda").to(bf)
    slot_of = torch.arange(cap, dtype=torch.int32, device="cuda")
    rope = torch.randn(cap, 128, device="cuda").to(bf)
    q = torch.randn(T, Hq * 128, device="cuda").to(bf)
    out = torch.empty_like(q)
    scale = 1 / math.sqrt(128)
    fn = lambda: o.prefill_attn(q, pool, 0, slot_of, rope, out, T, L, Hq, scale)
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 3
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / n
    fl = 4.0 * T * (L - T / 2) * Hq * 128
    print(f"T={T} L={L:6d} Hq={Hq} Hkv={Hkv}: {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s (rope_gather included)", flush=True)
    del pool
