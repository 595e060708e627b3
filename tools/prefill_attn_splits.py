import os, sys
sys.path.insert(0, "/root/repo")
import torch
from streaming_vlm_amd.ops import HipOps
o = HipOps(); bf = torch.bfloat16
r = lambda *s: torch.randn(*s, device="cuda").to(bf)
for Hq, Hkv, T, L in ((12, 2, 290, 2400), (28, 4, 290, 4600)):
    pools = [r(1, 2, Hkv, L + 64, 128) for _ in range(8)]
    slot = torch.arange(L + 64, dtype=torch.int32, device="cuda"); rope = r(L + 64, 128)
    q, out = r(T, Hq * 128), torch.empty(T, Hq * 128, dtype=bf, device="cuda")
    fn = lambda: [o.prefill_attn(q, p, 0, slot, rope, out, T, L, Hq, 128 ** -0.5) for p in pools]
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): g.replay()
    e.record(); torch.cuda.synchronize()
    print(f"splits={os.environ.get('SVLM_PREFILL_SPLITS','auto')} Hq{Hq} T{T} L{L}: {s.elapsed_time(e)*1e3/40:.1f} us")
