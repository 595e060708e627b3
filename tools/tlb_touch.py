#!/usr/bin/env python
"""Does the first response of a weight-streaming launch wait for address translation?  28 different gate/up matrices (1.5 GB, cold),
one graph: [touch] + dec_gate_up per layer, where the touch kernel reads ONE element per `stride` bytes of either the matrix the next
launch streams ("real") or a dummy buffer of the same size ("ctrl": same launch overhead, no help).  If "real" beats "ctrl" by a
fraction of a microsecond at a stride of 2 MB / 64 KB, page-table walks are part of every launch's first ~1.9 us and a TLB-warming
touch from the launch in front would remove them.
    python tools/tlb_touch.py [2b|7b]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streaming_vlm_amd.ops import HipOps

o = HipOps()
bf = torch.bfloat16
dev = "cuda"
H, I = (3584, 18944) if (len(sys.argv) > 1 and sys.argv[1] == "7b") else (1536, 8960)
NL = 28
Ws = [(torch.randn((2 * I, H), device=dev) * 0.05).to(bf) for _ in range(NL)]
Ds = [torch.zeros_like(Ws[0]) for _ in range(NL)]
x, lnw, h = torch.randn(H, device=dev).to(bf), (torch.randn(H, device=dev) * 0.1 + 1).to(bf), torch.empty(I, dtype=bf, device=dev)
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
sink = torch.zeros(NL, dtype=torch.float32, device=dev)


def run(mode, stride):
    def layers():
        for i in range(NL):
            if mode != "none":
                src = Ws[i] if mode == "real" else Ds[i]
                sink[i] = src.view(-1)[::stride // 2].sum(dtype=torch.float32)
            o.dec_gate_up(x, lnw, 1e-6, Ws[i], h)
    layers(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        layers()
    ts = []
    for _ in range(6):
        flush.fill_(1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3 / NL)
    return sum(ts[1:]) / len(ts[1:])


print(f"H={H} I={I}: us per layer (touch kernel + dec_gate_up)")
print(f"  no touch kernel                 {run('none', 0):7.2f}")
for stride in (2 << 20, 64 << 10, 4 << 10):
    c, rl = run("ctrl", stride), run("real", stride)
    print(f"  stride {stride >> 10:5d} KB   ctrl {c:7.2f}   real {rl:7.2f}   real - ctrl {rl - c:+6.2f}")
