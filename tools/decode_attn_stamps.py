#!/usr/bin/env python
"""Where the bounded-window decode attention's time goes: wall-clock stamps (s_memrealtime, 10 ns) of every workgroup of the split and
the combine kernel of ONE layer in the middle of a 28-layer replay (28 different cold pools, one graph, a GEMV in front of every pair
like the decode step's QKV).  Diagnostic library only:
    python tools/build_diag_lib.py && SVLM_LIB_PATH=streaming-vlm_amd/build/libsvlm_hip_diag.so python tools/decode_attn_stamps.py [Hq,Hkv,L [chunk]]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from streaming_vlm_amd.ops import HipOps

o = HipOps()
bf = torch.bfloat16
D = 128
Hq, Hkv, L = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "12,2,2129").split(","))
ch = int(sys.argv[2]) if len(sys.argv) > 2 else 48
NL, LAYER = 28, 14
cap = -(-(L + 64) // 64) * 64
lin_rows = -(-cap // 16) * 16
pools = [(torch.randn((1, 2, Hkv, cap, D), device="cuda") * 0.5).to(bf) for _ in range(NL)]
lins = [((torch.randn((1, 2, Hkv, lin_rows, D), device="cuda") * 0.5).to(bf), torch.tensor([L - 10, 1], dtype=torch.int32, device="cuda")) for _ in range(NL)]
slot = torch.arange(cap, dtype=torch.int32, device="cuda")
rope = torch.randn((cap, D), device="cuda").to(bf)
q = torch.randn(Hq * D, device="cuda").to(bf)
out = torch.empty(Hq * D, dtype=bf, device="cuda")
ws = o.decode_attn_ws(Hq, cap, ch, "cuda")
len_dev = torch.tensor([L - 1], dtype=torch.int32, device="cuda")
Wq = [torch.randn((2048, 1536), device="cuda").to(bf) for _ in range(NL)]      # a QKV-sized GEMV in front of every pair
x = torch.randn(1536, device="cuda").to(bf)
y = torch.empty(2048, dtype=bf, device="cuda")
stamps = torch.zeros((8192, 8), dtype=torch.int64, device="cuda")

lib = ctypes.CDLL(os.environ["SVLM_LIB_PATH"])
lib.svlm_diag_set_da_stamps.argtypes = [ctypes.c_void_p]


def layers(stamp):
    for i in range(NL):
        o.gemv(x, Wq[i], out=y)
        if stamp and i == LAYER:
            pass
        o.decode_attn(q, pools[i], 0, slot, rope, out, ws, Hq, cap, ch, D ** -0.5, length=1, len_dev=len_dev, lin=lins[i])


layers(False); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    layers(False)
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
ts = []
for _ in range(5):
    flush.fill_(1)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    ts.append(s.elapsed_time(e) * 1e3 / NL)
print(f"Hq{Hq} Hkv{Hkv} L{L} chunk {ch}: GEMV + split + combine {np.mean(ts):.2f} us per layer (graph of {NL})")
# stamped pass: eager launches of the same sequence with the stamp buffer armed around ONE layer (stamps of the last armed launch stay)
lib.svlm_diag_set_da_stamps(ctypes.c_void_p(stamps.data_ptr()))
flush.fill_(1)
g.replay(); torch.cuda.synchronize()          # every layer stamps; the buffer keeps the LAST layer's (layer NL - 1): cold like the others
lib.svlm_diag_set_da_stamps(ctypes.c_void_p(0))
st = stamps.cpu().numpy().astype(np.float64)
ns = -(-L // ch)
sp = st[:ns * Hkv]
sp = sp[sp[:, 6] > 0]          # (a workgroup beyond the cache's end leaves after its first stamp)
n_c = Hq * (2 if ns > 64 else 1) * (2 if ns > 192 else 1)
cb = st[4096:4096 + n_c]
t0 = sp[:, 0].min()
tick = 0.01      # 100 MHz -> us
names_s = ["entry", "length known", "K/V + q staged (loads landed)", "barrier 1", "QK + PV done", "wave merge done", "partials stored"]
print(f"split kernel, {sp.shape[0]} workgroups; us after the first workgroup's entry: min / median / max")
for i, n in enumerate(names_s):
    v = (sp[:, i] - t0) * tick
    print(f"  {n:32s} {v.min():6.2f} {np.median(v):6.2f} {v.max():6.2f}")
names_c = ["entry", "partials merged (loads landed)", "barrier", "out stored"]
print(f"combine kernel, {cb.shape[0]} workgroups:")
for i, n in enumerate(names_c):
    v = (cb[:, i] - t0) * tick
    print(f"  {n:32s} {v.min():6.2f} {np.median(v):6.2f} {v.max():6.2f}")
