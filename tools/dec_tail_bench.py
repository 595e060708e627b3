"""Persistent decode-layer tail against the four per-op launches it replaces, graph-replayed over a stack of layers with distinct
(cold) weights: `python tools/dec_tail_bench.py [2b|7b|3b] [layers]`.  Prints us per layer for both forms and the byte rate."""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import streaming_vlm_amd  # noqa
from streaming_vlm_amd.ops import HipOps
from streaming_vlm_amd import config as C

BF16 = torch.bfloat16
which = sys.argv[1] if len(sys.argv) > 1 else "2b"
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 28
tc = {"2b": C.qwen2_vl_2b, "7b": C.qwen2_vl_7b, "3b": C.qwen2_5_vl_3b}[which]().text
H, I, qd, kd, D = tc.hidden_size, tc.intermediate_size, tc.num_heads * 128, tc.num_kv_heads * 128, 128
ops = HipOps()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s, sc=0.03: (torch.randn(s, generator=g, device=dev) * sc).to(BF16)
Ls = [dict(o=r(H, qd), ln2=r(H) + 1, gu=r(2 * I, H), down=r(H, I), ln1=r(H) + 1, qkv=r(qd + 2 * kd, H), b=r(qd + 2 * kd, sc=0.1)) for _ in range(nl)]
attn, x0 = r(qd, sc=1.0), r(H, sc=2.0)
x, h, q = x0.clone(), torch.zeros(I, dtype=BF16, device=dev), torch.zeros(qd + 2 * kd, dtype=BF16, device=dev)
pool = torch.zeros((nl + 1, 2, tc.num_kv_heads, 64, D), dtype=BF16, device=dev)
slot = torch.arange(64, dtype=torch.int32, device=dev)
ln = torch.tensor([5], dtype=torch.int32, device=dev)
ws = ops.dec_tail_ws(H, I, nl, dev)
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def per_op():
    for i, L in enumerate(Ls):
        ops.gemv(attn, L["o"], residual=x, out=x)
        ops.dec_gate_up(x, L["ln2"], tc.rms_eps, L["gu"], h)
        ops.gemv(h, L["down"], residual=x, out=x)
        ops.dec_qkv(x, L["ln1"], tc.rms_eps, L["qkv"], L["b"], q, pool, i + 1, slot, qd, kd, len_dev=ln)


assert ops.dec_tail_supported(H, I, qd, kd), "no svlm_dec_tail build for this model (2B class only)"
stamps = torch.zeros((256, 2, 16), dtype=torch.int64, device=dev)
STAMP_LAYER = nl // 2


def tails():
    ops.dec_tail_reset(ws, H, I, nl)
    for i, L in enumerate(Ls):
        nxt = (L["ln1"], L["qkv"], L["b"], q, pool, i + 1, slot, qd, kd, 0, ln)
        ops.dec_tail(attn, x, L["o"], L["ln2"], L["gu"], L["down"], tc.rms_eps, ws, i, nl, nxt=nxt,
                     stamps=stamps if (i == STAMP_LAYER and "stamps" in sys.argv) else None)


def timed(fn, reps=20):
    gr = torch.cuda.CUDAGraph()
    x.copy_(x0)
    fn(); torch.cuda.synchronize()
    with torch.cuda.graph(gr):
        fn()
    ts = []
    for _ in range(reps):
        x.copy_(x0)
        flush.fill_(1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / nl)
    ts.sort()
    return ts[len(ts) // 2], ts[0], x.clone()


mb = (H * qd + 3 * H * I + (qd + 2 * kd) * H) * 2 / 1e6
a_med, a_min, xa = timed(per_op)
b_med, b_min, xb = timed(tails)
st = int(ws[0])
err = float((xa.float() - xb.float()).abs().max()) / (float(xa.float().abs().max()) + 1e-30)
print(json.dumps({"model": which, "layers": nl, "MB_per_layer": round(mb, 1), "per_op_us_per_layer": round(a_med, 2), "per_op_min": round(a_min, 2),
                  "tail_us_per_layer": round(b_med, 2), "tail_min": round(b_min, 2), "ratio": round(b_med / a_med, 3),
                  "tail_TBps": round(mb / b_med, 2), "per_op_TBps": round(mb / a_med, 2), "status": st, "x_rel_diff": err}))

if "stamps" in sys.argv:
    st = stamps.cpu().numpy().astype("float64")
    t0 = st[:, :, 0].min()
    names_g = ["start", "x1 published", "x1 gathered", "B1", "h gathered", "B2", "B3", "x2 gathered", "B4"]
    names_c = ["start", "loads issued", "B1", "GU done", "h published", "B2", "DOWN done", "x2 published", "B4", "end"]
    import numpy as np
    for role, names in ((0, names_g), (1, names_c)):
        for i, n in enumerate(names):
            v = (st[:, role, i] - t0) / 100.0            # 100 MHz ticks -> us
            print(f"{'gatherer' if role == 0 else 'consumer0'} {n:14s} min {v.min():7.2f}  median {np.median(v):7.2f}  max {v.max():7.2f} us")
