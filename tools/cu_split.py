#!/usr/bin/env python
"""Space-partition experiment for the look-ahead ViT: the decode graph on a stream masked to 8-k XCDs, the ViT pass on a stream
masked to the other k (hipExtStreamCreateWithCUMask; on a multi-XCC device mask bit b is CU b // 8 of XCC b % 8), against
today's unmasked pair.  Reports 19 decode replays alone / ViT alone / both together for every split."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import streaming_vlm_amd as S
from streaming_vlm_amd import config as C
from streaming_vlm_amd.weights import random_state_dict
from streaming_vlm_amd.synthetic import ResidentVideo, ResidentProcessor

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(xcds):
    bits = [0] * 8                       # 256 CUs -> 8 x uint32
    for b in range(256):
        if b % 8 in xcds:
            bits[b // 32] |= 1 << (b % 32)
    arr = (ctypes.c_uint32 * 8)(*bits)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


cfg = C.qwen2_vl_2b()
model = S.StreamingQwen2VL(cfg, random_state_dict(cfg, 0, "cuda"), "cuda", max_len=2700, max_new_tokens=20)
eng = model._svlm_engine
video = ResidentVideo(12, 448, 1.0, 0, "cuda")
S.streaming_inference(model=model, processor=ResidentProcessor(), video=video, model_base="Qwen2", duration=10, previous_text="",
                      kv_policy="sink_window", sink=4, window=2048, do_sample=False, max_new_tokens=20, suppress_eos=True, quiet=True)
torch.cuda.synchronize()
g = eng._graph
pix, grid = video.chunks[3].pixel_values, video.chunks[3].grid


def t(fn, n=5):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e))
    return best


def decode():
    eng.state.copy_(torch.tensor([2100, 0], dtype=torch.int32))
    for _ in range(19):
        g.replay()


def vit():
    eng.vision_forward(pix, grid)


def run(main, side):
    def on(st, fn):
        def f():
            cur = torch.cuda.current_stream()
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                fn()
            cur.wait_stream(st)
        return f

    def both():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur); main.wait_stream(cur)
        with torch.cuda.stream(side):
            vit()
        with torch.cuda.stream(main):
            decode()
        cur.wait_stream(side); cur.wait_stream(main)
    return t(on(main, decode)), t(on(side, vit)), t(both)


plain = run(torch.cuda.Stream(), torch.cuda.Stream())
print(f"unmasked          : decode x19 {plain[0]:.3f} ms | vit {plain[1]:.3f} ms | together {plain[2]:.3f} ms", flush=True)
for k in (1, 2, 3, 4):
    v_x = set(range(k))
    d_x = set(range(8)) - v_x
    r = run(masked_stream(d_x), masked_stream(v_x))
    print(f"vit on {k} XCD(s), decode on {8 - k}: decode x19 {r[0]:.3f} ms | vit {r[1]:.3f} ms | together {r[2]:.3f} ms", flush=True)
    r = run(torch.cuda.Stream(), masked_stream(v_x))
    print(f"vit on {k} XCD(s), decode unmasked  : decode x19 {r[0]:.3f} ms | vit {r[1]:.3f} ms | together {r[2]:.3f} ms", flush=True)
