#!/usr/bin/env python
"""Cost of hiding the look-ahead ViT under the decode steps: 19 decode-graph replays alone, with a ViT pass on the side stream,
and the ViT pass alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import streaming_vlm_amd as S
from streaming_vlm_amd import config as C
from streaming_vlm_amd.weights import random_state_dict
from streaming_vlm_amd.synthetic import ResidentVideo, ResidentProcessor

cfg = C.qwen2_vl_2b()
FP8 = "--fp8" in sys.argv          # the fp8 tower: half the operand bytes through L2 -- does the look-ahead interfere less?
model = S.StreamingQwen2VL(cfg, random_state_dict(cfg, 0, "cuda"), "cuda", max_len=2700, max_new_tokens=20, vit_fp8=FP8)
eng = model._svlm_engine
video = ResidentVideo(12, 448, 1.0, 0, "cuda")
S.streaming_inference(model=model, processor=ResidentProcessor(), video=video, model_base="Qwen2", duration=10, previous_text="",
                      kv_policy="sink_window", sink=4, window=2048, do_sample=False, max_new_tokens=20, suppress_eos=True, quiet=True)
torch.cuda.synchronize()
g = eng._graph
pix, grid = video.chunks[3].pixel_values, video.chunks[3].grid
side = torch.cuda.Stream()
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
    for _ in range(19): g.replay()
def vit():
    eng.vision_forward(pix, grid)
def both():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        eng.vision_forward(pix, grid)
    decode()
    torch.cuda.current_stream().wait_stream(side)
def both_decode_first():
    # the decode replays are enqueued FIRST (0.3 ms of host time), the ~290 eager ViT launches (~3 ms of host time) behind them:
    # the side stream only waits for what was on the main stream BEFORE the decode steps
    ev = torch.cuda.Event()
    ev.record()
    decode()
    side.wait_event(ev)
    with torch.cuda.stream(side):
        eng.vision_forward(pix, grid)
    torch.cuda.current_stream().wait_stream(side)
print(f"decode enqueued first, then the vit launches: {t(both_decode_first):.3f} ms", flush=True)
# the whole pass as ONE captured graph on the side stream: no host launch time at all
gv = torch.cuda.CUDAGraph()
eng.vision_forward(pix, grid); torch.cuda.synchronize()
with torch.cuda.graph(gv, stream=side):
    vis_static = eng.vision_forward(pix, grid)
torch.cuda.synchronize()
def both_graph():
    ev = torch.cuda.Event()
    ev.record()
    side.wait_event(ev)
    with torch.cuda.stream(side):
        gv.replay()
    decode()
    torch.cuda.current_stream().wait_stream(side)
def vit_graph():
    with torch.cuda.stream(side):
        gv.replay()
    torch.cuda.current_stream().wait_stream(side)
print(f"vit as one graph: alone {t(vit_graph):.3f} ms | beside the decode replays {t(both_graph):.3f} ms", flush=True)
print(f"[{'fp8' if FP8 else 'bf16'} vit] decode x19 alone {t(decode):.3f} ms | vit alone {t(vit):.3f} ms | decode with vit on a side stream {t(both):.3f} ms", flush=True)
if FP8:
    sys.exit(0)

# ---- the same question for the prefill: ViT of the next chunk beside THIS chunk's prefill (both GEMM streams, one tile per CU each)
import copy
from streaming_vlm_amd.engine import _VitRun
class _Proxy:
    def __init__(self, eng, ops):
        self.__dict__["_e"], self.__dict__["ops"] = eng, ops
    def __getattr__(self, k):
        return getattr(self._e, k)
o2 = copy.copy(eng.ops); o2._gemm_ws = {}              # separate split-K scratch for the side stream
proxy = _Proxy(eng, o2)
T, L_before = 290, 1840
cache = eng.new_cache(); cache.reserve(L_before + T + 20); cache.commit(L_before); cache.sync_device()
idx = torch.randint(0, 1000, (T,), dtype=torch.int32, device="cuda")
def prefill():
    eng._prefill(cache, idx, None, T, L_before)
def vit2():
    run = _VitRun(proxy, pix, grid); run.blocks(0, cfg.vision.depth); run.finish()
def both2():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        vit2()
    prefill()
    torch.cuda.current_stream().wait_stream(side)
prefill(); vit2(); torch.cuda.synchronize()
print(f"prefill alone {t(prefill):.3f} ms | vit alone {t(vit2):.3f} ms | prefill with vit on a side stream {t(both2):.3f} ms")
def all3():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        vit2()
    prefill(); decode()
    torch.cuda.current_stream().wait_stream(side)
def serial_then_overlap():
    prefill()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        vit2()
    decode()
    torch.cuda.current_stream().wait_stream(side)
print(f"prefill+decode with vit started at the prefill {t(all3):.3f} ms | vit started after the prefill (today) {t(serial_then_overlap):.3f} ms")
