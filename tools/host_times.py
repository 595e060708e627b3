"""Host wall time per chunk of each piece of the streaming loop (no profiler): what the GPU waits for between chunks."""
import os, sys, time, json, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import streaming_vlm_amd as S
from streaming_vlm_amd import config as C, engine as E, driver as D
from streaming_vlm_amd.weights import random_state_dict
from streaming_vlm_amd.synthetic import ResidentVideo, ResidentProcessor

acc = collections.defaultdict(float)
def wrap(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[label] += time.perf_counter() - t
    setattr(obj, name, g)

cfg = C.qwen2_vl_2b()
sd = random_state_dict(cfg, 0, "cuda")
model = S.StreamingQwen2VL(cfg, sd, "cuda", max_len=4 + 2048 + 512, max_new_tokens=20)
eng = model._svlm_engine
n = 40
video = ResidentVideo(n, 448, 1.0, 0, "cuda")
proc = ResidentProcessor()
for nm in ["_prefill", "vision_prefetch", "_decode_step", "_vision", "generate"]:
    wrap(eng, nm)
wrap(E, "rope_index_qwen2")
wrap(D, "sink_window_evict")
wrap(eng.ops, "mrope_table")
wrap(eng.ops, "mark_seen")
orig_new_cache = eng.new_cache
def new_cache(*a, **k):
    c = orig_new_cache(*a, **k)
    for nm in ["reserve", "sync_device", "commit", "release_reserved", "prune"]:
        wrap(c, nm, "cache." + nm)
    return c
eng.new_cache = new_cache
wrap(proc, "batch_decode")
mark = {}
def cb(i):
    if i == 15:
        torch.cuda.synchronize(); acc.clear(); mark["t"] = time.perf_counter()
S.streaming_inference(model=model, processor=proc, video=video, model_base="Qwen2", duration=n, previous_text="",
                      kv_policy="sink_window", sink=4, window=2048, do_sample=False, max_new_tokens=20, suppress_eos=True,
                      quiet=True, chunk_callback=cb)
torch.cuda.synchronize()
tot = (time.perf_counter() - mark["t"]) / (n - 15) * 1e3
print(f"wall {tot:.3f} ms/chunk")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:24s} {v / (n - 15) * 1e3:8.3f} ms/chunk")
