#!/usr/bin/env python
"""How much of a chunk's prefill is launch gaps: the same ~390 launches issued eagerly from Python vs replayed from a graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import streaming_vlm_amd as S
from streaming_vlm_amd import config as C
from streaming_vlm_amd.weights import random_state_dict

cfg = C.qwen2_vl_2b()
sd = random_state_dict(cfg, 0, "cuda")
model = S.StreamingQwen2VL(cfg, sd, "cuda", max_len=2700, max_new_tokens=20)
eng = model._svlm_engine
T, L_before = 290, 1840
cache = eng.new_cache()
cache.reserve(L_before + T + 20); cache.commit(L_before); cache.sync_device()
idx = torch.randint(0, 1000, (T,), dtype=torch.int32, device="cuda")
eng.rope_cs.normal_()
def run():
    eng._prefill(cache, idx, None, T, L_before)
run(); torch.cuda.synchronize()
def timeit(fn, n=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
t_eager = timeit(run)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    run()
t_graph = timeit(g.replay)
t0 = time.perf_counter(); run(); t_host = (time.perf_counter() - t0) * 1e3
print(f"prefill T={T} L={L_before + T}: eager {t_eager:.3f} ms, graph {t_graph:.3f} ms, host enqueue {t_host:.3f} ms")
