import sys, os, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import streaming_vlm_amd as S
from streaming_vlm_amd import config as C
from streaming_vlm_amd.weights import random_state_dict
from ref_ops import RefOps
torch.set_num_threads(4)
cfg = C.tiny()
sd = random_state_dict(cfg, 0, "cpu")
model = S.StreamingQwen2VL(cfg, sd, "cpu", ops=RefOps(), max_len=768, max_new_tokens=4, use_graph=False)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
marks = []
def cb(i):
    marks.append(time.perf_counter())
S.streaming_inference(model=model, processor=S.SyntheticProcessor(), video_path="synthetic://28x28@1fps", model_base="Qwen2", duration=N,
                      previous_text="hello", kv_policy="sink_window", sink=4, window=64, do_sample=False, max_new_tokens=4, suppress_eos=True,
                      quiet=True, chunk_callback=cb)
import numpy as np
d = np.diff(np.array(marks)) * 1e3
for a in range(0, len(d), max(len(d) // 10, 1)):
    print(f"chunks {a:6d}..: {d[a:a + len(d) // 10].mean():7.3f} ms per chunk")
