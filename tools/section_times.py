"""Per-section wall time of the streaming loop (PKV/VIDEO/INPUT/GEN/POST as the reference's time_test reports them)
plus a host-only profile of generate(): where the GPU sits idle waiting for the host."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import streaming_vlm_amd as S
from streaming_vlm_amd import config as C
from streaming_vlm_amd.weights import random_state_dict
from streaming_vlm_amd.synthetic import ResidentVideo, ResidentProcessor

cfg = C.qwen2_vl_2b()
sd = random_state_dict(cfg, 0, "cuda")
model = S.StreamingQwen2VL(cfg, sd, "cuda", max_len=4 + 2048 + 512, max_new_tokens=20)
n = 30
video = ResidentVideo(n, 448, 1.0, 0, "cuda")
proc = ResidentProcessor()
res = S.streaming_inference(model=model, processor=proc, video=video, model_base="Qwen2", duration=n, previous_text="",
                            kv_policy="sink_window", sink=4, window=2048, do_sample=False, max_new_tokens=20, suppress_eos=True,
                            quiet=True, time_test=True)
tail = res[12:]
avg = {k: 1e3 * sum(r[k] for r in tail) / len(tail) for k in tail[0]}
print("section ms/chunk:", json.dumps({k: round(v, 3) for k, v in avg.items()}), "sum", round(sum(avg.values()), 3))
if os.environ.get("PROFILE") == "1":
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    S.streaming_inference(model=model, processor=proc, video=video, model_base="Qwen2", duration=n, previous_text="",
                          kv_policy="sink_window", sink=4, window=2048, do_sample=False, max_new_tokens=20, suppress_eos=True, quiet=True)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
