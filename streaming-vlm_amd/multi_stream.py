"""One independent stream per GPU -- the only parallelism the path has (reference:
eval/livesports3kcc/distributed_generate_streaming.py:55,62,127-143: N processes, `cuda:{rank}`, strided
sample sharding, results joined outside the model; no data-path communication).

`torch.distributed` ("nccl" == RCCL over xGMI on ROCm, "gloo" in CPU tests) is used for exactly two things,
both OUTSIDE the timed region: a barrier on each side of it and one all_gather of three doubles per rank.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_distributed(backend: Optional[str] = None):
    """Returns (dist or None, rank, world, local_rank).  Rendezvous comes from MASTER_ADDR/MASTER_PORT."""
    rank, world, local_rank = env_rank()
    if world == 1:
        return None, 0, 1, local_rank
    import torch.distributed as dist
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        kw["device_id"] = torch.device("cuda", local_rank)
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return dist, rank, world, local_rank


def shard(items: List, rank: int, world: int) -> List:
    """Strided sharding of independent units (streams / samples): item i -> rank i mod world."""
    return items[rank::world]


def fence(dist, device=None):
    """barrier bracketed by device syncs: everything enqueued before is finished on every rank."""
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def aggregate(frames: float, tokens: float, seconds: float, dist, device="cpu") -> dict:
    """Whole-job throughput = units of ALL ranks / slowest rank's time."""
    mine = torch.tensor([frames, tokens, seconds], dtype=torch.float64, device=device)
    if dist is None:
        allst = mine.cpu().unsqueeze(0)
    else:
        parts = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, mine)
        allst = torch.stack(parts).cpu()
    t_max = float(allst[:, 2].max())
    return {"frames_per_sec": float(allst[:, 0].sum()) / t_max, "tokens_per_sec": float(allst[:, 1].sum()) / t_max,
            "t_max": t_max, "per_rank_frames_per_sec": (allst[:, 0] / allst[:, 2]).tolist(), "world": allst.shape[0]}


# ----------------------------------------------------------------------------- sample-sharded generation
def streaming_worker(rank: int, world: int, records: List[dict], model, processor, save_dir: str, simple_ctx: bool = False,
                     repetition_penalty: float = 1.05, temperature: float = 0.9, **stream_kw) -> List[int]:
    """One rank of the reference's LiveSports-3K-CC generation (eval/livesports3kcc/distributed_generate_streaming.py:44-121):
    record i goes to rank i mod world (:62), every finished record is one `{save_dir}/{i}.json` (resumable, :69-71), prompts are
    built as at :87-96.  `records` replaces the `datasets` download (needs network): dicts with video, video_id, event_id,
    begin, end, event_title, preasr_text.  Returns the indices this rank produced."""
    import json
    from .driver import streaming_inference
    os.makedirs(save_dir, exist_ok=True)
    done = []
    for idx in shard(list(range(len(records))), rank, world):
        save_path = os.path.join(save_dir, f"{idx}.json")
        if os.path.exists(save_path):
            continue
        r = records[idx]
        title, preasr = r.get("event_title"), r.get("preasr_text")
        if simple_ctx:
            title = "" if preasr else title
            prompt = f"{title}\n{preasr}".strip()
        else:
            prompt = ("You are an expert video commentator providing real-time, insightful, and engaging commentary on visual content.\n")
            if title:
                prompt += f"This is a video titled \"{title}\".\n"
        responses = streaming_inference(model=model, processor=processor, query=prompt, previous_text=preasr if preasr else "",
                                        video_path=r["video"], skip_first_chunk=r["begin"], duration=r["end"] - r["begin"],
                                        temperature=temperature, repetition_penalty=repetition_penalty, quiet=True, **stream_kw)
        with open(save_path, "w") as wf:
            json.dump({"video_id": r.get("video_id"), "event_id": r.get("event_id"), "begin": r["begin"], "end": r["end"],
                       "pred": "".join(x["response"] for x in responses)}, wf)
        done.append(idx)
    return done


def join_jsonl(save_dir: str) -> str:
    """jsons -> jsonl, as the reference does after its workers finish (:138-150)."""
    import json
    out = save_dir.rstrip("/") + ".jsonl"
    with open(out, "w") as wf:
        for name in sorted(os.listdir(save_dir), key=lambda n: int(n.split(".")[0]) if n.split(".")[0].isdigit() else 1 << 30):
            try:
                wf.write(json.dumps(json.load(open(os.path.join(save_dir, name)))) + "\n")
            except Exception:
                continue
    return out
