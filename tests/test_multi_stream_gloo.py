"""N > 1 path on CPU: world_size-2 `gloo`, one independent stream per rank (the reference's own data-parallel
inference pattern), barrier + one all_gather outside the timed region, whole-job aggregation."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import torch
    torch.set_num_threads(2)
    import streaming_vlm_amd as S
    from streaming_vlm_amd import config as C, multi_stream as MS
    from streaming_vlm_amd.weights import random_state_dict
    from ref_ops import RefOps
    dist, rank, world, _ = MS.init_distributed("gloo")
    cfg = C.tiny()
    model = S.StreamingQwen2VL(cfg, random_state_dict(cfg, 0, "cpu"), "cpu", ops=RefOps(), max_len=512, max_new_tokens=4, use_graph=False)
    streams = MS.shard(list(range(4)), rank, world)            # 4 streams over 2 ranks: [0, 2] and [1, 3]
    MS.fence(dist)
    t0 = time.perf_counter()
    frames = tokens = 0
    logs = {{}}
    for s in streams:
        log, counts = [], []
        S.streaming_inference(model=model, processor=S.SyntheticProcessor(), video_path=f"synthetic://56x56@1fps?stream={{s}}",
                              model_base="Qwen2", duration=3, kv_policy="sink_window", sink=4, window=64, do_sample=False,
                              max_new_tokens=4, suppress_eos=True, quiet=True, ids_log=log, token_counts=counts)
        frames += 3; tokens += sum(counts); logs[s] = [e["new"] for e in log]
    MS.fence(dist)
    dt = time.perf_counter() - t0
    agg = MS.aggregate(frames, tokens, dt, dist)
    json.dump({{"rank": rank, "streams": streams, "agg": agg, "dt": dt, "logs": logs}}, open(os.path.join({out!r}, f"r{{rank}}.json"), "w"))
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()
""")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_independent_streams(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=str(tmp_path)))
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    res = [json.load(open(tmp_path / f"r{r}.json")) for r in range(2)]
    assert res[0]["streams"] == [0, 2] and res[1]["streams"] == [1, 3]
    a0, a1 = res[0]["agg"], res[1]["agg"]
    assert a0["world"] == 2 and abs(a0["frames_per_sec"] - a1["frames_per_sec"]) < 1e-9       # every rank sees the same aggregate
    t_max = max(r["dt"] for r in res)
    assert abs(a0["frames_per_sec"] - 12 / t_max) < 1e-6 * 12 / t_max                          # all frames / slowest rank
    assert abs(a0["tokens_per_sec"] - 48 / t_max) < 1e-6 * 48 / t_max
    # different streams -> different frames -> (almost surely) different tokens; same stream id -> identical tokens across runs
    assert res[0]["logs"]["0"] != res[1]["logs"]["1"]


def test_bench_gpus_2_launches_itself_and_refuses_a_wrong_world_size():
    """`python bench.py --gpus 2` WITHOUT a launcher starts its two ranks itself (before any GPU call) and the collective sees both;
    a WORLD_SIZE that disagrees with --gpus is refused.  (--dry-run: rendezvous + aggregation only -- the hot path has no CPU form.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and len(line["per_gpu_frames_per_sec"]) == 2
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=dict(env, WORLD_SIZE="4", RANK="0"),
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=4" in bad.stderr
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"], env=env, capture_output=True, text=True, timeout=120)
    assert one.returncode == 0 and json.loads(one.stdout.strip().splitlines()[-1])["ranks_seen"] == 1


def test_single_process_aggregate_and_shard():
    sys.path.insert(0, ROOT)
    from streaming_vlm_amd import multi_stream as MS
    assert MS.shard(list(range(7)), 1, 3) == [1, 4]
    a = MS.aggregate(10, 200, 2.0, None)
    assert a == {"frames_per_sec": 5.0, "tokens_per_sec": 100.0, "t_max": 2.0, "per_rank_frames_per_sec": [5.0], "world": 1}


def test_sample_sharded_worker_is_resumable_and_joins_jsonl(tmp_path):
    """The LiveSports-3K-CC worker pattern (distributed_generate_streaming.py:44-150) on synthetic records: strided shards,
    one JSON per record, finished records are skipped on a re-run, jsons -> jsonl."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import streaming_vlm_amd as S
    from streaming_vlm_amd import config as C, multi_stream as MS
    from streaming_vlm_amd.weights import random_state_dict
    from ref_ops import RefOps
    cfg = C.tiny()
    model = S.StreamingQwen2VL(cfg, random_state_dict(cfg, 0, "cpu"), "cpu", ops=RefOps(), max_len=512, max_new_tokens=20, use_graph=False)
    recs = [dict(video=f"synthetic://56x56@1fps?stream={i}", video_id=f"v{i}", event_id=i, begin=i, end=i + 2,
                 event_title="t" if i % 2 else "", preasr_text="before" if i == 2 else "") for i in range(5)]
    save = str(tmp_path / "tiny")
    kw = dict(model_base="Qwen2", do_sample=False, max_new_tokens=4, suppress_eos=True, window_size=4, text_round=4)
    d0 = MS.streaming_worker(0, 2, recs, model, S.SyntheticProcessor(), save, **kw)
    d1 = MS.streaming_worker(1, 2, recs, model, S.SyntheticProcessor(), save, simple_ctx=True, **kw)
    assert d0 == [0, 2, 4] and d1 == [1, 3]
    assert MS.streaming_worker(0, 2, recs, model, S.SyntheticProcessor(), save, **kw) == []          # resumable: nothing left
    out = MS.join_jsonl(save)
    rows = [json.loads(l) for l in open(out)]
    assert [r["event_id"] for r in rows] == [0, 1, 2, 3, 4] and all(isinstance(r["pred"], str) and r["pred"] for r in rows)
    assert rows[3]["begin"] == 3 and rows[3]["end"] == 5
