"""GPU end-to-end parity: the streaming loop on HIP kernels vs the CPU oracle on the same synthetic
stream and the same seeded weights.

Bars (north_star: "logits within 1e-3 bf16, token ids exact under greedy, identical eviction indices")
  * eviction traces (every prune/move with its closed interval) and KV lengths: identical;
  * greedy token ids: EXACT, every step of every chunk, no waiver.  The weights are `decisive_state_dict` (random weights of
    the real shapes + one planted copy head, weights.py), whose post-penalty top-2 margin is asserted on the oracle to be
    >= 10x the logit noise measured in the same test -- with plain iid weights the top two of 152k logits are closer than
    the bf16 noise of a 28-layer forward in a fixed share of the steps, whatever the kernels do;
  * last-row logits of EVERY forward (the engine is teacher-forced with the oracle's tokens, so a flip cannot hide the
    forwards behind it): both sides round to bf16 at the same ~12 points per layer and flash attention rounds
    P = exp(s - running max) tile by tile, so any two valid tilings differ by single bf16 flips that propagate.
    (a) absolute anchor: against the oracle run in fp32 ("truth": same weights, no rounding points) the HIP path may not be
        further away than 1.25x what the bf16 oracle itself is, forward by forward (+1e-4 of the logit range);
    (b) against the bf16 oracle: within 2.5x of the oracle's own re-tiling noise (global max vs 32-key tiles)
        (mean|d|/max|ref| <= 2.5*floor + 5e-4, max <= 2.5*floor + 5e-3; DESIGN.md "Numerics").
"""
import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu


def _rel(a, b):
    """(max, mean) of |a - b| over max|b|."""
    d = (a - b).abs()
    sc = float(b.abs().max())
    return float(d.max()) / sc, float(d.mean()) / sc


def _compare(cfg, sd, n_chunks, model, floor=True, truth=True, exact_tokens=True, **kw):
    """`sd`: CPU state dict the model was built from.  Returns the measured numbers (printed as one [e2e] line)."""
    from oracle import model as om
    ref = H.run_oracle_stream(cfg, sd, n_chunks, keep_logits=True, **kw)
    force = ref["generated"]
    _, trace, counts, ids_log = H.run_engine_stream(model, n_chunks, keep_logits=True, force_tokens=force, **kw)
    assert trace == ref["trace"], f"eviction indices differ:\n{trace}\n{ref['trace']}"
    n_stream, n_chunks = n_chunks, len(force)          # answered chunks (a dense prefill piles the first ones into one answer)
    for i in range(n_chunks):
        assert ids_log[i]["kv_len"] == ref["kv_len"][i], (i, ids_log[i]["kv_len"], ref["kv_len"][i])
        assert len(ids_log[i]["logits"]) == len(ref["logits"][i]) == len(force[i])
    # ---- logits of every forward vs the bf16 oracle; absolute noise on the deciding logits
    worst_max = worst_mean = noise_abs = 0.0
    for i in range(n_chunks):
        for a, b in zip(ids_log[i]["logits"], ref["logits"][i]):
            mx, mn = _rel(a, b)
            worst_max, worst_mean = max(worst_max, mx), max(worst_mean, mn)
            noise_abs = max(noise_abs, float((a - b).abs().max()))
    # ---- greedy tokens
    margins = H.greedy_margins(ref, suppress_eos=kw.get("suppress_eos", True))
    own = [e["own"] for e in ids_log]
    n_steps = sum(len(f) for f in force)
    n_same = sum(int(a == b) for o, f in zip(own, force) for a, b in zip(o, f))
    if exact_tokens:
        assert min(margins) >= 10 * noise_abs, f"weights are not decisive: min top-2 margin {min(margins):.3e} vs logit noise {noise_abs:.3e}"
        assert own == force, f"greedy tokens differ: {n_same}/{n_steps} equal\n{own}\n{force}"
    else:           # iid weights: a token may only differ where the oracle's own margin is inside 2x the measured noise
        k = 0
        for o, f in zip(own, force):
            for a, b in zip(o, f):
                assert a == b or margins[k] <= 2 * noise_abs, (k, a, b, margins[k], noise_abs)
                k += 1
    msg = (f"[e2e] {n_steps} forwards, tokens {n_same}/{n_steps} equal, min margin {min(margins):.3e} vs logit noise {noise_abs:.3e}; "
           f"vs bf16 oracle: max rel {worst_max:.3e} mean rel {worst_mean:.3e}")
    out = dict(worst_max=worst_max, worst_mean=worst_mean, noise_abs=noise_abs, min_margin=min(margins), same=n_same, steps=n_steps)
    # ---- (a) fp32 truth
    if truth:
        sd32 = {k: v.float() for k, v in sd.items()}
        tru = H.run_oracle_stream(cfg, sd32, n_stream, keep_logits=True, force_tokens=force, **kw)
        assert tru["trace"] == ref["trace"]
        hip_sum = ora_sum = worst_ratio = 0.0
        for i in range(n_chunks):
            for a, b, t in zip(ids_log[i]["logits"], ref["logits"][i], tru["logits"][i]):
                eh, eo = _rel(a, t)[1], _rel(b, t)[1]
                hip_sum, ora_sum = hip_sum + eh, ora_sum + eo
                worst_ratio = max(worst_ratio, eh / (eo + 1e-4))
                assert eh <= 1.25 * eo + 1e-4, f"chunk {i}: HIP is {eh:.3e} from the fp32 truth, the bf16 oracle {eo:.3e}"
        msg += (f"; vs fp32 truth: HIP mean rel {hip_sum / n_steps:.3e}, bf16 oracle {ora_sum / n_steps:.3e}, worst per-forward ratio "
                f"{worst_ratio:.2f}, truth tokens {'==' if tru['own'] == force else '!='} oracle tokens")
        out.update(hip_truth=hip_sum / n_steps, oracle_truth=ora_sum / n_steps, worst_ratio=worst_ratio)
    # ---- (b) the oracle's own re-tiling noise
    if floor:
        om.ATTN_TILE = 32
        try:
            alt = H.run_oracle_stream(cfg, sd, n_stream, keep_logits=True, force_tokens=force, **kw)
        finally:
            om.ATTN_TILE = None
        fmax = fmean = 0.0
        for i in range(n_chunks):
            for a, b in zip(alt["logits"][i], ref["logits"][i]):
                mx, mn = _rel(a, b)
                fmax, fmean = max(fmax, mx), max(fmean, mn)
        msg += f"; oracle re-tiling floor max {fmax:.3e} mean {fmean:.3e}"
        print(msg)
        assert worst_mean <= 2.5 * fmean + 5e-4, (worst_mean, fmean)
        assert worst_max <= 2.5 * fmax + 5e-3, (worst_max, fmax)
    else:
        print(msg)
    return out


def _tiny_model(use_graph=True, family="qwen2", decisive=True, stream=None, **kw):
    """`stream`: the kwargs of the stream the weights must be decisive for (frame size, tokens per chunk, all_text)."""
    import streaming_vlm_amd as S
    from streaming_vlm_amd import config as C
    from streaming_vlm_amd.weights import random_state_dict
    cfg = C.tiny_2_5(**kw) if family == "qwen2_5" else C.tiny(**kw)
    sd = H.decisive_weights(cfg, **(stream or {})) if decisive else random_state_dict(cfg, 0, "cpu")
    model = S.StreamingQwen2VL(cfg, {k: v.cuda() for k, v in sd.items()}, "cuda", max_len=768, max_new_tokens=8, use_graph=use_graph)
    return cfg, sd, model


def test_tiny_stream_sink_window():
    cfg, sd, model = _tiny_model()
    _compare(cfg, sd, 6, model)


def test_tiny_stream_on_the_persistent_decode_tail():
    """The same stream with the decode step built on svlm_dec_tail (3 launches per layer): the bars of every other stream, graph
    replay included (the reset memset is a node of the step's graph)."""
    import streaming_vlm_amd as S
    cfg, sd, _ = _tiny_model()
    model = S.StreamingQwen2VL(cfg, {k: v.cuda() for k, v in sd.items()}, "cuda", max_len=768, max_new_tokens=8, decode_tail=True)
    assert model._svlm_engine.decode_tail
    _compare(cfg, sd, 6, model)
    eager = S.StreamingQwen2VL(cfg, {k: v.cuda() for k, v in sd.items()}, "cuda", max_len=768, max_new_tokens=8, decode_tail=True, use_graph=False)
    a = H.run_engine_stream(model, 4, keep_logits=True)[3]
    b = H.run_engine_stream(eager, 4, keep_logits=True)[3]
    for x, y in zip(a, b):
        assert x["ids"] == y["ids"] and all(torch.equal(u, v) for u, v in zip(x["logits"], y["logits"]))


def test_tiny_qwen2_5_stream_sink_window_ragged_windows():
    """Qwen2.5-VL family: windowed RMSNorm/SwiGLU tower (112x84 frames -> ragged attention windows), float temporal M-RoPE."""
    cfg, sd, model = _tiny_model(family="qwen2_5", stream=dict(size=(112, 84)))
    _compare(cfg, sd, 6, model, size=(112, 84), window=96)


def test_tiny_qwen2_5_stream_structural_and_all_text():
    cfg, sd, model = _tiny_model(family="qwen2_5")
    _compare(cfg, sd, 7, model, policy="structural", text_round=2, window_size=3, text_sink=4, text_sliding_window=8,
             previous_text="a b c d e f g h i j k l m n o p")
    cfg, sd, model = _tiny_model(family="qwen2_5", stream=dict(all_text=True))
    _compare(cfg, sd, 4, model, all_text=True)


def test_tiny_streams_append_mode():
    """pos_mode="append" on both families: un-rotated keys + per-row position history vs the oracle's rotated-K cache."""
    for fam in ("qwen2", "qwen2_5"):
        cfg, sd, model = _tiny_model(family=fam)
        _compare(cfg, sd, 7, model, policy="structural", text_round=2, window_size=3, text_sink=4, text_sliding_window=8,
                 previous_text="a b c d e f g h i j k l m n o p", pos_mode="append")
    cfg, sd, model = _tiny_model()
    _compare(cfg, sd, 6, model, pos_mode="append")


def test_tiny_stream_iid_weights_teacher_forced():
    """Plain N(0, 0.02) weights (no planted head): near-ties between the top two logits are part of the data, so the engine is
    teacher-forced with the oracle's tokens and every forward's logits are held to the same bars; a token may differ only
    where the oracle's own margin is inside twice the measured logit noise."""
    cfg, sd, model = _tiny_model(decisive=False)
    out = _compare(cfg, sd, 8, model, exact_tokens=False)
    assert out["same"] >= 0.9 * out["steps"], out


def test_dense_prefill_then_live_stream():
    """BASELINE configs[4] shape: 6 chunks of frames piled into ONE forward (ViT in passes of 2 grids, LLM prefill in passes of 48 rows,
    both forced small here), then live chunks under sink/window eviction -- same bars as every other stream, bf16 and fp8-free."""
    cfg, sd, model = _tiny_model()
    eng = model._svlm_engine
    eng.PREFILL_ROWS, eng.VIT_BATCH_SEQS = 48, 2
    out = _compare(cfg, sd, 10, model, window=96, dense_prefill_chunks=6)
    assert out["steps"] == 5 * 8


def test_dense_prefill_one_long_pass_runs_two_query_blocks():
    """A dense prefill whose prompt (13 chunks of 224 x 224 frames: ~1.1k rows) goes through in ONE pass: the prefill attention then runs
    two query blocks per wave on 64-key super tiles (T >= 1024, no key splits) -- the configuration of the 4096-row passes of configs[4]
    -- inside the engine, held to the same bars; live chunks follow under sink/window eviction."""
    import streaming_vlm_amd as S
    from streaming_vlm_amd import config as C
    cfg = C.tiny()
    sd = H.decisive_weights(cfg, size=224)
    model = S.StreamingQwen2VL(cfg, {k: v.cuda() for k, v in sd.items()}, "cuda", max_len=2048, max_new_tokens=8)
    out = _compare(cfg, sd, 15, model, size=224, window=512, dense_prefill_chunks=13)
    assert out["steps"] == 3 * 8
    assert model._svlm_engine._last_cache.get_seq_length() <= 4 + 512 + 100


def test_tiny_stream_structural():
    cfg, sd, model = _tiny_model()
    _compare(cfg, sd, 8, model, policy="structural", text_round=2, window_size=3, text_sink=4, text_sliding_window=8,
             previous_text="a b c d e f g h i j k l m n o p")


def test_graph_replay_equals_eager_launches():
    """The captured decode-step graph must reproduce eager launches bit for bit."""
    outs = []
    for g in (True, False):
        cfg, sd, model = _tiny_model(use_graph=g)
        _, trace, counts, ids_log = H.run_engine_stream(model, 4, keep_logits=True)
        outs.append(ids_log)
    for a, b in zip(*outs):
        assert a["ids"] == b["ids"]
        for x, y in zip(a["logits"], b["logits"]):
            assert torch.equal(x, y)


@pytest.mark.parametrize("family", ["qwen2", "qwen2_5"])
def test_decode_steps_on_the_prefills_rotated_keys_equal_rotation_at_every_step(family):
    """The engine's decode attention streams the rotated keys the chunk's prefill left in the cache's linear planes and takes only the
    rows decoded since from the pool (kv_pool.py); an engine without the planes rotates every pool row at every step, as the reference
    does (language_forward.py:55-63).  Same bits: ids and every forward's logits, over chunks with eviction in between."""
    import streaming_vlm_amd as S
    cfg, sd, _ = _tiny_model(family=family)
    logs = []
    for lp in (True, False):
        model = S.StreamingQwen2VL(cfg, {k: v.cuda() for k, v in sd.items()}, "cuda", max_len=768, max_new_tokens=8, linear_planes=lp)
        assert model._svlm_engine.linear_planes is lp and (model._svlm_engine.new_cache().lin is not None) is lp
        logs.append(H.run_engine_stream(model, 6, keep_logits=True)[3])
    for a, b in zip(*logs):
        assert a["ids"] == b["ids"]
        for x, y in zip(a["logits"], b["logits"]):
            assert torch.equal(x, y)


def test_two_streams_alternating_on_one_engine_keep_their_graphs():
    """Two KV pools taking turns on one engine (two streams sharing a model replica): each pool's decode graph is captured once and
    found again by the pool's serial -- the engine holds the pool while it holds the graph -- and both streams produce what they
    produce alone."""
    import streaming_vlm_amd as S
    cfg, sd, model = _tiny_model()
    eng = model._svlm_engine
    alone = []
    for stream in (0, 1):
        c2, s2, m2 = _tiny_model()
        log = []
        S.streaming_inference(model=m2, processor=S.SyntheticProcessor(), video_path=f"synthetic://56x56@1fps?stream={stream}", model_base="Qwen2",
                              duration=4, kv_policy="sink_window", sink=4, window=64, do_sample=False, max_new_tokens=8, suppress_eos=True,
                              quiet=True, ids_log=log, vision_lookahead=False)
        alone.append([e["new"] for e in log])
    proc = S.SyntheticProcessor()
    vids = [S.SyntheticVideo(56, 1.0, s) for s in (0, 1)]
    srcs = [H.chunk_source(proc, v, "") for v in vids]
    caches, hist, got, grids = [None, None], [None, None], [[], []], [[], []]
    for i in range(4):
        for s in (0, 1):
            ids, pix, grid = srcs[s](i)
            if hist[s] is not None:
                ids = hist[s] + (ids[1:] if hist[s][-1] == 198 else ids)
            grids[s] = grids[s] + [list(g) for g in grid]
            out = eng.generate(ids, caches[s], grids[s], pix.cuda(), grid, max_new_tokens=8, repetition_penalty=1.05, suppress_eos=True)
            caches[s] = out.past_key_values
            seq = out.sequences + ([151645] if out.sequences[-1] != 151645 else [])
            got[s].append(seq[len(ids):])
            hist[s] = seq
    assert len(eng._graphs) == 2, list(eng._graphs)                     # one capture per pool, none dropped or redone
    assert {k[0] for k in eng._graphs} == {caches[0].serial, caches[1].serial}
    # (the hand-driven pair never evicts: compare the chunks that come before the driver's first eviction)
    for s in (0, 1):
        assert got[s][:2] == alone[s][:2], (s, got[s][:2], alone[s][:2])


def test_golden_streams_eviction_trace_and_tokens():
    """Committed oracle streams (tests/golden/oracle_streams.json, minted in the build container by oracle/make_golden.py
    with the decisive weights) replayed on the HIP engine with NO oracle in the loop: eviction indices, KV lengths and EVERY
    greedy token of every chunk must be identical (cfg0: 32 chunks of 224x224, sink 4 / window 256)."""
    import json, os
    with open(os.path.join(os.path.dirname(__file__), "golden", "oracle_streams.json")) as f:
        gold = json.load(f)
    for name, g in gold.items():
        kw = dict(g["kwargs"])
        assert g["min_margin"] >= 0.5, (name, g["min_margin"])          # minted with decisive weights
        cfg, sd, model = _tiny_model(family="qwen2_5" if g.get("model") == "tiny_2_5" else "qwen2",
                                     stream=dict(size=kw.get("size", 56), all_text=kw.get("all_text", False)))
        _, trace, counts, ids_log = H.run_engine_stream(model, g["n_chunks"], **kw)
        assert [[list(t) for t in c] for c in trace] == g["trace"], name
        assert [e["kv_len"] for e in ids_log] == g["kv_len"], name
        got = [e["new"] for e in ids_log]
        same = sum(int(x == y) for x, y in zip(got, g["new_tokens"]))
        print(f"[golden] {name}: {same}/{g['n_chunks']} chunks token-identical")
        assert got == g["new_tokens"], (name, same)


def test_golden_streams_with_the_device_side_eviction_plan():
    """the same committed streams with the spans and eviction indices computed ON THE DEVICE (svlm_evict_plan: the
    reference's get_qwen_range + process_past_kv / the sink-window cut as one kernel over the device copy of the ids) -- every
    prune / move interval, KV length and token must again equal the oracle's."""
    import json, os
    with open(os.path.join(os.path.dirname(__file__), "golden", "oracle_streams.json")) as f:
        gold = json.load(f)
    n_ops = 0
    for name, g in gold.items():
        kw = dict(g["kwargs"])
        cfg, sd, model = _tiny_model(family="qwen2_5" if g.get("model") == "tiny_2_5" else "qwen2",
                                     stream=dict(size=kw.get("size", 56), all_text=kw.get("all_text", False)))
        _, trace, counts, ids_log = H.run_engine_stream(model, g["n_chunks"], device_policy=True, **kw)
        assert [[list(t) for t in c] for c in trace] == g["trace"], name
        assert [e["kv_len"] for e in ids_log] == g["kv_len"], name
        assert [e["new"] for e in ids_log] == g["new_tokens"], name
        n_ops += sum(len(c) for c in trace)
    print(f"[device policy] {len(gold)} streams, {n_ops} device-computed prune / move intervals identical to the oracle's")
    assert n_ops > 100


def test_evict_plan_kernel_edits_the_ids_like_the_host():
    """svlm_evict_plan as a pure function: op list AND edited ids equal the host policy's on a hand-built history, incl. the
    move of the assistant text, a vision span at the sink/window cut, and the no-op cases."""
    import numpy as np
    import streaming_vlm_amd as S
    from streaming_vlm_amd.ops import HipOps
    from streaming_vlm_amd.driver import process_past_kv, sink_window_evict
    o = HipOps()
    IM_S, IM_E, U, A, VS, VE, VP, LF, TIME = 151644, 151645, 872, 77091, 151652, 151653, 151656, 198, 1462
    sys_ = [IM_S, 8948, LF, 2610, 525, 264, 10950, 17847, 13, IM_E, LF]
    prev = [IM_S, 19702, 1467, LF] + list(range(300, 330)) + [IM_E, LF]
    def user(n, q=()): return [IM_S, U, LF, TIME, 28, 15, 13, 15, 82, VS] + [VP] * n + [VE] + list(q) + [IM_E, LF]
    def asst(t): return [IM_S, A, LF] + list(t) + [2503, IM_E, LF]
    ids = sys_ + prev
    for r in range(5):
        ids = ids + user(6, q=[900, 901] if r == 0 else ()) + asst([40 + r, 50 + r, 60 + r])
    ids = ids[:-1]                                           # the history ends with <|im_end|>
    for rnd, tr, vr, ts, tsw in [(5, 2, 3, 4, 8), (5, 3, 2, 2, 6), (5, 4, 4, None, None), (1, 4, 4, 4, 8), (0, 2, 2, 4, 8), (5, 2, 2, None, 10)]:
        t = torch.tensor([ids])
        hist = [{"role": "previous text", "content": "x"}] + [{"role": "user" if k % 2 == 0 else "assistant", "content": [{"type": "text", "text": "t"}, {"type": "video"}] if k % 2 == 0 else "abcd ..."} for k in range(10)]
        trace = []
        _, want_ids, _, _ = process_past_kv(None, rnd, tr, vr, hist, t, 3, 2, [0] * 8, [0] * 8, ts, tsw, trace)
        ops, got_ids = o.evict_plan(ids, "structural", rnd, tr, vr, ts, tsw, 3, 2)
        assert [tuple(x) for x in ops] == [tuple(x) for x in trace], (rnd, tr, vr, ops, trace)
        assert np.array_equal(got_ids, want_ids[0].numpy()), (rnd, tr, vr)
    for sink, window in [(4, 40), (4, 37), (4, 200), (0, 10), (4, 33)]:
        trace = []
        class KV:          # the host function only asks for the length
            def __init__(self, n): self.n = n
            def get_seq_length(self): return self.n
            def release_reserved(self): pass
            def prune(self, s, e): self.n -= e - s + 1
        kv = KV(len(ids) - 1)
        _, want_ids = sink_window_evict(kv, torch.tensor([ids]), sink, window, trace)
        ops, got_ids = o.evict_plan(ids, "sink_window", sink=sink, window=window, kv_len=len(ids) - 1)
        assert [tuple(x) for x in ops] == [tuple(x) for x in trace], (sink, window, ops, trace)
        assert np.array_equal(got_ids, want_ids[0].numpy()), (sink, window)


def test_golden_full_size_streams_tokens_exact():
    """Full-size models against token streams minted by the CPU oracle in the build container (tests/golden/
    full_size_streams.json): BASELINE configs[1] -- Qwen2-VL-2B, 448x448 @1 fps, sink 4 / window 2048, 20 tokens per chunk,
    12 chunks so that the window is full and evicts --, configs[2]'s model at its frame rate (Qwen2-VL-7B, 448x448 @2 fps, window 512
    so that 3 chunks evict), configs[2] AT SPEC (window 4096, 18 chunks: the window fills at chunk 14 and the last four evict) and
    configs[4] at the 7B's widths, mid-size (10 chunks = 20 frames = 2.8k rows piled into one forward -- three prefill passes of
    1024 rows, the ViT in passes of 8 + 2 grids --, then three live chunks, the first of which compacts the cache; bf16 and fp8
    tower).  Every eviction index and every greedy token must be identical; the logit behind each token must sit within 5 % of
    the oracle's margin of the oracle's value."""
    import json, os
    import streaming_vlm_amd as S
    from streaming_vlm_amd import config as C
    with open(os.path.join(os.path.dirname(__file__), "golden", "full_size_streams.json")) as f:
        gold = json.load(f)
    for name, g in gold.items():
        kw = dict(g["kwargs"])
        cfg = {"2b": C.qwen2_vl_2b, "7b": C.qwen2_vl_7b}[g["model"]]()
        sd = H.decisive_weights(cfg, size=kw["size"], max_new=kw["max_new"])
        model = S.StreamingQwen2VL(cfg, {k: v.cuda() for k, v in sd.items()}, "cuda", max_len=g["max_len"], max_new_tokens=kw["max_new"],
                                   vit_fp8=bool(g.get("vit_fp8", False)))
        if g.get("prefill_rows") or "dense_prefill_chunks" in kw:       # mid-size dense prefill: several prefill passes over the 2.8k-row prompt
            model._svlm_engine.PREFILL_ROWS = int(g.get("prefill_rows") or 1024)
        del sd
        margins = g.get("margins")
        # a stream minted with per-step margins (the dense prefill: a few near-ties in its first answered turn) is replayed
        # teacher-forced with the oracle's tokens, so that a near-tie cannot send the rest of the stream elsewhere; the engine's OWN
        # argmax is what is compared
        # ("new_tokens" carries the <|im_end|> the loop appends behind a turn that ran to max_new_tokens; the generated ones come first)
        force = [c[:kw["max_new"]] for c in g["new_tokens"]] if margins else None
        _, trace, counts, ids_log = H.run_engine_stream(model, g["n_chunks"], keep_logits=True, force_tokens=force, **kw)
        assert [[list(t) for t in c] for c in trace] == g["trace"], name
        assert [e["kv_len"] for e in ids_log] == g["kv_len"], name
        got = [e["own"] if margins else e["new"] for e in ids_log]
        same = sum(int(x == y) for x, y in zip(got, g["new_tokens"]))
        worst = 0.0
        for e, tops, toks in zip(ids_log, g["top_logit"], g["new_tokens"]):
            for lg, top, tok in zip(e["logits"], tops, toks):
                worst = max(worst, abs(float(lg[tok]) - top))
        if margins:
            flat_got = [t for c in got for t in c]
            flat_want = [t for c in force for t in c]
            assert len(flat_got) == len(flat_want) == len(margins), name
            decisive = [m for m in margins if m >= 1.0]
            n_dec = sum(int(a == b) for a, b, m in zip(flat_got, flat_want, margins) if m >= 1.0)
            print(f"[golden-full] {name}: {n_dec}/{len(decisive)} decisive steps token-identical ({len(margins) - len(decisive)} near-ties skipped), "
                  f"|top logit - oracle| <= {worst:.3e}, smallest decisive margin {min(decisive):.3f}")
            assert len(decisive) >= 0.8 * len(margins), (name, len(decisive), len(margins))
            assert n_dec == len(decisive), (name, n_dec, len(decisive))
            assert worst <= 0.05 * min(decisive), (name, worst, min(decisive))
        else:
            print(f"[golden-full] {name}: {same}/{g['n_chunks']} chunks token-identical, kv_len max {max(g['kv_len'])}, "
                  f"|top logit - oracle| <= {worst:.3e}, oracle min margin {g['min_margin']:.3f}")
            assert got == g["new_tokens"], (name, same)
            assert worst <= 0.05 * g["min_margin"], (name, worst, g["min_margin"])
        del model
        torch.cuda.empty_cache()


def test_vision_lookahead_is_bitwise_neutral():
    """Encoding chunk i+1's frames on the side stream under chunk i's decode steps must not change a single bit."""
    outs = []
    for ahead in (True, False):
        cfg, sd, model = _tiny_model()
        _, trace, counts, ids_log = H.run_engine_stream(model, 5, keep_logits=True, vision_lookahead=ahead)
        outs.append((trace, ids_log))
    assert outs[0][0] == outs[1][0]
    for a, b in zip(outs[0][1], outs[1][1]):
        assert a["ids"] == b["ids"]
        for x, y in zip(a["logits"], b["logits"]):
            assert torch.equal(x, y)


def test_device_frame_ingest_equals_host_processor():
    """uint8 frames -> H2D -> svlm_patchify_u8 gives the same stream, bit for bit, as host patchify + upload."""
    import streaming_vlm_amd as S
    outs = []
    for dev_ingest in (False, True):
        cfg, sd, model = _tiny_model()
        proc = S.DeviceFrameProcessor(model._svlm_engine.ops) if dev_ingest else S.SyntheticProcessor()
        video = S.PinnedVideo(5, 56, 1.0, 0) if dev_ingest else None
        ids_log = []
        S.streaming_inference(model=model, processor=proc, video=video, video_path="synthetic://56x56@1fps", model_base="Qwen2",
                              duration=5, previous_text="hello world", kv_policy="sink_window", sink=4, window=64, do_sample=False,
                              max_new_tokens=8, suppress_eos=True, quiet=True, ids_log=ids_log, keep_logits=True)
        outs.append(ids_log)
    for a, b in zip(*outs):
        assert a["ids"] == b["ids"]
        for x, y in zip(a["logits"], b["logits"]):
            assert torch.equal(x, y)


def test_eos_truncates_and_rolls_back_kv():
    import streaming_vlm_amd as S
    cfg, sd, model = _tiny_model()
    _, trace, counts, ids_log = H.run_engine_stream(model, 3, suppress_eos=False)
    # KV always trails the ids by the not-yet-forwarded suffix (>= 1 token)
    for e, n in zip(ids_log, counts):
        assert len(e["ids"]) - e["kv_len"] in (1, 2)


def test_real_shape_2b_layer_stack_224():
    """BASELINE configs[0] geometry (Qwen2-VL-2B, 224x224 -> 64 vision tokens), 2 chunks, full parity.
    4 LLM layers / 4 ViT blocks of the REAL widths keep the CPU oracle within seconds."""
    import streaming_vlm_amd as S
    from streaming_vlm_amd import config as C
    from streaming_vlm_amd.weights import random_state_dict
    cfg = C.qwen2_vl_2b()
    cfg.vision.depth = 4
    cfg.text.num_layers = 4
    sd = H.decisive_weights(cfg, size=224)
    model = S.StreamingQwen2VL(cfg, {k: v.cuda() for k, v in sd.items()}, "cuda", max_len=1024, max_new_tokens=8)
    _compare(cfg, sd, 3, model, size=224, window=128)


def test_full_size_2b_two_chunks_448():
    """BASELINE configs[1] at FULL size: Qwen2-VL-2B (28 LLM layers, 32 ViT blocks), 448x448 frames (1024 patches ->
    256 vision tokens), 2 chunks x 4 tokens, sink 4 / window 256 so that the second chunk evicts.  Same bars as the
    tiny model: identical eviction trace, exact greedy tokens, logits anchored on the fp32 truth."""
    import streaming_vlm_amd as S
    from streaming_vlm_amd import config as C
    from streaming_vlm_amd.weights import random_state_dict
    torch.set_num_threads(min(16, torch.get_num_threads()))
    cfg = C.qwen2_vl_2b()
    sd = H.decisive_weights(cfg, size=448, max_new=4)
    model = S.StreamingQwen2VL(cfg, {k: v.cuda() for k, v in sd.items()}, "cuda", max_len=1024, max_new_tokens=8)
    _compare(cfg, sd, 2, model, floor=False, size=448, window=256, max_new=4)


def test_full_size_2b_two_chunks_448_on_the_persistent_decode_tail():
    """Full-size Qwen2-VL-2B with the decode step on svlm_dec_tail: exact tokens, logits anchored on the fp32 truth."""
    import streaming_vlm_amd as S
    from streaming_vlm_amd import config as C
    torch.set_num_threads(min(16, torch.get_num_threads()))
    cfg = C.qwen2_vl_2b()
    sd = H.decisive_weights(cfg, size=448, max_new=4)
    model = S.StreamingQwen2VL(cfg, {k: v.cuda() for k, v in sd.items()}, "cuda", max_len=1024, max_new_tokens=8, decode_tail=True)
    _compare(cfg, sd, 2, model, floor=False, size=448, window=256, max_new=4)


def test_full_size_7b_two_chunks_448():
    """BASELINE configs[2]'s model at FULL size: Qwen2-VL-7B (28 layers of 3584 / 18944, 28 query / 4 kv heads, separate
    lm_head), 448x448 frames at 2 fps (one 2-frame temporal patch per chunk), 2 chunks x 4 tokens, sink 4 / window 256 so that
    the second chunk evicts.  Same bars: identical eviction trace, exact greedy tokens, logits anchored on the fp32 truth."""
    import streaming_vlm_amd as S
    from streaming_vlm_amd import config as C
    torch.set_num_threads(min(16, torch.get_num_threads()))
    cfg = C.qwen2_vl_7b()
    sd = H.decisive_weights(cfg, size=448, max_new=4)
    model = S.StreamingQwen2VL(cfg, {k: v.cuda() for k, v in sd.items()}, "cuda", max_len=1024, max_new_tokens=8)
    _compare(cfg, sd, 2, model, floor=False, size=448, fps=2.0, window=256, max_new=4)


def test_full_size_qwen2_5_vl_3b_two_chunks_448():
    """Qwen2.5-VL-3B at FULL size (36 LLM layers, 32 windowed ViT blocks, intermediate 3420 padded to 3424), 448x448 frames, two
    chunks with an eviction in between: same bars as the Qwen2-VL-2B full-size test."""
    import streaming_vlm_amd as S
    from streaming_vlm_amd import config as C
    from streaming_vlm_amd.weights import random_state_dict
    torch.set_num_threads(min(16, torch.get_num_threads()))
    cfg = C.qwen2_5_vl_3b()
    sd = H.decisive_weights(cfg, size=448, max_new=4)
    model = S.StreamingQwen2VL(cfg, {k: v.cuda() for k, v in sd.items()}, "cuda", max_len=1024, max_new_tokens=8)
    _compare(cfg, sd, 2, model, floor=False, size=448, window=256, max_new=4)


def test_tight_pool_defragments_in_place_and_stays_exact():
    """A KV pool with 5 % head-room under the structural policy (many 1-3 row prunes and moves) runs out of whole pages:
    svlm_kv_move_rows packs the sparse ones in place, and the stream still matches the oracle."""
    import streaming_vlm_amd as S
    from streaming_vlm_amd import config as C
    from streaming_vlm_amd.weights import random_state_dict
    cfg = C.tiny()
    sd = H.decisive_weights(cfg)
    model = S.StreamingQwen2VL(cfg, {k: v.cuda() for k, v in sd.items()}, "cuda", max_len=200, max_new_tokens=8, kv_slack=0.05,
                               kv_page_tokens=16)
    _compare(cfg, sd, 12, model, policy="structural", text_round=2, window_size=3, text_sink=4, text_sliding_window=8,
             previous_text="a b c d e f g h i j k l m n o p")
    stats = model._svlm_engine._last_cache.stats
    print("[defrag]", stats)
    assert stats["defrags"] >= 1 and stats["moved_rows"] > 0


def test_sampling_recompute_and_teacher_forcing_run_on_the_device(tmp_path):
    """The remaining switches of streaming_inference on the HIP path: do_sample (seeded multinomial), recompute (mode c of the
    efficiency harness) and gt_json teacher forcing (inference.py:483-487)."""
    import json
    import streaming_vlm_amd as S
    kw = dict(processor=S.SyntheticProcessor(), video_path="synthetic://56x56@1fps", model_base="Qwen2", duration=4, quiet=True,
              max_new_tokens=6, window_size=2, text_round=2)
    outs = []
    for _ in range(2):
        cfg, sd, model = _tiny_model()
        g = torch.Generator(device="cuda").manual_seed(5)
        log = []
        S.streaming_inference(model=model, do_sample=True, temperature=0.9, generator=g, ids_log=log, **kw)
        outs.append([e["new"] for e in log])
    assert outs[0] == outs[1]
    # the captured decode graph draws fresh noise at every step and every call: no token pattern repeats across the chunks
    assert len({tuple(t) for t in outs[0]}) == len(outs[0])
    # top_k = 1 (what stock Qwen2-VL checkpoints ship in their generation_config) is the greedy stream, token for token
    cfg, sd, model = _tiny_model()
    ga, gb = [], []
    S.streaming_inference(model=model, do_sample=True, temperature=0.9, top_k=1, suppress_eos=True, ids_log=ga, **kw)
    cfg, sd, model = _tiny_model()
    S.streaming_inference(model=model, do_sample=False, suppress_eos=True, ids_log=gb, **kw)
    assert [e["new"] for e in ga] == [e["new"] for e in gb]
    cfg, sd, model = _tiny_model()
    a, b = [], []
    S.streaming_inference(model=model, do_sample=False, suppress_eos=True, recompute=True, ids_log=a, **kw)
    cfg, sd, model = _tiny_model()
    S.streaming_inference(model=model, do_sample=False, suppress_eos=True, ids_log=b, **kw)
    assert [e["new"] for e in a][0] == [e["new"] for e in b][0]          # first chunk identical; later ones agree up to bf16 noise
    gt = tmp_path / "gt.jsonl"
    gt.write_text(json.dumps({f"Time={i}.0-{i + 1}.0s": {"phrase": "x y"} for i in range(4)}) + "\n")
    cfg, sd, model = _tiny_model()
    res = S.streaming_inference(model=model, do_sample=False, suppress_eos=True, gt_json=str(gt), gt_idx=0, **kw)
    assert len(res) == 4


def test_tiny_stream_with_the_7b_head_grouping():
    """7 query heads per kv head (Qwen2-VL-7B / Qwen2.5-VL-7B: 28 / 4) end to end: the MFMA decode attention pads the group
    to 16 columns, the fused QKV kernel appends 1 kv head."""
    import streaming_vlm_amd as S
    from streaming_vlm_amd import config as C
    from streaming_vlm_amd.weights import random_state_dict
    cfg = C.tiny()
    cfg.text.num_heads, cfg.text.num_kv_heads = 7, 1
    sd = H.decisive_weights(cfg)
    model = S.StreamingQwen2VL(cfg, {k: v.cuda() for k, v in sd.items()}, "cuda", max_len=768, max_new_tokens=8)
    _compare(cfg, sd, 5, model)


def test_native_size_frames_are_resized_on_the_device():
    """A source of native-size frames (`synthetic-raw://`) goes through the reference's per-chunk _spatial_resize_video
    (inference.py:342) on the GPU: the stream equals, bit for bit, the stream over the ORACLE-resized frames."""
    import streaming_vlm_amd as S
    from streaming_vlm_amd import ingest
    from oracle import resize as R

    class Presized:
        spatial_resize = False

        def __init__(self, raw):
            self.raw = raw

        def chunk(self, start_s, duration_s):
            f = self.raw.chunk(start_s, duration_s)
            h, w = ingest.resized_shape(f.shape[2], f.shape[3], f.shape[0])
            return torch.from_numpy(R.resize_bicubic_aa_u8(f.numpy(), h, w))

    path = "synthetic-raw://330x250@1fps"
    assert ingest.resized_shape(250, 330, 1) == (252, 336)
    outs = []
    for presized in (False, True):
        cfg, sd, model = _tiny_model()
        proc = S.DeviceFrameProcessor(model._svlm_engine.ops)
        raw = S.SyntheticVideo.from_path(path)
        ids_log = []
        S.streaming_inference(model=model, processor=proc, video=Presized(raw) if presized else raw, video_path=path, model_base="Qwen2",
                              duration=3, previous_text="hello world", kv_policy="sink_window", sink=4, window=160, do_sample=False,
                              max_new_tokens=8, suppress_eos=True, quiet=True, ids_log=ids_log, keep_logits=True)
        outs.append(ids_log)
    assert len(outs[0]) == 3
    for a, b in zip(*outs):
        assert a["ids"] == b["ids"] and a["ids"].count(151656) > 0
        for x, y in zip(a["logits"], b["logits"]):
            assert torch.equal(x, y)
    n_pad = outs[0][0]["ids"].count(151656)
    assert n_pad % ((252 // 28) * (336 // 28)) == 0, n_pad           # 108 video tokens per chunk: the resized grid
