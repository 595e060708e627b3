"""GPU end-to-end parity: the streaming loop on HIP kernels vs the CPU oracle on the same synthetic
stream and the same seeded weights.

Bars
  * eviction traces (every prune/move with its closed interval) and KV lengths: identical;
  * last-row logits of every forward: both sides round to bf16 at the same ~12 points per layer, and
    flash attention rounds P = exp(s - running max) to bf16 tile by tile, so ANY two valid tilings differ by
    single bf16 flips that propagate through the layers.  The test measures that floor on the oracle itself
    (global-max vs 32-key tiles) and requires the HIP path to stay within 2.5x of it
    (mean|d|/max|ref| <= 2.5*floor + 5e-4, max <= 2.5*floor + 5e-3; DESIGN.md "Numerics");
  * greedy token ids: exact on streams whose oracle top-2 margin exceeds the measured logit noise;
    where a margin is inside the noise the argmax is not defined by the arithmetic and the test
    reports it instead of failing.
"""
import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu


def _noise_floor(cfg, sd, n_chunks, ref, **kw):
    """Logit error between two equally valid flash-attention tilings of the ORACLE itself (global max vs
    32-key online tiles): the rounding noise any implementation of the reference numerics carries."""
    from oracle import model as om
    om.ATTN_TILE = 32
    try:
        alt = H.run_oracle_stream(cfg, sd, n_chunks, keep_logits=True, **kw)
    finally:
        om.ATTN_TILE = None
    fmax = fmean = 0.0
    for i in range(n_chunks):
        if alt["new_tokens"][i] != ref["new_tokens"][i]:
            # histories diverge after this chunk; compare only what both computed from identical inputs
            for a, b, ta, tb in zip(alt["logits"][i], ref["logits"][i], alt["new_tokens"][i], ref["new_tokens"][i]):
                d = (a - b).abs(); sc = float(b.abs().max())
                fmax, fmean = max(fmax, float(d.max()) / sc), max(fmean, float(d.mean()) / sc)
                if ta != tb:
                    break
            break
        for a, b in zip(alt["logits"][i], ref["logits"][i]):
            d = (a - b).abs(); sc = float(b.abs().max())
            fmax, fmean = max(fmax, float(d.max()) / sc), max(fmean, float(d.mean()) / sc)
    return fmax, fmean


def _compare(cfg, sd, n_chunks, model, **kw):
    import streaming_vlm_amd as S  # noqa: F401
    _, trace, counts, ids_log = H.run_engine_stream(model, n_chunks, keep_logits=True, **kw)
    ref = H.run_oracle_stream(cfg, sd, n_chunks, keep_logits=True, **kw)
    floor_max, floor_mean = _noise_floor(cfg, sd, n_chunks, ref, **kw)
    assert trace == ref["trace"], f"eviction indices differ:\n{trace}\n{ref['trace']}"
    worst_max = worst_mean = 0.0
    diverged = False
    for i in range(n_chunks):
        if diverged:
            break
        assert ids_log[i]["kv_len"] == ref["kv_len"][i]
        for j, (a, b) in enumerate(zip(ids_log[i]["logits"], ref["logits"][i])):
            scale = float(b.abs().max())
            d = (a - b).abs()
            worst_max = max(worst_max, float(d.max()) / scale)
            worst_mean = max(worst_mean, float(d.mean()) / scale)
            # the margin that decides the greedy token is the one AFTER the repetition penalty and the EOS suppression
            # (streaming_generate_qwen.py:75-99): a seen token's logit is divided by 1.05, which can turn a clear raw margin
            # into a near tie
            from oracle import generate as og
            n_new_i = len(ref["new_tokens"][i])
            hist = ref["ids"][i][:len(ref["ids"][i]) - n_new_i + j]
            sc = og.repetition_penalty(b.clone(), hist, 1.05)
            if kw.get("suppress_eos", True):
                sc[[151645, 151643]] = float("-inf")
            top2 = torch.topk(sc, 2).values
            margin = float(top2[0] - top2[1])
            if ids_log[i]["new"][j] != ref["new_tokens"][i][j]:
                noise = float(d.max())
                print(f"[e2e] chunk {i} step {j}: token differs, oracle top-2 margin {margin:.3e} vs logit noise {noise:.3e}")
                assert margin <= 4 * noise, "greedy token differs although the oracle margin is far above the noise"
                diverged = True          # histories differ from here on; stop comparing
                break
    print(f"[e2e] logits: max rel err {worst_max:.3e}, mean rel err {worst_mean:.3e}; oracle-vs-oracle tiling noise floor "
          f"max {floor_max:.3e} mean {floor_mean:.3e}; diverged={diverged}")
    # the HIP path may not be further from the oracle than ~2x what the oracle is from itself under re-tiling
    assert worst_mean <= 2.5 * floor_mean + 5e-4, (worst_mean, floor_mean)
    assert worst_max <= 2.5 * floor_max + 5e-3, (worst_max, floor_max)
    return diverged


def _tiny_model(use_graph=True, family="qwen2", **kw):
    import streaming_vlm_amd as S
    from streaming_vlm_amd import config as C
    from streaming_vlm_amd.weights import random_state_dict
    cfg = C.tiny_2_5(**kw) if family == "qwen2_5" else C.tiny(**kw)
    sd = random_state_dict(cfg, 0, "cpu")
    model = S.StreamingQwen2VL(cfg, {k: v.cuda() for k, v in sd.items()}, "cuda", max_len=768, max_new_tokens=8, use_graph=use_graph)
    return cfg, sd, model


def test_tiny_stream_sink_window():
    cfg, sd, model = _tiny_model()
    _compare(cfg, sd, 6, model)


def test_tiny_qwen2_5_stream_sink_window_ragged_windows():
    """Qwen2.5-VL family: windowed RMSNorm/SwiGLU tower (112x84 frames -> ragged attention windows), float temporal M-RoPE."""
    cfg, sd, model = _tiny_model(family="qwen2_5")
    _compare(cfg, sd, 6, model, size=(112, 84), window=96)


def test_tiny_qwen2_5_stream_structural_and_all_text():
    cfg, sd, model = _tiny_model(family="qwen2_5")
    _compare(cfg, sd, 7, model, policy="structural", text_round=2, window_size=3, text_sink=4, text_sliding_window=8,
             previous_text="a b c d e f g h i j k l m n o p")
    cfg, sd, model = _tiny_model(family="qwen2_5")
    _compare(cfg, sd, 4, model, all_text=True)


def test_tiny_streams_append_mode():
    """pos_mode="append" on both families: un-rotated keys + per-row position history vs the oracle's rotated-K cache."""
    for fam in ("qwen2", "qwen2_5"):
        cfg, sd, model = _tiny_model(family=fam)
        _compare(cfg, sd, 7, model, policy="structural", text_round=2, window_size=3, text_sink=4, text_sliding_window=8,
                 previous_text="a b c d e f g h i j k l m n o p", pos_mode="append")
    cfg, sd, model = _tiny_model()
    _compare(cfg, sd, 6, model, pos_mode="append")


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


def test_golden_streams_eviction_trace_and_tokens():
    """Committed oracle streams (tests/golden/oracle_streams.json, minted by oracle/make_golden.py) replayed on the HIP
    engine with no oracle in the loop: eviction indices and KV lengths must be identical; greedy tokens are compared
    chunk by chunk up to the first bf16-noise flip (histories differ from there on)."""
    import json, os
    with open(os.path.join(os.path.dirname(__file__), "golden", "oracle_streams.json")) as f:
        gold = json.load(f)
    for name, g in gold.items():
        cfg, sd, model = _tiny_model(family="qwen2_5" if g.get("model") == "tiny_2_5" else "qwen2")
        _, trace, counts, ids_log = H.run_engine_stream(model, g["n_chunks"], **dict(g["kwargs"]))
        assert [[list(t) for t in c] for c in trace] == g["trace"], name
        assert [e["kv_len"] for e in ids_log] == g["kv_len"], name
        same = 0
        for e, want in zip(ids_log, g["new_tokens"]):
            if e["new"] != want:
                break
            same += 1
        print(f"[golden] {name}: {same}/{g['n_chunks']} chunks token-identical before the first flip")
        assert same >= 1, name


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
    sd = random_state_dict(cfg, 0, "cpu")
    model = S.StreamingQwen2VL(cfg, {k: v.cuda() for k, v in sd.items()}, "cuda", max_len=1024, max_new_tokens=8)
    _compare(cfg, sd, 3, model, size=224, window=128)


def test_full_size_2b_two_chunks_448():
    """BASELINE configs[1] at FULL size: Qwen2-VL-2B (28 LLM layers, 32 ViT blocks), 448x448 frames (1024 patches ->
    256 vision tokens), 2 chunks x 8 tokens, sink 4 / window 256 so that the second chunk evicts.  Same bars as the
    tiny model: identical eviction trace, logits within 2.5x of the oracle's own tiling noise."""
    import streaming_vlm_amd as S
    from streaming_vlm_amd import config as C
    from streaming_vlm_amd.weights import random_state_dict
    torch.set_num_threads(min(16, torch.get_num_threads()))
    cfg = C.qwen2_vl_2b()
    sd = random_state_dict(cfg, 0, "cpu")
    model = S.StreamingQwen2VL(cfg, {k: v.cuda() for k, v in sd.items()}, "cuda", max_len=1024, max_new_tokens=8)
    _compare(cfg, sd, 2, model, size=448, window=256, max_new=4)


def test_full_size_qwen2_5_vl_3b_two_chunks_448():
    """Qwen2.5-VL-3B at FULL size (36 LLM layers, 32 windowed ViT blocks, intermediate 3420 padded to 3424), 448x448 frames, two
    chunks with an eviction in between: same bars as the Qwen2-VL-2B full-size test."""
    import streaming_vlm_amd as S
    from streaming_vlm_amd import config as C
    from streaming_vlm_amd.weights import random_state_dict
    torch.set_num_threads(min(16, torch.get_num_threads()))
    cfg = C.qwen2_5_vl_3b()
    sd = random_state_dict(cfg, 0, "cpu")
    model = S.StreamingQwen2VL(cfg, {k: v.cuda() for k, v in sd.items()}, "cuda", max_len=1024, max_new_tokens=8)
    _compare(cfg, sd, 2, model, size=448, window=256, max_new=4)


def test_tight_pool_defragments_in_place_and_stays_exact():
    """A KV pool with 5 % head-room under the structural policy (many 1-3 row prunes and moves) runs out of whole pages:
    svlm_kv_move_rows packs the sparse ones in place, and the stream still matches the oracle."""
    import streaming_vlm_amd as S
    from streaming_vlm_amd import config as C
    from streaming_vlm_amd.weights import random_state_dict
    cfg = C.tiny()
    sd = random_state_dict(cfg, 0, "cpu")
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
    sd = random_state_dict(cfg, 0, "cpu")
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
