"""Host logic of the product (no GPU): the engine driven by the TEST-ONLY CPU ops backend must reproduce the
oracle's streaming loop exactly -- tokens, KV lengths, eviction traces -- and the committed oracle traces."""
import json
import os

import numpy as np
import pytest
import torch

import helpers as H
from ref_ops import RefOps

import streaming_vlm_amd as S
from streaming_vlm_amd import config as C
from streaming_vlm_amd.weights import random_state_dict


@pytest.fixture(scope="module")
def tiny():
    cfg = C.tiny()
    return cfg, random_state_dict(cfg, 0, "cpu")


def _model(cfg, sd, **kw):
    return S.StreamingQwen2VL(cfg, sd, "cpu", ops=RefOps(), max_len=1024, max_new_tokens=8, use_graph=False, **kw)


def test_engine_equals_oracle_and_golden_traces(tiny, golden_dir):
    cfg, sd = tiny
    with open(os.path.join(golden_dir, "oracle_streams.json")) as f:
        gold = json.load(f)
    cfg25 = C.tiny_2_5()
    for name, g in gold.items():
        if "default" in name:
            continue
        kw = dict(g["kwargs"])
        # the streams were minted on the decisive weights of their geometry (oracle/make_golden.py)
        c = cfg25 if g.get("model") == "tiny_2_5" else cfg                                     # Qwen2.5-VL family streams
        model = _model(c, H.decisive_weights(c, size=kw.get("size", 56), all_text=kw.get("all_text", False)))
        res, trace, counts, log = H.run_engine_stream(model, g["n_chunks"], **kw)
        assert [[list(t) for t in c] for c in trace] == g["trace"], name          # identical eviction indices
        assert [e["kv_len"] for e in log] == g["kv_len"], name
        assert [e["new"] for e in log] == g["new_tokens"], name
        assert len(res) == g["n_chunks"] and all(isinstance(r["response"], str) for r in res)


def test_decode_tail_path_equals_per_op_path_on_the_host_backend(tiny):
    """Engine sequencing of the persistent-tail decode step (QKV of layer 0, then [attention, tail] per layer, the tail computing the
    NEXT layer's QKV): on the host backend the tail is the four ops it fuses, so tokens, logits and KV lengths must be identical."""
    cfg, sd = tiny
    sd = H.decisive_weights(cfg)
    outs = []
    for tail in (False, True):
        model = _model(cfg, sd, decode_tail=tail)
        res, trace, counts, log = H.run_engine_stream(model, 4, keep_logits=True)
        outs.append((trace, [e["ids"] for e in log], [e["kv_len"] for e in log], [torch.stack(e["logits"]) for e in log]))
    assert outs[0][:3] == outs[1][:3]
    for a, b in zip(outs[0][3], outs[1][3]):
        assert torch.equal(a, b)


def test_default_structural_policy_trace(tiny, golden_dir):
    """Reference defaults (16 vision rounds, 16 text rounds, 512+512 previous-text): trace equals the oracle's."""
    cfg, sd = tiny
    with open(os.path.join(golden_dir, "oracle_streams.json")) as f:
        g = json.load(f)["structural_default_16"]
    res, trace, counts, log = H.run_engine_stream(_model(cfg, H.decisive_weights(cfg)), g["n_chunks"], **g["kwargs"])
    assert [[list(t) for t in c] for c in trace] == g["trace"]
    assert [e["new"] for e in log] == g["new_tokens"]
    assert any(len(c) for c in trace), "20 chunks must trigger the 16-round eviction"


def test_recompute_mode_equals_fresh_prefill(tiny):
    """efficiency mode (c): no KV reuse -- every chunk re-encodes the retained frames (inference.py:423-438)."""
    cfg, sd = tiny
    res, trace, counts, log = H.run_engine_stream(_model(cfg, sd), 4, policy="structural", text_round=100, window_size=100, recompute=True)
    res2, trace2, counts2, log2 = H.run_engine_stream(_model(cfg, sd), 4, policy="structural", text_round=100, window_size=100)
    assert [e["new"] for e in log] == [e["new"] for e in log2]      # shrink mode: cached == recomputed


def test_grid_history_stays_bounded_on_an_unbounded_stream(tiny, monkeypatch):
    """The reference appends a row to streaming_args.video_grid_thw per chunk for ever (inference.py:415) and only ever reads its first
    rows (one per surviving vision span) and, when recomputing, its last ones: the driver keeps both ends and drops the middle.  Same
    tokens and eviction trace as the ever-growing tensor, sliding window and recompute."""
    from streaming_vlm_amd import driver as D
    from streaming_vlm_amd import model as M
    cfg, sd = tiny
    seen = []
    real = M.streaming_generate

    def spy(self, *a, **kw):
        seen.append(int(kw["streaming_args"].video_grid_thw.shape[0]))
        return real(self, *a, **kw)

    runs = {}
    assert D.GRID_ROWS_KEPT == 4096
    for kept in (4096, 8):
        monkeypatch.setattr(D, "GRID_ROWS_KEPT", kept)
        monkeypatch.setattr(M, "streaming_generate", spy)          # (bound to every model object when it is built)
        seen.clear()
        a = H.run_engine_stream(_model(cfg, sd), 24)
        n_a = list(seen)
        seen.clear()
        b = H.run_engine_stream(_model(cfg, sd), 12, policy="structural", text_round=100, window_size=3, recompute=True)
        runs[kept] = ([e["new"] for e in a[3]], a[1], [e["new"] for e in b[3]], n_a, list(seen))
    full, cut = runs[4096], runs[8]
    assert full[0] == cut[0] and full[1] == cut[1] and full[2] == cut[2]
    assert max(full[3]) == 24 and max(cut[3]) <= 8 and max(cut[4]) <= 8


def test_time_test_returns_section_times(tiny):
    cfg, sd = tiny
    out = S.streaming_inference(model=_model(cfg, sd), processor=S.SyntheticProcessor(), video_path="synthetic://56x56@1fps",
                                model_base="Qwen2", duration=2, do_sample=False, max_new_tokens=4, quiet=True, time_test=True)
    assert len(out) == 2 and set(out[0]) == {"PKV", "CHECK", "VIDEO", "INPUT", "GEN", "POST"}


def test_sampling_path_runs_and_is_seeded(tiny):
    cfg, sd = tiny
    outs = []
    for _ in range(2):
        g = torch.Generator().manual_seed(5)
        log = []
        S.streaming_inference(model=_model(cfg, sd), processor=S.SyntheticProcessor(), video_path="synthetic://56x56@1fps",
                              model_base="Qwen2", duration=2, do_sample=True, temperature=0.9, max_new_tokens=4, quiet=True,
                              generator=g, ids_log=log)
        outs.append([e["new"] for e in log])
    assert outs[0] == outs[1]


def test_vtt_and_json_outputs(tiny, tmp_path, capsys):
    cfg, sd = tiny
    p = str(tmp_path / "o.vtt")
    S.streaming_inference(model=_model(cfg, sd), processor=S.SyntheticProcessor(), video_path="synthetic://56x56@1fps", output_dir=p,
                          model_base="Qwen2", duration=2, do_sample=False, max_new_tokens=4, quiet=True, emit_json=True)
    txt = open(p).read()
    assert txt.startswith("WEBVTT\n\n00:00:00.000 --> 00:00:01.000\n Infer Time:")
    lines = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert [l["type"] for l in lines] == ["segment", "segment"] and lines[1]["start"] == 1.0


def test_error_behaviour(tiny):
    cfg, sd = tiny
    m = _model(cfg, sd)
    with pytest.raises(AssertionError):
        S.streaming_inference(model=m, processor=S.SyntheticProcessor(), window_size=5, chunk_duration=2, video_path="synthetic://56x56@1fps")
    with pytest.raises(FileNotFoundError):
        S.streaming_inference(model=m, processor=S.SyntheticProcessor(), video_path="/no/such/video.mp4", model_base="Qwen2", quiet=True)
    with pytest.raises(AssertionError):   # StreamingArgs: pos_mode must be in ['append', 'shrink'] (streaming_args.py:4)
        S.streaming_inference(model=m, processor=S.SyntheticProcessor(), video_path="synthetic://56x56@1fps", pos_mode="grow",
                              model_base="Qwen2", duration=1, quiet=True)
    with pytest.raises(ValueError):       # pixel/token mismatch is the reference's ValueError (model_forward.py:56-61)
        m._svlm_engine.generate([151652, 151656, 151653, 5], None, [[1, 4, 4]], torch.zeros(16, 1176), [[1, 4, 4]], max_new_tokens=2)
    with pytest.raises(MemoryError):
        m._svlm_engine.generate(list(range(10, 1100)), None, [], max_new_tokens=2)
    # keyword arguments the path does not implement are refused, not dropped: image inputs (qwen2/model_forward.py:36-50) and typos
    args = dict(input_ids=torch.tensor([[1, 2, 3]]), streaming_args=S.StreamingArgs("shrink"), max_new_tokens=2)
    with pytest.raises(NotImplementedError):
        m.generate(pixel_values=torch.zeros(4, 1176), image_grid_thw=torch.tensor([[1, 2, 2]]), **args)
    with pytest.raises(TypeError, match="temperatur"):
        m.generate(temperatur=0.5, **args)


# ----------------------------------------------------------------------------- KV pool
def _pool(max_len=256, pages=16):
    return S.KVPool(2, 2, 128, max_len, "cpu", RefOps(), page_tokens=pages, slack=0.25)


def _fill(pool, n, base):
    start = pool.length
    pool.reserve(n)
    k = torch.arange(base, base + n, dtype=torch.float32).view(n, 1).expand(n, 256).to(torch.bfloat16).contiguous()
    for layer in range(2):
        pool.ops.kv_append(k, -k, pool.pool, layer, torch.from_numpy(pool.slot_of), start, n)
    pool.commit(start + n)


def _rows(pool, layer=0):
    pool_k, pool_v = pool.layer_kv(layer)
    return pool_k[0, 0, :, 0].float().tolist(), pool_v[0, 1, :, 5].float().tolist()


def test_cache_object_contract_of_the_reference():
    """SURVEY 8b "Cache object": iteration yields (k, v) with shape[2] == L, key_cache[i] / value_cache[i] are readable
    and ASSIGNABLE (the reference's prune writes index_select results back, inference.py:54-59), get_seq_length(),
    update(k, v, layer_idx, cache_kwargs) appends and returns the full layer (streaming_cache.py:30-74)."""
    pool = _pool()
    pool.slot_of_dev = torch.from_numpy(pool.slot_of)
    g = torch.Generator().manual_seed(3)
    mk = lambda n: torch.randn(1, 2, n, 128, generator=g).to(torch.bfloat16)
    # update(): two chunks, every layer; the returned tensors are the full layer in logical order
    ref = [[None, None] for _ in range(2)]
    for T in (37, 5):
        for layer in range(2):
            k, v = mk(T), mk(T)
            ref[layer][0] = k if ref[layer][0] is None else torch.cat([ref[layer][0], k], 2)
            ref[layer][1] = v if ref[layer][1] is None else torch.cat([ref[layer][1], v], 2)
            ko, vo = pool.update(k, v, layer, None)
            assert torch.equal(ko, ref[layer][0]) and torch.equal(vo, ref[layer][1])
    assert pool.get_seq_length() == 42 and len(pool) == 2 and len(pool.key_cache) == 2
    for layer, (k, v) in enumerate(pool):
        assert k.shape == (1, 2, 42, 128) and torch.equal(k, ref[layer][0]) and torch.equal(v, ref[layer][1])
        assert torch.equal(pool.key_cache[layer], k) and torch.equal(pool.value_cache[layer], v)
    # the reference's prune, verbatim in shape: index_select on dim 2, assign back per layer
    keep = torch.tensor([i for i in range(42) if not 4 <= i <= 20])
    for i, (k, v) in enumerate(list(pool)):
        pool.key_cache[i] = torch.index_select(k, 2, keep)
        pool.value_cache[i] = torch.index_select(v, 2, keep)
    assert pool.get_seq_length() == 25
    for layer, (k, v) in enumerate(pool):
        assert torch.equal(k, ref[layer][0][:, :, keep]) and torch.equal(v, ref[layer][1][:, :, keep])
    # ... and it agrees with the in-place edit the product uses instead
    other = _pool()
    other.slot_of_dev = torch.from_numpy(other.slot_of)
    for layer in range(2):
        other.update(ref[layer][0], ref[layer][1], layer)
    other.prune(4, 20)
    for (k0, v0), (k1, v1) in zip(pool, other):
        assert torch.equal(k0, k1) and torch.equal(v0, v1)
    # half-assigned layers are an error at the next read, not silent garbage
    pool.key_cache[0] = mk(25)
    with pytest.raises(RuntimeError):
        pool.layer_kv(0)
    with pytest.raises(ValueError):
        pool.value_cache[0] = mk(24)


def test_append_mode_positions_travel_with_their_rows(tiny):
    """pos_mode="append": the product keeps un-rotated keys + each row's ORIGINAL position (KVPool.pos_rows, edited with the
    slot table); the oracle follows the reference literally (keys rotated BEFORE caching, language_forward.py:89-97).  Same bits."""
    for cfg in (tiny[0], C.tiny_2_5()):
        sd = random_state_dict(cfg, 0, "cpu")
        kw = dict(policy="structural", text_round=2, window_size=3, text_sink=4, text_sliding_window=8,
                  previous_text="a b c d e f g h i j", pos_mode="append")
        ref = H.run_oracle_stream(cfg, sd, 7, keep_logits=True, **kw)
        res, trace, counts, log = H.run_engine_stream(_model(cfg, sd), 7, keep_logits=True, **kw)
        assert [[list(t) for t in c] for c in trace] == [[list(t) for t in c] for c in ref["trace"]]
        assert [e["new"] for e in log] == ref["new_tokens"]
        for e, want in zip(log, ref["logits"]):
            for a, b in zip(e["logits"], want):
                assert torch.equal(a, b)
        shr = H.run_oracle_stream(cfg, sd, 7, keep_logits=True, **{**kw, "pos_mode": "shrink"})
        assert any(not torch.equal(a, b) for x, y in zip(ref["logits"][3:], shr["logits"][3:]) for a, b in zip(x, y)), \
            "after the first eviction append and shrink must see different positions"


def test_pool_position_history_follows_prune_and_move():
    pool = _pool()
    pool.slot_of_dev = torch.from_numpy(pool.slot_of)
    _fill(pool, 60, 0)
    pool.pos_rows[:, :60] = np.arange(60)[None, :] + np.array([[0.0], [100.0], [200.0]])
    ref = list(range(60))
    pool.prune(4, 13); del ref[4:14]
    pool.move(30, 35, 7); ref = ref[:8] + ref[30:36] + ref[8:30] + ref[36:]
    assert pool.pos_rows[0, :pool.length].tolist() == ref
    assert pool.pos_rows[2, :pool.length].tolist() == [r + 200 for r in ref]
    assert _rows(pool)[0] == ref                        # the K rows moved the same way


def test_linear_planes_validity_follows_every_edit_of_the_logical_order():
    """KVPool.lin_valid / *lin_len_dev: what the decode kernels may stream from the prefill's rotated copy (kv_pool.py).  A prefill
    raises it to its length; prune / move / truncate / plane assignment / update() lower it to the first row they disturb; appends
    (reserve / commit / release_reserved) and defragmentation (slots move, logical rows do not) leave it; the device copy follows at
    the next sync_device()."""
    pool = _pool(max_len=256, pages=16)
    pool.slot_of_dev = torch.from_numpy(pool.slot_of)
    assert pool.lin is not None and pool.lin.shape == (2, 2, 2, 256, 128) and pool.lin_args()[1] is pool.lin_len_dev
    dev_state = lambda: pool.lin_len_dev.tolist()                # {rows rotated, appended rows follow}
    _fill(pool, 200, 0)
    pool.lin_written(200); pool.lin_len_dev.copy_(torch.tensor([200, 1], dtype=torch.int32))   # what the gather launches of a prefill do
    pool.reserve(20); pool.commit(210); pool.release_reserved()  # decode steps: appended rows, unused tail given back
    pool.sync_device()
    assert pool.lin_valid == 200 and pool.lin_fresh and dev_state() == [200, 1]
    pool.move(150, 160, 99)
    assert pool.lin_valid == 100 and not pool.lin_fresh and dev_state() == [200, 1]      # host first ...
    pool.sync_device()
    assert dev_state() == [100, 0]                                                        # ... the device at the next sync
    pool.lin_written(100); pool.lin_appends_off(); pool.sync_device()                     # a decode step that does not maintain the planes
    assert pool.lin_valid == 100 and dev_state() == [100, 0]
    pool.prune(120, 130)
    assert pool.lin_valid == 100
    pool.prune(40, 50)
    assert pool.lin_valid == 40
    pool.defragment()
    assert pool.lin_valid == 40
    pool.truncate(30); pool.sync_device()
    assert pool.lin_valid == 30 and dev_state() == [30, 0]
    pool.lin_written(30)
    g = torch.Generator().manual_seed(1)
    k, v = (torch.randn(1, 2, 7, 128, generator=g).to(torch.bfloat16) for _ in range(2))
    for layer in range(2):
        pool.update(k, v, layer)                                 # rows that exist in the pool only
    assert pool.lin_valid == 30 and not pool.lin_fresh and pool.get_seq_length() == 37
    pool.lin_written(37)
    for i, (kk, vv) in enumerate(list(pool)):
        pool.key_cache[i] = kk[:, :, :20]
        pool.value_cache[i] = vv[:, :, :20]
    pool.sync_device()
    assert pool.lin_valid == 0 and dev_state() == [0, 0]
    bare = S.KVPool(2, 2, 128, 64, "cpu", RefOps(), linear_planes=False)
    assert bare.lin is None and bare.lin_args() is None
    bare.lin_written(10)
    assert bare.lin_valid == 0


def test_pool_prune_move_truncate_match_list_semantics():
    pool = _pool()
    pool.slot_of_dev = torch.from_numpy(pool.slot_of)      # CPU "device" mirror shares memory
    _fill(pool, 100, 0)
    ref = list(range(100))
    pool.prune(4, 23); del ref[4:24]
    pool.move(50, 56, 10); ref = ref[:11] + ref[50:57] + ref[11:50] + ref[57:]
    pool.truncate(70); ref = ref[:70]
    _fill(pool, 30, 130); ref += list(range(130, 160))       # bf16 holds integers < 256 exactly
    k, v = _rows(pool)
    assert k == [float(r) for r in ref] and v == [-float(r) for r in ref]
    assert pool.get_seq_length() == 100 and len(list(iter(pool))) == 2


def test_pool_frees_pages_and_defragments_in_place():
    pool = _pool(max_len=256, pages=16)
    pool.slot_of_dev = torch.from_numpy(pool.slot_of)
    _fill(pool, 240, 0)
    free0 = pool.free_slots_available()
    keep = list(range(240))
    # evict 12 of every 16 rows -> every page keeps 4 live rows (pinned, fragmented)
    for p in reversed(range(15)):
        pool.prune(p * 16 + 2, p * 16 + 13)
        del keep[p * 16 + 2:p * 16 + 14]
    assert pool.fragmentation() > 0.5 and pool.free_slots_available() == free0
    moved = pool.defragment()
    assert moved > 0 and pool.free_slots_available() > free0 and pool.stats["defrags"] == 1
    k, v = _rows(pool, 1)
    assert k == [float(r) for r in keep] and v == [-float(r) for r in keep]
    # reserve() defragments on demand instead of failing
    pool2 = _pool(max_len=256, pages=16)
    pool2.slot_of_dev = torch.from_numpy(pool2.slot_of)
    _fill(pool2, 250, 0)
    for p in reversed(range(15)):
        pool2.prune(p * 16 + 1, p * 16 + 14)
    _fill(pool2, 150, 1000)
    assert pool2.length == 250 - 15 * 14 + 150


def test_pool_limits():
    pool = _pool(max_len=64)
    with pytest.raises(MemoryError):
        pool.reserve(65)
    pool.reserve(10)
    with pytest.raises(AssertionError):
        pool.prune(0, 3)             # reserved-but-uncommitted rows must be released first
    pool.release_reserved()
    assert pool.reserved == 0


def test_patchify_matches_hf_layout():
    """Row order must be merge-block-major (what the merger's view(-1, 4*embed) relies on)."""
    frames = torch.arange(2 * 3 * 28 * 56, dtype=torch.float32).reshape(2, 3, 28, 56).to(torch.uint8)
    pix, grid = S.patchify(frames)
    assert grid == [[1, 2, 4]] and tuple(pix.shape) == (8, 1176)
    x = frames.float() / 255.0
    mean = torch.tensor(S.synthetic.OPENAI_CLIP_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(S.synthetic.OPENAI_CLIP_STD).view(1, 3, 1, 1)
    x = (x - mean) / std
    # patch (h=1, w=2) is row 4*? in merge-block-major order: blocks (0,0):[(0,0),(0,1),(1,0),(1,1)], (0,1):[(0,2),(0,3),(1,2),(1,3)]
    want = x[:, :, 14:28, 28:42].permute(1, 0, 2, 3).reshape(-1)          # (C, T, P, P) flattening of patch (1, 2)
    assert torch.allclose(pix[6], want)
    one, g1 = S.patchify(frames[:1])                                        # a single frame is duplicated to fill the temporal patch
    assert g1 == [[1, 2, 4]] and torch.allclose(one[:, :], S.patchify(torch.cat([frames[:1], frames[:1]]))[0])


def test_cli_mirrors_the_reference_flags(tiny, tmp_path, monkeypatch, capsys):
    """`python -m streaming_vlm_amd.driver` takes the reference's flags (inference.py:524-561), writes WebVTT and JSON lines."""
    from streaming_vlm_amd import driver as drv
    cfg, sd = tiny
    built = []

    def load(path, base, max_len=None, max_new_tokens=None):          # the CLI's own sizing, on the CPU test backend
        built.append(max_len)
        return S.StreamingQwen2VL(cfg, sd, "cpu", ops=RefOps(), max_len=max_len, max_new_tokens=max_new_tokens, use_graph=False), S.SyntheticProcessor()
    monkeypatch.setattr(drv, "load_model_and_processor", load)
    vtt = tmp_path / "out.vtt"
    out = drv._cli(["--model_path", "random:tiny", "--model_base", "Qwen2", "--video_path", "synthetic://56x56@1fps", "--duration", "2",
                    "--window_size", "4", "--text_round", "4", "--output_dir", str(vtt), "--emit_json", "--quiet", "--greedy"])
    assert len(out) == 2 and out[1]["start_time"] == 1 and out[1]["end_time"] == 2
    lines = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert [l["start"] for l in lines] == [0.0, 1.0] and all(l["type"] == "segment" for l in lines)
    text = vtt.read_text()
    assert text.startswith("WEBVTT\n\n00:00:00.000 --> 00:00:01.000\n") and "00:00:01.000 --> 00:00:02.000" in text
    # the reference's DEFAULT settings (16 vision rounds, 16 text rounds, 512 + 512 previous text, 20 tokens per chunk) past the
    # first structural eviction: the engine the CLI sizes for itself must hold the whole stream
    out = drv._cli(["--model_path", "random:tiny", "--model_base", "Qwen2", "--video_path", "synthetic://56x56@1fps", "--duration", "19",
                    "--output_dir", str(tmp_path / "d.vtt"), "--quiet", "--greedy"])
    assert len(out) == 19 and built[-1] is not None and built[-1] < 4096, built
    # two-second chunks: the loop retains window_size ROUNDS (inference.py:320), so an engine sized for window_size // chunk_duration
    # rounds would overflow once more than that many chunks are retained
    out = drv._cli(["--model_path", "random:tiny", "--model_base", "Qwen2", "--video_path", "synthetic://56x56@1fps", "--duration", "16",
                    "--chunk_duration", "2", "--window_size", "4", "--text_round", "4", "--output_dir", str(tmp_path / "e.vtt"), "--quiet", "--greedy"])
    assert len(out) == 8 and out[-1]["end_time"] == 16


def test_required_max_len_bounds_every_golden_stream(golden_dir):
    """The engine capacity `streaming_inference` derives from the eviction policy (driver.required_max_len) must cover the longest
    sequence each committed stream reaches (cached rows + un-cached suffix + generated tokens), and stay within 2x of it."""
    from streaming_vlm_amd.driver import required_max_len
    with open(os.path.join(golden_dir, "oracle_streams.json")) as f:
        gold = json.load(f)
    for name, g in gold.items():
        kw = dict(g["kwargs"])
        size = kw.get("size", 56)
        h, w = (size, size) if isinstance(size, int) else size
        n_tok = (h // 28) * (w // 28)
        prev = kw.get("previous_text", "hello world")
        need = required_max_len(n_tok, 8, kw["policy"], kw.get("window_size", 16), kw.get("text_round", 16), kw.get("text_sink"),
                                kw.get("text_sliding_window"), kw.get("sink", 4), kw.get("window", 64), g["n_chunks"], len(prev.split()))
        longest = max(g["kv_len"]) + 2
        assert longest <= need, (name, longest, need)
        assert need <= 2 * longest + 64, (name, longest, need)


def test_efficiency_harness_emits_the_reference_document(tiny, tmp_path):
    """tools/efficiency_modes.py: the four modes of eval/efficiency/efficiency_test.py on the CPU test backend; the saved
    document carries the reference's keys (efficiency_test.py:87-136) and mode (b) shows its saw-tooth once the stream is
    longer than its window (here a 6-chunk window so that the CPU run stays short)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("efficiency_modes", os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools", "efficiency_modes.py"))
    em = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(em)
    cfg, sd = tiny
    for mode, over, chunks in (("a", {}, 6), ("b", dict(window_size=6, text_round=6), 16), ("c", dict(window_size=3, text_round=3), 6), ("d", {}, 6)):
        mcfg = dict(em.MODES[mode], **over)
        model = S.StreamingQwen2VL(cfg, sd, "cpu", ops=RefOps(), max_len=em.mode_max_len(mcfg, chunks, 4), max_new_tokens=20, use_graph=False)
        p = em.efficiency_payload(mode, model, S.SyntheticProcessor(), None, chunks, model_path="random:tiny", video_path="synthetic://56x56@1fps", **over)
        assert set(p) >= {"meta", "per_chunk", "summary"}
        assert set(p["meta"]) == {"timestamp", "model_path", "model_base", "video_path", "pos_mode", "all_text", "skip_first_chunk", "temperature",
                                  "mode", "window_size", "chunk_duration", "text_round", "text_sink", "text_sliding_window", "recompute",
                                  "duration_tested_sec"}
        assert p["meta"]["mode"] == em.MODE_NAMES[mode] and p["meta"]["recompute"] == (mode == "c")
        assert len(p["per_chunk"]) == chunks == p["summary"]["num_chunks"]
        assert set(p["per_chunk"][0]) == {"chunk_index", "time_start_sec", "video_len_sec", "gen_time_sec", "decoded_tokens", "gen_time_per_token"}
        assert all(r["decoded_tokens"] == 20 and r["gen_time_per_token"] > 0 for r in p["per_chunk"])
        assert p["summary"]["avg_gen_time_per_token"] > 0
        path = em.save_payload(p, str(tmp_path))
        assert os.path.basename(path).startswith(em.MODE_NAMES[mode] + "__Qwen2__random:tiny__") and json.load(open(path))["meta"] == p["meta"]
        if mode == "b":          # the KV cache stops growing once the 6-chunk window is full
            assert p["svlm"]["kv_len_last"] < 16 * 40


def test_sampling_with_top_k_1_is_greedy_in_oracle_and_engine(tiny):
    """HF merges the checkpoint's generation_config into generate(): stock Qwen2-VL checkpoints carry top_k = 1, which turns the
    reference's do_sample=True call into the greedy stream.  Oracle warpers and the engine's sampling path both must say so."""
    from oracle import generate as og
    cfg, sd = tiny
    sc = torch.randn(1000, generator=torch.Generator().manual_seed(0))
    w = og.warp_scores(sc, 0.9, 1, 0.001)
    assert int(torch.isfinite(w).sum()) == 1 and int(torch.argmax(w)) == int(torch.argmax(sc))
    w = og.warp_scores(sc, 1.0, 5, 1.0)
    assert int(torch.isfinite(w).sum()) == 5
    p = torch.softmax(og.warp_scores(sc, 1.0, 0, 0.5), -1)
    kept = torch.sort(p[p > 0], descending=True).values
    full = torch.sort(torch.softmax(sc, -1), descending=True).values
    assert abs(float(full[:kept.numel()].sum()) - 0.5) < float(full[kept.numel() - 1]) + 1e-6          # smallest prefix reaching 0.5
    kw = dict(processor=S.SyntheticProcessor(), video_path="synthetic://56x56@1fps", model_base="Qwen2", duration=3, quiet=True,
              max_new_tokens=6, suppress_eos=True)
    a, b = [], []
    S.streaming_inference(model=_model(cfg, sd), do_sample=True, temperature=0.9, top_k=1, ids_log=a, **kw)
    S.streaming_inference(model=_model(cfg, sd), do_sample=False, ids_log=b, **kw)
    assert [e["new"] for e in a] == [e["new"] for e in b]


def test_dense_prefill_chunks_equal_one_forward_of_the_piled_turns(tiny):
    """BASELINE configs[4] shape on the CPU backend: 5 chunks piled into one generate() call (ViT in passes, prefill in passes of
    PREFILL_ROWS), then live chunks with sink/window eviction -- tokens, KV lengths and eviction trace equal the oracle's, whose
    piled forward is one plain pass."""
    cfg, sd = tiny
    sd = H.decisive_weights(cfg)
    model = _model(cfg, sd)
    eng = model._svlm_engine
    eng.PREFILL_ROWS, eng.VIT_BATCH_SEQS = 32, 2                 # force several passes on the tiny stream
    kw = dict(window=64, dense_prefill_chunks=5)
    res, trace, counts, log = H.run_engine_stream(model, 9, **kw)
    ref = H.run_oracle_stream(cfg, sd, 9, **kw)
    assert len(res) == 5 and len(log) == 5                      # chunks 4 .. 8 answer; 0 .. 3 only pile up
    live = [t for t in ref["trace"] if True]
    assert [e["new"] for e in log] == ref["new_tokens"]
    assert [e["kv_len"] for e in log] == ref["kv_len"]
    assert [t for t in trace if t] == [t for t in live if t] and any(t for t in trace)
    assert log[0]["ids"].count(151652) == 5                     # five vision spans in the first answered sequence
