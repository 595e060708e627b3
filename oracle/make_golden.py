#!/usr/bin/env python
"""Mint the golden vectors under tests/golden/ (run in the BUILD container, where /root/reference and the
stock `transformers` are importable; the GPU box has neither the reference nor a need for this script).

Sources, in order of authority:
  ref_*      outputs of the reference's OWN importable modules (PYTHONPATH=/root/reference/src):
             utils.get_qwen_range.get_qwen_range, inference.qwen2.pos_emb.get_rope_index, utils.vtt_utils.sec2ts.
             The reference ships no tests or fixtures, so these are the only vectors it can pin itself.
  hf_*       outputs of the installed transformers' Qwen2-VL modules (the third-party code the reference calls)
             on a tiny seeded config: vision tower, decoder with DynamicCache, rotary tables, M-RoPE apply.
  oracle_*   the oracle's own eviction traces / greedy tokens (regression anchors for the host logic).

Only DATA is written (ids, shapes, numbers): no reference source text.
    python oracle/make_golden.py
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF_SRC = "/root/reference/src"

IM_START, IM_END, USER, ASSISTANT = 151644, 151645, 872, 77091
VSTART, VEND, VPAD, LF, TIME = 151652, 151653, 151656, 198, 1462
PT = [19702, 1467]


def build_sequences():
    """Hand-built chat-template id sequences covering the edge cases of inference.py:114-119,402-405."""
    def user(n_vid, with_time=True, query=(), lf=True):
        s = [IM_START, USER, LF] + ([TIME, 28, 15, 13, 15, 12, 16, 13, 15, 82] if with_time else []) + [VSTART] + [VPAD] * n_vid + [VEND]
        return s + list(query) + [IM_END] + ([LF] if lf else [])

    def asst(toks, lf=True, end=True):
        return [IM_START, ASSISTANT, LF] + list(toks) + ([IM_END] if end else []) + ([LF] if lf and end else [])

    sys_ = [IM_START, 8948, LF, 2610, 525, 264, 10950, 17847, 13, IM_END, LF]
    prev = lambda toks: [IM_START] + PT + [LF] + list(toks) + [IM_END, LF]
    seqs = {}
    seqs["two_rounds"] = sys_ + prev([100, 101, 102]) + user(4, query=[500, 501]) + asst([7, 8, 2503]) + user(4) + asst([9, 2503])
    seqs["no_trailing_lf"] = sys_ + prev([]) + user(2) + asst([7, 2503], lf=False)
    seqs["open_assistant"] = sys_ + prev([5]) + user(2) + asst([7], end=False)          # unterminated last segment
    seqs["many_rounds"] = sys_ + prev(list(range(300, 340)))
    for r in range(6):
        seqs["many_rounds"] = seqs["many_rounds"] + user(6, query=[900] if r == 0 else ()) + asst([40 + r, 41 + r, 2503])
    seqs["text_only"] = sys_ + prev([1, 2, 3])
    seqs["vision_first"] = [VSTART] + [VPAD] * 4 + [VEND, 11, 12]
    return seqs


def gen_reference_vectors():
    sys.path.insert(0, REF_SRC)
    from streaming_vlm.utils.get_qwen_range import get_qwen_range as ref_range
    from streaming_vlm.inference.qwen2.pos_emb import get_rope_index as ref_rope
    from streaming_vlm.utils.vtt_utils import sec2ts as ref_sec2ts

    seqs = build_sequences()
    ranges = {}
    for name, ids in seqs.items():
        t = torch.tensor([ids])
        cases = []
        for label in ["user", "previous text", "assistant", "vision", "user_text"]:
            for lf in (True, False):
                for index in (0, 1, 2, -1, -2, 5):
                    try:
                        r = ref_range(t, label, index, contain_lf=lf)
                        cases.append({"label": label, "index": index, "contain_lf": lf, "range": [int(r[0]), int(r[1])]})
                    except IndexError:
                        cases.append({"label": label, "index": index, "contain_lf": lf, "range": None})
        ranges[name] = {"ids": ids, "cases": cases}
    with open(os.path.join(OUT, "ref_qwen_ranges.json"), "w") as f:
        json.dump(ranges, f)

    class _Stub:      # get_rope_index only reads self.config.*
        class config:
            class vision_config:
                spatial_merge_size = 2
            image_token_id = 151655
            video_token_id = VPAD
            vision_start_token_id = VSTART

    rope = {}
    grids = {"two_rounds": [[1, 4, 4], [1, 4, 4]], "no_trailing_lf": [[1, 2, 4]], "open_assistant": [[1, 4, 2]],
             "many_rounds": [[1, 4, 6]] * 6, "vision_first": [[1, 4, 4]]}
    for name, g in grids.items():
        ids = torch.tensor([seqs[name]])
        pos, delta = ref_rope(_Stub(), ids, None, torch.tensor(g), None, torch.ones_like(ids, dtype=torch.bool))
        rope[name] = {"ids": seqs[name], "grid": g, "pos": pos[:, 0].tolist(), "delta": int(delta.flatten()[0])}
    # a 448x448 chunk: grid (1, 32, 32) -> 256 tokens
    big = [IM_START, USER, LF, TIME, 28, 15, VSTART] + [VPAD] * 256 + [VEND, IM_END, LF, IM_START, ASSISTANT, LF, 5, 6]
    ids = torch.tensor([big])
    pos, _ = ref_rope(_Stub(), ids, None, torch.tensor([[1, 32, 32]]), None, torch.ones_like(ids, dtype=torch.bool))
    rope["chunk_448"] = {"ids": big, "grid": [[1, 32, 32]], "pos": pos[:, 0].tolist(), "delta": 0}
    with open(os.path.join(OUT, "ref_rope_index.json"), "w") as f:
        json.dump(rope, f)
    with open(os.path.join(OUT, "ref_sec2ts.json"), "w") as f:
        json.dump({str(s): ref_sec2ts(s) for s in [0, 0.5, 1.0, 59.999, 61.25, 3600, 3661.5, 86399.001]}, f)
    # Qwen2.5 `all_text` positions (StreamingArgs.all_text, qwen2_5/model_forward.py:6-28,99)
    from streaming_vlm.inference.qwen2_5.model_forward import get_1d_rope_index as ref_1d
    one_d = {}
    for name in ("two_rounds", "vision_first"):
        ids = torch.tensor([seqs[name]])
        pos, delta = ref_1d(ids, None, None, None, torch.ones_like(ids))
        one_d[name] = {"n": len(seqs[name]), "pos": pos[:, 0].tolist(), "delta": int(delta.flatten()[0])}
    with open(os.path.join(OUT, "ref_rope_1d.json"), "w") as f:
        json.dump(one_d, f)
    print("reference vectors: ranges", sum(len(v["cases"]) for v in ranges.values()), "cases; rope", len(rope), "sequences")


def gen_hf_vectors():
    """Tiny stock Qwen2-VL modules (transformers as installed) on seeded inputs.  Weights are regenerated from the
    seed by the test (streaming_vlm_amd.weights.random_state_dict), only inputs/outputs are stored."""
    from transformers import Qwen2VLConfig, Qwen2VLForConditionalGeneration
    from transformers.cache_utils import DynamicCache
    import transformers
    import helpers as H
    from streaming_vlm_amd import config as C
    from streaming_vlm_amd.weights import random_state_dict

    cfg = C.tiny()
    cfg.text.num_heads, cfg.text.num_kv_heads = 2, 1          # HF derives head_dim = hidden / heads
    hf_cfg = Qwen2VLConfig(
        text_config=dict(hidden_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1, intermediate_size=512,
                         vocab_size=cfg.text.vocab_size, rms_norm_eps=1e-6, tie_word_embeddings=True,
                         rope_parameters={"rope_type": "default", "rope_theta": 1e6, "mrope_section": [16, 24, 24]}),
        vision_config=dict(depth=2, embed_dim=160, num_heads=2, hidden_size=256, mlp_ratio=2, patch_size=14, temporal_patch_size=2,
                           spatial_merge_size=2, in_channels=3),
        tie_word_embeddings=True)
    hf_cfg._attn_implementation = "eager"
    model = Qwen2VLForConditionalGeneration(hf_cfg).to(torch.float32).eval()
    sd = random_state_dict(cfg, 7, "cpu", dtype=torch.float32)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not [m for m in missing if "lm_head" not in m and "inv_freq" not in m], missing
    assert not unexpected, unexpected
    g = torch.Generator().manual_seed(11)
    out = {}
    with torch.no_grad():
        # ---- vision tower: 2 temporal grids of 4x4 patches
        grid = torch.tensor([[2, 4, 4]])
        pix = torch.randn(32, cfg.vision.patch_dim, generator=g)
        vis = model.model.visual(pix, grid_thw=grid)
        vis = vis.pooler_output if hasattr(vis, "pooler_output") else vis
        out["vit_pix"], out["vit_grid"], out["vit_out"] = pix.numpy(), grid.numpy(), vis.float().numpy()
        # ---- decoder: prefill 10 tokens then 1 token, 3-axis positions, DynamicCache
        x = torch.randn(1, 11, 256, generator=g) * 0.5
        pos = torch.stack([torch.arange(11), torch.tensor([0, 1, 2, 2, 2, 3, 3, 4, 5, 6, 7]), torch.tensor([0, 1, 2, 2, 3, 2, 3, 4, 5, 6, 7])]).unsqueeze(1)
        cache = DynamicCache(config=hf_cfg.text_config) if "config" in DynamicCache.__init__.__code__.co_varnames else DynamicCache()
        lm = model.model.language_model
        h1 = lm(inputs_embeds=x[:, :10], position_ids=pos[:, :, :10], past_key_values=cache, use_cache=True).last_hidden_state
        h2 = lm(inputs_embeds=x[:, 10:], position_ids=pos[:, :, 10:], past_key_values=cache, use_cache=True).last_hidden_state
        out["lm_x"], out["lm_pos"] = x[0].numpy(), pos[:, 0].numpy()
        out["lm_h_prefill"], out["lm_h_decode"] = h1[0].float().numpy(), h2[0].float().numpy()
        # ---- rotary table + M-RoPE apply
        from transformers.models.qwen2_vl import modeling_qwen2_vl as M
        cos, sin = lm.rotary_emb(x, pos)
        q = torch.randn(1, 2, 11, 128, generator=g)
        k = torch.randn(1, 1, 11, 128, generator=g)
        qe, ke = M.apply_multimodal_rotary_pos_emb(q, k, cos, sin, [16, 24, 24])
        out["rope_cos"], out["rope_sin"] = cos[:, 0].numpy(), sin[:, 0].numpy()
        out["rope_q"], out["rope_k"], out["rope_qe"], out["rope_ke"] = q[0].numpy(), k[0].numpy(), qe[0].numpy(), ke[0].numpy()
    np.savez_compressed(os.path.join(OUT, "hf_tiny_modules.npz"), **out)
    with open(os.path.join(OUT, "hf_tiny_modules.json"), "w") as f:
        json.dump({"transformers": transformers.__version__, "torch": torch.__version__, "weights_seed": 7, "dtype": "float32",
                   "note": "weights = streaming_vlm_amd.weights.random_state_dict(tiny(heads=2,kv=1), seed 7, fp32)"}, f)
    print("hf vectors:", {k: v.shape for k, v in out.items()})


def gen_hf_vectors_2_5():
    """Tiny stock Qwen2.5-VL vision tower + get_rope_index (transformers as installed): the third-party code behind the
    reference's qwen2_5/ patches.  Ragged windows on purpose: a (2, 8, 6) grid with 56-px windows."""
    from transformers import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration
    import transformers
    from streaming_vlm_amd import config as C
    from streaming_vlm_amd.weights import random_state_dict

    cfg = C.tiny_2_5()
    cfg.text.num_heads, cfg.text.num_kv_heads = 2, 1
    vc = cfg.vision
    hf_cfg = Qwen2_5_VLConfig(
        text_config=dict(hidden_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1, intermediate_size=512,
                         vocab_size=cfg.text.vocab_size, rms_norm_eps=1e-6, tie_word_embeddings=True,
                         rope_parameters={"rope_type": "default", "rope_theta": 1e6, "mrope_section": [16, 24, 24]}),
        vision_config=dict(depth=vc.depth, hidden_size=vc.embed_dim, num_heads=vc.num_heads, intermediate_size=vc.mlp_hidden,
                           out_hidden_size=vc.out_hidden, patch_size=14, temporal_patch_size=2, spatial_merge_size=2,
                           in_channels=3, window_size=vc.window_size, fullatt_block_indexes=list(vc.fullatt_block_indexes),
                           hidden_act="silu", tokens_per_second=2),
        tie_word_embeddings=True)
    hf_cfg._attn_implementation = "eager"
    model = Qwen2_5_VLForConditionalGeneration(hf_cfg).to(torch.float32).eval()
    sd = random_state_dict(cfg, 7, "cpu", dtype=torch.float32)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not [m for m in missing if "lm_head" not in m and "inv_freq" not in m], missing
    assert not unexpected, unexpected
    g = torch.Generator().manual_seed(13)
    out = {}
    with torch.no_grad():
        for tag, grid in (("a", [[2, 8, 6]]), ("b", [[1, 4, 4], [1, 4, 4]])):
            gt = torch.tensor(grid)
            n = int(sum(t * h * w for t, h, w in grid))
            pix = torch.randn(n, vc.patch_dim, generator=g)
            vis = model.model.visual(pix, grid_thw=gt)
            vis = vis.pooler_output if hasattr(vis, "pooler_output") else vis
            out[f"vit_pix_{tag}"], out[f"vit_grid_{tag}"], out[f"vit_out_{tag}"] = pix.numpy(), gt.numpy(), vis.float().numpy()
        # M-RoPE ids with a float temporal step: second_per_grid_ts = 1.0 and 0.5, tokens_per_second = 2
        VS, VP, VE = cfg.vision_start_token_id, cfg.video_token_id, 151653
        ids = [1, 2, 3, VS] + [VP] * 12 + [VE, 7, 8, VS] + [VP] * 24 + [VE, 9]
        grids = [[1, 8, 6], [2, 8, 6]]
        for tag, spg in (("1", 1.0), ("h", 0.5)):
            kw = dict(input_ids=torch.tensor([ids]), video_grid_thw=torch.tensor(grids), second_per_grid_ts=torch.tensor([spg, spg]),
                      attention_mask=torch.ones(1, len(ids), dtype=torch.long))
            try:
                pos, _ = model.model.get_rope_index(**kw)
            except TypeError:      # newer signatures want the modality map
                mm = torch.tensor([[2 if t == VP else 0 for t in ids]])
                pos, _ = model.model.get_rope_index(mm_token_type_ids=mm, **kw)
            out[f"rope_pos_{tag}"] = pos[:, 0].float().numpy()
        out["rope_ids"], out["rope_grids"] = np.array(ids), np.array(grids)
    np.savez_compressed(os.path.join(OUT, "hf_tiny_modules_2_5.npz"), **out)
    with open(os.path.join(OUT, "hf_tiny_modules_2_5.json"), "w") as f:
        json.dump({"transformers": transformers.__version__, "torch": torch.__version__, "weights_seed": 7, "dtype": "float32",
                   "note": "weights = streaming_vlm_amd.weights.random_state_dict(tiny_2_5(heads=2,kv=1), seed 7, fp32)"}, f)
    print("hf 2.5 vectors:", {k: v.shape for k, v in out.items()})


def gen_oracle_vectors():
    """Oracle streams on the DECISIVE weights (streaming_vlm_amd.weights.decisive_state_dict with the copy distance
    tests/helpers.py:decisive_offset picks for the stream's geometry): every greedy token is decided by a margin far above the
    bf16 noise of any implementation, so the GPU replay (tests/test_engine_gpu.py) demands token-for-token equality."""
    import helpers as H
    from streaming_vlm_amd import config as C
    cfg = C.tiny()
    runs = {}

    def mint(name, cfg, kw, n, model=None):
        sd = H.decisive_weights(cfg, size=kw.get("size", 56), all_text=kw.get("all_text", False))
        o = H.run_oracle_stream(cfg, sd, n, keep_logits=True, **kw)
        mm = min(H.greedy_margins(o))
        assert mm >= 0.5, (name, mm)
        runs[name] = {"kwargs": kw, "n_chunks": n, "trace": o["trace"], "kv_len": o["kv_len"], "new_tokens": o["new_tokens"],
                      "min_margin": round(mm, 4), "offset": H.decisive_offset(kw.get("size", 56), 8, kw.get("all_text", False))}
        if model:
            runs[name]["model"] = model

    for name, kw in {
        "sink4_win64": dict(policy="sink_window", sink=4, window=64),
        "sink4_win256": dict(policy="sink_window", sink=4, window=256, size=112),
        "structural_t2_v3": dict(policy="structural", text_round=2, window_size=3, text_sink=4, text_sliding_window=8,
                                 previous_text="a b c d e f g h i j k l m n o p"),
        "structural_t3_v2": dict(policy="structural", text_round=3, window_size=2, text_sink=2, text_sliding_window=6,
                                 previous_text="a b c d e f g h i j k l m n o p"),
        "structural_default_16": dict(policy="structural", text_round=16, window_size=16, text_sink=512, text_sliding_window=512),
        # BASELINE configs[0] geometry: 32 frames of 224x224 (64 vision tokens each), sink 4 / window 256, greedy, rep-pen 1.05
        "cfg0_224_sink4_win256_32": dict(policy="sink_window", sink=4, window=256, size=224),
        # pos_mode="append": positions travel with their rows (rotated-K cache in the reference), nothing is re-indexed
        "append_sink4_win64": dict(policy="sink_window", sink=4, window=64, pos_mode="append"),
        "append_structural_t2_v3": dict(policy="structural", text_round=2, window_size=3, text_sink=4, text_sliding_window=8,
                                        previous_text="a b c d e f g h i j k l m n o p", pos_mode="append"),
    }.items():
        mint(name, cfg, kw, 20 if "default" in name else (32 if name.startswith("cfg0") else 10))
    # Qwen2.5-VL family (tiny_2_5: windowed RMSNorm/SwiGLU tower, float temporal M-RoPE); frames of 112x84 have ragged windows
    cfg25 = C.tiny_2_5()
    for name, kw in {
        "q25_sink4_win64": dict(policy="sink_window", sink=4, window=64),
        "q25_ragged_sink4_win96": dict(policy="sink_window", sink=4, window=96, size=[112, 84]),
        "q25_structural_t2_v3": dict(policy="structural", text_round=2, window_size=3, text_sink=4, text_sliding_window=8,
                                     previous_text="a b c d e f g h i j k l m n o p"),
        "q25_all_text": dict(policy="sink_window", sink=4, window=64, all_text=True),
        "q25_append_structural": dict(policy="structural", text_round=2, window_size=3, text_sink=4, text_sliding_window=8,
                                      previous_text="a b c d e f g h i j k l m n o p", pos_mode="append"),
    }.items():
        mint(name, cfg25, kw, 10, model="tiny_2_5")
    with open(os.path.join(OUT, "oracle_streams.json"), "w") as f:
        json.dump(runs, f)
    print("oracle vectors:", {k: v["min_margin"] for k, v in runs.items()})


RESIZE_CASES = [(100, 80, 56, 42), (37, 53, 28, 28), (90, 120, 28, 56), (30, 40, 56, 70), (64, 64, 64, 64)]
RESIZE_AXES = [(640, 448), (360, 252), (1280, 448), (320, 448), (53, 28)]


def resize_case_input(H, W, h, w):
    g = torch.Generator().manual_seed(H * 7 + w)
    x = torch.randint(0, 256, (2, 3, H, W), generator=g, dtype=torch.uint8)
    x[1] = (torch.arange(W).view(1, 1, W) * 255 // max(W - 1, 1) + torch.arange(H).view(1, H, 1)).clamp(0, 255).to(torch.uint8)   # smooth ramp: many exact ties
    return x


def gen_resize_vectors():
    """tests/golden/resize_torch_cpu.npz: what the INSTALLED torch's CPU kernel returns for the reference's resize call
    (F.interpolate(x.float(), size, mode="bicubic", antialias=True, align_corners=False), the call under torchvision's v1
    transforms.functional.resize): fp32 outputs of seeded clips, and the tap weights of whole axes read off one-hot rows."""
    import torch.nn.functional as F
    out = {}
    for (H, W, h, w) in RESIZE_CASES:
        y = F.interpolate(resize_case_input(H, W, h, w).float(), size=(h, w), mode="bicubic", antialias=True, align_corners=False)
        out[f"f32_{H}x{W}_{h}x{w}"] = y.numpy()
    for (n_in, n_out) in RESIZE_AXES:
        eye = torch.eye(n_in).view(1, 1, n_in, n_in)
        wts = F.interpolate(eye, size=(n_in, n_out), mode="bicubic", antialias=True, align_corners=False)[0, 0].numpy().T      # [out][in]
        nz = [np.nonzero(r)[0] for r in wts]
        lo = np.array([r[0] for r in nz], np.int32)
        K = max(int(r[-1] - r[0] + 1) for r in nz)
        tab = np.zeros((n_out, K), np.float32)
        for i, r in enumerate(nz):
            tab[i, :r[-1] - r[0] + 1] = wts[i, r[0]:r[-1] + 1]
        out[f"w_{n_in}_{n_out}"] = tab
        out[f"lo_{n_in}_{n_out}"] = lo
    np.savez_compressed(os.path.join(OUT, "resize_torch_cpu.npz"), torch_version=np.array(torch.__version__), **out)
    print("resize vectors:", sorted(out)[:4], "...")


def gen_full_size_vectors(which=("2b", "7b")):
    """FULL-size token streams (tests/golden/full_size_streams.json): the CPU oracle in bf16 on the decisive weights, minutes of
    host time each, so they are minted here once and replayed on the GPU without the oracle in the loop.
      cfg1_2b_448_win2048   BASELINE configs[1]: Qwen2-VL-2B, 448x448 @1 fps, sink 4 / window 2048, 20 tokens, 12 chunks
                            (the window fills at chunk 7 and evicts from there on)
      cfg2_7b_448_2fps      configs[2]'s model and frame rate: Qwen2-VL-7B, 448x448 @2 fps, 20 tokens, window 512, 3 chunks
    Stored per chunk: eviction trace, KV length, tokens, and the raw logit of each chosen token (to bound the GPU's logit there)."""
    import helpers as H
    from streaming_vlm_amd import config as C
    path = os.path.join(OUT, "full_size_streams.json")
    runs = {}
    if os.path.exists(path):
        with open(path) as f:
            runs = json.load(f)
    plans = {"2b": ("cfg1_2b_448_win2048", C.qwen2_vl_2b, dict(size=448, fps=1.0, policy="sink_window", sink=4, window=2048, max_new=20,
                                                              previous_text=""), 12),
             "7b": ("cfg2_7b_448_2fps", C.qwen2_vl_7b, dict(size=448, fps=2.0, policy="sink_window", sink=4, window=512, max_new=20,
                                                           previous_text=""), 3),
             # BASELINE configs[2] AT SPEC: Qwen2-VL-7B, 448x448 @2 fps, sink 4 / window 4096, 20 tokens; ~295 rows per chunk, so the
             # window fills at chunk 14 and the last four chunks each evict (about an hour of host time)
             "7b_spec": ("cfg2_7b_448_2fps_win4096", C.qwen2_vl_7b, dict(size=448, fps=2.0, policy="sink_window", sink=4, window=4096,
                                                                        max_new=20, previous_text=""), 18),
             # BASELINE configs[4] at the real 7B widths, mid-size: 10 chunks (20 frames, 2.8k prompt rows) piled into ONE forward, then 3
             # live chunks; the first live chunk compacts the cache to sink + window (2048 here, so that it does).  The GPU replay runs
             # the engine with PREFILL_ROWS = 1024 (three prefill passes) and its 8-grid ViT batching (8 + 2 grids).  Once with the bf16
             # tower, once with the fp8 tower recipe (oracle/model.py:linear_fp8).  The full-size shape (300 chunks, 83k rows) costs the
             # CPU oracle days (its attention materialises Hq x T x L scores); 32 chunks took 2 h 20 min and are not needed to run
             # every mechanism.
             "7b_dense": ("cfg4_7b_dense10", C.qwen2_vl_7b, dict(size=448, fps=2.0, policy="sink_window", sink=4, window=2048, max_new=20,
                                                                 previous_text="", dense_prefill_chunks=10), 13),
             "7b_dense_fp8": ("cfg4_7b_dense10_fp8vit", C.qwen2_vl_7b, dict(size=448, fps=2.0, policy="sink_window", sink=4, window=2048,
                                                                            max_new=20, previous_text="", dense_prefill_chunks=10), 13)}
    for key in which:
        name, mk, kw, n = plans[key]
        dense_chunks = kw.get("dense_prefill_chunks", 0)
        cfg = mk()
        sd = H.decisive_weights(cfg, size=kw["size"], max_new=kw["max_new"])
        from oracle import model as om
        om.VIT_FP8 = key.endswith("_fp8")
        try:
            o = H.run_oracle_stream(cfg, sd, n, keep_logits=True, **kw)
        finally:
            om.VIT_FP8 = False
        margins = H.greedy_margins(o)
        mm = min(margins)
        # Streams without a dense prefill: every step must be decisive (the replay runs free and requires every token).  A dense
        # prefill's first answered turn has no previous answer for the planted copy head to read (it lands on header tokens or inside a
        # vision span), so a few of its steps are near-ties: per-step margins are stored, the replay is teacher-forced with these
        # tokens and requires the engine's own argmax wherever the oracle's margin is >= 1.
        assert mm >= 1.0 or dense_chunks, (name, mm)
        tops = [[round(float(lg[t]), 4) for lg, t in zip(lgs, gen)] for lgs, gen in zip(o["logits"], o["generated"])]
        dense = kw.get("dense_prefill_chunks", 0)
        runs[name] = {"model": key.split("_")[0], "vit_fp8": key.endswith("_fp8"), "kwargs": kw, "n_chunks": n, "trace": o["trace"], "kv_len": o["kv_len"], "new_tokens": o["new_tokens"],
                      "top_logit": tops, "min_margin": round(mm, 4), "prefill_rows": 1024 if dense else None,
                      "margins": [round(float(m), 4) for m in margins] if dense else None,
                      "max_len": max(kw["sink"] + kw["window"] + 2 * 320 + 64, 64 + dense * 300 + 2 * 320),
                      "torch": torch.__version__}
        print(name, "min margin", mm, "kv_len", o["kv_len"])
        del sd, o
        with open(path, "w") as f:
            json.dump(runs, f)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    if "--resize" in sys.argv:        # only the torch-CPU resize fixture
        gen_resize_vectors()
        sys.exit(0)
    if "--full-only" in sys.argv:     # one full-size plan by key (2b / 7b / 7b_spec), nothing else re-minted
        gen_full_size_vectors(tuple(sys.argv[sys.argv.index("--full-only") + 1].split(",")))
        sys.exit(0)
    gen_reference_vectors()
    gen_hf_vectors()
    gen_hf_vectors_2_5()
    gen_oracle_vectors()
    gen_resize_vectors()
    if "--full" in sys.argv:          # minutes of CPU time and ~20 GB of RAM for the 7B
        gen_full_size_vectors()
