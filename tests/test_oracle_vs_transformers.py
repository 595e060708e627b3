"""Pin the oracle's model math against the stock transformers Qwen2-VL modules (the third-party code the
reference calls) via vectors minted in the build container (tests/golden/hf_tiny_modules.npz, fp32, tiny
seeded config).  Also checks the shrink-mode invariant the reference relies on: post-cache RoPE over
un-rotated keys with the full position table == the stock pre-cache RoPE path."""
import os

import numpy as np
import pytest
import torch

from oracle import kv_policy, model as om

import helpers as H
from streaming_vlm_amd import config as C
from streaming_vlm_amd.weights import random_state_dict


@pytest.fixture(scope="module")
def hf(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "hf_tiny_modules.npz")))


@pytest.fixture(scope="module")
def setup():
    cfg = C.tiny()
    cfg.text.num_heads, cfg.text.num_kv_heads = 2, 1
    sd = random_state_dict(cfg, 7, "cpu", dtype=torch.float32)
    return cfg, H.oracle_cfg(cfg), sd


def test_vision_tower(hf, setup):
    cfg, ocfg, sd = setup
    out = om.vit_forward(sd, ocfg, torch.from_numpy(hf["vit_pix"]), hf["vit_grid"].tolist())
    err = float((out - torch.from_numpy(hf["vit_out"])).abs().max())
    print("vit max abs err", err)
    assert err < 2e-4 * float(np.abs(hf["vit_out"]).max()) + 1e-5


def test_decoder_prefill_then_decode_equals_stock_precache_rope(hf, setup):
    cfg, ocfg, sd = setup
    x = torch.from_numpy(hf["lm_x"])
    pos = hf["lm_pos"]
    kv = kv_policy.ListKV(ocfg.text.num_layers)
    h1 = om.decoder_forward(sd, ocfg, x[:10], kv, pos[:, :10])
    h2 = om.decoder_forward(sd, ocfg, x[10:], kv, pos)          # shrink mode: FULL table, keys re-rotated
    for got, want, name in ((h1, hf["lm_h_prefill"], "prefill"), (h2, hf["lm_h_decode"], "decode")):
        err = float((got - torch.from_numpy(want)).abs().max())
        print(name, "max abs err", err)
        assert err < 2e-4 * float(np.abs(want).max()) + 1e-5


def test_mrope_table_and_apply(hf, setup):
    cfg, ocfg, sd = setup
    cos, sin = om.mrope_cos_sin(hf["lm_pos"], 128, 1e6, [16, 24, 24], torch.float32)
    # stock returns the un-selected (3, L, D) tables; selecting sections must reproduce ours
    sec = [16, 24, 24] * 2
    sel = lambda t: torch.cat([m[i % 3] for i, m in enumerate(torch.from_numpy(t).split(sec, dim=-1))], dim=-1)
    assert torch.allclose(cos, sel(hf["rope_cos"]), atol=1e-6) and torch.allclose(sin, sel(hf["rope_sin"]), atol=1e-6)
    q, k = torch.from_numpy(hf["rope_q"]), torch.from_numpy(hf["rope_k"])
    assert torch.allclose(om.apply_rope(q, cos, sin), torch.from_numpy(hf["rope_qe"]), atol=1e-5)
    assert torch.allclose(om.apply_rope(k, cos, sin), torch.from_numpy(hf["rope_ke"]), atol=1e-5)


def test_live_transformers_agrees_when_importable(setup):
    """Same check against the LIVE installed transformers (skipped where it is absent / incompatible)."""
    tf = pytest.importorskip("transformers")
    try:
        from transformers.models.qwen2_vl import modeling_qwen2_vl as M
    except Exception as e:          # pragma: no cover
        pytest.skip(f"qwen2_vl modules unavailable: {e}")
    x = torch.randn(5, 256)
    w = torch.randn(256) * 0.1 + 1
    norm = M.Qwen2VLRMSNorm(256, eps=1e-6)
    norm.weight.data.copy_(w)
    assert torch.allclose(norm(x), om.rms_norm(x, w, 1e-6), atol=1e-6)
    assert torch.equal(M.rotate_half(x), om.rotate_half(x))


def test_tiled_attention_equals_global_in_fp32():
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(3, n, 128, generator=g) for n in (7, 150, 150))
    a = om.flash_attention(q, k, v, 143, 0.088)
    b = om._flash_attention_tiled(q, k, v, 143, 0.088, 32)
    assert torch.allclose(a, b, atol=1e-5)


# ----------------------------------------------------------------------------- Qwen2.5-VL
@pytest.fixture(scope="module")
def hf25(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "hf_tiny_modules_2_5.npz")))


@pytest.fixture(scope="module")
def setup25():
    cfg = C.tiny_2_5()
    cfg.text.num_heads, cfg.text.num_kv_heads = 2, 1
    return cfg, H.oracle_cfg(cfg), random_state_dict(cfg, 7, "cpu", dtype=torch.float32)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_qwen2_5_vision_tower(hf25, setup25, tag):
    """Windowed RMSNorm/SwiGLU tower incl. ragged windows ((2,8,6) grid, 56-px windows) and two videos in one call,
    against the stock Qwen2_5_VisionTransformerPretrainedModel."""
    cfg, ocfg, sd = setup25
    out = om.vit_forward(sd, ocfg, torch.from_numpy(hf25[f"vit_pix_{tag}"]), hf25[f"vit_grid_{tag}"].tolist())
    want = torch.from_numpy(hf25[f"vit_out_{tag}"])
    err = float((out - want).abs().max())
    print("qwen2.5 vit max abs err", err)
    assert out.shape == want.shape and err < 2e-4 * float(want.abs().max()) + 1e-5


@pytest.mark.parametrize("tag,spg", [("1", 1.0), ("h", 0.5)])
def test_qwen2_5_rope_index(hf25, setup25, tag, spg):
    """Float temporal M-RoPE ids: oracle restatement and the product's vectorised builder against the stock
    Qwen2_5_VLModel.get_rope_index (second_per_grid_ts 1.0 and 0.5, tokens_per_second 2)."""
    from oracle.rope_index import get_rope_index_2_5
    from streaming_vlm_amd.positions import rope_index_qwen2_5
    cfg, ocfg, sd = setup25
    ids, grids = hf25["rope_ids"].tolist(), hf25["rope_grids"].tolist()
    want = hf25[f"rope_pos_{tag}"]
    got_o = get_rope_index_2_5(ids, grids, 2, cfg.video_token_id, cfg.vision_start_token_id, spg, 2.0)
    got_p, nxt = rope_index_qwen2_5(ids, grids, 2, cfg.video_token_id, cfg.vision_start_token_id, spg, 2.0)
    assert np.array_equal(got_p, got_o) and nxt == float(got_o.max()) + 1       # product == line-by-line restatement
    if spg == 1.0:
        assert np.array_equal(got_o, want), (got_o[0], want[0])
    else:
        # the stock module truncates the temporal index to integers; the reference REPLACES it with its own float32
        # version (qwen2_5/pos_emb.py:121-126, bound at patch_model.py:37), which the restatement follows: only the temporal
        # axis of the second grid step may differ from the stock vector, and only by the truncated fraction
        diff = np.flatnonzero((got_o != want).any(0))
        assert set(diff.tolist()) <= set(range(32, 44)) and np.array_equal(got_o[1:], want[1:])


def test_qwen2_5_window_plan_matches_oracle():
    """The engine's numpy window plan == the oracle's restatement of get_window_index for divisible, ragged and
    multi-video grids."""
    from streaming_vlm_amd.engine import _window_plan
    for grid, ws in (([[1, 32, 32]], 112), ([[2, 8, 6]], 56), ([[1, 4, 4], [3, 10, 14]], 56), ([[1, 36, 20]], 112)):
        idx, cu = om.get_window_index(grid, 2, ws, 14)
        index, win_len, frame_len = _window_plan([tuple(g) for g in grid], 2, ws, 14)
        assert np.array_equal(index, idx.numpy())
        assert np.cumsum([0] + win_len).tolist() == cu
        assert sum(frame_len) == sum(t * h * w for t, h, w in grid)
