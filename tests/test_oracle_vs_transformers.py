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
