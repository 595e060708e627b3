"""Pin the oracle AND the product's host logic against vectors produced by the reference's own importable
modules (tests/golden/ref_*.json, minted by oracle/make_golden.py from /root/reference)."""
import json
import os

import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from oracle import qwen_range as oqr
from oracle.rope_index import get_rope_index as oracle_rope

import streaming_vlm_amd as S
from streaming_vlm_amd.spans import all_ranges
from streaming_vlm_amd.positions import rope_index_qwen2


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def test_qwen_range_matches_reference(golden_dir):
    data = _load(golden_dir, "ref_qwen_ranges.json")
    n = 0
    for name, entry in data.items():
        ids = entry["ids"]
        for c in entry["cases"]:
            for impl in (oqr.get_qwen_range, S.get_qwen_range):
                try:
                    got = list(impl(ids, c["label"], c["index"], contain_lf=c["contain_lf"]))
                except IndexError:
                    got = None
                assert got == c["range"], (name, c, impl.__module__, got)
            n += 1
    assert n >= 300


def test_product_range_accepts_tensors():
    ids = torch.tensor([[151644, 872, 198, 1, 151645, 198]])
    assert S.get_qwen_range(ids, "user", 0) == (0, 5)
    assert S.get_qwen_range(ids, "user", 0, contain_lf=False) == (0, 4)
    with pytest.raises(IndexError):
        S.get_qwen_range(ids, "assistant", 0)


_TOK = st.sampled_from([151644, 151645, 872, 77091, 151652, 151653, 151656, 198, 19702, 1467, 1462, 5, 6])


@settings(max_examples=300, deadline=None)
@given(st.lists(_TOK, min_size=0, max_size=60), st.sampled_from(["user", "previous text", "assistant", "vision", "user_text"]), st.booleans())
def test_product_range_equals_oracle_on_random_sequences(ids, label, lf):
    assert all_ranges(ids, label, lf) == oqr.all_ranges(ids, label, lf)


def test_rope_index_matches_reference(golden_dir):
    data = _load(golden_dir, "ref_rope_index.json")
    for name, e in data.items():
        want = np.asarray(e["pos"])
        got_o = oracle_rope(e["ids"], e["grid"])
        got_p, nxt = rope_index_qwen2(e["ids"], e["grid"], 2, 151656, 151652)
        assert np.array_equal(got_o, want), name
        assert np.array_equal(got_p, want), name
        assert nxt == int(want.max()) + 1
        # mrope delta of the reference = max + 1 - len
        if name != "chunk_448":
            assert e["delta"] == int(want.max()) + 1 - len(e["ids"])


def test_rope_index_text_only_and_errors():
    pos, nxt = rope_index_qwen2([1, 2, 3, 4], [], 2, 151656, 151652)
    assert pos.tolist() == [[0, 1, 2, 3]] * 3 and nxt == 4
    with pytest.raises(ValueError):
        rope_index_qwen2([151652, 151656, 151656], [[1, 4, 4]], 2, 151656, 151652)     # truncated vision span
    with pytest.raises(ValueError):
        oracle_rope([151652, 151656, 151656], [[1, 4, 4]])


def test_sec2ts_matches_reference(golden_dir):
    for s, want in _load(golden_dir, "ref_sec2ts.json").items():
        assert S.sec2ts(float(s)) == want


def test_vtt_writer(tmp_path):
    p = str(tmp_path / "a.vtt")
    with S.open_vtt(p) as f:
        f.write("x\n")
    with S.open_vtt(p) as f:
        f.write("y\n")
    assert open(p).read() == "WEBVTT\n\nx\ny\n"


def test_streaming_args_surface():
    a = S.StreamingArgs("shrink")
    assert a.pos_mode == "shrink" and a.all_text is False and a.input_ids is None and a.video_grid_thw is None
    with pytest.raises(AssertionError):
        S.StreamingArgs("nope")


def test_all_text_positions_match_reference_1d_rope(golden_dir):
    """StreamingArgs.all_text: vectors minted from the reference's importable get_1d_rope_index
    (qwen2_5/model_forward.py:6-28) -- oracle restatement and product builder."""
    import json, os
    import numpy as np
    from oracle.rope_index import get_1d_rope_index
    from streaming_vlm_amd.positions import rope_index_1d
    with open(os.path.join(golden_dir, "ref_rope_1d.json")) as f:
        gold = json.load(f)
    for name, g in gold.items():
        want = np.array(g["pos"])
        assert np.array_equal(get_1d_rope_index(g["n"]), want), name
        pos, nxt = rope_index_1d(g["n"])
        assert np.array_equal(pos, want) and nxt == g["n"] and g["delta"] == 0
