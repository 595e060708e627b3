"""Frame resize in front of the path: the oracle pinned TO THE BIT against torch's own CPU interpolate (the call
under the reference's torchvision resize) -- live against the installed torch and against a committed fixture of its outputs --
host logic, and the host half of the C ABI (tap tables).  The kernel itself is compared with the oracle in tests/test_kernels_gpu.py."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import resize as R
from oracle.make_golden import RESIZE_AXES, RESIZE_CASES, resize_case_input

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "resize_torch_cpu.npz")


def _torch_f32(x: torch.Tensor, h: int, w: int) -> np.ndarray:
    """what torchvision's v1 tensor resize(BICUBIC, antialias=True) computes for a uint8 clip before it rounds: float32 interpolate"""
    return F.interpolate(x.float(), size=(h, w), mode="bicubic", antialias=True, align_corners=False).numpy()


@pytest.mark.parametrize("H,W,h,w", [(360, 640, 252, 448), (240, 320, 336, 448), (100, 80, 56, 42), (37, 53, 28, 28), (64, 64, 64, 64),
                                     (90, 120, 28, 56), (30, 40, 56, 70), (720, 1280, 252, 448)])
def test_oracle_resize_equals_torch_interpolate_bit_for_bit(H, W, h, w):
    """The reference's resize is torch's antialiased bicubic CPU kernel in float32.  The oracle restates that kernel's arithmetic
    (formulas, double intermediates, FMA contraction, tap order: oracle/_c/resize_ref.c): the fp32 intermediate must be IDENTICAL
    in every bit, hence the rounded uint8 too -- on noise and on a ramp whose values sit on exact .5 ties."""
    x = resize_case_input(H, W, h, w)
    t = _torch_f32(x, h, w)
    f = R.resize_bicubic_aa_f32(x.numpy(), h, w)
    assert np.array_equal(f.view(np.uint32), t.view(np.uint32)), f"{int((f.view(np.uint32) != t.view(np.uint32)).sum())} of {t.size} fp32 values differ from torch"
    want = torch.from_numpy(t).clamp(0, 255).round().to(torch.uint8).numpy()
    assert np.array_equal(R.resize_bicubic_aa_u8(x.numpy(), h, w), want)
    if (H, W) == (h, w):
        assert np.array_equal(want, x.numpy()), "same-size resize must be the identity"


def test_oracle_resize_equals_the_committed_torch_outputs():
    """The same pin without the installed torch's kernel in the loop: fp32 outputs and whole-axis tap weights (read off one-hot rows)
    minted by oracle/make_golden.py --resize with the build container's torch."""
    g = np.load(GOLD)
    for (H, W, h, w) in RESIZE_CASES:
        want = g[f"f32_{H}x{W}_{h}x{w}"]
        got = R.resize_bicubic_aa_f32(resize_case_input(H, W, h, w).numpy(), h, w)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (H, W, h, w)
    for (n_in, n_out) in RESIZE_AXES:
        xmin, xsize, wt = R.aa_tables(n_in, n_out)
        tab, lo = g[f"w_{n_in}_{n_out}"], g[f"lo_{n_in}_{n_out}"]
        for i in range(n_out):            # torch's zero end taps cannot be told from padding in a probe: compare from the first non-zero one
            k = int(lo[i] - xmin[i])
            n = min(tab.shape[1], wt.shape[1] - k)
            assert k >= 0 and np.array_equal(wt[i, k:k + n].view(np.uint32), tab[i, :n].view(np.uint32)), (n_in, n_out, i)
            assert not wt[i, :k].any() and not wt[i, k + n:].any() and not tab[i, n:].any()


@pytest.mark.parametrize("n_in,n_out", [(640, 448), (360, 252), (854, 448), (1280, 448), (720, 252), (320, 448), (240, 336), (448, 448),
                                        (53, 28), (1920, 1008), (17, 200), (3, 7), (500, 1)])
def test_host_tap_tables_equal_oracle(n_in, n_out):
    """svlm_resize_aa_tables is host arithmetic behind the C ABI: first tap, tap count and fp32 weights, bit for bit."""
    from streaming_vlm_amd.ops import aa_resize_tables
    x1, n1, w1 = aa_resize_tables(n_in, n_out)
    x2, n2, w2 = R.aa_tables(n_in, n_out)
    assert np.array_equal(x1, x2) and np.array_equal(n1, n2) and w1.shape == w2.shape
    assert np.array_equal(w1.view(np.uint32), w2.view(np.uint32))
    assert np.all(n1 >= 1) and np.all(x1 + n1 <= n_in)
    assert np.allclose(w1.sum(1), 1.0, atol=1e-5)


def test_host_tap_tables_reject_bad_arguments():
    from streaming_vlm_amd import _lib
    lib = _lib.load()
    assert lib.svlm_resize_aa_tables(0, 10, None, None, None, 0) < 0
    assert lib.svlm_resize_aa_tables(10, -1, None, None, None, 0) < 0
    K = lib.svlm_resize_aa_tables(640, 448, None, None, None, 0)
    xmin = np.zeros(448, np.int32)
    wt = np.zeros((448, K), np.float32)
    assert lib.svlm_resize_aa_tables(640, 448, xmin.ctypes.data, xmin.ctypes.data, wt.ctypes.data, K - 1) < 0      # stride too small
    assert lib.svlm_resize_ws_bytes(6, 360, 448) == 6 * 360 * 448 * 4 and lib.svlm_resize_ws_bytes(0, 1, 1) < 0


def test_smart_resize_and_budgets():
    from streaming_vlm_amd import ingest
    # known answers of qwen_vl_utils.smart_resize at the reference's budgets (min 100*28*28, max 768*28*28)
    assert ingest.resized_shape(448, 448, 2) == (448, 448)            # BASELINE frames pass through
    assert ingest.resized_shape(720, 1280, 2) == (560, 1008)          # 720p: scaled down into 602112 px, multiples of 28
    assert ingest.resized_shape(360, 640, 2) == (364, 644)            # rounded to the factor only
    assert ingest.resized_shape(224, 224, 2) == (280, 280)            # below the minimum: scaled up
    assert ingest.resized_shape(1080, 1920, 480) == (420, 728)        # long clip: the TOTAL budget binds (77 070 336 / 480 * 2)
    with pytest.raises(ValueError):
        ingest.smart_resize(10, 4000)
    for H in range(100, 1300, 97):
        for W in range(120, 2000, 131):
            for n in (1, 2, 16, 480):
                mp = max(min(ingest.VIDEO_MAX_PIXELS, ingest.VIDEO_TOTAL_PIXELS / n * 2), int(ingest.VIDEO_MIN_PIXELS * 1.05))
                want = R.smart_resize(H, W, 28, ingest.VIDEO_MIN_PIXELS, mp)
                got = ingest.resized_shape(H, W, n)
                assert got == want and got[0] % 28 == 0 and got[1] % 28 == 0
                assert ingest.VIDEO_MIN_PIXELS <= got[0] * got[1] <= mp


def test_spatial_resize_video_passthrough_and_loud_failure():
    from streaming_vlm_amd import ingest
    clip = torch.zeros((2, 3, 448, 448), dtype=torch.uint8)
    assert ingest.spatial_resize_video(clip, None) is clip               # nothing to do: no launch, no ops needed
    with pytest.raises(RuntimeError, match="resize runs on the GPU"):
        ingest.spatial_resize_video(torch.zeros((2, 3, 360, 640), dtype=torch.uint8), None)
    with pytest.raises(ValueError):
        ingest.spatial_resize_video(torch.zeros((2, 3, 360, 640)), None)
    from streaming_vlm_amd.synthetic import SyntheticVideo
    assert SyntheticVideo.from_path("synthetic://448x448@1fps").spatial_resize is False
    raw = SyntheticVideo.from_path("synthetic-raw://640x360@2fps?stream=3")
    assert raw.spatial_resize is True and raw.chunk(0, 1).shape == (2, 3, 360, 640)
