"""Frame resize in front of the path: oracle pinned against torch's own interpolate, host logic, and the host
half of the C ABI (tap tables).  The kernel itself is compared with the oracle in tests/test_kernels_gpu.py."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import resize as R


def _torch_resize_u8(x: torch.Tensor, h: int, w: int) -> np.ndarray:
    """torchvision's tensor resize(BICUBIC, antialias=True) on uint8: float32 interpolate, clamp, round, cast back."""
    y = F.interpolate(x.float(), size=(h, w), mode="bicubic", antialias=True, align_corners=False)
    return y.clamp(0, 255).round().to(torch.uint8).numpy()


@pytest.mark.parametrize("H,W,h,w", [(360, 640, 252, 448), (240, 320, 336, 448), (100, 80, 56, 42), (37, 53, 28, 28), (64, 64, 64, 64),
                                     (90, 120, 28, 56), (30, 40, 56, 70)])
def test_oracle_resize_against_torch_interpolate(H, W, h, w):
    """The reference's resize is torch's antialiased bicubic interpolate; torch's CPU kernel sums its taps in an unpublished
    association, so the pin is: never more than ONE grey level apart, at most 1 pixel in 10 000 apart on noise and 3 in 1000 on a ramp full of exact ties."""
    g = torch.Generator().manual_seed(H * 7 + w)
    x = torch.randint(0, 256, (2, 3, H, W), generator=g, dtype=torch.uint8)
    x[1] = (torch.arange(W).view(1, 1, W) * 255 // max(W - 1, 1) + torch.arange(H).view(1, H, 1)).clamp(0, 255).to(torch.uint8)   # smooth ramp
    want = _torch_resize_u8(x, h, w)
    got = R.resize_bicubic_aa_u8(x.numpy(), h, w)
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() <= 1, f"oracle differs from torch by {d.max()} levels"
    assert (d[0] != 0).mean() <= 1e-4, f"noise frame: {(d[0] != 0).sum()} of {d[0].size} pixels differ"
    # the ramp interpolates to many EXACT .5 values, where the last bit of the fp32 sum decides the rounding
    assert (d[1] != 0).mean() <= 3e-3, f"ramp frame: {(d[1] != 0).sum()} of {d[1].size} pixels differ"
    if (H, W) == (h, w):
        assert np.array_equal(got, x.numpy()), "same-size resize must be the identity"
    f = R.resize_bicubic_aa_f32(x.numpy(), h, w)
    t = F.interpolate(x.float(), size=(h, w), mode="bicubic", antialias=True, align_corners=False).numpy()
    assert np.abs(f - t).max() < 2e-3, "fp32 intermediate drifts from torch beyond summation-order noise"


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
