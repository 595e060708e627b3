"""GPU parity of the persistent decode-layer tail (csrc/dec_tail.hip), stage by stage.

Every fused stage is checked AT ITS OWN SCALE (cdna_hip_programming.md section 5.6: "a whole-layer tolerance hides an O(1)-wrong
sub-stage"): the kernel's hand-off granules are its intermediates (x', h, x''), so each stage's output is compared with the oracle
math (tests/ref_ops.py) applied to the kernel's OWN input of that stage, at the per-kernel bar of tests/test_kernels_gpu.py
(max |err| <= 2^-7 max|ref|, mean <= 1e-3 max|ref|).  The tail sums its products pairwise (v_dot2c_f32_bf16), the per-op kernels one
FMA at a time, so the two agree to fp32 rounding, not bit for bit: each stage is also held within one bf16 flip per thousand
outputs of svlm_gemv_bf16 / svlm_dec_gate_up / svlm_dec_qkv on the same stage input.  Hand-offs are exercised idle, replayed from a
graph, beside a GEMM stream on a second stream (uneven load), on grids smaller than the chip, and with the status word poisoned.
"""
import pytest
import torch

from test_kernels_gpu import BF16, close, rnd

pytestmark = pytest.mark.gpu

SHAPES = [(1536, 12, 2, 8960), (256, 4, 2, 512), (1024, 4, 2, 1000)]          # layers that fit the register files: 2B class and test widths
TOO_BIG = [(3584, 28, 4, 18944), (2048, 16, 2, 11008)]                      # 7B / 3B: refused (their per-op kernels already stream at 0.8 of peak)


@pytest.fixture(scope="module")
def ops():
    from streaming_vlm_amd.ops import HipOps
    return HipOps()


@pytest.fixture(scope="module")
def ref():
    from ref_ops import RefOps
    return RefOps()


def _unpack(gran):
    """int64 granules -> (bf16 values, tags)"""
    g = gran.cpu()
    lo = (g & 0xFFFFFFFF).to(torch.int64)
    vals = torch.stack([lo & 0xFFFF, lo >> 16], dim=1).reshape(-1).to(torch.int32).to(torch.int16).view(BF16)
    return vals, (g >> 32) & 0xFFFFFFFF


class Layer:
    def __init__(self, H, Hq, Hkv, I, seed, n_slots=48):
        D = 128
        self.H, self.I, self.qd, self.kd, self.D = H, I, Hq * D, Hkv * D, D
        s = seed * 100
        self.o_w = rnd((H, self.qd), s + 1, 0.03)
        self.ln2 = rnd((H,), s + 2, 0.1) + 1
        self.gu_w = rnd((2 * I, H), s + 3, 0.03)
        self.down_w = rnd((H, I), s + 4, 0.03)
        self.ln1 = rnd((H,), s + 5, 0.1) + 1
        self.qkv_w = rnd((self.qd + 2 * self.kd, H), s + 6, 0.03)
        self.qkv_b = rnd((self.qd + 2 * self.kd,), s + 7, 0.1)

    def cuda(self):
        for k, v in list(self.__dict__.items()):
            if isinstance(v, torch.Tensor):
                setattr(self, k + "_g", v.cuda())
        return self


def _run_tail(ops, L, nxt_layer, attn_g, x_g, ws, layer, n_layers, q_g, pool_g, slot_g, len_dev, grid=0):
    nxt = None
    if nxt_layer is not None:
        nxt = (nxt_layer.ln1_g, nxt_layer.qkv_w_g, nxt_layer.qkv_b_g, q_g, pool_g, layer + 1, slot_g, L.qd, L.kd, 17, len_dev)
    ops.dec_tail(attn_g, x_g, L.o_w_g, L.ln2_g, L.gu_w_g, L.down_w_g, 1e-6, ws, layer, n_layers, nxt=nxt, grid=grid)


def _check_stages(ops, ref, L, Ln, attn, x0, ws, layer, q_g, pool_g, pool0, slot_of, x_after):
    """All four stages of one launched tail against the oracle math and the per-op kernels, each on the kernel's own stage input."""
    H, I, qd, kd = L.H, L.I, L.qd, L.kd
    st, g1, gh, g2 = ops.dec_tail_views(ws, H, I, layer)
    assert int(st[0]) == 0, "a gatherer gave up"
    x1, t1 = _unpack(g1)
    h, th = _unpack(gh)
    x2, t2 = _unpack(g2)
    assert bool((t1 == 1).all()) and bool((th == 1).all()) and bool((t2 == 1).all()), "unpublished granules"
    # --- O: x' = x + W_o attn
    want = ref.gemv(attn, L.o_w, residual=x0, out=torch.zeros(H, dtype=BF16))
    close("tail x'", x1, want)
    # --- GU on the kernel's x'
    want = torch.zeros(I, dtype=BF16)
    ref.dec_gate_up(x1, L.ln2, 1e-6, L.gu_w, want)
    close("tail h", h, want)
    # --- DOWN on the kernel's h and x'
    want = ref.gemv(h, L.down_w, residual=x1, out=torch.zeros(H, dtype=BF16))
    close("tail x''", x2, want)
    assert torch.equal(x_after.cpu().view(torch.int16), x2.view(torch.int16)), "x buffer != published x''"
    if Ln is not None:
        q_c = torch.zeros(qd + 2 * kd, dtype=BF16)
        pool_c = pool0.clone()
        ref.dec_qkv(x2, Ln.ln1, 1e-6, Ln.qkv_w, Ln.qkv_b, q_c, pool_c, layer + 1, slot_of, qd, kd, length=17)
        close("tail q", q_g[:qd], q_c[:qd])
        close("tail pool", pool_g, pool_c)
        others = [i for i in range(pool0.shape[0]) if i != layer + 1]
        assert torch.equal(pool_g.cpu()[others], pool0[others]), "other layers' planes untouched"
    # --- against the per-op kernels on the same stage inputs: same rounding points, fp32 sums in a different order -> at most a
    # stray last-bit flip (<= 0.5 % of the outputs may differ at all, none by more than one bf16 ulp of the largest value)
    def same(name, a, b):
        a, b = a.float().cpu(), b.float().cpu()
        diff = (a != b).float().mean().item()
        assert diff <= 5e-3 and float((a - b).abs().max()) <= 2 ** -7 * float(b.abs().max()), f"{name}: {diff:.4f} of the outputs differ from the per-op kernel"
    same("O", ops.gemv(attn.cuda(), L.o_w_g, residual=x0.cuda()), x1)
    hh = torch.zeros(I, dtype=BF16, device="cuda")
    ops.dec_gate_up(x1.cuda(), L.ln2_g, 1e-6, L.gu_w_g, hh)
    same("GU", hh, h)
    same("DOWN", ops.gemv(h.cuda(), L.down_w_g, residual=x1.cuda()), x2)
    if Ln is not None:
        q2 = torch.zeros(qd + 2 * kd, dtype=BF16, device="cuda")
        pool2 = pool0.clone().cuda()
        ops.dec_qkv(x2.cuda(), Ln.ln1_g, 1e-6, Ln.qkv_w_g, Ln.qkv_b_g, q2, pool2, layer + 1, slot_of.cuda(), qd, kd,
                    len_dev=torch.tensor([17], dtype=torch.int32, device="cuda"))
        same("QKV q", q2[:qd], q_g[:qd])
        same("QKV pool", pool2, pool_g)


@pytest.mark.parametrize("H,Hq,Hkv,I", SHAPES)
@pytest.mark.parametrize("grid", [0, 200])
def test_tail_stage_by_stage(ops, ref, H, Hq, Hkv, I, grid):
    if not ops.dec_tail_supported(H, I, Hq * 128, Hkv * 128, grid):
        pytest.skip("no build for this geometry at this grid")
    L, Ln = Layer(H, Hq, Hkv, I, 1).cuda(), Layer(H, Hq, Hkv, I, 2).cuda()
    attn, x0 = rnd((L.qd,), 11, 1.0), rnd((H,), 12, 2.0)
    pool0 = rnd((3, 2, Hkv, 48, 128), 13)
    slot_of = torch.randperm(48, generator=torch.Generator().manual_seed(6)).to(torch.int32)
    len_dev = torch.tensor([17], dtype=torch.int32, device="cuda")
    ws = ops.dec_tail_ws(H, I, 2, "cuda")
    x_g, q_g, pool_g = x0.cuda(), torch.zeros(L.qd + 2 * L.kd, dtype=BF16, device="cuda"), pool0.clone().cuda()
    ops.dec_tail_reset(ws, H, I, 2)
    _run_tail(ops, L, Ln, attn.cuda(), x_g, ws, 0, 2, q_g, pool_g, slot_of.cuda(), len_dev, grid=grid)
    torch.cuda.synchronize()
    _check_stages(ops, ref, L, Ln, attn, x0, ws, 0, q_g, pool_g, pool0, slot_of, x_g)
    # last layer: no QKV phase
    x_g2 = x0.cuda()
    _run_tail(ops, L, None, attn.cuda(), x_g2, ws, 1, 2, None, None, None, None, grid=grid)
    torch.cuda.synchronize()
    _check_stages(ops, ref, L, None, attn, x0, ws, 1, None, None, None, None, x_g2)
    assert torch.equal(x_g2, x_g)


def test_tail_graph_replay_and_uneven_load(ops, ref):
    """Three chained layers (2B widths) captured with their reset node: every replay gives the eager launches' bits, idle and beside a
    GEMM stream that occupies part of the chip (hand-offs must not depend on timing or placement)."""
    H, Hq, Hkv, I = 1536, 12, 2, 8960
    Ls = [Layer(H, Hq, Hkv, I, 3 + i).cuda() for i in range(3)]
    n_layers = 3
    attn = [rnd((Ls[0].qd,), 30 + i, 1.0).cuda() for i in range(3)]
    x0 = rnd((H,), 40, 2.0)
    pool0 = rnd((3, 2, Hkv, 48, 128), 41)
    slot_g = torch.randperm(48, generator=torch.Generator().manual_seed(6)).to(torch.int32).cuda()
    len_dev = torch.tensor([17], dtype=torch.int32, device="cuda")
    ws = ops.dec_tail_ws(H, I, n_layers, "cuda")
    x_g, q_g, pool_g = x0.cuda(), torch.zeros(Ls[0].qd + 2 * Ls[0].kd, dtype=BF16, device="cuda"), pool0.clone().cuda()

    x0_g = x0.cuda()

    def step():
        x_g.copy_(x0_g)
        ops.dec_tail_reset(ws, H, I, n_layers)
        for i in range(n_layers):
            _run_tail(ops, Ls[i], Ls[i + 1] if i + 1 < n_layers else None, attn[i], x_g, ws, i, n_layers, q_g, pool_g, slot_g, len_dev)

    step()
    torch.cuda.synchronize()
    want_x, want_ws, want_pool = x_g.clone(), ws.clone(), pool_g.clone()
    assert int(ws[0]) == 0
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    A, B = rnd((1024, 1280), 50).cuda(), rnd((5120, 1280), 51).cuda()
    for it in range(12):
        ws[32:].fill_(-1)                       # poison: a replay that skipped the reset node would read stale tags
        if it >= 4:                             # uneven load: GEMMs of the ViT's shape on a second stream
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(6):
                    ops.gemm(A, B)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(x_g, want_x) and torch.equal(ws, want_ws) and torch.equal(pool_g, want_pool), f"replay {it} differs"


def test_tail_poisoned_status_does_not_spin(ops):
    """A non-zero status word (an earlier give-up) switches every gatherer to no-poll mode: the launch must drain at once (no granule is
    ever awaited), and the word stays set for the host to see."""
    H, Hq, Hkv, I = 1536, 12, 2, 8960
    L = Layer(H, Hq, Hkv, I, 9).cuda()
    ws = ops.dec_tail_ws(H, I, 1, "cuda")
    ws[0] = 1
    x_g = rnd((H,), 1).cuda()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    _run_tail(ops, L, None, rnd((L.qd,), 2).cuda(), x_g, ws, 0, 1, None, None, None, None)
    t1.record()
    torch.cuda.synchronize()
    assert int(ws[0]) == 1 and t0.elapsed_time(t1) < 50.0


@pytest.mark.parametrize("H,Hq,Hkv,I", TOO_BIG + [(8192, 8, 8, 1024)])
def test_tail_refuses_layers_that_do_not_fit(ops, H, Hq, Hkv, I):
    from streaming_vlm_amd._lib import SvlmError
    assert not ops.dec_tail_supported(H, I, Hq * 128, Hkv * 128)
    ws = ops.dec_tail_ws(H, I, 1, "cuda")
    z = lambda *s: torch.zeros(s, dtype=BF16, device="cuda")
    with pytest.raises(SvlmError):
        ops.dec_tail(z(Hq * 128), z(H), z(H, Hq * 128), z(H), z(2 * I, H), z(H, I), 1e-6, ws, 0, 1)
