"""Python face of the C ABI: one method per entry point of include/svlm.h, taking torch tensors.

torch is used for device memory and streams only; every method validates shapes on the host
(a faulting kernel can take the whole node down) and launches on torch's current stream, so the
calls can be captured in a HIP graph (`torch.cuda.graph`).
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_SWIGLU, check

BF16 = torch.bfloat16


def _ptr(t):
    return 0 if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    # raw handle of torch's current stream; the C accessor is ~20x cheaper than building a torch.cuda.Stream object
    # (a prefill is ~300 launches from Python, so this is per-chunk host time the GPU can end up waiting for)
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _req(t, dtype, name, dims=None):
    if not t.is_cuda:
        raise _lib.SvlmError(f"{name}: tensor must live on the GPU (got {t.device}); the HIP path has no CPU fallback")
    if t.dtype != dtype:
        raise _lib.SvlmError(f"{name}: dtype {t.dtype} != {dtype}")
    if dims is not None and t.dim() != dims:
        raise _lib.SvlmError(f"{name}: expected {dims}-D tensor, got shape {tuple(t.shape)}")
    if t.dim() and t.stride(-1) != 1:
        raise _lib.SvlmError(f"{name}: innermost dimension must be contiguous")


def aa_resize_tables(in_size, out_size):
    """Tap tables (first tap, tap count, weights) of one axis of the antialiased bicubic resize, from the library's HOST
    function svlm_resize_aa_tables (no GPU involved)."""
    import numpy as np
    lib = _lib.load()
    K = lib.svlm_resize_aa_tables(in_size, out_size, None, None, None, 0)
    if K <= 0:
        raise _lib.SvlmError(f"svlm_resize_aa_tables({in_size}, {out_size}): bad arguments")
    xmin, xsize = np.zeros(out_size, np.int32), np.zeros(out_size, np.int32)
    wt = np.zeros((out_size, K), np.float32)
    rc = lib.svlm_resize_aa_tables(in_size, out_size, xmin.ctypes.data, xsize.ctypes.data, wt.ctypes.data, K)
    if rc != K:
        raise _lib.SvlmError(f"svlm_resize_aa_tables({in_size}, {out_size}) -> {rc}")
    return xmin, xsize, wt


class HipOps:
    """The only ops backend the product ships."""

    name = "hip"

    def __init__(self):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.SvlmError("no HIP device visible: the svlm hot path only runs on an MI355X (gfx950) GPU")
        self._gemm_ws = {}          # per-device fp32 scratch for split-K slabs
        self._resize_tabs = {}      # (device, H, W, h, w) -> device tap tables of the frame resize

    GEMM_WS_BYTES = 96 << 20

    def _ws(self, device):
        key = str(device)
        if key not in self._gemm_ws:
            self._gemm_ws[key] = torch.empty(self.GEMM_WS_BYTES // 4, dtype=torch.float32, device=device)
        return self._gemm_ws[key]

    # ------------------------------------------------------------------ dense
    def gemm(self, A, W, bias=None, residual=None, out=None, act=ACT_NONE):
        _req(A, BF16, "gemm.A", 2); _req(W, BF16, "gemm.W", 2)
        M, K = A.shape
        N, K2 = W.shape
        if K != K2:
            raise _lib.SvlmError(f"gemm: A is {tuple(A.shape)} but W is {tuple(W.shape)}")
        if act == ACT_SWIGLU:          # W = [gate rows; up rows]: the output has half as many columns
            if N % 2 or residual is not None or (bias is not None and bias.numel() != N):
                raise _lib.SvlmError("gemm: ACT_SWIGLU takes W = [gate; up] (even row count), bias = [gate; up], no residual")
            N //= 2
        if out is None:
            out = torch.empty((M, N), dtype=BF16, device=A.device)
        _req(out, BF16, "gemm.out", 2)
        if tuple(out.shape) != (M, N):
            raise _lib.SvlmError(f"gemm: out shape {tuple(out.shape)} != {(M, N)}")
        if bias is not None:
            _req(bias, BF16, "gemm.bias", 1)
            assert bias.numel() == (2 * N if act == ACT_SWIGLU else N)
        ldr = 0
        if residual is not None:
            _req(residual, BF16, "gemm.residual", 2)
            assert tuple(residual.shape) == (M, N)
            ldr = residual.stride(0)
        ws = self._ws(A.device)
        check(self.lib.svlm_gemm_bf16(_ptr(A), A.stride(0), _ptr(W), W.stride(0), _ptr(bias), _ptr(residual), ldr,
                                      _ptr(out), out.stride(0), M, N, K, act, _ptr(ws), ws.numel() * 4, _stream()), "svlm_gemm_bf16")
        return out

    def gemm_norm(self, A, W, norm_w, eps, out, out_norm, bias=None, residual=None, act=ACT_NONE, norm_b=None):
        """out = epi(A @ W^T) as `gemm`; out_norm = RMSNorm(out) * norm_w (norm_b None) or LayerNorm(out; norm_w, norm_b), inside
        the split-K reduce when there is one."""
        _req(A, BF16, "gemm_norm.A", 2); _req(W, BF16, "gemm_norm.W", 2); _req(out, BF16, "gemm_norm.out", 2)
        _req(out_norm, BF16, "gemm_norm.out_norm", 2); _req(norm_w, BF16, "gemm_norm.norm_w", 1)
        M, K = A.shape
        N = W.shape[0]
        assert W.shape[1] == K and tuple(out.shape) == (M, N) == tuple(out_norm.shape) and norm_w.numel() == N
        assert A.stride(1) == 1 and W.stride(1) == 1 and out.stride(1) == 1 and out_norm.stride(1) == 1
        ldr = 0
        if residual is not None:
            _req(residual, BF16, "gemm_norm.residual", 2)
            assert tuple(residual.shape) == (M, N) and residual.stride(1) == 1
            ldr = residual.stride(0)
        if bias is not None:
            _req(bias, BF16, "gemm_norm.bias", 1)
        ws = self._ws(A.device)
        check(self.lib.svlm_gemm_bf16_norm(_ptr(A), A.stride(0), _ptr(W), W.stride(0), _ptr(bias), _ptr(residual), ldr, _ptr(out),
                                           out.stride(0), M, N, K, act, _ptr(ws), ws.numel() * 4, _ptr(norm_w), _ptr(norm_b), float(eps),
                                           _ptr(out_norm), out_norm.stride(0), _stream()), "svlm_gemm_bf16_norm")
        return out, out_norm

    # ------------------------------------------------------------------ fp8 (BASELINE configs[4]: ViT GEMMs on the fp8 MFMA path)
    FP8 = torch.float8_e4m3fn

    def quant_rows_fp8(self, x, q=None, scale=None):
        """bf16 (rows, cols) -> (fp8 e4m3 (rows, cols), fp32 scale per row): scale = max|row| / 448, q = rne(x / scale)."""
        _req(x, BF16, "quant_fp8.x", 2)
        rows, cols = x.shape
        if q is None:
            q = torch.empty((rows, cols), dtype=self.FP8, device=x.device)
        if scale is None:
            scale = torch.empty((rows,), dtype=torch.float32, device=x.device)
        _req(q, self.FP8, "quant_fp8.q", 2); _req(scale, torch.float32, "quant_fp8.scale", 1)
        assert tuple(q.shape) == (rows, cols) and scale.numel() == rows
        check(self.lib.svlm_quant_rows_fp8(_ptr(x), x.stride(0), _ptr(q), q.stride(0), _ptr(scale), rows, cols, _stream()), "svlm_quant_rows_fp8")
        return q, scale

    def gemm_fp8(self, A8, a_scale, W8, w_scale, bias=None, residual=None, out=None, act=ACT_NONE, norm_w=None, norm_b=None, eps=1e-6,
                 out_norm=None, out_norm_q=None):
        """out = epi((A8 @ W8^T) * a_scale[:, None] * w_scale[None, :]); with norm_w: out_norm = norm(out) as `gemm_norm`;
        `out_norm_q` = (fp8 tensor, fp32 row scales): out_norm quantised for the next fp8 GEMM inside the same call."""
        _req(A8, self.FP8, "gemm_fp8.A", 2); _req(W8, self.FP8, "gemm_fp8.W", 2)
        _req(a_scale, torch.float32, "gemm_fp8.a_scale", 1); _req(w_scale, torch.float32, "gemm_fp8.w_scale", 1)
        M, K = A8.shape
        N, K2 = W8.shape
        if K != K2 or a_scale.numel() != M or w_scale.numel() != N:
            raise _lib.SvlmError(f"gemm_fp8: A {tuple(A8.shape)} W {tuple(W8.shape)} scales {a_scale.numel()}/{w_scale.numel()}")
        if out is None:
            out = torch.empty((M, N), dtype=BF16, device=A8.device)
        _req(out, BF16, "gemm_fp8.out", 2)
        assert tuple(out.shape) == (M, N)
        ldr = 0
        if residual is not None:
            _req(residual, BF16, "gemm_fp8.residual", 2); assert tuple(residual.shape) == (M, N)
            ldr = residual.stride(0)
        if bias is not None:
            _req(bias, BF16, "gemm_fp8.bias", 1); assert bias.numel() == N
        ldxn = 0
        if norm_w is not None:
            _req(norm_w, BF16, "gemm_fp8.norm_w", 1); _req(out_norm, BF16, "gemm_fp8.out_norm", 2)
            assert norm_w.numel() == N and tuple(out_norm.shape) == (M, N)
            ldxn = out_norm.stride(0)
        ws = self._ws(A8.device)
        if out_norm_q is not None:
            q8, qs = out_norm_q
            _req(q8, self.FP8, "gemm_fp8.out_norm_q", 2); _req(qs, torch.float32, "gemm_fp8.out_norm_scale", 1)
            assert norm_w is not None and tuple(q8.shape) == (M, N) and qs.numel() == M
            check(self.lib.svlm_gemm_fp8_normq(_ptr(A8), A8.stride(0), _ptr(a_scale), _ptr(W8), W8.stride(0), _ptr(w_scale), _ptr(bias), _ptr(residual),
                                               ldr, _ptr(out), out.stride(0), M, N, K, act, _ptr(ws), ws.numel() * 4, _ptr(norm_w), _ptr(norm_b),
                                               float(eps), _ptr(out_norm), ldxn, _ptr(q8), q8.stride(0), _ptr(qs), _stream()), "svlm_gemm_fp8_normq")
            return out
        check(self.lib.svlm_gemm_fp8(_ptr(A8), A8.stride(0), _ptr(a_scale), _ptr(W8), W8.stride(0), _ptr(w_scale), _ptr(bias), _ptr(residual), ldr,
                                     _ptr(out), out.stride(0), M, N, K, act, _ptr(ws), ws.numel() * 4, _ptr(norm_w), _ptr(norm_b), float(eps),
                                     _ptr(out_norm), ldxn, _stream()), "svlm_gemm_fp8")
        return out

    def gemv(self, x, W, bias=None, residual=None, out=None, out_f32=None, act=ACT_NONE):
        _req(x, BF16, "gemv.x"); _req(W, BF16, "gemv.W", 2)
        N, K = W.shape
        if x.numel() != K:
            raise _lib.SvlmError(f"gemv: x has {x.numel()} elements, W is {tuple(W.shape)}")
        if out is None and out_f32 is None:
            out = torch.empty((N,), dtype=BF16, device=x.device)
        if out is not None:
            _req(out, BF16, "gemv.out"); assert out.numel() == N
        if out_f32 is not None:
            _req(out_f32, torch.float32, "gemv.out_f32"); assert out_f32.numel() == N
        if bias is not None:
            _req(bias, BF16, "gemv.bias"); assert bias.numel() == N
        if residual is not None:
            _req(residual, BF16, "gemv.residual"); assert residual.numel() == N
        check(self.lib.svlm_gemv_bf16(_ptr(x), _ptr(W), W.stride(0), _ptr(bias), _ptr(residual), _ptr(out), _ptr(out_f32),
                                      N, K, act, _stream()), "svlm_gemv_bf16")
        return out if out is not None else out_f32

    def prefetch(self, t, n_wgs=256):
        """Infinity-Cache warm-up of a contiguous tensor on the CURRENT stream (callers put it on a side stream)."""
        assert t.is_cuda and t.is_contiguous()
        check(self.lib.svlm_prefetch(_ptr(t), t.numel() * t.element_size(), int(n_wgs), _stream()), "svlm_prefetch")

    def rmsnorm(self, x, w, eps, out=None):
        _req(x, BF16, "rmsnorm.x"); _req(w, BF16, "rmsnorm.w", 1)
        cols = x.shape[-1]
        rows = x.numel() // cols
        assert x.is_contiguous() and w.numel() == cols
        if out is None:
            out = torch.empty_like(x)
        _req(out, BF16, "rmsnorm.out"); assert out.is_contiguous() and out.numel() == x.numel()
        check(self.lib.svlm_rmsnorm(_ptr(x), _ptr(w), _ptr(out), rows, cols, float(eps), _stream()), "svlm_rmsnorm")
        return out

    def layernorm(self, x, w, b, eps, out=None):
        _req(x, BF16, "layernorm.x"); _req(w, BF16, "layernorm.w", 1); _req(b, BF16, "layernorm.b", 1)
        cols = x.shape[-1]
        rows = x.numel() // cols
        assert x.is_contiguous() and w.numel() == cols and b.numel() == cols
        if out is None:
            out = torch.empty_like(x)
        _req(out, BF16, "layernorm.out"); assert out.is_contiguous() and out.numel() == x.numel()
        check(self.lib.svlm_layernorm(_ptr(x), _ptr(w), _ptr(b), _ptr(out), rows, cols, float(eps), _stream()), "svlm_layernorm")
        return out

    def add(self, a, b, out=None):
        _req(a, BF16, "add.a"); _req(b, BF16, "add.b")
        assert a.is_contiguous() and b.is_contiguous() and a.numel() == b.numel()
        if out is None:
            out = torch.empty_like(a)
        check(self.lib.svlm_add(_ptr(a), _ptr(b), _ptr(out), a.numel(), _stream()), "svlm_add")
        return out

    def silu_mul(self, gu, out=None):
        _req(gu, BF16, "silu_mul.gu", 2)
        rows, two_i = gu.shape
        assert gu.is_contiguous() and two_i % 2 == 0
        if out is None:
            out = torch.empty((rows, two_i // 2), dtype=BF16, device=gu.device)
        assert out.is_contiguous() and tuple(out.shape) == (rows, two_i // 2)
        check(self.lib.svlm_silu_mul(_ptr(gu), _ptr(out), rows, two_i // 2, _stream()), "svlm_silu_mul")
        return out

    def gather_rows(self, table, alt, idx, out, idx_off=None):
        _req(table, BF16, "gather.table", 2); _req(idx, torch.int32, "gather.idx", 1); _req(out, BF16, "gather.out", 2)
        rows, cols = out.shape
        assert table.is_contiguous() and out.is_contiguous() and table.shape[1] == cols
        if alt is not None:
            _req(alt, BF16, "gather.alt", 2); assert alt.is_contiguous() and alt.shape[1] == cols
        if idx_off is None:
            assert idx.numel() >= rows
        else:
            _req(idx_off, torch.int32, "gather.idx_off")
        check(self.lib.svlm_gather_rows(_ptr(table), _ptr(alt), _ptr(idx), _ptr(idx_off), _ptr(out), rows, cols, _stream()),
              "svlm_gather_rows")
        return out

    def patchify_u8(self, frames, patch=14, temporal=2, merge=2, mean=(0.48145466, 0.4578275, 0.40821073),
                    std=(0.26862954, 0.26130258, 0.27577711), out=None):
        """uint8 (T, 3, H, W) frames on the device -> (bf16 patches (N, 3*temporal*patch^2), [[gt, gh, gw]])."""
        _req(frames, torch.uint8, "patchify.frames", 4)
        T, C, H, W = frames.shape
        assert C == 3 and frames.is_contiguous()
        gt, gh, gw = (T + temporal - 1) // temporal, H // patch, W // patch
        cols = 3 * temporal * patch * patch
        if out is None:
            out = torch.empty((gt * gh * gw, cols), dtype=BF16, device=frames.device)
        _req(out, BF16, "patchify.out", 2)
        assert out.is_contiguous() and tuple(out.shape) == (gt * gh * gw, cols)
        f32 = lambda v: float(torch.tensor(v, dtype=torch.float32))          # the exact fp32 constants torch.tensor(list) holds
        check(self.lib.svlm_patchify_u8(_ptr(frames), _ptr(out), T, H, W, patch, temporal, merge, f32(mean[0]), f32(mean[1]), f32(mean[2]),
                                        f32(std[0]), f32(std[1]), f32(std[2]), _stream()), "svlm_patchify_u8")
        return out, [[gt, gh, gw]]

    def resize_tables(self, in_size, out_size):
        return aa_resize_tables(in_size, out_size)

    def resize_u8(self, frames, h, w, out=None):
        """uint8 (T, C, H, W) frames on the device -> uint8 (T, C, h, w): torchvision's resize(BICUBIC, antialias=True)."""
        _req(frames, torch.uint8, "resize.frames", 4)
        T, Cc, H, W = frames.shape
        assert frames.is_contiguous()
        if out is None:
            out = torch.empty((T, Cc, h, w), dtype=torch.uint8, device=frames.device)
        _req(out, torch.uint8, "resize.out", 4)
        assert out.is_contiguous() and tuple(out.shape) == (T, Cc, h, w)
        key = (str(frames.device), H, W, h, w)
        tabs = self._resize_tabs.get(key)
        if tabs is None:
            tx, ty = self.resize_tables(W, w), self.resize_tables(H, h)
            tabs = tuple(torch.from_numpy(a).to(frames.device) for a in tx + ty)
            if len(self._resize_tabs) > 16:
                self._resize_tabs.clear()
            self._resize_tabs[key] = tabs
        xmin, xsize, wx, ymin, ysize, wy = tabs
        need = self.lib.svlm_resize_ws_bytes(T * Cc, H, w)
        ws = self._gemm_ws.get("rz" + str(frames.device))
        if ws is None or ws.numel() * 4 < need:
            ws = torch.empty(max(need // 4, 1 << 18), dtype=torch.float32, device=frames.device)
            self._gemm_ws["rz" + str(frames.device)] = ws
        check(self.lib.svlm_resize_bicubic_aa_u8(_ptr(frames), _ptr(out), T * Cc, H, W, h, w, _ptr(xmin), _ptr(xsize), _ptr(wx),
                                                 wx.shape[1], _ptr(ymin), _ptr(ysize), _ptr(wy), wy.shape[1], _ptr(ws),
                                                 ws.numel() * 4, _stream()), "svlm_resize_bicubic_aa_u8")
        return out

    # ------------------------------------------------------------------ ViT
    def vit_rope(self, qkv, cosT, sinT, H, d):
        _req(qkv, BF16, "vit_rope.qkv", 2); _req(cosT, torch.float32, "vit_rope.cos", 2); _req(sinT, torch.float32, "vit_rope.sin", 2)
        N = qkv.shape[0]
        assert qkv.is_contiguous() and qkv.shape[1] == 3 * H * d
        assert cosT.is_contiguous() and sinT.is_contiguous() and tuple(cosT.shape) == (N, d // 2) == tuple(sinT.shape)
        check(self.lib.svlm_vit_rope(_ptr(qkv), _ptr(cosT), _ptr(sinT), N, H, d, _stream()), "svlm_vit_rope")
        return qkv

    def vit_attn(self, qkv, n_seq, seq_len, H, d, scale, out=None):
        _req(qkv, BF16, "vit_attn.qkv", 2)
        N = qkv.shape[0]
        assert qkv.is_contiguous() and qkv.shape[1] == 3 * H * d and N == n_seq * seq_len
        if out is None:
            out = torch.empty((N, H * d), dtype=BF16, device=qkv.device)
        assert out.is_contiguous() and tuple(out.shape) == (N, H * d)
        check(self.lib.svlm_vit_attn(_ptr(qkv), _ptr(out), n_seq, seq_len, H, d, float(scale), _stream()), "svlm_vit_attn")
        return out

    # ------------------------------------------------------------------ rope table / KV pool
    def mrope_table(self, pos3, inv_freq, rope_cs, start, count, sections):
        """pos3: (3, stride) int32 or float32; rope_cs: (cap, D) bf16."""
        _req(inv_freq, torch.float32, "mrope.inv_freq", 1); _req(rope_cs, BF16, "mrope.rope_cs", 2)
        assert pos3.is_cuda and pos3.dim() == 2 and pos3.shape[0] == 3 and pos3.is_contiguous()
        D = rope_cs.shape[1]
        assert rope_cs.is_contiguous() and inv_freq.numel() == D // 2
        assert 0 <= start and start + count <= min(pos3.shape[1], rope_cs.shape[0]), (start, count, pos3.shape, rope_cs.shape)
        is_f = pos3.dtype == torch.float32
        assert is_f or pos3.dtype == torch.int32
        check(self.lib.svlm_mrope_table(0 if is_f else _ptr(pos3), _ptr(pos3) if is_f else 0, pos3.shape[1], _ptr(inv_freq),
                                        _ptr(rope_cs), start, count, D, sections[0], sections[1], sections[2], _stream()),
              "svlm_mrope_table")

    def rope_index_ws(self, max_len, max_spans, device):
        return torch.zeros((self.lib.svlm_rope_index_ws_bytes(int(max_len), int(max_spans)) + 3) // 4, dtype=torch.int32, device=device)

    def rope_index(self, ids, L, grids, n_grids, merge, video_token_id, vision_start_token_id, pos3, ws, n_extra=0, second_per_grid_t=1.0,
                   tokens_per_second=2.0):
        """pos3 (3, stride) int32 or fp32 <- M-RoPE ids of ids[:L] (+ n_extra rows continuing the text run); ws[0] = status."""
        _req(ids, torch.int32, "rope_index.ids", 1); _req(grids, torch.int32, "rope_index.grids", 2); _req(ws, torch.int32, "rope_index.ws", 1)
        assert ids.numel() >= L and grids.shape[0] >= n_grids and grids.shape[1] == 3 and grids.is_contiguous()
        assert pos3.is_cuda and pos3.dim() == 2 and pos3.shape[0] == 3 and pos3.is_contiguous() and L + n_extra <= pos3.shape[1]
        is_f = pos3.dtype == torch.float32
        assert is_f or pos3.dtype == torch.int32
        assert ws.numel() * 4 >= self.lib.svlm_rope_index_ws_bytes(max(int(L), 1), int(n_grids))
        check(self.lib.svlm_rope_index(_ptr(ids), int(L), _ptr(grids), int(n_grids), int(merge), int(video_token_id), int(vision_start_token_id),
                                       0 if is_f else _ptr(pos3), _ptr(pos3) if is_f else 0, pos3.shape[1], float(second_per_grid_t),
                                       float(tokens_per_second), int(n_extra), _ptr(ws), ws.numel() * 4, _stream()), "svlm_rope_index")

    def evict_plan(self, ids, policy, round_i=0, text_round=16, visual_round=16, text_sink=None, text_sliding_window=None,
                   assistant_start_bias=3, assistant_end_bias=2, sink=4, window=2048, kv_len=0, device="cuda"):
        """Span finder + eviction policy on the device (svlm_evict_plan).  ids: host int sequence (uploaded here) -> (ops, new ids) with
        ops = [("prune", s, e) | ("move", s, e, dst)], the reference's eviction indices in order.  One small D2H at the end."""
        import ctypes
        import numpy as np
        from .spans import TOKEN_IDS
        arr = np.asarray(ids, dtype=np.int32).reshape(-1)
        L = int(arr.shape[0])
        d_ids = torch.from_numpy(arr).to(device)
        ws = torch.zeros((self.lib.svlm_evict_plan_ws_bytes(L) + 3) // 4, dtype=torch.int32, device=device)
        pt = TOKEN_IDS["previous text"]
        toks = (ctypes.c_int * 11)(TOKEN_IDS["<|im_start|>"], TOKEN_IDS["<|im_end|>"], TOKEN_IDS["user"], TOKEN_IDS["assistant"],
                                   TOKEN_IDS["<|vision_start|>"], TOKEN_IDS["<|vision_end|>"], TOKEN_IDS["<|video_pad|>"], TOKEN_IDS["\n"],
                                   pt[0], pt[1], TOKEN_IDS["Time"])
        check(self.lib.svlm_evict_plan(_ptr(d_ids), L, {"sink_window": 0, "structural": 1}[policy], int(round_i), int(text_round), int(visual_round),
                                       -1 if text_sink is None else int(text_sink), -1 if text_sliding_window is None else int(text_sliding_window),
                                       int(assistant_start_bias), int(assistant_end_bias), int(sink), int(window), int(kv_len),
                                       ctypes.cast(toks, ctypes.c_void_p), _ptr(ws), ws.numel() * 4, _stream()), "svlm_evict_plan")
        out = ws[:68].cpu().numpy()
        status, n_ops, new_len, which = int(out[0]), int(out[1]), int(out[2]), int(out[3])
        if status:
            raise _lib.SvlmError(f"svlm_evict_plan: status {status} (1: a span the policy needs is missing, 2: more than 16 edits)")
        ops = []
        for k in range(n_ops):
            t, a, b, c = (int(v) for v in out[4 + 4 * k: 8 + 4 * k])
            ops.append(("prune", a, b) if t == 1 else ("move", a, b, c))
        res = (ws[68:68 + new_len] if which == 1 else d_ids[:new_len]).cpu().numpy().astype(np.int64)
        return ops, res

    @staticmethod
    def _planes(pool, layer):
        # pool (layers, 2, Hkv, n_slots, D)
        return pool[layer, 0], pool[layer, 1]

    def kv_append(self, k_new, v_new, pool, layer, slot_of, start, T, len_dev=None):
        """k_new/v_new: (T, Hkv*D) row views (may be column slices of the fused qkv buffer)."""
        _req(pool, BF16, "kv_append.pool", 5); _req(slot_of, torch.int32, "kv_append.slot_of", 1)
        _req(k_new, BF16, "kv_append.k", 2); _req(v_new, BF16, "kv_append.v", 2)
        _, _, Hkv, n_slots, D = pool.shape
        assert pool.is_contiguous() and k_new.shape[0] >= T and k_new.shape[1] == Hkv * D == v_new.shape[1]
        if len_dev is None:
            assert 0 <= start and start + T <= slot_of.numel(), (start, T, slot_of.numel())
        kp, vp = self._planes(pool, layer)
        check(self.lib.svlm_kv_append(_ptr(k_new), k_new.stride(0), _ptr(v_new), v_new.stride(0), _ptr(kp), _ptr(vp), _ptr(slot_of),
                                      _ptr(len_dev), start, T, Hkv, D, n_slots, _stream()), "svlm_kv_append")

    def kv_move_rows(self, pool, src, dst):
        _req(pool, BF16, "kv_move.pool", 5); _req(src, torch.int32, "kv_move.src", 1); _req(dst, torch.int32, "kv_move.dst", 1)
        Ly, two, Hkv, n_slots, D = pool.shape
        n = src.numel()
        assert pool.is_contiguous() and dst.numel() == n
        check(self.lib.svlm_kv_move_rows(_ptr(pool), Ly * two * Hkv, n_slots, D, _ptr(src), _ptr(dst), n, _stream()), "svlm_kv_move_rows")

    def kv_gather(self, pool, layer, which, slot_of, L):
        _req(pool, BF16, "kv_gather.pool", 5); _req(slot_of, torch.int32, "kv_gather.slot_of", 1)
        _, _, Hkv, n_slots, D = pool.shape
        assert slot_of.numel() >= L
        out = torch.empty((Hkv, L, D), dtype=BF16, device=pool.device)
        check(self.lib.svlm_kv_gather(_ptr(pool[layer, which]), _ptr(slot_of), _ptr(out), L, Hkv, D, n_slots, _stream()), "svlm_kv_gather")
        return out

    # ------------------------------------------------------------------ attention
    def decode_attn_ws(self, Hq, max_len, chunk, device):
        nbytes = self.lib.svlm_decode_attn_ws_bytes(Hq, max_len, chunk)
        if nbytes < 0:
            raise _lib.SvlmError("svlm_decode_attn_ws_bytes: bad arguments")
        return torch.empty((nbytes // 4,), dtype=torch.float32, device=device)

    @staticmethod
    def _lin_args(lin, layer, Hkv, D, need_rows, what):
        """lin = (planes (n_layers, 2, Hkv, lin_rows, D) bf16, lin_state int32[2]) or None -> (k_lin, v_lin, lin_rows, lin_state)"""
        if lin is None:
            return None, None, 0, None
        planes, lin_len = lin
        _req(planes, BF16, what + ".lin", 5); _req(lin_len, torch.int32, what + ".lin_state", 1)
        assert lin_len.numel() == 2, "lin_state = {rows rotated, appended rows follow}"
        assert planes.shape[1] == 2 and planes.shape[2] == Hkv and planes.shape[4] == D and planes.is_contiguous(), tuple(planes.shape)
        assert planes.shape[3] % 16 == 0 and planes.shape[3] >= need_rows, (planes.shape[3], need_rows)
        return planes[layer, 0], planes[layer, 1], planes.shape[3], lin_len

    def decode_attn(self, q, pool, layer, slot_of, rope_cs, out, ws, Hq, max_len, chunk, scale, length=0, len_dev=None, lin=None):
        """length = host-known KV length INCLUDING the appended row (len_dev None), or the constant
        added to *len_dev (normally 1).  lin: the cache's linear planes (KVPool.lin_args()): key ranges below *lin_len are streamed
        from the rotated copy the prefill left there (svlm_decode_attn_lin)."""
        _req(q, BF16, "decode_attn.q"); _req(out, BF16, "decode_attn.out"); _req(ws, torch.float32, "decode_attn.ws", 1)
        _req(rope_cs, BF16, "decode_attn.rope_cs", 2); _req(slot_of, torch.int32, "decode_attn.slot_of", 1)
        _, _, Hkv, n_slots, D = pool.shape
        assert q.numel() == Hq * D == out.numel() and q.is_contiguous() and out.is_contiguous()
        assert max_len <= slot_of.numel() and max_len <= rope_cs.shape[0] and rope_cs.shape[1] == D
        need = self.lib.svlm_decode_attn_ws_bytes(Hq, max_len, chunk)
        assert ws.numel() * 4 >= need, (ws.numel() * 4, need)
        if len_dev is None:
            assert 0 < length <= max_len, (length, max_len)
        kp, vp = self._planes(pool, layer)
        kl, vl, lin_rows, lin_len = self._lin_args(lin, layer, Hkv, D, max_len, "decode_attn")
        check(self.lib.svlm_decode_attn_lin(_ptr(q), _ptr(kp), _ptr(vp), _ptr(slot_of), _ptr(rope_cs), _ptr(len_dev), length,
                                            _ptr(kl), _ptr(vl), lin_rows, _ptr(lin_len), _ptr(out), _ptr(ws), Hq, Hkv, D, n_slots, max_len,
                                            chunk, float(scale), _stream()), "svlm_decode_attn_lin")
        return out

    def prefill_attn(self, q, pool, layer, slot_of, rope_cs, out, T, L, Hq, scale, k_new=None, v_new=None, lin=None):
        """k_new / v_new (T, Hkv*D) row views: the chunk's un-rotated K/V rows, appended to their slots by the same launch that
        rotates the keys (otherwise they must already be in the pool, `kv_append`).  lin: the cache's linear planes
        (KVPool.lin_args()): the rotated keys / gathered values of rows [0, L) are left there for the decode steps."""
        _req(q, BF16, "prefill_attn.q", 2); _req(out, BF16, "prefill_attn.out", 2)
        _req(rope_cs, BF16, "prefill_attn.rope_cs", 2); _req(slot_of, torch.int32, "prefill_attn.slot_of", 1)
        _, _, Hkv, n_slots, D = pool.shape
        assert q.shape[0] >= T and q.shape[1] == Hq * D and out.shape[0] >= T and out.shape[1] == Hq * D
        assert T <= L <= slot_of.numel() and L <= rope_cs.shape[0] and rope_cs.shape[1] == D
        kp, vp = self._planes(pool, layer)
        need = self.lib.svlm_prefill_attn_ws_bytes(T, L, Hq, Hkv)
        key = "pf" + str(q.device)
        ws = self._gemm_ws.get(key)
        if ws is None or ws.numel() * 2 < need:
            ws = torch.empty(max(need // 2, 1 << 20), dtype=BF16, device=q.device)
            self._gemm_ws[key] = ws
        kv_stride = 0
        if k_new is not None or v_new is not None:
            _req(k_new, BF16, "prefill_attn.k_new", 2); _req(v_new, BF16, "prefill_attn.v_new", 2)
            assert k_new.shape[0] >= T and k_new.shape[1] == Hkv * D == v_new.shape[1] and k_new.stride(1) == 1 == v_new.stride(1)
            assert k_new.stride(0) == v_new.stride(0)
            kv_stride = k_new.stride(0)
        kl, vl, lin_rows, lin_len = self._lin_args(lin, layer, Hkv, D, L, "prefill_attn")
        check(self.lib.svlm_prefill_attn_ropeload_lin(_ptr(q), q.stride(0), _ptr(k_new), _ptr(v_new), kv_stride, _ptr(kp), _ptr(vp),
                                                      _ptr(slot_of), _ptr(rope_cs), _ptr(out), out.stride(0), T, L, Hq, Hkv, D, n_slots,
                                                      float(scale), _ptr(ws), ws.numel() * 2, _ptr(kl), _ptr(vl), lin_rows, _ptr(lin_len),
                                                      _stream()), "svlm_prefill_attn_ropeload_lin")
        return out

    # ------------------------------------------------------------------ sampling
    def mark_seen(self, ids, n, seen):
        _req(ids, torch.int32, "mark_seen.ids", 1); _req(seen, torch.uint8, "mark_seen.seen", 1)
        assert ids.numel() >= n
        check(self.lib.svlm_mark_seen(_ptr(ids), n, _ptr(seen), seen.numel(), _stream()), "svlm_mark_seen")

    def sampling_ws(self, V, device):
        n = max(self.lib.svlm_argmax_ws_bytes(), self.lib.svlm_dec_lm_head_ws_bytes(V))
        return torch.empty((n // 4,), dtype=torch.float32, device=device)

    def penalty_argmax(self, logits, seen, penalty, suppress, tok_buf, state, advance_kv, ws):
        _req(logits, torch.float32, "argmax.logits", 1); _req(tok_buf, torch.int32, "argmax.tok_buf", 1)
        _req(state, torch.int32, "argmax.state", 1)
        if seen is not None:
            _req(seen, torch.uint8, "argmax.seen", 1); assert seen.numel() == logits.numel()
        n_sup = 0
        if suppress is not None:
            _req(suppress, torch.int32, "argmax.suppress", 1)
            n_sup = suppress.numel()
        assert state.numel() >= 2
        _req(ws, torch.float32, "argmax.ws", 1)
        assert ws.numel() * 4 >= self.lib.svlm_argmax_ws_bytes()
        check(self.lib.svlm_penalty_argmax(_ptr(logits), logits.numel(), _ptr(seen), float(penalty), _ptr(suppress), n_sup,
                                           _ptr(tok_buf), _ptr(state), int(advance_kv), _ptr(ws), _stream()), "svlm_penalty_argmax")

    def penalty_sample(self, logits, seen, penalty, suppress, temperature, top_k, top_p, rng, tok_buf, state, advance_kv, ws):
        """One draw from the processed distribution (repetition penalty -> temperature -> top-k -> top-p -> multinomial) + token
        feedback; rng: uint32[2] seed in device memory."""
        _req(logits, torch.float32, "sample.logits", 1); _req(tok_buf, torch.int32, "sample.tok_buf", 1)
        _req(state, torch.int32, "sample.state", 1); _req(rng, torch.int32, "sample.rng", 1); _req(ws, torch.float32, "sample.ws", 1)
        if seen is not None:
            _req(seen, torch.uint8, "sample.seen", 1); assert seen.numel() == logits.numel()
        n_sup = 0
        if suppress is not None:
            _req(suppress, torch.int32, "sample.suppress", 1)
            n_sup = suppress.numel()
        assert state.numel() >= 2 and rng.numel() >= 2 and ws.numel() * 4 >= self.lib.svlm_argmax_ws_bytes()
        check(self.lib.svlm_penalty_sample(_ptr(logits), logits.numel(), _ptr(seen), float(penalty), _ptr(suppress), n_sup, float(temperature),
                                           int(top_k), float(top_p), _ptr(rng), _ptr(tok_buf), _ptr(state), int(advance_kv), _ptr(ws),
                                           _stream()), "svlm_penalty_sample")

    # ------------------------------------------------------------------ fused decode step
    def dec_qkv(self, x, ln_w, eps, W, bias, q_out, pool, layer, slot_of, qd, kd, length=0, len_dev=None, lin=None):
        """lin: the cache's linear planes (KVPool.lin_args()): the new row goes there too (key un-rotated), so that the decode attention
        finds the rows appended since the prefill without the slot table."""
        _req(x, BF16, "dec_qkv.x", 1); _req(ln_w, BF16, "dec_qkv.ln_w", 1); _req(W, BF16, "dec_qkv.W", 2)
        _req(bias, BF16, "dec_qkv.bias", 1); _req(q_out, BF16, "dec_qkv.q_out", 1); _req(slot_of, torch.int32, "dec_qkv.slot_of", 1)
        _, _, Hkv, n_slots, D = pool.shape
        N, K = W.shape
        assert N == qd + 2 * kd == bias.numel() and kd == Hkv * D and x.numel() == K == ln_w.numel() and q_out.numel() >= qd
        if len_dev is None:
            assert 0 <= length < slot_of.numel()
        kp, vp = self._planes(pool, layer)
        kl, vl, lin_rows, _ = self._lin_args(lin, layer, Hkv, D, length + 1 if len_dev is None else 1, "dec_qkv")
        check(self.lib.svlm_dec_qkv_lin(_ptr(x), _ptr(ln_w), float(eps), _ptr(W), W.stride(0), _ptr(bias), _ptr(q_out), _ptr(kp), _ptr(vp),
                                        _ptr(slot_of), _ptr(len_dev), int(length), K, qd, kd, D, n_slots, _ptr(kl), _ptr(vl), lin_rows,
                                        _stream()), "svlm_dec_qkv_lin")

    def dec_gate_up(self, x, ln_w, eps, W, h):
        _req(x, BF16, "dec_gate_up.x", 1); _req(ln_w, BF16, "dec_gate_up.ln_w", 1); _req(W, BF16, "dec_gate_up.W", 2); _req(h, BF16, "dec_gate_up.h", 1)
        N, K = W.shape
        assert N % 2 == 0 and h.numel() == N // 2 and x.numel() == K == ln_w.numel()
        check(self.lib.svlm_dec_gate_up(_ptr(x), _ptr(ln_w), float(eps), _ptr(W), W.stride(0), _ptr(h), N // 2, K, _stream()), "svlm_dec_gate_up")

    # ---- persistent decode-layer tail (csrc/dec_tail.hip)
    def dec_tail_supported(self, H, I, qd, kd, grid=0):
        """True when svlm_dec_tail has a build for this layer geometry (its weights must fit the CUs' register files)."""
        if not grid:
            grid = self.lib.svlm_device_cus()
        return self.lib.svlm_dec_tail_supported(int(H), int(I), int(qd), int(kd), int(grid)) == 1

    def dec_tail_ws(self, H, I, n_layers, device):
        """Granule workspace of a decode step's tails: [256-B status block | one granule block per layer], zero-initialised."""
        n = self.lib.svlm_dec_tail_ws_bytes(int(H), int(I), int(n_layers))
        if n <= 0:
            raise _lib.SvlmError(f"svlm_dec_tail_ws_bytes({H}, {I}, {n_layers}) -> {n}")
        return torch.zeros(n // 8, dtype=torch.int64, device=device)

    def dec_tail_reset(self, ws, H, I, n_layers):
        _req(ws, torch.int64, "dec_tail_reset.ws", 1)
        assert ws.numel() * 8 >= self.lib.svlm_dec_tail_ws_bytes(int(H), int(I), int(n_layers))
        check(self.lib.svlm_dec_tail_reset(_ptr(ws), int(H), int(I), int(n_layers), _stream()), "svlm_dec_tail_reset")

    def dec_tail(self, attn, x, o_w, ln2, gu_w, down_w, eps, ws, layer, n_layers, nxt=None, grid=0, stamps=None):
        """One layer's tail.  `nxt` = (ln1, qkv_w, qkv_b, q_out, pool, layer_index, slot_of, qd, kd, length, len_dev) of the NEXT layer's
        QKV phase, or None for the last layer."""
        _req(attn, BF16, "dec_tail.attn", 1); _req(x, BF16, "dec_tail.x", 1); _req(o_w, BF16, "dec_tail.o_w", 2); _req(ln2, BF16, "dec_tail.ln2", 1)
        _req(gu_w, BF16, "dec_tail.gu_w", 2); _req(down_w, BF16, "dec_tail.down_w", 2); _req(ws, torch.int64, "dec_tail.ws", 1)
        H, qd = o_w.shape
        I = gu_w.shape[0] // 2
        assert gu_w.shape == (2 * I, H) and down_w.shape == (H, I) and attn.numel() == qd and x.numel() == H == ln2.numel()
        assert ws.numel() * 8 >= self.lib.svlm_dec_tail_ws_bytes(H, I, int(n_layers)) > 0 and 0 <= layer < n_layers
        if nxt is None:
            args = (0, 0, 0, 0, 0, 0, 0, 0, 0, 0, H, I, qd, 0, 0, 0)
        else:
            ln1, qkv_w, qkv_b, q_out, pool, li, slot_of, qd2, kd, length, len_dev = nxt
            _req(ln1, BF16, "dec_tail.ln1", 1); _req(qkv_w, BF16, "dec_tail.qkv_w", 2); _req(qkv_b, BF16, "dec_tail.qkv_b", 1)
            _req(q_out, BF16, "dec_tail.q_out", 1); _req(slot_of, torch.int32, "dec_tail.slot_of", 1); _req(pool, BF16, "dec_tail.pool", 5)
            _, _, Hkv, n_slots, D = pool.shape
            assert qkv_w.shape == (qd2 + 2 * kd, H) and qkv_b.numel() == qd2 + 2 * kd and kd == Hkv * D and qd2 == qd
            assert ln1.numel() == H and q_out.numel() >= qd
            if len_dev is None:
                assert 0 <= length < slot_of.numel()
            kp, vp = self._planes(pool, li)
            args = (_ptr(ln1), _ptr(qkv_w), qkv_w.stride(0), _ptr(qkv_b), _ptr(q_out), _ptr(kp), _ptr(vp), _ptr(slot_of), _ptr(len_dev),
                    int(length), H, I, qd, kd, D, n_slots)
        check(self.lib.svlm_dec_tail(_ptr(attn), _ptr(x), _ptr(o_w), o_w.stride(0), _ptr(ln2), _ptr(gu_w), gu_w.stride(0), _ptr(down_w),
                                     down_w.stride(0), *args, float(eps), _ptr(ws), int(layer), int(n_layers), int(grid), _ptr(stamps), _stream()),
              "svlm_dec_tail")

    @staticmethod
    def dec_tail_views(ws, H, I, layer):
        """(status, x' granules, h granules, x'' granules) of one layer's block, for tests: int64 views, low half = 2 x bf16, high = tag."""
        per = ((H // 2 * 2 + I // 2) * 8 + 255) // 256 * 256 // 8
        base = 32 + per * layer
        return ws[:1], ws[base:base + H // 2], ws[base + H // 2:base + H // 2 + I // 2], ws[base + H // 2 + I // 2:base + H + I // 2]

    def dec_lm_head(self, x, ln_w, eps, W, logits, seen, penalty, suppress, ws, temperature=None, rng=None, state=None):
        """`temperature` + `rng` (uint32[2] seed on the device) + `state`: Gumbel-max temperature sampling in the candidates."""
        _req(x, BF16, "dec_lm_head.x", 1); _req(ln_w, BF16, "dec_lm_head.ln_w", 1); _req(W, BF16, "dec_lm_head.W", 2)
        _req(logits, torch.float32, "dec_lm_head.logits", 1); _req(ws, torch.float32, "dec_lm_head.ws", 1)
        V, K = W.shape
        assert logits.numel() == V and x.numel() == K == ln_w.numel()
        assert ws.numel() * 4 >= self.lib.svlm_dec_lm_head_ws_bytes(V)
        n_sup = 0
        if seen is not None:
            _req(seen, torch.uint8, "dec_lm_head.seen", 1); assert seen.numel() == V
        if suppress is not None:
            _req(suppress, torch.int32, "dec_lm_head.suppress", 1)
            n_sup = suppress.numel()
        if rng is not None:
            _req(rng, torch.int32, "dec_lm_head.rng", 1); _req(state, torch.int32, "dec_lm_head.state", 1)
            check(self.lib.svlm_dec_lm_head_sample(_ptr(x), _ptr(ln_w), float(eps), _ptr(W), W.stride(0), _ptr(logits), _ptr(seen), float(penalty),
                                                   _ptr(suppress), n_sup, _ptr(ws), V, K, float(temperature), _ptr(rng), _ptr(state), _stream()),
                  "svlm_dec_lm_head_sample")
            return
        check(self.lib.svlm_dec_lm_head(_ptr(x), _ptr(ln_w), float(eps), _ptr(W), W.stride(0), _ptr(logits), _ptr(seen), float(penalty),
                                        _ptr(suppress), n_sup, _ptr(ws), V, K, _stream()), "svlm_dec_lm_head")

    def argmax_finish(self, ws, V, seen, tok_buf, state, advance_kv):
        _req(ws, torch.float32, "argmax_finish.ws", 1); _req(tok_buf, torch.int32, "argmax_finish.tok_buf", 1)
        _req(state, torch.int32, "argmax_finish.state", 1)
        check(self.lib.svlm_argmax_finish(_ptr(ws), V, _ptr(seen), _ptr(tok_buf), _ptr(state), int(advance_kv), _stream()), "svlm_argmax_finish")
