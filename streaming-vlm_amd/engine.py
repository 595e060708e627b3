"""The per-chunk hot path on HIP kernels: ViT -> merger -> LLM prefill -> greedy decode.

Host-side restatement of the reference's model forwards with every arithmetic step routed to the
C ABI (``ops.HipOps``):
  ViT        qwen2/vision_forward.py:53-80 (+ block :36-50, attention :6-34)
  model      qwen2/model_forward.py:6-150 (embed gather, vision splice, shrink-mode positions)
  decoder    qwen2/language_forward.py:66-334
  generate   generate/streaming_generate_qwen.py:8-127 (_sample loop)
Differences that do not change results: lm_head runs on the last row only (the reference computes
all T rows, model_forward.py:243, and reads one, streaming_generate_qwen.py:73); decode steps feed
the sampled token back on the device and replay one captured HIP graph instead of returning to
Python per token; positions are computed once per chunk instead of once per forward.
"""
from __future__ import annotations

import math
import os
import weakref
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch

from ._lib import ACT_GELU_ERF, ACT_QUICK_GELU, ACT_SWIGLU
from .config import ModelConfig
from .kv_pool import KVPool
from .positions import rope_index_1d, rope_index_qwen2, rope_index_qwen2_5
from .weights import EngineWeights

BF16 = torch.bfloat16


@dataclass
class GenerateOutput:
    sequences: List[int]
    past_key_values: KVPool
    logits: Optional[List[torch.Tensor]] = None      # fp32 last-row logits per forward (tests)
    n_new: int = 0
    own: Optional[List[int]] = None                  # with force_tokens: the engine's own greedy choice at every step


def _window_plan(grid, merge: int, window_size: int, patch: int):
    """Window order of a Qwen2.5-VL tower (transformers.vision_utils.get_vision_window_index, the helper behind
    Qwen2_5_VisionTransformerPretrainedModel.get_window_index that the reference calls at qwen2_5/vision_forward.py:65):
    returns (window_index over merged tokens, window lengths in PATCHES, frame lengths in patches)."""
    vw = window_size // merge // patch
    unit = merge * merge
    index, win_len, frame_len = [], [], []
    base = 0
    for t, h, w in grid:
        gh, gw = h // merge, w // merge
        idx = np.arange(t * gh * gw).reshape(t, gh, gw)
        ph, pw = vw - gh % vw, vw - gw % vw                  # the helper pads a FULL window when the size divides evenly
        padded = np.full((t, gh + ph, gw + pw), -100, dtype=np.int64)
        padded[:, :gh, :gw] = idx
        nh, nw = (gh + ph) // vw, (gw + pw) // vw
        padded = padded.reshape(t, nh, vw, nw, vw).transpose(0, 1, 3, 2, 4).reshape(t, nh * nw, vw * vw)
        for row in padded.reshape(-1, vw * vw):
            keep = row[row != -100]
            if keep.size:                                    # empty windows vanish (unique_consecutive on the cu_seqlens)
                index.append(keep + base)
                win_len.append(int(keep.size) * unit)
        base += t * gh * gw
        frame_len += [h * w] * t
    return np.concatenate(index).astype(np.int32), win_len, frame_len


def _runs(lengths):
    """Consecutive equal lengths -> [(first row, count, length)]: each run is one svlm_vit_attn launch."""
    out, row, i = [], 0, 0
    while i < len(lengths):
        j = i
        while j < len(lengths) and lengths[j] == lengths[i]:
            j += 1
        out.append((row, j - i, lengths[i]))
        row += (j - i) * lengths[i]
        i = j
    return out


class _VitRun:
    """One ViT + merger pass in resumable stages (patch embed at construction, `blocks(lo, hi)`, `finish()`), so that a
    look-ahead pass can be cut in two: most blocks underneath the current chunk's decode steps, the tail in the host
    turnaround between chunks when the GPU has nothing else to do."""

    def __init__(self, eng, pixel_values, grid_thw):
        o, w, vc = eng.ops, eng.w, eng.cfg.vision
        grid = [[int(v) for v in g] for g in (grid_thw.tolist() if hasattr(grid_thw, "tolist") else grid_thw)]
        N = sum(t * h * ww for t, h, ww in grid)
        if pixel_values.shape[0] != N or pixel_values.shape[1] != vc.patch_dim:
            raise ValueError(f"pixel_values {tuple(pixel_values.shape)} does not match grid {grid} (N={N}, patch_dim={vc.patch_dim})")
        sizes = {h * ww for _, h, ww in grid}
        if len(sizes) != 1:
            raise ValueError("all temporal grids of one call must share h*w")
        self.eng, self.N = eng, N
        self.seq_len = sizes.pop()
        self.n_seq = N // self.seq_len
        pix = pixel_values.to(device=eng.device, dtype=BF16).contiguous()
        self.cosT, self.sinT = eng._vit_rope(grid)
        E = vc.embed_dim
        self.x = o.gemm(pix, w.patch_embed)
        self.h = torch.empty_like(self.x)
        self.qkv = torch.empty((N, 3 * E), dtype=BF16, device=eng.device)
        self.a = torch.empty((N, E), dtype=BF16, device=eng.device)
        self.q25 = vc.arch == "qwen2_5"
        self._h_is_norm1 = False
        if self.q25:
            # window order (qwen2_5/vision_forward.py:65-83): groups of merge^2 patches are gathered so that every
            # attention window is one contiguous run of rows; the rotary tables follow the same permutation
            plan = eng._vit_windows(grid)
            m2 = vc.spatial_merge_size ** 2
            self.x = o.gather_rows(self.x.view(N // m2, m2 * E), None, plan["index"],
                                   torch.empty((N // m2, m2 * E), dtype=BF16, device=eng.device)).view(N, E)
            self.cosT, self.sinT = plan["cos"], plan["sin"]
            self.win_runs, self.full_runs, self.rev = plan["win_runs"], plan["full_runs"], plan["reverse"]
            self.f = torch.empty((N, 2 * vc.mlp_padded), dtype=BF16, device=eng.device)
            self.g = torch.empty((N, vc.mlp_padded), dtype=BF16, device=eng.device)
        else:
            self.f = torch.empty((N, vc.mlp_hidden), dtype=BF16, device=eng.device)
        self.fp8 = eng.vit_fp8
        if self.fp8:            # fp8 operand images + one fp32 scale per row of every GEMM input
            f8 = torch.float8_e4m3fn
            self.q_e = torch.empty((N, E), dtype=f8, device=eng.device)
            self.q_f = torch.empty((N, vc.mlp_hidden), dtype=f8, device=eng.device)
            self.s_row = torch.empty((N,), dtype=torch.float32, device=eng.device)

    def _blocks_fp8(self, lo: int, hi: int):
        """The Qwen2 tower with its four Linears per block on the fp8 MFMA path (BASELINE configs[4]): every GEMM input is
        quantised per row right before its GEMM (svlm_quant_rows_fp8), weights were quantised per output channel at load."""
        o, vc = self.eng.ops, self.eng.cfg.vision
        Hh, d = vc.num_heads, vc.head_dim
        scale = 1.0 / math.sqrt(d)
        x, h, qkv, a, f = self.x, self.h, self.qkv, self.a, self.f
        blocks = self.eng.w.vit
        w8 = self.eng.vit_w8
        # The two LayerNorm outputs of a block leave their producer (the split-K reduce of proj / fc2) already quantised: e4m3 + row
        # scales beside the bf16 row (svlm_gemm_fp8_normq), so only the attention output and the GELU output still go through the
        # stand-alone quantiser.  `_hq_ready`: q_e / s_row hold the quantised norm1 row of the next block.
        if lo < hi and not self._h_is_norm1:
            o.layernorm(x, blocks[lo]["n1w"], blocks[lo]["n1b"], 1e-6, out=h)
            self._hq_ready = False
        for bi in range(lo, hi):
            bw, b8 = blocks[bi], w8[bi]
            if not getattr(self, "_hq_ready", False):
                o.quant_rows_fp8(h, self.q_e, self.s_row)
            o.gemm_fp8(self.q_e, self.s_row, *b8["qkv"], bias=bw["qkv_b"], out=qkv)
            o.vit_rope(qkv, self.cosT, self.sinT, Hh, d)
            o.vit_attn(qkv, self.n_seq, self.seq_len, Hh, d, scale, out=a)
            o.quant_rows_fp8(a, self.q_e, self.s_row)
            o.gemm_fp8(self.q_e, self.s_row, *b8["proj"], bias=bw["proj_b"], residual=x, out=x, norm_w=bw["n2w"], norm_b=bw["n2b"], out_norm=h,
                       out_norm_q=(self.q_e, self.s_row))
            o.gemm_fp8(self.q_e, self.s_row, *b8["fc1"], bias=bw["fc1_b"], out=f, act=ACT_QUICK_GELU)
            o.quant_rows_fp8(f, self.q_f, self.s_row)
            if bi + 1 < len(blocks):
                nb = blocks[bi + 1]
                o.gemm_fp8(self.q_f, self.s_row, *b8["fc2"], bias=bw["fc2_b"], residual=x, out=x, norm_w=nb["n1w"], norm_b=nb["n1b"], out_norm=h,
                           out_norm_q=(self.q_e, self.s_row))
                self._hq_ready = True
            else:
                o.gemm_fp8(self.q_f, self.s_row, *b8["fc2"], bias=bw["fc2_b"], residual=x, out=x)
                self._hq_ready = False
        if lo < hi:
            self._h_is_norm1 = hi < len(blocks)

    def _blocks_2_5(self, lo: int, hi: int):
        """Qwen2_5_VLVisionBlock x (hi - lo): RMSNorm -> qkv -> 2-D rope -> window / full attention -> proj + residual ->
        RMSNorm -> SwiGLU (gate|up fused, biases) -> down + residual (qwen2_5/vision_forward.py:6-50,89-92)."""
        o, vc = self.eng.ops, self.eng.cfg.vision
        Hh, d, E = vc.num_heads, vc.head_dim, vc.embed_dim
        scale = 1.0 / math.sqrt(d)
        x, h, qkv, a = self.x, self.h, self.qkv, self.a
        blocks = self.eng.w.vit
        if lo < hi and not self._h_is_norm1:
            o.rmsnorm(x, blocks[lo]["n1w"], 1e-6, out=h)
        for bi in range(lo, hi):
            bw = blocks[bi]
            o.gemm(h, bw["qkv_w"], bias=bw["qkv_b"], out=qkv)
            o.vit_rope(qkv, self.cosT, self.sinT, Hh, d)
            for row, n, ln in (self.full_runs if bi in vc.fullatt_block_indexes else self.win_runs):
                o.vit_attn(qkv[row:row + n * ln], n, ln, Hh, d, scale, out=a[row:row + n * ln])
            # proj / down hand their output row to the RMSNorm that follows (inside their split-K reduce)
            o.gemm_norm(a, bw["proj_w"], bw["n2w"], 1e-6, x, h, bias=bw["proj_b"], residual=x)
            if E % 64 == 0:            # SwiGLU (with its biases) in the gate/up GEMM's epilogue
                o.gemm(h, bw["gu_w"], bias=bw["gu_b"], out=self.g, act=ACT_SWIGLU)
            else:
                o.gemm(h, bw["gu_w"], bias=bw["gu_b"], out=self.f)
                o.silu_mul(self.f, out=self.g)
            if bi + 1 < len(blocks):
                o.gemm_norm(self.g, bw["down_w"], blocks[bi + 1]["n1w"], 1e-6, x, h, bias=bw["down_b"], residual=x)
            else:
                o.gemm(self.g, bw["down_w"], bias=bw["down_b"], residual=x, out=x)
        if lo < hi:
            self._h_is_norm1 = hi < len(blocks)

    def blocks(self, lo: int, hi: int):
        if self.q25:
            return self._blocks_2_5(lo, hi)
        if self.fp8:
            return self._blocks_fp8(lo, hi)
        o, vc = self.eng.ops, self.eng.cfg.vision
        Hh, d = vc.num_heads, vc.head_dim
        scale = 1.0 / math.sqrt(d)
        x, h, qkv, a, f = self.x, self.h, self.qkv, self.a, self.f
        blocks = self.eng.w.vit
        if lo < hi and not self._h_is_norm1:
            o.layernorm(x, blocks[lo]["n1w"], blocks[lo]["n1b"], 1e-6, out=h)
        for bi in range(lo, hi):
            bw = blocks[bi]
            o.gemm(h, bw["qkv_w"], bias=bw["qkv_b"], out=qkv)
            o.vit_rope(qkv, self.cosT, self.sinT, Hh, d)
            o.vit_attn(qkv, self.n_seq, self.seq_len, Hh, d, scale, out=a)
            # proj / fc2 hand their output row to the LayerNorm that follows (inside their split-K reduce when there is one)
            o.gemm_norm(a, bw["proj_w"], bw["n2w"], 1e-6, x, h, bias=bw["proj_b"], residual=x, norm_b=bw["n2b"])
            o.gemm(h, bw["fc1_w"], bias=bw["fc1_b"], out=f, act=ACT_QUICK_GELU)
            if bi + 1 < len(blocks):
                nb = blocks[bi + 1]
                o.gemm_norm(f, bw["fc2_w"], nb["n1w"], 1e-6, x, h, bias=bw["fc2_b"], residual=x, norm_b=nb["n1b"])
            else:
                o.gemm(f, bw["fc2_w"], bias=bw["fc2_b"], residual=x, out=x)
        if lo < hi:
            self._h_is_norm1 = hi < len(blocks)    # h already holds norm1 of the next block (for a resumed pass)

    def finish(self):
        o, vc, mg = self.eng.ops, self.eng.cfg.vision, self.eng.w.merger
        if self.q25:
            o.rmsnorm(self.x, mg["ln_w"], 1e-6, out=self.h)
        else:
            o.layernorm(self.x, mg["ln_w"], mg["ln_b"], 1e-6, out=self.h)
        m2 = vc.spatial_merge_size ** 2
        hm = self.h.view(self.N // m2, vc.embed_dim * m2)
        if self.fp8:
            m8 = self.eng.vit_w8[-1]
            g1 = o.gemm_fp8(*o.quant_rows_fp8(hm), *m8["w0"], bias=mg["b0"], act=ACT_GELU_ERF)
            out = o.gemm_fp8(*o.quant_rows_fp8(g1), *m8["w2"], bias=mg["b2"])
            return out
        g1 = o.gemm(hm, mg["w0"], bias=mg["b0"], act=ACT_GELU_ERF)
        out = o.gemm(g1, mg["w2"], bias=mg["b2"])
        if self.q25:           # back from window order to token order (qwen2_5/vision_forward.py:96-97)
            out = o.gather_rows(out, None, self.rev, torch.empty_like(out))
        return out


class SvlmEngine:
    def __init__(self, cfg: ModelConfig, state_dict, device="cuda", ops=None, max_len: int = 4096, max_new_tokens: int = 32,
                 decode_chunk: Optional[int] = None, use_graph: Optional[bool] = None, kv_slack: float = 1.0, kv_page_tokens: int = 16,
                 decode_tail: Optional[bool] = None,
                 vit_fp8: bool = False, linear_planes: Optional[bool] = None):
        if ops is None:
            from .ops import HipOps
            ops = HipOps()                      # raises when the HIP extension / GPU is missing
        self.ops = ops
        self.cfg = cfg
        self.device = torch.device(device)
        # the cache keeps the rotated keys / values the prefill gathers (kv_pool.py: linear planes) and the decode steps stream them;
        # False: no second copy of the K/V rows, every decode step rotates the pool rows (same bits)
        self.linear_planes = (os.environ.get("SVLM_LINEAR_PLANES", "1") != "0") if linear_planes is None else bool(linear_planes)
        self.w = EngineWeights(state_dict, cfg, self.device)
        tc, vc = cfg.text, cfg.vision
        if tc.head_dim != 128 or sum(tc.mrope_section) * 2 != tc.head_dim:
            raise ValueError("LLM head_dim must be 128 with mrope sections summing to 64")
        # BASELINE configs[4]: the vision tower's Linears on the fp8 MFMA path (svlm_gemm_fp8), weights quantised once here
        self.vit_fp8 = bool(vit_fp8)
        if self.vit_fp8:
            if vc.arch != "qwen2" or vc.embed_dim % 128 or vc.mlp_hidden % 128:
                raise ValueError("vit_fp8 needs the Qwen2-VL tower with widths that are multiples of 128 (LiveCC-7B / Qwen2-VL)")

            def q8(wt):          # per output channel: s = max|row| / 448, q = rne_e4m3(w / s)   (oracle/model.py:quant_rows_fp8)
                wf = wt.float()
                amax = wf.abs().amax(dim=1, keepdim=True)
                s8 = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
                return (wf / s8).to(torch.float8_e4m3fn).contiguous(), s8.reshape(-1).contiguous()
            self.vit_w8 = [dict(qkv=q8(b["qkv_w"]), proj=q8(b["proj_w"]), fc1=q8(b["fc1_w"]), fc2=q8(b["fc2_w"])) for b in self.w.vit]
            self.vit_w8.append(dict(w0=q8(self.w.merger["w0"]), w2=q8(self.w.merger["w2"])))
        self.max_len = int(max_len)
        self.max_new = int(max_new_tokens)
        self.kv_slack, self.kv_page_tokens = float(kv_slack), int(kv_page_tokens)      # KV pool head-room over max_len, page size
        # keys per decode-attention workgroup: fixed when given (or SVLM_DECODE_CHUNK), else picked per generate() call from the
        # CURRENT sequence length -- an engine sized for a 90k-row dense prefill decodes its 4k-row live window with the
        # bounded-window geometry, not with the long-cache one
        self._fixed_chunk = decode_chunk if decode_chunk is not None else (int(os.environ["SVLM_DECODE_CHUNK"]) if "SVLM_DECODE_CHUNK" in os.environ else None)
        self.decode_chunk = int(self._fixed_chunk if self._fixed_chunk is not None else self.pick_decode_chunk(self.max_len, tc.num_kv_heads, self.linear_planes))
        self._attn_len = self.max_len          # bound on the sequence length the current call's decode attention covers
        self.use_graph = (os.environ.get("SVLM_NO_GRAPH", "0") != "1") if use_graph is None else bool(use_graph)
        dev = self.device
        H, V = tc.hidden_size, tc.vocab_size
        self.qd, self.kd = tc.num_heads * tc.head_dim, tc.num_kv_heads * tc.head_dim
        # rope: inv_freq exactly as Qwen2VLRotaryEmbedding computes it (fp32, host)
        inv = 1.0 / (tc.rope_theta ** (torch.arange(0, tc.head_dim, 2, dtype=torch.float) / tc.head_dim))
        self.inv_freq = inv.to(dev)
        self.pos3_dev = torch.zeros((3, self.max_len), dtype=torch.int32, device=dev)
        self.posf_dev = torch.zeros((3, self.max_len), dtype=torch.float32, device=dev) if cfg.family == "qwen2_5" else None
        self.rope_cs = torch.zeros((self.max_len, tc.head_dim), dtype=BF16, device=dev)
        # sampling / feedback state
        self.tok_buf = torch.zeros(self.max_new + 1, dtype=torch.int32, device=dev)
        if dev.type == "cuda":
            self._tok_host = torch.zeros(self.max_new + 1, dtype=torch.int32).pin_memory()
            self._pos_status_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._tok_ev = torch.cuda.Event()
            self._poll_host = [torch.zeros(self.max_new + 1, dtype=torch.int32).pin_memory() for _ in range(2)]      # EOS looks, one behind
            self._poll_ev = [torch.cuda.Event(), torch.cuda.Event()]
        # host -> device copies of a chunk's start (ids, grids, positions, gather indices, state) go through this page-locked arena and are
        # truly asynchronous: a copy from pageable memory makes the host wait for everything enqueued in front of it -- the look-ahead
        # ViT's tail, which exists to run UNDERNEATH the host's turnaround, not in front of it.  One bump allocation per copy, reset per
        # generate() (the previous call ended with a host sync on its tokens, so its copies have been read).
        self._stage_buf = torch.empty(4 * self.max_len + 4096, dtype=torch.int32).pin_memory() if dev.type == "cuda" else None
        self._stage_top = 0
        self.state = torch.zeros(2, dtype=torch.int32, device=dev)          # [kv_len, cur]
        self.seen = torch.zeros(V, dtype=torch.uint8, device=dev)
        self.logits = torch.zeros(V, dtype=torch.float32, device=dev)
        self.eos_dev = torch.tensor(list(cfg.eos_token_ids), dtype=torch.int32, device=dev)
        self.ids_dev = torch.zeros(self.max_len, dtype=torch.int32, device=dev)
        # device-side position bookkeeping: M-RoPE ids are derived from ids_dev + the span grid table by svlm_rope_index
        self.max_spans = max(64, self.max_len // 64)
        self.grids_dev = torch.zeros((self.max_spans, 3), dtype=torch.int32, device=dev)
        self.pos_ws = ops.rope_index_ws(self.max_len, self.max_spans, dev) if hasattr(ops, "rope_index") else None
        self.device_positions = self.pos_ws is not None and os.environ.get("SVLM_HOST_POSITIONS", "0") != "1"
        # decode-step static activations
        self.d_x = torch.zeros(H, dtype=BF16, device=dev)
        self.d_xn = torch.zeros(H, dtype=BF16, device=dev)
        self.d_qkv = torch.zeros(self.qd + 2 * self.kd, dtype=BF16, device=dev)
        self.d_attn = torch.zeros(self.qd, dtype=BF16, device=dev)
        self.d_gu = torch.zeros(2 * tc.intermediate_size, dtype=BF16, device=dev)
        self.d_h = torch.zeros(tc.intermediate_size, dtype=BF16, device=dev)
        self.d_ws = ops.decode_attn_ws(tc.num_heads, self.max_len, 16 if self._fixed_chunk is None else self.decode_chunk, dev)
        self.d_sws = ops.sampling_ws(V, dev)
        # decode-step structure: per-op launches (6 per layer), or the persistent layer tail (csrc/dec_tail.hip: o_proj -> gate/up ->
        # down_proj -> next layer's QKV in ONE launch per layer, 3 launches per layer with the attention pair; 2B-class layers).  Measured slower than
        # the per-op launches on MI355X (profiles/r03_dec_tail_*.json), so it is opt-in.
        self.decode_tail = (os.environ.get("SVLM_DECODE_TAIL", "0") == "1") if decode_tail is None else bool(decode_tail)
        self.tail_ws = None
        if self.decode_tail:
            if not hasattr(ops, "dec_tail"):
                raise ValueError("decode_tail=True needs an ops backend with dec_tail (the HIP library)")
            if hasattr(ops, "dec_tail_supported") and not ops.dec_tail_supported(H, tc.intermediate_size, self.qd, self.kd):
                raise ValueError(f"decode_tail=True: no build of the persistent layer tail for hidden {H} / intermediate {tc.intermediate_size} "
                                 "(a layer's weights must fit the register files: Qwen2-VL-2B class); use the per-op decode step")
            self.tail_ws = ops.dec_tail_ws(H, tc.intermediate_size, len(self.w.layers), dev)
            self._tail_status_host = torch.zeros(1, dtype=torch.int64).pin_memory() if dev.type == "cuda" else None
        self._vit_rope_cache = {}
        self._vis_stream = None            # side stream the NEXT chunk's ViT runs on while this chunk decodes
        self._vis_pending = None           # [pixel tensor, grid, event, features, unfinished _VitRun]
        # ViT blocks of a look-ahead pass held back for the gap between chunks (~0.19 ms each at 448x448 on the 2B tower)
        self.vit_tail = int(os.environ.get("SVLM_VIT_TAIL", 5))
        self.section_events = None         # bench: [] -> generate() appends (name, start event, end event) of its ViT / prefill phases
        # captured decode-step graphs, keyed by (pool serial, sampling configuration, attention geometry); each entry keeps a strong
        # reference to its pool (the graph holds raw pointers into it); a few streams can alternate on one engine without re-capturing
        self._graphs = {}
        self._graph = None                 # the one replayed last (tools read it)
        self.max_graphs = 4
        self._penalty = 1.0
        self._suppress = None
        # token choice: None = greedy, else (temperature, top_k, top_p); HF GenerationConfig's own defaults for a model that does not
        # say otherwise (convert_* reads the checkpoint's generation_config, model.py)
        self._sampling = None
        self.default_top_k, self.default_top_p = 50, 1.0
        self.rng_dev = torch.zeros(2, dtype=torch.int32, device=dev)       # Philox seed of the current generate() call
        self._sample_calls = 0

    @staticmethod
    def pick_decode_chunk(max_len: int, n_kv_heads: int, linear_planes: bool = True) -> int:
        """Keys per decode-attention workgroup (measured on MI355X, tools/decode_attn_sweep.py): 48 at the bounded windows
        of the streaming configs (2B @ 2048: 8.5 us, 7B @ 4096: 12.3 us; fewer, fatter splits starve the chip, more of them
        bloat the combine).  Long caches run the barrier-free streaming kernel (chunk > 64): 128 keys while the cache is small
        enough that workgroup count matters (2 kv heads x 32k: 17.9 us; x 8k: 11.3), 192 from ~100k head-keys (4 x 32k: 26.0 us;
        256 is 10 % slower there: 64 KB strides between workgroups), 512 for the very long ones (4 x 131k: 73 us = 3.7 TB/s)."""
        if max_len <= 1024:
            return max(16, 16 * int(math.ceil(max_len / 64 / 16)))
        if max_len <= 6144:
            return 48
        hk = max_len * n_kv_heads
        if linear_planes:
            # streaming from the linear planes (no cos/sin rows, no rotation: each workgroup is done sooner, fewer and fatter ones pay):
            # 2 kv heads x 8k 128: 10.0 us; x 32k 192: 15.1 (128: 16.3); 4 x 32k 320-384: 19.7 (192: 21.2); 2 x 131k 512: 29.4 (256: 33.4);
            # 4 x 131k 512: 52.1 (384: 56.7)
            return 128 if hk < 40_000 else (192 if hk < 100_000 else (384 if hk < 200_000 else 512))
        return 128 if hk < 100_000 else (192 if hk < 400_000 else 512)

    # ------------------------------------------------------------------ host -> device staging
    def _h2d(self, dst: torch.Tensor, src) -> None:
        """dst.copy_(src) for a 4-byte numpy / torch source, asynchronous on a GPU (staged through the page-locked arena)."""
        t = torch.from_numpy(np.ascontiguousarray(src)) if isinstance(src, np.ndarray) else src.contiguous()
        buf = self._stage_buf
        if buf is None or t.element_size() != 4 or self._stage_top + t.numel() > buf.numel():
            dst.copy_(t.view(dst.shape) if t.numel() == dst.numel() else t)
            return
        view = buf[self._stage_top:self._stage_top + t.numel()].view(t.dtype).view(t.shape)
        self._stage_top += t.numel()
        view.copy_(t)
        dst.copy_(view, non_blocking=True)

    # ------------------------------------------------------------------ cache
    def new_cache(self, page_tokens: Optional[int] = None, slack: Optional[float] = None) -> KVPool:
        tc = self.cfg.text
        return KVPool(tc.num_layers, tc.num_kv_heads, tc.head_dim, self.max_len, self.device, self.ops,
                      self.kv_page_tokens if page_tokens is None else page_tokens, self.kv_slack if slack is None else slack,
                      linear_planes=self.linear_planes)

    # ------------------------------------------------------------------ ViT
    def _vit_rope(self, grid_thw):
        key = tuple(tuple(int(v) for v in g) for g in grid_thw)
        if key not in self._vit_rope_cache:
            vc = self.cfg.vision
            dim = vc.head_dim // 2
            inv = 1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float) / dim))
            m = vc.spatial_merge_size
            out = []
            for t, h, w in key:
                hp = torch.arange(h).unsqueeze(1).expand(-1, w).reshape(h // m, m, w // m, m).permute(0, 2, 1, 3).flatten()
                wp = torch.arange(w).unsqueeze(0).expand(h, -1).reshape(h // m, m, w // m, m).permute(0, 2, 1, 3).flatten()
                pos = torch.stack([hp, wp], dim=-1).repeat(t, 1)
                out.append((pos.unsqueeze(-1).float() * inv).flatten(1))
            fr = torch.cat(out, 0)                                   # (N, d/2) = [h freqs | w freqs]
            self._vit_rope_cache[key] = (fr.cos().contiguous().to(self.device), fr.sin().contiguous().to(self.device))
        return self._vit_rope_cache[key]

    def _vit_windows(self, grid):
        """Per-grid window plan of the Qwen2.5 tower, cached: gather index, its inverse, permuted rotary tables and the
        (first row, count, length) launch runs of the windowed and of the full-attention blocks."""
        key = ("win",) + tuple(tuple(int(v) for v in g) for g in grid)
        if key not in self._vit_rope_cache:
            vc = self.cfg.vision
            m2 = vc.spatial_merge_size ** 2
            index, win_len, frame_len = _window_plan(key[1:], vc.spatial_merge_size, vc.window_size, vc.patch_size)
            cosT, sinT = self._vit_rope(grid)
            rows = (torch.from_numpy(index).long().unsqueeze(1) * m2 + torch.arange(m2)).reshape(-1).to(self.device)
            dev_i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(self.device)
            self._vit_rope_cache[key] = dict(index=dev_i32(index), reverse=dev_i32(np.argsort(index, kind="stable")),
                                             cos=cosT[rows].contiguous(), sin=sinT[rows].contiguous(),
                                             win_runs=_runs(win_len), full_runs=_runs(frame_len))
        return self._vit_rope_cache[key]

    VIT_BATCH_SEQS = 8            # attention sequences (frames' temporal grids) per ViT pass of a many-frame call

    def vision_forward(self, pixel_values, grid_thw):
        """streaming_visual_encoder_forward: (N, C*T*P*P) patches -> (N / merge^2, hidden).  A call with many grids (dense-frame
        prefill, recompute) runs in passes of VIT_BATCH_SEQS sequences: the GEMMs see 8k rows instead of 1k, the activations of
        600 frames never exist at once."""
        grid = [[int(v) for v in g] for g in (grid_thw.tolist() if hasattr(grid_thw, "tolist") else grid_thw)]
        if sum(g[0] for g in grid) <= self.VIT_BATCH_SEQS or len({g[1] * g[2] for g in grid}) != 1:
            run = _VitRun(self, pixel_values, grid)
            run.blocks(0, self.cfg.vision.depth)
            return run.finish()
        outs, row, i = [], 0, 0
        while i < len(grid):
            j, seqs = i, 0
            while j < len(grid) and (seqs == 0 or seqs + grid[j][0] <= self.VIT_BATCH_SEQS):
                seqs += grid[j][0]
                j += 1
            n = sum(t * h * w for t, h, w in grid[i:j])
            run = _VitRun(self, pixel_values[row:row + n], grid[i:j])
            run.blocks(0, self.cfg.vision.depth)
            outs.append(run.finish())
            row, i = row + n, j
        return torch.cat(outs, 0)

    def vision_prefetch(self, pixel_values, grid_thw, after=None):
        """Enqueue the ViT + merger of the NEXT chunk's frames.  Frames do not depend on generated text, so most of the
        pass runs on a side stream underneath this chunk's decode steps (HBM/latency-bound GEMVs that leave the MFMA
        pipes idle).  The side stream starts after everything already enqueued on the current stream (this chunk's
        prefill: the two never run GEMMs at the same time, so the split-K scratch is not shared).  The last `vit_tail`
        blocks and the merger are held back for `vision_prefetch_finish`, which puts them on the MAIN stream behind the
        last decode step: they run during the host's turnaround between chunks, when the GPU would otherwise idle.
        `generate` of the next chunk picks the features up when it is handed the SAME pixel tensor.
        `after`: an event on the main stream the pass may start behind (default: everything enqueued there so far).  `generate`
        enqueues its decode replays FIRST and these ~290 eager launches behind them, with `after` = the point in front of the decode
        steps: the launches cost the host ~3 ms, and issued ahead of the replays they delayed the first decode step by as much
        (tools/vit_overlap.py: 22.0 -> 21.1 ms for 19 steps + one pass)."""
        if self.device.type != "cuda":
            return
        self._vision_drain()
        if self._vis_stream is None:
            # (stream priorities were measured and make no difference here: the device offers only normal/high and the
            # decode workgroups are not dispatch-starved, they share HBM and CUs with the ViT tiles)
            self._vis_stream = torch.cuda.Stream(device=self.device)
        main = torch.cuda.current_stream()
        if after is not None:
            self._vis_stream.wait_event(after)
        else:
            self._vis_stream.wait_stream(main)
        depth = self.cfg.vision.depth
        tail = min(max(self.vit_tail, 0), depth)
        with torch.cuda.stream(self._vis_stream):
            run = _VitRun(self, pixel_values, grid_thw)
            run.blocks(0, depth - tail)
            out = run.finish() if self.vit_tail < 0 else None          # < 0: nothing held back
            ev = torch.cuda.Event()
            ev.record(self._vis_stream)
        grid = [[int(v) for v in g] for g in (grid_thw.tolist() if hasattr(grid_thw, "tolist") else grid_thw)]
        self._vis_pending = [pixel_values, grid, ev, out, run if out is None else None]

    def vision_prefetch_finish(self):
        """Held-back tail of a look-ahead pass, on the CURRENT stream; call once the chunk's decode steps are enqueued."""
        pend = self._vis_pending
        if pend is None or pend[4] is None:
            return
        run, pend[4] = pend[4], None
        main = torch.cuda.current_stream()
        main.wait_event(pend[2])                       # the side stream's blocks: long finished by now
        for t in (run.x, run.h, run.qkv, run.a, run.f) + ((run.g,) if run.q25 else ()) + ((run.q_e, run.q_f, run.s_row) if run.fp8 else ()):
            t.record_stream(main)                      # allocated on the side stream, used here
        depth = self.cfg.vision.depth
        run.blocks(depth - min(self.vit_tail, depth), depth)
        pend[3] = run.finish()
        pend[2] = None

    def _vision_drain(self):
        """Drop a look-ahead nobody consumed (the caller changed its mind): let the side stream finish first."""
        pend, self._vis_pending = self._vis_pending, None
        if pend is not None:
            self._vis_stream.synchronize()

    def _vision(self, pixel_values, grid_thw):
        pend = self._vis_pending
        if pend is not None:
            grid = [[int(v) for v in g] for g in (grid_thw.tolist() if hasattr(grid_thw, "tolist") else grid_thw)]
            if pend[0] is pixel_values and pend[1] == grid:
                self.vision_prefetch_finish()          # no-op when generate() already issued the tail
                self._vis_pending = None
                if pend[2] is not None:                # whole pass ran on the side stream
                    main = torch.cuda.current_stream()
                    main.wait_event(pend[2])
                    pend[3].record_stream(main)
                return pend[3]
            self._vision_drain()
        return self.vision_forward(pixel_values, grid_thw)

    # ------------------------------------------------------------------ LLM
    PREFILL_ROWS = 4096           # rows per pass of a long prefill (dense-frame prompts are tens of thousands of rows)

    def _prefill(self, c: KVPool, idx_dev, vis, T: int, L_before: int, head: bool = True):
        """Rows L_before .. L_before+T-1 through all layers (their K/V rows go to the pool); `head`: last-row logits."""
        o, w, tc = self.ops, self.w, self.cfg.text
        H, qd, kd = tc.hidden_size, self.qd, self.kd
        L = L_before + T
        dev = self.device
        x = torch.empty((T, H), dtype=BF16, device=dev)
        o.gather_rows(w.embed, vis, idx_dev, x)
        xn = torch.empty_like(x)
        qkv = torch.empty((T, qd + 2 * kd), dtype=BF16, device=dev)
        attn = torch.empty((T, qd), dtype=BF16, device=dev)
        gu = torch.empty((T, 2 * tc.intermediate_size), dtype=BF16, device=dev)
        hm = torch.empty((T, tc.intermediate_size), dtype=BF16, device=dev)
        scale = 1.0 / math.sqrt(tc.head_dim)
        n_layers = len(w.layers)
        # the gather launch of every layer also leaves the rotated keys / values of rows [0, L) in the cache's linear planes, which the
        # decode steps of this chunk stream instead of rotating the pool rows again (kv_pool.py)
        lin = c.lin_args() if hasattr(c, "lin_args") else None
        o.rmsnorm(x, w.layers[0]["ln1"], tc.rms_eps, out=xn)
        for li, lw in enumerate(w.layers):
            o.gemm(xn, lw["qkv_w"], bias=lw["qkv_b"], out=qkv)
            # the chunk's K/V rows go to their pool slots inside the launch that rotates and gathers the keys
            o.prefill_attn(qkv[:, :qd], c.pool, li, c.slot_of_dev, self.rope_cs, attn, T, L, tc.num_heads, scale,
                           k_new=qkv[:, qd:qd + kd], v_new=qkv[:, qd + kd:], lin=lin)
            # the two residual-stream GEMMs hand their output row to the RMSNorm that follows inside their split-K reduce
            o.gemm_norm(attn, lw["o_w"], lw["ln2"], tc.rms_eps, x, xn, residual=x)
            if H % 64 == 0:            # gate and up columns paired inside the GEMM tile: SwiGLU in its epilogue, no (T, 2I) round trip
                o.gemm(xn, lw["gu_w"], out=hm, act=ACT_SWIGLU)
            else:
                o.gemm(xn, lw["gu_w"], out=gu)
                o.silu_mul(gu, out=hm)
            if li + 1 < n_layers:
                o.gemm_norm(hm, lw["down_w"], w.layers[li + 1]["ln1"], tc.rms_eps, x, xn, residual=x)
            else:
                o.gemm(hm, lw["down_w"], residual=x, out=x)
        if lin is not None:
            c.lin_written(L)
            if self.decode_tail:          # svlm_dec_tail's QKV appends to the pool only: the attention takes the appended rows from there
                c.lin_appends_off()
                c.sync_device()
        if not head:
            return
        last = x[T - 1:T].contiguous()
        o.rmsnorm(last, w.final_norm, tc.rms_eps, out=self.d_xn.view(1, H))
        o.gemv(self.d_xn, w.lm_head, out_f32=self.logits)

    def _decode_step_launch(self, c: KVPool):
        """One token, 6 launches per layer: [norm+QKV+append] [attn split] [attn combine] [o+res] [norm+gate/up+SwiGLU]
        [down+res]; every kernel reads the KV length / token index from `self.state` on the device."""
        o, w, tc = self.ops, self.w, self.cfg.text
        H, qd, kd = tc.hidden_size, self.qd, self.kd
        kv_len = self.state[0:1]
        scale = 1.0 / math.sqrt(tc.head_dim)
        lin = c.lin_args() if hasattr(c, "lin_args") else None        # rows below *lin_len: streamed from the prefill's rotated copy
        o.gather_rows(w.embed, None, self.tok_buf, self.d_x.view(1, H), idx_off=self.state[1:2])
        if self.decode_tail:
            # 3 launches per layer: [attn split] [attn combine] [tail: o+res, norm+gate/up+SwiGLU, down+res, next layer's norm+QKV+append]
            nl = len(w.layers)
            o.dec_tail_reset(self.tail_ws, H, tc.intermediate_size, nl)
            l0 = w.layers[0]
            o.dec_qkv(self.d_x, l0["ln1"], tc.rms_eps, l0["qkv_w"], l0["qkv_b"], self.d_qkv, c.pool, 0, c.slot_of_dev, qd, kd, len_dev=kv_len)
            for li, lw in enumerate(w.layers):
                o.decode_attn(self.d_qkv[:qd], c.pool, li, c.slot_of_dev, self.rope_cs, self.d_attn, self.d_ws, tc.num_heads,
                              self._attn_len, self.decode_chunk, scale, length=1, len_dev=kv_len, lin=lin)
                nxt = None
                if li + 1 < nl:
                    ln = w.layers[li + 1]
                    nxt = (ln["ln1"], ln["qkv_w"], ln["qkv_b"], self.d_qkv, c.pool, li + 1, c.slot_of_dev, qd, kd, 0, kv_len)
                o.dec_tail(self.d_attn, self.d_x, lw["o_w"], lw["ln2"], lw["gu_w"], lw["down_w"], tc.rms_eps, self.tail_ws, li, nl, nxt=nxt)
        for li, lw in enumerate(w.layers if not self.decode_tail else ()):
            o.dec_qkv(self.d_x, lw["ln1"], tc.rms_eps, lw["qkv_w"], lw["qkv_b"], self.d_qkv, c.pool, li, c.slot_of_dev, qd, kd,
                      len_dev=kv_len, lin=lin)
            o.decode_attn(self.d_qkv[:qd], c.pool, li, c.slot_of_dev, self.rope_cs, self.d_attn, self.d_ws, tc.num_heads,
                          self._attn_len, self.decode_chunk, scale, length=1, len_dev=kv_len, lin=lin)
            o.gemv(self.d_attn, lw["o_w"], residual=self.d_x, out=self.d_x)
            o.dec_gate_up(self.d_x, lw["ln2"], tc.rms_eps, lw["gu_w"], self.d_h)
            o.gemv(self.d_h, lw["down_w"], residual=self.d_x, out=self.d_x)
        smp = self._sampling
        if smp is not None and smp[1] == 0 and smp[2] >= 1.0:     # plain temperature sampling: Gumbel-max inside the candidates
            o.dec_lm_head(self.d_x, w.final_norm, tc.rms_eps, w.lm_head, self.logits, self.seen if self._penalty != 1.0 else None,
                          self._penalty, self._suppress, self.d_sws, temperature=smp[0], rng=self.rng_dev, state=self.state)
        else:
            o.dec_lm_head(self.d_x, w.final_norm, tc.rms_eps, w.lm_head, self.logits, self.seen if self._penalty != 1.0 else None,
                          self._penalty, self._suppress, self.d_sws)

    def _sample_launch(self, advance_kv: int, fused: bool = False):
        """Token choice + device-side feedback (tok_buf / state / seen).  `fused`: dec_lm_head already left its candidates in
        d_sws (greedy, or Gumbel-max sampling); top-k / top-p draws always go through the filter kernel on the logits row."""
        seen = self.seen if self._penalty != 1.0 else None
        smp = self._sampling
        if smp is not None and not (smp[1] == 0 and smp[2] >= 1.0 and fused):
            self.ops.penalty_sample(self.logits, seen, self._penalty, self._suppress, smp[0], smp[1], smp[2], self.rng_dev, self.tok_buf,
                                    self.state, advance_kv, self.d_sws)
        elif fused:      # candidates already left in d_sws by dec_lm_head
            self.ops.argmax_finish(self.d_sws, self.cfg.text.vocab_size, seen, self.tok_buf, self.state, advance_kv)
        else:
            self.ops.penalty_argmax(self.logits, seen, self._penalty, self._suppress, self.tok_buf, self.state, advance_kv, self.d_sws)

    def _decode_step(self, c: KVPool):
        if not self.use_graph:
            self._decode_step_launch(c)
            self._sample_launch(1, fused=True)
            return
        key = (c.serial, self._penalty, self._suppress is not None, self._sampling, self.decode_chunk, self._attn_len, self.decode_tail)
        hit = self._graphs.get(key)
        if hit is not None and hit[1]() is not c:         # (a serial is never reused; belt and braces)
            hit = None
        if hit is None:
            # capture once per (cache, sampling config); state is restored because capture does not execute
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._decode_step_launch(c)
                self._sample_launch(1, fused=True)
            # a graph holds raw pointers into its pool but must not keep the pool alive: modes that open a fresh cache per chunk
            # (recompute, the efficiency harness's mode c) would otherwise pin max_graphs full pools.  Entries of pools that are gone
            # are dropped here; a serial is unique, so a dead pool's graph can never be replayed.
            for k in [k for k, v in self._graphs.items() if v[1]() is None]:
                del self._graphs[k]
            if len(self._graphs) >= self.max_graphs:
                self._graphs.pop(next(iter(self._graphs)))          # oldest capture goes
            hit = self._graphs[key] = (g, weakref.ref(c))
        self._graph = hit[0]
        self._graph.replay()

    def _mark(self, name=None, start=None):
        """Section timing for bench.py (off unless `section_events` is a list): an event now; with `name`, the (start, now) pair."""
        if self.section_events is None or self.device.type != "cuda":
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        if name is not None:
            self.section_events.append((name, start, ev))
        return ev

    # ------------------------------------------------------------------ generate
    def generate(self, ids: Sequence[int], cache: Optional[KVPool], video_grid_thw, pixel_values=None, grid_thw=None,
                 max_new_tokens: int = 20, repetition_penalty: float = 1.05, do_sample: bool = False, temperature: float = 1.0,
                 suppress_eos: bool = False, keep_logits: bool = False, generator=None, next_vision=None,
                 all_text: bool = False, second_per_grid_t: Optional[float] = None, pos_mode: str = "shrink",
                 last_cache_position: float = -1, force_tokens: Optional[Sequence[int]] = None, top_k: Optional[int] = None,
                 top_p: Optional[float] = None) -> GenerateOutput:
        """`next_vision=(pixel_values, grid_thw)` of the FOLLOWING chunk, when the caller already has its frames, is
        encoded on a side stream underneath this chunk's decode steps (see `vision_prefetch`).
        `force_tokens` (greedy only): teacher forcing for parity tests -- every step still takes its own argmax on the device
        (returned in `.own`), then the given token is fed back instead, so that the logits of EVERY forward can be compared
        with a reference even after a near-tie went the other way."""
        cfg, tc, o = self.cfg, self.cfg.text, self.ops
        ids = np.asarray(ids, dtype=np.int64).reshape(-1)
        if cache is None:
            cache = self.new_cache()
        cache.release_reserved()
        self._last_cache = cache
        L_before, L_ids = cache.length, int(ids.shape[0])
        T = L_ids - L_before
        if T < 1:
            raise ValueError(f"nothing to forward: {L_ids} ids but {L_before} rows already cached")
        if max_new_tokens < 1 or max_new_tokens > self.max_new:
            raise ValueError(f"max_new_tokens={max_new_tokens} outside [1, {self.max_new}]")
        if L_ids + max_new_tokens > self.max_len:
            raise MemoryError(f"sequence {L_ids}+{max_new_tokens} exceeds engine max_len {self.max_len}")
        dev = self.device
        if self._fixed_chunk is None:
            # decode-attention geometry of THIS call: split size from the current length, grid bounded by it.  The bound moves in
            # steps of 1k rows up to 8k and of 8k rows beyond (a cache that only grows -- full-attention mode -- would otherwise
            # re-capture its 171-kernel decode graph every few chunks: 5-8 ms each time, seen as 7 ms/token spikes in
            # tools/efficiency_modes.py mode a), and the split size is picked for the bound, so both change together.
            need = L_ids + max_new_tokens
            step = 1024 if need <= 8192 else 8192
            self._attn_len = min(self.max_len, -(-need // step) * step)
            self.decode_chunk = self.pick_decode_chunk(self._attn_len, tc.num_kv_heads, self.linear_planes)
        # ---- shrink-mode positions for the whole (pruned) sequence + the tokens to be generated
        n_rows = L_ids + max_new_tokens
        is_f = cfg.family == "qwen2_5" and not all_text
        spg = 2.0 / float(os.environ.get("QWENVL_FPS", "2.0")) if second_per_grid_t is None else float(second_per_grid_t)

        def index(seq, grids):
            """(3, len) positions of a self-contained id sequence + the position after it."""
            if all_text:       # 1-D rope on all three axes (qwen2_5/model_forward.py:6-28,99)
                return rope_index_1d(len(seq))
            if is_f:           # float temporal axis: second_per_grid_t * tokens_per_second per grid step (qwen2_5/pos_emb.py:107-127)
                return rope_index_qwen2_5(seq, grids, cfg.vision.spatial_merge_size, cfg.video_token_id,
                                          cfg.vision_start_token_id, spg, cfg.vision.tokens_per_second)
            return rope_index_qwen2(seq, grids, cfg.vision.spatial_merge_size, cfg.video_token_id, cfg.vision_start_token_id)

        pos_full = np.empty((3, n_rows), dtype=np.float32 if is_f else np.int32)
        on_device = pos_mode == "shrink" and self.device_positions and len(video_grid_thw) <= self.max_spans
        self._stage_top = 0
        self._h2d(self.ids_dev[:L_ids], ids.astype(np.int32))
        if on_device:
            # shrink mode: positions re-derived from the pruned ids on every chunk (qwen2/model_forward.py:119-126) -- on the device,
            # from the ids and the span grid table; nothing of size L is computed or copied by the host
            n_g = 0 if all_text else len(video_grid_thw)
            if n_g:
                self._h2d(self.grids_dev[:n_g], np.asarray(video_grid_thw, dtype=np.int32).reshape(n_g, 3))
            pos_dev = self.posf_dev if is_f else self.pos3_dev
            o.rope_index(self.ids_dev, L_ids, self.grids_dev, n_g, cfg.vision.spatial_merge_size, -1 if all_text else cfg.video_token_id,
                         -1 if all_text else cfg.vision_start_token_id, pos_dev, self.pos_ws, n_extra=max_new_tokens, second_per_grid_t=spg,
                         tokens_per_second=cfg.vision.tokens_per_second)
        elif pos_mode == "shrink":
            pos, nxt = index(ids, video_grid_thw)
            pos_full[:, :L_ids] = pos
        elif pos_mode == "append":
            # positions are assigned once and travel with their rows (qwen2/model_forward.py:75-117): the un-cached suffix is
            # indexed on its own with THIS call's grids and shifted behind the last forwarded position; cached rows keep
            # theirs (the reference caches rotated keys; un-rotated keys + their original positions give the same bits)
            if L_before == 0:
                pos, nxt = index(ids, grid_thw if grid_thw is not None else video_grid_thw)
            else:
                pos, nxt = index(ids[L_before:], grid_thw if grid_thw is not None else [])
                off = last_cache_position + 1
                pos = pos + (np.float32(off) if is_f else int(off))
                pos_full[:, :L_before] = cache.pos_rows[:, :L_before]
            pos_full[:, L_before:L_ids] = pos
            # decode tokens continue on the TEMPORAL axis of the last forwarded token (:113-117)
            nxt = float(pos[0, -1]) + 1
        else:
            raise ValueError(f"pos_mode must be 'shrink' or 'append', not {pos_mode!r}")
        if not on_device:
            pos_full[:, L_ids:] = (np.float32(nxt) if is_f else int(nxt)) + np.arange(max_new_tokens, dtype=pos_full.dtype)
            pos_dev = self.posf_dev if is_f else self.pos3_dev
            self._h2d(pos_dev[:, :n_rows], pos_full)
        o.mrope_table(pos_dev, self.inv_freq, self.rope_cs, 0, n_rows, tc.mrope_section)
        cache.reserve(T + max_new_tokens)
        cache.sync_device()
        # ---- embeddings of the un-cached suffix (vision rows spliced in order)
        new_ids = ids[L_before:]
        vmask = new_ids == cfg.video_token_id
        vis = None
        idx = new_ids.astype(np.int32)
        if vmask.any():
            if pixel_values is None:
                raise ValueError("video tokens in the un-cached suffix but no pixel_values_videos")
            ev0 = self._mark()
            vis = self._vision(pixel_values, grid_thw)
            self._mark("vit", ev0)
            n_tok = int(vmask.sum())
            if n_tok != vis.shape[0]:
                raise ValueError(f"Video features and video tokens do not match: tokens: {n_tok}, features {vis.shape[0]}")
            idx = idx.copy()
            idx[vmask] = -1 - np.arange(n_tok, dtype=np.int32)
        idx_dev = torch.empty(idx.shape[0], dtype=torch.int32, device=dev)
        self._h2d(idx_dev, idx)
        # ---- sampling state
        self._penalty = float(repetition_penalty)
        self._suppress = self.eos_dev if suppress_eos else None
        if self._penalty != 1.0:
            self.seen.zero_()
            o.mark_seen(self.ids_dev, L_ids, self.seen)
        self._h2d(self.state, np.array([L_ids, -1], dtype=np.int32))
        logits_out = [] if keep_logits else None
        # ---- token choice (streaming_generate_qwen.py:75-99): greedy, or HF's warpers + one multinomial draw per token, on the device
        self._sampling = None
        if do_sample:
            if force_tokens is not None:
                raise ValueError("force_tokens is a greedy-parity aid; it cannot be combined with do_sample")
            k = self.default_top_k if top_k is None else int(top_k)
            p = self.default_top_p if top_p is None else float(top_p)
            if temperature <= 0 or k < 0 or not (0.0 < p <= 1.0):
                raise ValueError(f"bad sampling settings: temperature={temperature} top_k={k} top_p={p}")
            if k != 1:                                   # top_k = 1 leaves one survivor: the argmax
                self._sampling = (float(temperature), k, p)
                # one Philox seed per call: the caller's generator seed (or 42, the reference's set_seed, inference.py:24-25)
                # mixed with the call count -- no device round trip, reproducible per engine instance
                base = (generator.initial_seed() if generator is not None else 42) & 0xFFFFFFFFFFFFFFFF
                z = (base + 0x9E3779B97F4A7C15 * (self._sample_calls + 1)) & 0xFFFFFFFFFFFFFFFF
                z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
                z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
                z ^= z >> 31
                self._sample_calls += 1
                self._h2d(self.rng_dev, np.array([z & 0xFFFFFFFF, z >> 32], dtype=np.uint32).view(np.int32))

        # a prompt of tens of thousands of rows (dense-frame prefill) goes through in passes: causal attention over the rows
        # already in the pool makes the passes arithmetically the same forward
        ev0 = self._mark()
        for t0 in range(0, T, self.PREFILL_ROWS):
            tc_rows = min(self.PREFILL_ROWS, T - t0)
            self._prefill(cache, idx_dev[t0:t0 + tc_rows], vis, tc_rows, L_before + t0, head=t0 + tc_rows == T)
        self._mark("prefill", ev0)
        if keep_logits:
            logits_out.append(self.logits.detach().cpu().clone())
        self._sample_launch(0)
        own = None
        if force_tokens is not None:
            if len(force_tokens) < 1 or len(force_tokens) > max_new_tokens:
                raise ValueError(f"force_tokens has {len(force_tokens)} entries for max_new_tokens={max_new_tokens}")
            max_new_tokens = len(force_tokens)
            own, seen_host = [], set(int(t) for t in ids)
            self._force(0, force_tokens, own, seen_host)
        ev_pre = None
        if next_vision is not None and self.device.type == "cuda":
            ev_pre = torch.cuda.Event()
            ev_pre.record()                 # the look-ahead ViT may start here: behind the prefill, beside the decode steps
        eos_poll = 0 if (suppress_eos or self.device.type != "cuda") else 4
        poll, prefetched = None, False       # poll: (event, host buffer, token count) of the look that is still in flight
        for step in range(1, max_new_tokens):
            self._decode_step(cache)
            if keep_logits:
                logits_out.append(self.logits.detach().cpu().clone())
            if own is not None:
                self._force(step, force_tokens, own, seen_host)
            # a real caption ends well before max_new_tokens: every 4th step the tokens so far are copied out, and the remaining steps
            # are not launched once an end-of-turn token is among them.  The look is ONE POLL BEHIND: the copy queued at step s is read
            # at step s + 4, when it has long landed, so the GPU never drains at a poll (at most 2 x 4 steps run past the EOS; their
            # KV rows are rolled back with the rest, cache.commit below).  The look-ahead ViT is enqueued in front of the first poll,
            # underneath the steps already queued -- not behind the whole loop, where it would no longer overlap anything.  With EOS
            # suppressed (benchmarks: fixed token counts) nothing is polled and the chunk keeps its single host sync.
            if eos_poll and step % eos_poll == 0 and step + 1 < max_new_tokens:
                if next_vision is not None and not prefetched:
                    self.vision_prefetch(*next_vision, after=ev_pre)
                    prefetched = True
                if poll is not None:
                    poll[0].synchronize()
                    if any(int(t) in cfg.eos_token_ids for t in poll[1][:poll[2]].tolist()):
                        break
                k = (step // eos_poll) & 1
                self._poll_host[k][:step + 1].copy_(self.tok_buf[:step + 1], non_blocking=True)
                self._poll_ev[k].record()
                poll = (self._poll_ev[k], self._poll_host[k], step + 1)
        # the one host sync of the chunk waits for the TOKENS only: the held-back ViT tail of the look-ahead pass is enqueued
        # behind the copy and keeps the GPU busy while the host turns the chunk around
        if self.device.type == "cuda":
            self._tok_host[:max_new_tokens].copy_(self.tok_buf[:max_new_tokens], non_blocking=True)
            if on_device:
                self._pos_status_host.copy_(self.pos_ws[:1], non_blocking=True)
            if self.tail_ws is not None:
                self._tail_status_host.copy_(self.tail_ws[:1], non_blocking=True)
            self._tok_ev.record()
            if next_vision is not None and not prefetched:      # decode replays are in the queue: now the host can spend its 3 ms on the ViT launches
                self.vision_prefetch(*next_vision, after=ev_pre)
            self.vision_prefetch_finish()
            self._tok_ev.synchronize()
            toks = self._tok_host[:max_new_tokens].numpy().copy()
        else:                                                            # host-side test backend
            toks = self.tok_buf[:max_new_tokens].numpy().copy()
        if self.tail_ws is not None and self.device.type == "cuda" and int(self._tail_status_host[0]) & 0xFFFFFFFF:
            self.tail_ws[:1].zero_()
            raise RuntimeError("decode tail: a hand-off wait gave up (status %d): the workgroups of a svlm_dec_tail launch were not all "
                               "resident together; this chunk's tokens are invalid" % (int(self._tail_status_host[0]) & 0xFFFFFFFF))
        n_new = max_new_tokens
        for j, t in enumerate(toks):
            if int(t) in cfg.eos_token_ids:
                n_new = j + 1
                break
        cache.commit(L_ids + n_new - 1)
        cache.release_reserved()
        if on_device:
            if self.device.type == "cuda":
                st = int(self._pos_status_host[0])
                if st:           # the reference raises here too (a span without its tokens / grid row: qwen2/pos_emb.py:88,132)
                    raise ValueError(f"rope index: malformed vision spans in the id sequence (device status {st})")
            self.last_position = -1.0            # shrink mode never reads streaming_args.last_cache_position
        else:
            cache.pos_rows[:, L_before:cache.length] = pos_full[:, L_before:cache.length]
            self.last_position = float(pos_full[0, cache.length - 1])           # streaming_args.last_cache_position
        seq = ids.tolist() + [int(t) for t in toks[:n_new]]
        return GenerateOutput(seq, cache, logits_out, n_new, own)

    def _force(self, step: int, force_tokens, own: list, seen_host: set):
        """Teacher forcing: note the token the device just chose for `step`, put the given one in its place."""
        mine, tok = int(self.tok_buf[step]), int(force_tokens[step])
        own.append(mine)
        if mine != tok:
            self.tok_buf[step:step + 1].copy_(torch.tensor([tok], dtype=torch.int32))
            if self._penalty != 1.0:
                if mine not in seen_host:
                    self.seen[mine] = 0
                self.seen[tok] = 1
        seen_host.add(tok)
