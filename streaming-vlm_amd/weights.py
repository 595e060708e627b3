"""Weight containers: HF state-dict names in, device-resident fused bf16 matrices out.

State-dict keys follow ``Qwen2VLForConditionalGeneration`` (transformers >= 4.52 layout, the one
the reference patches: ``model.model.language_model`` / ``model.model.visual``,
qwen2/patch_model.py:24-26), so ``convert_qwen2_to_streaming(hf_model)`` can pass
``hf_model.state_dict()`` straight in.
"""
from __future__ import annotations

from typing import Dict

import torch

from .config import ModelConfig

V_PREFIX = "model.visual."
L_PREFIX = "model.language_model."


def state_dict_shapes(cfg: ModelConfig) -> Dict[str, tuple]:
    vc, tc = cfg.vision, cfg.text
    E, Hd = vc.embed_dim, tc.hidden_size
    sh = {V_PREFIX + "patch_embed.proj.weight": (E, vc.in_channels, vc.temporal_patch_size, vc.patch_size, vc.patch_size)}
    M = E * vc.spatial_merge_size ** 2
    m = V_PREFIX + "merger."
    if vc.arch == "qwen2_5":
        # Qwen2_5_VLVisionBlock: RMSNorm (no bias), SwiGLU MLP with biases; merger ln_q is an RMSNorm
        I = vc.mlp_hidden
        for b in range(vc.depth):
            p = f"{V_PREFIX}blocks.{b}."
            sh.update({p + "norm1.weight": (E,), p + "norm2.weight": (E,),
                       p + "attn.qkv.weight": (3 * E, E), p + "attn.qkv.bias": (3 * E,),
                       p + "attn.proj.weight": (E, E), p + "attn.proj.bias": (E,),
                       p + "mlp.gate_proj.weight": (I, E), p + "mlp.gate_proj.bias": (I,),
                       p + "mlp.up_proj.weight": (I, E), p + "mlp.up_proj.bias": (I,),
                       p + "mlp.down_proj.weight": (E, I), p + "mlp.down_proj.bias": (E,)})
        sh.update({m + "ln_q.weight": (E,), m + "mlp.0.weight": (M, M), m + "mlp.0.bias": (M,),
                   m + "mlp.2.weight": (Hd, M), m + "mlp.2.bias": (Hd,)})
    else:
        for b in range(vc.depth):
            p = f"{V_PREFIX}blocks.{b}."
            sh.update({p + "norm1.weight": (E,), p + "norm1.bias": (E,), p + "norm2.weight": (E,), p + "norm2.bias": (E,),
                       p + "attn.qkv.weight": (3 * E, E), p + "attn.qkv.bias": (3 * E,),
                       p + "attn.proj.weight": (E, E), p + "attn.proj.bias": (E,),
                       p + "mlp.fc1.weight": (vc.mlp_hidden, E), p + "mlp.fc1.bias": (vc.mlp_hidden,),
                       p + "mlp.fc2.weight": (E, vc.mlp_hidden), p + "mlp.fc2.bias": (E,)})
        sh.update({m + "ln_q.weight": (E,), m + "ln_q.bias": (E,), m + "mlp.0.weight": (M, M), m + "mlp.0.bias": (M,),
                   m + "mlp.2.weight": (Hd, M), m + "mlp.2.bias": (Hd,)})
    sh[L_PREFIX + "embed_tokens.weight"] = (tc.vocab_size, Hd)
    qd, kd = tc.num_heads * tc.head_dim, tc.num_kv_heads * tc.head_dim
    for i in range(tc.num_layers):
        p = f"{L_PREFIX}layers.{i}."
        sh.update({p + "input_layernorm.weight": (Hd,), p + "post_attention_layernorm.weight": (Hd,),
                   p + "self_attn.q_proj.weight": (qd, Hd), p + "self_attn.q_proj.bias": (qd,),
                   p + "self_attn.k_proj.weight": (kd, Hd), p + "self_attn.k_proj.bias": (kd,),
                   p + "self_attn.v_proj.weight": (kd, Hd), p + "self_attn.v_proj.bias": (kd,),
                   p + "self_attn.o_proj.weight": (Hd, qd),
                   p + "mlp.gate_proj.weight": (tc.intermediate_size, Hd), p + "mlp.up_proj.weight": (tc.intermediate_size, Hd),
                   p + "mlp.down_proj.weight": (Hd, tc.intermediate_size)})
    sh[L_PREFIX + "norm.weight"] = (Hd,)
    if not tc.tie_word_embeddings:
        sh["lm_head.weight"] = (tc.vocab_size, Hd)
    return sh


def random_state_dict(cfg: ModelConfig, seed: int = 0, device="cpu", dtype=torch.bfloat16, std: float = 0.02):
    """Random-init weights of the real shapes (no checkpoint is available offline): matrices and
    biases N(0, std), norm gains 1 + N(0, std)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    sd = {}
    for name, shape in state_dict_shapes(cfg).items():
        t = torch.randn(shape, generator=g, device=device, dtype=torch.float32) * std
        if name.endswith("norm.weight") or "norm1.weight" in name or "norm2.weight" in name or name.endswith("ln_q.weight") \
                or name.endswith("layernorm.weight"):
            t = t + 1.0
        sd[name] = t.to(dtype)
    return sd


def decisive_state_dict(cfg: ModelConfig, seed: int = 0, device="cpu", dtype=torch.bfloat16, offset: int = 45,
                        gain: float = 20.0, sharpness: float = 4.0):
    """Random weights of the real shapes whose GREEDY TOKEN is decided by a wide margin (tens of times the bf16 noise of a
    28-layer forward) while still depending on the cached context: `random_state_dict` plus one planted attention head.

    With iid random weights the top two of ~152k logits are closer than the rounding noise in a fixed fraction of the
    steps whatever the kernels do, so "token ids exact under greedy" cannot be asserted on them.  The planted head makes
    the network digital the way a trained copy/induction head is:
      * layer 0, kv head 0 and its query heads: q and k come from their BIASES only (weights zeroed) and carry one
        amplitude on the 16 temporal M-RoPE frequency pairs, phased so that after the rotation q(p) . k(p') peaks at
        p - p' == `offset` (`sharpness` = score gap per unit of sum(cos): 1.35 units to the neighbours, > 4.6 beyond);
      * its value / output projections are gamma * P^T P with one random P (128 x hidden): the head writes a copy of the
        (layer-0, i.e. exact) embedding of the token `offset` positions back into the residual stream with RMS `gain`;
      * the tied lm_head reads that copy back as the same token (the 7B's separate lm_head is set equal to the embedding);
      * the chat-template markers the span finder keys on (<|im_start|>, <|im_end|>, <|endoftext|>, the vision markers, "Time")
        are embedded as HALF the embedding of an ordinary stand-in token, so a copied marker decodes to its stand-in and
        the generated text never contains one.
    Every other weight, the whole ViT included, stays N(0, 0.02) and runs at its usual magnitudes; which token comes out
    depends on the KV row `offset` positions back, its slot mapping and its re-indexed position."""
    from .config import ENDOFTEXT, IM_END, IM_START, VIDEO_PAD, VISION_END, VISION_START
    tc = cfg.text
    sd = random_state_dict(cfg, seed, device, dtype)
    g = torch.Generator(device=device)
    g.manual_seed(seed + 7919)
    D, H, G = tc.head_dim, tc.hidden_size, tc.num_heads // tc.num_kv_heads
    n_t = tc.mrope_section[0]                                   # temporal frequency pairs: text tokens' 1-D positions
    inv = 1.0 / (tc.rope_theta ** (torch.arange(0, D, 2, dtype=torch.float64) / D))
    amp = (sharpness * (D ** 0.5)) ** 0.5                       # b^2 * sum(cos) / sqrt(D) = sharpness * sum(cos)
    kb = torch.zeros(D, dtype=torch.float64)
    qb = torch.zeros(D, dtype=torch.float64)
    kb[:n_t] = amp
    qb[:n_t] = amp * torch.cos(inv[:n_t] * offset)
    qb[D // 2:D // 2 + n_t] = -amp * torch.sin(inv[:n_t] * offset)
    p = f"{L_PREFIX}layers.0.self_attn."
    sd[p + "q_proj.weight"][:G * D] = 0
    sd[p + "k_proj.weight"][:D] = 0
    sd[p + "q_proj.bias"][:G * D] = qb.float().repeat(G).to(device=device, dtype=dtype)
    sd[p + "k_proj.bias"][:D] = kb.float().to(device=device, dtype=dtype)
    P = torch.randn((D, H), generator=g, device=device, dtype=torch.float32) * 0.02
    sd[p + "v_proj.weight"][:D] = P.to(dtype)
    sd[p + "v_proj.bias"][:D] = 0
    # |P^T P x| for a unit-RMS x: sigma^2 * sqrt(D^2 + D*H) per coordinate
    gamma = gain / (0.02 ** 2 * (D * D + D * H) ** 0.5)
    sd[p + "o_proj.weight"][:, :G * D] = ((gamma / G) * P.t().repeat(1, G)).to(dtype)
    emb = sd[L_PREFIX + "embed_tokens.weight"]
    for k, s in enumerate((ENDOFTEXT, IM_START, IM_END, VISION_START, VISION_END, VIDEO_PAD, 1462)):      # 1462 = "Time", a span marker
        emb[s] = (0.5 * emb[1000 + 37 * k].float()).to(dtype)
    if not tc.tie_word_embeddings:
        sd["lm_head.weight"] = emb.clone()
    return sd


class EngineWeights:
    """Fused, contiguous, device-resident views the kernels consume."""

    def __init__(self, sd: Dict[str, torch.Tensor], cfg: ModelConfig, device):
        vc, tc = cfg.vision, cfg.text
        want = state_dict_shapes(cfg)
        missing = [k for k in want if k not in sd]
        if missing:
            raise KeyError(f"state dict lacks {len(missing)} tensors, e.g. {missing[:3]}")
        for k, shp in want.items():
            if tuple(sd[k].shape) != tuple(shp):
                raise ValueError(f"{k}: shape {tuple(sd[k].shape)} != expected {shp}")

        def dev(t):
            return t.detach().to(device=device, dtype=torch.bfloat16).contiguous()

        self.patch_embed = dev(sd[V_PREFIX + "patch_embed.proj.weight"].reshape(vc.embed_dim, -1))
        self.vit = []
        m = V_PREFIX + "merger."
        if vc.arch == "qwen2_5":
            I, Ip, E = vc.mlp_hidden, vc.mlp_padded, vc.embed_dim
            for b in range(vc.depth):
                p = f"{V_PREFIX}blocks.{b}."
                blk = {k: dev(sd[p + n]) for k, n in [
                    ("n1w", "norm1.weight"), ("n2w", "norm2.weight"), ("qkv_w", "attn.qkv.weight"), ("qkv_b", "attn.qkv.bias"),
                    ("proj_w", "attn.proj.weight"), ("proj_b", "attn.proj.bias"), ("down_b", "mlp.down_proj.bias")]}
                # gate and up fused row-wise, each padded with zero rows to a multiple of 8 (3420 -> 3424): the pad columns of
                # h = silu(0) * 0 are exactly 0 and meet zero columns of down_proj, so the result is unchanged bit for bit
                gu_w = torch.zeros((2 * Ip, E), dtype=torch.bfloat16, device=device)
                gu_b = torch.zeros(2 * Ip, dtype=torch.bfloat16, device=device)
                gu_w[:I].copy_(sd[p + "mlp.gate_proj.weight"]); gu_w[Ip:Ip + I].copy_(sd[p + "mlp.up_proj.weight"])
                gu_b[:I].copy_(sd[p + "mlp.gate_proj.bias"]); gu_b[Ip:Ip + I].copy_(sd[p + "mlp.up_proj.bias"])
                down_w = torch.zeros((E, Ip), dtype=torch.bfloat16, device=device)
                down_w[:, :I].copy_(sd[p + "mlp.down_proj.weight"])
                blk.update(gu_w=gu_w, gu_b=gu_b, down_w=down_w)
                self.vit.append(blk)
            self.merger = {k: dev(sd[m + n]) for k, n in [("ln_w", "ln_q.weight"), ("w0", "mlp.0.weight"), ("b0", "mlp.0.bias"),
                                                           ("w2", "mlp.2.weight"), ("b2", "mlp.2.bias")]}
        else:
            for b in range(vc.depth):
                p = f"{V_PREFIX}blocks.{b}."
                self.vit.append({k: dev(sd[p + n]) for k, n in [
                    ("n1w", "norm1.weight"), ("n1b", "norm1.bias"), ("n2w", "norm2.weight"), ("n2b", "norm2.bias"),
                    ("qkv_w", "attn.qkv.weight"), ("qkv_b", "attn.qkv.bias"), ("proj_w", "attn.proj.weight"),
                    ("proj_b", "attn.proj.bias"), ("fc1_w", "mlp.fc1.weight"), ("fc1_b", "mlp.fc1.bias"),
                    ("fc2_w", "mlp.fc2.weight"), ("fc2_b", "mlp.fc2.bias")]})
            self.merger = {k: dev(sd[m + n]) for k, n in [("ln_w", "ln_q.weight"), ("ln_b", "ln_q.bias"), ("w0", "mlp.0.weight"),
                                                           ("b0", "mlp.0.bias"), ("w2", "mlp.2.weight"), ("b2", "mlp.2.bias")]}
        self.embed = dev(sd[L_PREFIX + "embed_tokens.weight"])
        self.lm_head = self.embed if tc.tie_word_embeddings else dev(sd["lm_head.weight"])
        self.final_norm = dev(sd[L_PREFIX + "norm.weight"])
        self.layers = []
        for i in range(tc.num_layers):
            p = f"{L_PREFIX}layers.{i}."
            qkv_w = torch.cat([sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.v_proj.weight"]], 0)
            qkv_b = torch.cat([sd[p + "self_attn.q_proj.bias"], sd[p + "self_attn.k_proj.bias"], sd[p + "self_attn.v_proj.bias"]], 0)
            gu_w = torch.cat([sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"]], 0)
            # the four matrices of a layer live back to back in ONE buffer (streamed in this order by a decode step),
            # so the next layer can be pulled into the Infinity Cache with a single range prefetch
            mats = [("qkv_w", qkv_w), ("o_w", sd[p + "self_attn.o_proj.weight"]), ("gu_w", gu_w), ("down_w", sd[p + "mlp.down_proj.weight"])]
            flat = torch.empty(sum(m.numel() for _, m in mats), dtype=torch.bfloat16, device=device)
            layer = dict(ln1=dev(sd[p + "input_layernorm.weight"]), ln2=dev(sd[p + "post_attention_layernorm.weight"]), qkv_b=dev(qkv_b), flat=flat)
            off = 0
            for name, m in mats:
                view = flat[off:off + m.numel()].view(m.shape)
                view.copy_(m)
                layer[name] = view
                off += m.numel()
            self.layers.append(layer)

    def nbytes_llm_decode(self) -> int:
        """Weight bytes one decode step streams (layers + final norm + lm_head)."""
        n = self.lm_head.numel() + self.final_norm.numel()
        for l in self.layers:
            n += sum(t.numel() for k, t in l.items() if k != "flat")
        return n * 2
