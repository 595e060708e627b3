"""Model geometry of the Qwen2-VL family the streaming path runs (SURVEY section 8 header).

Public model-card values; `from_hf_config` asserts them against a real config.json when one is
available (no checkpoint exists offline, so BASELINE runs use random weights of these shapes).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List

# hard-coded Qwen token ids the reference relies on (src/streaming_vlm/utils/get_qwen_range.py:2-13)
IM_START, IM_END = 151644, 151645
ENDOFTEXT = 151643
VISION_START, VISION_END, VIDEO_PAD = 151652, 151653, 151656
LF = 198


@dataclass
class VisionConfig:
    depth: int = 32
    embed_dim: int = 1280
    num_heads: int = 16
    mlp_hidden: int = 5120
    patch_size: int = 14
    temporal_patch_size: int = 2
    spatial_merge_size: int = 2
    in_channels: int = 3
    # Qwen2.5-VL tower: RMSNorm + SwiGLU blocks, windowed attention except in `fullatt_block_indexes`,
    # merger output `out_hidden`; `mlp_hidden` is then the SwiGLU intermediate size (3420 for the released checkpoints)
    arch: str = "qwen2"
    window_size: int = 112
    fullatt_block_indexes: tuple = (7, 15, 23, 31)
    out_hidden: int = 0
    tokens_per_second: float = 2.0

    @property
    def mlp_padded(self):
        """SwiGLU intermediate size rounded up to 8 columns (16-B rows for the GEMM); the pad rows/columns are zero."""
        return (self.mlp_hidden + 7) // 8 * 8

    @property
    def head_dim(self):
        return self.embed_dim // self.num_heads

    @property
    def patch_dim(self):
        return self.in_channels * self.temporal_patch_size * self.patch_size * self.patch_size


@dataclass
class TextConfig:
    hidden_size: int = 1536
    num_layers: int = 28
    num_heads: int = 12
    num_kv_heads: int = 2
    head_dim: int = 128
    intermediate_size: int = 8960
    vocab_size: int = 151936
    rms_eps: float = 1e-6
    rope_theta: float = 1e6
    mrope_section: List[int] = field(default_factory=lambda: [16, 24, 24])
    tie_word_embeddings: bool = True


@dataclass
class ModelConfig:
    vision: VisionConfig = field(default_factory=VisionConfig)
    text: TextConfig = field(default_factory=TextConfig)
    video_token_id: int = VIDEO_PAD
    vision_start_token_id: int = VISION_START
    eos_token_ids: tuple = (IM_END, ENDOFTEXT)
    name: str = "qwen2-vl"

    @property
    def family(self) -> str:
        return "qwen2_5" if self.vision.arch == "qwen2_5" else "qwen2"


def qwen2_vl_2b() -> ModelConfig:
    return ModelConfig(VisionConfig(), TextConfig(), name="Qwen2-VL-2B")


def qwen2_vl_7b() -> ModelConfig:
    return ModelConfig(VisionConfig(), TextConfig(hidden_size=3584, num_layers=28, num_heads=28, num_kv_heads=4,
                                                  intermediate_size=18944, vocab_size=152064, tie_word_embeddings=False),
                       name="Qwen2-VL-7B")


def qwen2_5_vl_7b() -> ModelConfig:
    """Qwen2.5-VL-7B / the released StreamingVLM checkpoint (model-card values)."""
    return ModelConfig(VisionConfig(arch="qwen2_5", mlp_hidden=3420, out_hidden=3584),
                       TextConfig(hidden_size=3584, num_layers=28, num_heads=28, num_kv_heads=4, intermediate_size=18944,
                                  vocab_size=152064, tie_word_embeddings=False), name="Qwen2.5-VL-7B")


def qwen2_5_vl_3b() -> ModelConfig:
    return ModelConfig(VisionConfig(arch="qwen2_5", mlp_hidden=3420, out_hidden=2048),
                       TextConfig(hidden_size=2048, num_layers=36, num_heads=16, num_kv_heads=2, intermediate_size=11008,
                                  vocab_size=151936, tie_word_embeddings=True), name="Qwen2.5-VL-3B")


def tiny_2_5(depth=4, layers=2, vocab=151680) -> ModelConfig:
    """Qwen2.5-style tiny geometry: RMSNorm/SwiGLU tower with an intermediate size that needs padding (100 -> 104),
    windowed blocks with one full-attention block, 56-px windows (2x2 merged tokens)."""
    return ModelConfig(VisionConfig(arch="qwen2_5", depth=depth, embed_dim=160, num_heads=2, mlp_hidden=100, out_hidden=256,
                                    window_size=56, fullatt_block_indexes=(1,)),
                       TextConfig(hidden_size=256, num_layers=layers, num_heads=4, num_kv_heads=2, intermediate_size=512,
                                  vocab_size=vocab, tie_word_embeddings=True), name="tiny-2.5")


def tiny(depth=2, layers=2, vocab=151680) -> ModelConfig:
    """Small geometry with the real head sizes (ViT d=80, LLM d=128, mrope [16,24,24]) for parity tests."""
    return ModelConfig(VisionConfig(depth=depth, embed_dim=160, num_heads=2, mlp_hidden=320),
                       TextConfig(hidden_size=256, num_layers=layers, num_heads=4, num_kv_heads=2, intermediate_size=512,
                                  vocab_size=vocab, tie_word_embeddings=True), name="tiny")


def from_hf_config(hf) -> ModelConfig:
    """Build from a transformers Qwen2VLConfig (4.52+ nests text_config / vision_config)."""
    tc = getattr(hf, "text_config", hf)
    vc = hf.vision_config
    rp = getattr(tc, "rope_parameters", None) or getattr(tc, "rope_scaling", None) or {}
    section = list(rp.get("mrope_section", [16, 24, 24]))
    theta = rp.get("rope_theta", getattr(tc, "rope_theta", 1e6))
    head_dim = getattr(tc, "head_dim", None) or tc.hidden_size // tc.num_attention_heads
    text = TextConfig(hidden_size=tc.hidden_size, num_layers=tc.num_hidden_layers, num_heads=tc.num_attention_heads,
                      num_kv_heads=tc.num_key_value_heads, head_dim=head_dim, intermediate_size=tc.intermediate_size,
                      vocab_size=tc.vocab_size, rms_eps=tc.rms_norm_eps, rope_theta=float(theta), mrope_section=section,
                      tie_word_embeddings=bool(getattr(hf, "tie_word_embeddings", getattr(tc, "tie_word_embeddings", False))))
    if hasattr(vc, "fullatt_block_indexes"):           # Qwen2_5_VLVisionConfig
        if vc.hidden_act != "silu":
            raise ValueError(f"unsupported Qwen2.5 ViT activation {vc.hidden_act}")
        vision = VisionConfig(arch="qwen2_5", depth=vc.depth, embed_dim=vc.hidden_size, num_heads=vc.num_heads,
                              mlp_hidden=vc.intermediate_size, patch_size=vc.patch_size,
                              temporal_patch_size=vc.temporal_patch_size, spatial_merge_size=vc.spatial_merge_size,
                              in_channels=getattr(vc, "in_channels", getattr(vc, "in_chans", 3)), window_size=vc.window_size,
                              fullatt_block_indexes=tuple(vc.fullatt_block_indexes), out_hidden=vc.out_hidden_size,
                              tokens_per_second=float(getattr(vc, "tokens_per_second", 2)))
        if vc.out_hidden_size != tc.hidden_size:
            raise ValueError(f"merger output {vc.out_hidden_size} != LLM hidden {tc.hidden_size}")
    else:
        vision = VisionConfig(depth=vc.depth, embed_dim=vc.embed_dim, num_heads=vc.num_heads,
                              mlp_hidden=int(vc.embed_dim * vc.mlp_ratio), patch_size=vc.patch_size,
                              temporal_patch_size=vc.temporal_patch_size, spatial_merge_size=vc.spatial_merge_size,
                              in_channels=vc.in_channels)
        if vc.hidden_size != tc.hidden_size:
            raise ValueError(f"merger output {vc.hidden_size} != LLM hidden {tc.hidden_size}")
        if vc.hidden_act != "quick_gelu":
            raise ValueError(f"unsupported ViT activation {vc.hidden_act}")
    return ModelConfig(vision, text, video_token_id=hf.video_token_id, vision_start_token_id=hf.vision_start_token_id,
                       name=getattr(hf, "name_or_path", "") or "qwen2-vl")
