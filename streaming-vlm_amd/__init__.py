"""streaming-vlm_amd: MI355X-native (gfx950) hot path of rahim-xelpmoc/streaming-vlm.

Importable as ``streaming_vlm_amd`` (the sibling shim package maps the name onto this directory,
whose hyphen is not a valid identifier).  Public surface = the reference's:

    from streaming_vlm_amd import streaming_inference, load_model_and_processor
    from streaming_vlm_amd import convert_qwen2_to_streaming, StreamingArgs, get_qwen_range

All arithmetic runs in ``libsvlm_hip.so`` (C ABI: include/svlm.h); importing the package does not
need a GPU, constructing an engine does and fails loudly otherwise.
"""
from .config import ModelConfig, TextConfig, VisionConfig, qwen2_vl_2b, qwen2_vl_7b, tiny  # noqa: F401
from .spans import SYSTEM_PROMPT_OFFSET, TOKEN_IDS, get_qwen_range  # noqa: F401
from .kv_pool import KVPool  # noqa: F401
from .engine import SvlmEngine  # noqa: F401
from .model import (StreamingArgs, StreamingQwen2VL, convert_qwen2_5_to_streaming, convert_qwen2_to_streaming,  # noqa: F401
                    streaming_generate)
from .driver import (contiguous_id_and_kv, load_model_and_processor, open_vtt, process_past_kv, prune_id_and_kv_cache,  # noqa: F401
                     resort_id_and_kv, sec2ts, sink_window_evict, streaming_inference)
from .synthetic import (DeviceFrameProcessor, PinnedVideo, SyntheticProcessor, SyntheticVideo, patchify,  # noqa: F401
                        synthetic_frame)
from .weights import random_state_dict  # noqa: F401

__version__ = "0.1.0"
