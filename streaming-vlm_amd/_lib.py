"""ctypes binding of libsvlm_hip.so (C ABI declared in include/svlm.h).

The product path has NO fallback: if the shared library is missing or a call fails, this module
raises.  Build with ``python -c "import __graft_entry__ as g; g.build()"`` (hipcc, gfx950).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SVLM_LIB_PATH") or os.path.join(_HERE, "libsvlm_hip.so")      # override: diagnostic builds (tools/)

SVLM_OK = 0
ACT_NONE, ACT_QUICK_GELU, ACT_GELU_ERF, ACT_SILU, ACT_SWIGLU = 0, 1, 2, 3, 4


class SvlmError(RuntimeError):
    pass


_p, _i, _f, _ll = C.c_void_p, C.c_int, C.c_float, C.c_longlong

# name -> (restype, argtypes); must list every symbol of include/svlm.h (tests/test_abi.py checks it)
SIGNATURES = {
    "svlm_abi_version": (_i, []),
    "svlm_last_error": (C.c_char_p, []),
    "svlm_device_cus": (_i, []),
    "svlm_gemm_bf16": (_i, [_p, _i, _p, _i, _p, _p, _i, _p, _i, _i, _i, _i, _i, _p, _ll, _p]),
    "svlm_gemm_bf16_norm": (_i, [_p, _i, _p, _i, _p, _p, _i, _p, _i, _i, _i, _i, _i, _p, _ll, _p, _p, _f, _p, _i, _p]),
    "svlm_quant_rows_fp8": (_i, [_p, _i, _p, _i, _p, _i, _i, _p]),
    "svlm_gemm_fp8": (_i, [_p, _i, _p, _p, _i, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _p, _ll, _p, _p, _f, _p, _i, _p]),
    "svlm_gemm_fp8_normq": (_i, [_p, _i, _p, _p, _i, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _p, _ll, _p, _p, _f, _p, _i, _p, _i, _p, _p]),
    "svlm_gemv_bf16": (_i, [_p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _p]),
    "svlm_prefetch": (_i, [_p, _ll, _i, _p]),
    "svlm_rmsnorm": (_i, [_p, _p, _p, _i, _i, _f, _p]),
    "svlm_layernorm": (_i, [_p, _p, _p, _p, _i, _i, _f, _p]),
    "svlm_add": (_i, [_p, _p, _p, _ll, _p]),
    "svlm_silu_mul": (_i, [_p, _p, _i, _i, _p]),
    "svlm_gather_rows": (_i, [_p, _p, _p, _p, _p, _i, _i, _p]),
    "svlm_patchify_u8": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f, _p]),
    "svlm_resize_aa_tables": (_i, [_i, _i, _p, _p, _p, _i]),
    "svlm_resize_ws_bytes": (_ll, [_i, _i, _i]),
    "svlm_resize_bicubic_aa_u8": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p, _p, _i, _p, _ll, _p]),
    "svlm_vit_rope": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "svlm_vit_attn": (_i, [_p, _p, _i, _i, _i, _i, _f, _p]),
    "svlm_mrope_table": (_i, [_p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "svlm_rope_index_ws_bytes": (_ll, [_i, _i]),
    "svlm_rope_index": (_i, [_p, _i, _p, _i, _i, _i, _i, _p, _p, _i, _f, _f, _i, _p, _ll, _p]),
    "svlm_evict_plan_ws_bytes": (_ll, [_i]),
    "svlm_evict_plan": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _ll, _p]),
    "svlm_kv_append": (_i, [_p, _i, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "svlm_kv_move_rows": (_i, [_p, _ll, _i, _i, _p, _p, _i, _p]),
    "svlm_kv_gather": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "svlm_decode_attn_ws_bytes": (_ll, [_i, _i, _i]),
    "svlm_decode_attn_ropeload": (_i, [_p, _p, _p, _p, _p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p]),
    "svlm_prefill_attn_ws_bytes": (_ll, [_i, _i, _i, _i]),
    "svlm_prefill_attn_ropeload": (_i, [_p, _i, _p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p, _ll, _p]),
    "svlm_prefill_attn_ropeload_lin": (_i, [_p, _i, _p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p, _ll, _p, _p, _i, _p, _p]),
    "svlm_decode_attn_lin": (_i, [_p, _p, _p, _p, _p, _p, _i, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p]),
    "svlm_mark_seen": (_i, [_p, _i, _p, _i, _p]),
    "svlm_argmax_ws_bytes": (_ll, []),
    "svlm_penalty_argmax": (_i, [_p, _i, _p, _f, _p, _i, _p, _p, _i, _p, _p]),
    "svlm_penalty_sample": (_i, [_p, _i, _p, _f, _p, _i, _f, _i, _f, _p, _p, _p, _i, _p, _p]),
    "svlm_dec_lm_head_sample": (_i, [_p, _p, _f, _p, _i, _p, _p, _f, _p, _i, _p, _i, _i, _f, _p, _p, _p]),
    "svlm_dec_qkv": (_i, [_p, _p, _f, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "svlm_dec_qkv_lin": (_i, [_p, _p, _f, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _i, _p]),
    "svlm_dec_gate_up": (_i, [_p, _p, _f, _p, _i, _p, _i, _i, _p]),
    "svlm_dec_tail_supported": (_i, [_i, _i, _i, _i, _i]),
    "svlm_dec_tail_ws_bytes": (_ll, [_i, _i, _i]),
    "svlm_dec_tail_reset": (_i, [_p, _i, _i, _i, _p]),
    "svlm_dec_tail": (_i, [_p, _p, _p, _i, _p, _p, _i, _p, _i, _p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p, _i, _i, _i, _p, _p]),
    "svlm_dec_lm_head_ws_bytes": (_ll, [_i]),
    "svlm_dec_lm_head": (_i, [_p, _p, _f, _p, _i, _p, _p, _f, _p, _i, _p, _i, _i, _p]),
    "svlm_argmax_finish": (_i, [_p, _i, _p, _p, _p, _i, _p]),
}

_lib = None


def load():
    """Load the library once; raise loudly when it is absent (no CPU / torch fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SvlmError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c \"import __graft_entry__ as g; "
            f"g.build()\"` at the repo root (needs hipcc). There is no fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError = ABI mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.svlm_abi_version() != 1:
        raise SvlmError(f"libsvlm_hip.so ABI version {lib.svlm_abi_version()} != 1")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != SVLM_OK:
        msg = load().svlm_last_error()
        raise SvlmError(f"{what} failed with status {rc}: {msg.decode() if msg else ''}")
