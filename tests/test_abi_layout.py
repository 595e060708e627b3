"""The C-ABI boundary and the repo layout contract (no GPU needed, no compute calls)."""
import ast
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "streaming-vlm_amd")


@pytest.fixture(scope="module")
def built():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build()                       # hipcc cross-compiles gfx950 without a GPU
    return os.path.join(PKG, "libsvlm_hip.so")


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "svlm.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(svlm_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound(built):
    from streaming_vlm_amd import _lib
    declared = _header_symbols()
    assert len(declared) >= 25
    assert sorted(_lib.SIGNATURES) == declared, "ctypes table and include/svlm.h disagree"
    lib = ctypes.CDLL(built)
    for name in declared:
        getattr(lib, name)                                   # AttributeError = missing export
    nm = subprocess.run(["nm", "-D", "--defined-only", built], capture_output=True, text=True).stdout
    exported = sorted(set(re.findall(r" T (svlm_[a-z0-9_]+)", nm)))
    assert exported == declared, f"library exports differ from the header: {set(exported) ^ set(declared)}"
    assert _lib.load().svlm_abi_version() == 1
    assert _lib.load().svlm_decode_attn_ws_bytes(12, 2716, 32) > 0        # pure host arithmetic, safe without a GPU
    # rotated q / gathered k, v (bf16) + 8 key splits of fp32 (O, m, l) partials: 60 query tiles alone cannot fill 256 CUs
    # (flash_attn.hip aims at ~480 workgroups -> 8 splits of the 2B chunk)
    assert _lib.load().svlm_prefill_attn_ws_bytes(275, 2330, 12, 2) == (275 * 12 + 2 * 2 * 2330) * 256 + 8 * 275 * 12 * 130 * 4
    # 252 query tiles: 2 splits; 700 tiles: none
    assert _lib.load().svlm_prefill_attn_ws_bytes(560, 4600, 28, 4) == (560 * 28 + 2 * 4 * 4600) * 256 + 2 * 560 * 28 * 130 * 4
    assert _lib.load().svlm_prefill_attn_ws_bytes(1600, 4600, 28, 4) == (1600 * 28 + 2 * 4 * 4600) * 256


def test_header_cites_the_reference_for_every_entry_point():
    txt = open(os.path.join(ROOT, "include", "svlm.h")).read()
    assert txt.count("replaces:") >= 15
    assert "language_forward.py" in txt and "vision_forward.py" in txt and "streaming_cache.py" in txt and "inference.py" in txt


def test_product_library_takes_no_behaviour_from_the_environment(built):
    """Tuning switches (SVLM_GEMM_BM, SVLM_DA_DIAG, ...) and the timing-only DIAG kernels exist in the diagnostic build only
    (-DSVLM_TUNING, tools/build_diag_lib.py): the product library neither imports getenv nor carries a switch name."""
    nm = subprocess.run(["nm", "-D", "--undefined-only", built], capture_output=True, text=True).stdout
    assert "getenv" not in nm
    blob = open(built, "rb").read()
    assert b"SVLM_DA_DIAG" not in blob and b"SVLM_GEMM_BM" not in blob and b"SVLM_PREFILL_SPLITS" not in blob
    assert b"decode_attn_long_kernel" not in blob


def test_library_targets_gfx950_only(built):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", built], capture_output=True, text=True).stdout
    archs = set(re.findall(r"gfx[0-9a-f]+", out))
    assert archs == {"gfx950"}, archs


def test_product_never_imports_the_oracle_or_a_fallback():
    bad = []
    for dirpath, _, files in os.walk(PKG):
        for fn in files:
            if not fn.endswith(".py"):
                continue
            path = os.path.join(dirpath, fn)
            tree = ast.parse(open(path).read())
            for node in ast.walk(tree):
                mods = []
                if isinstance(node, ast.Import):
                    mods = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom) and node.module:
                    mods = [node.module]
                for m in mods:
                    if m.split(".")[0] in ("oracle", "ref_ops", "helpers", "tests"):
                        bad.append((path, m))
    assert not bad, bad
    for fn in ("bench.py", "__graft_entry__.py"):
        src = open(os.path.join(ROOT, fn)).read()
        # the oracle may only appear in the cpu_baseline leg / smoke()
        assert "import oracle" not in src.split("def cpu_baseline")[0].split("def smoke")[0].replace("import oracle  # noqa", "")


def test_hip_path_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from streaming_vlm_amd import _lib
    from streaming_vlm_amd.ops import HipOps
    with pytest.raises(_lib.SvlmError, match="no HIP device"):
        HipOps()
    import streaming_vlm_amd as S
    from streaming_vlm_amd import config as C
    with pytest.raises(_lib.SvlmError):
        S.StreamingQwen2VL(C.tiny(), {}, "cpu")              # default ops = HIP; never a silent CPU path


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    from streaming_vlm_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.SvlmError, match="not built"):
        _lib.load()


def test_layout_contract():
    for p in ("bench.py", "__graft_entry__.py", "DESIGN.md", "INTEGRATION.md", "include/svlm.h", "oracle/__init__.py",
              "oracle/make_golden.py", "tests/golden", "profiles", "streaming-vlm_amd/csrc"):
        assert os.path.exists(os.path.join(ROOT, p)), p
    assert "TEST INFRASTRUCTURE" in open(os.path.join(ROOT, "oracle", "__init__.py")).read()
    gi = open(os.path.join(ROOT, ".gitignore")).read()
    assert "*.so" in gi
