"""Diagnostic build of the HIP library: the same sources with -DSVLM_TUNING, i.e. WITH the environment-driven tuning switches
(SVLM_GEMM_BM, SVLM_PREFILL_SPLITS, SVLM_DA_DIAG, ...) and the timing-only DIAG kernels the product library does not contain.
    python tools/build_diag_lib.py        ->  streaming-vlm_amd/build/libsvlm_hip_diag.so
    SVLM_LIB_PATH=streaming-vlm_amd/build/libsvlm_hip_diag.so python tools/gemm_sweep.py ...
"""
import glob, os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CSRC = os.path.join(ROOT, "streaming-vlm_amd", "csrc")
OUT = os.path.join(ROOT, "streaming-vlm_amd", "build")
os.makedirs(os.path.join(OUT, "diag"), exist_ok=True)
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
objs = []
jobs = []
for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
    obj = os.path.join(OUT, "diag", os.path.basename(src)[:-4] + ".o")
    objs.append(obj)
    jobs.append([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-DSVLM_TUNING"] + os.environ.get("SVLM_DIAG_DEFS", "").split() + ["-c", src, "-o", obj])
with ThreadPoolExecutor(max_workers=4) as ex:
    list(ex.map(lambda j: subprocess.check_call(j, cwd=CSRC), jobs))
lib = os.path.join(OUT, os.environ.get("SVLM_DIAG_NAME", "libsvlm_hip_diag.so"))
subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, cwd=CSRC)
print(lib)
