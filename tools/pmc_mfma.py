#!/usr/bin/env python
"""MFMA-pipe utilisation of the path's matrix kernels from one rocprofv3 --pmc pass per microbenchmark.
Usage: pmc_mfma.py <label> <kernel-substring> <counter_collection.csv> <flop per launch> <kernel_trace.csv>
Counters: SQ_VALU_MFMA_BUSY_CYCLES (MFMA pipe busy cycles, summed over the chip's SIMDs), SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE.
utilisation = flop / (duration x 2.5 PFLOP/s dense bf16 peak) is the figure bench.py reports; the counter ratio
MFMA_BUSY / (1024 SIMDs x GRBM_GUI_ACTIVE per XCD) -- rocprofv3's MfmaUtil -- is printed beside it, with the busy cycles per wave-level MFMA instruction it implies
(16 for v_mfma_f32_16x16x32_bf16 when the counter counts what the guide says it counts)."""
import csv, json, os, sys

label, name, pmc_csv, flop, trace_csv = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4]), sys.argv[5]
vals = {}
with open(pmc_csv) as f:
    for r in csv.DictReader(f):
        if name in r["Kernel_Name"]:
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
if not vals:
    raise SystemExit(f"no rows for kernel *{name}* in {pmc_csv}")
avg = {k: sum(v[len(v) // 4:]) / len(v[len(v) // 4:]) for k, v in vals.items()}
dur = []
with open(trace_csv) as f:
    for r in csv.DictReader(f):
        if name in r["Kernel_Name"]:
            dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
dur = dur[len(dur) // 4:]
us = sum(dur) / len(dur) / 1e3
n_mfma = flop / (2 * 16 * 16 * 32)
out = {"kernel": name, "launches": len(dur), "avg_us_under_pmc": round(us, 2), "flop_per_launch": flop,
       "tflops": round(flop / us / 1e6, 1), "mfma_util_vs_2500_tflops": round(flop / us / 1e6 / 2500, 4), "counters": avg}
if "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
    out["mfma_busy_cycles_per_wave_mfma"] = round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / n_mfma, 2)
    if avg.get("GRBM_GUI_ACTIVE"):
        # rocprofv3's own MfmaUtil = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max over XCDs of GRBM_GUI_ACTIVE x SIMD_NUM); the CSV row is the
        # SUM over the 8 XCDs' GRBM instances, and GUI_ACTIVE spans the dispatch's launch and drain as well as the kernel
        out["mfma_util_counter"] = round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * avg["GRBM_GUI_ACTIVE"] / 8), 4)
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_mfma.json")
try:
    data = json.load(open(path))
except Exception:
    data = {}
data[label] = out
json.dump(data, open(path, "w"), indent=1)
print(json.dumps(out))
