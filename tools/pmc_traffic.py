#!/usr/bin/env python
"""HBM traffic of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950).
Usage: pmc_traffic.py <kernel-substring> <fetch_counter_collection.csv> <write_counter_collection.csv> <algorithmic bytes> <command> [model name]
(the summary is keyed by model name first -- bench.py only quotes a traffic figure measured on the same model's shapes)
FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: gfx950 tallies the 128-B requests of wide coalesced streaming reads
at 64 B); both counters are in KiB."""
import csv, json, os, sys

name, f_csv, w_csv, algo, cmd = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
model = sys.argv[6] if len(sys.argv) > 6 else "Qwen2-VL-2B"


def avg(path, counter):
    vals = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if name in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no {counter} rows for kernel *{name}* in {path}")
    vals = vals[len(vals) // 4:]            # skip warm-up launches
    return sum(vals) / len(vals), len(vals)


fetch, n = avg(f_csv, "FETCH_SIZE")
write, _ = avg(w_csv, "WRITE_SIZE")
out_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
try:
    data = json.load(open(out_path))
except Exception:
    data = {}
data.setdefault(model, {})[name] = {"hbm_bytes_per_launch": int(round(fetch * 2 * 1024 + write * 1024)), "fetch_size_kb_raw_avg": fetch,
              "write_size_kb_raw_avg": write,
              "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B for wide coalesced streaming reads, "
                            "MI355X_MICROARCH.md HBM section) x1024; WRITE_SIZE x1024",
              "algorithmic_bytes_per_launch": algo, "command": cmd, "launches": n}
json.dump(data, open(out_path, "w"), indent=1)
print(json.dumps(data[model][name]))
