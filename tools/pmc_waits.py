#!/usr/bin/env python
"""Reduce a tools/run_pmc_waits.sh pass: per kernel, the share of wave cycles spent parked (SQ_WAIT_ANY: s_waitcnt / barrier),
issue-stalled (SQ_WAIT_INST_ANY; of which LDS: SQ_WAIT_INST_LDS) and issuing (SQ_ACTIVE_INST_ANY), LDS bank-conflict cycles per
LDS instruction.  Usage: pmc_waits.py <label> <kernel-substring> <counter_collection.csv>  -> profiles/pmc_waits.json"""
import csv, json, os, sys

label, name, path = sys.argv[1:4]
vals = {}
with open(path) as f:
    for r in csv.DictReader(f):
        if name in r["Kernel_Name"]:
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
if not vals:
    raise SystemExit(f"no rows for *{name}*")
a = {k: sum(v[len(v) // 4:]) / len(v[len(v) // 4:]) for k, v in vals.items()}
wc = a["SQ_WAVE_CYCLES"]
out = {"kernel": name, "launches": len(next(iter(vals.values()))),
       "parked_waitcnt_or_barrier": round(a["SQ_WAIT_ANY"] / wc, 3), "issue_stalled": round(a["SQ_WAIT_INST_ANY"] / wc, 3),
       "issue_stalled_on_lds": round(a["SQ_WAIT_INST_LDS"] / wc, 3), "issuing": round(a["SQ_ACTIVE_INST_ANY"] / wc, 3),
       "issuing_lds": round(a["SQ_ACTIVE_INST_LDS"] / wc, 3),
       "lds_bank_conflict_cycles_per_lds_inst": round(a["SQ_LDS_BANK_CONFLICT"] / max(a["SQ_INSTS_LDS"], 1), 2), "raw": a}
p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_waits.json")
try:
    data = json.load(open(p))
except Exception:
    data = {}
data[label] = out
json.dump(data, open(p, "w"), indent=1)
print(label, {k: v for k, v in out.items() if k != "raw"})
