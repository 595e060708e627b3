"""Idle-gap analysis of a rocprofv3 --kernel-trace CSV: per chunk (delimited by argmax/penalty kernels count), how long the
GPU had NO kernel running, and the biggest gaps with the kernels either side."""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
rows = rows[skip:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy_end = rows[0][0]
idle = 0
gaps = []
for s, e, n in rows:
    if s > busy_end:
        idle += s - busy_end
        gaps.append((s - busy_end, prev, n))
    if e > busy_end:
        busy_end, prev = e, n
print(f"span {1e-6*(t1-t0):.2f} ms, idle {1e-6*idle:.2f} ms ({100*idle/(t1-t0):.1f}%), kernels {len(rows)}")
hist = collections.Counter()
for g, a, b in gaps:
    hist[(a, b)] += g
print("idle by (prev kernel -> next kernel), ms:")
for (a, b), g in hist.most_common(25):
    cnt = sum(1 for x in gaps if x[1] == a and x[2] == b)
    print(f"  {1e-6*g:8.3f}  n={cnt:5d} avg {1e-3*g/cnt:7.2f} us   {a}  ->  {b}")
