#!/usr/bin/env python3
"""usage: timeline_gaps.py kernel_trace.csv  -- per MSM call: kernel busy time, span, idle gaps."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"].split("(")[0].split("::")[-1][:28], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# split into calls at msm_digits
calls, cur = [], []
for k in ks:
    if "msm_digits" in k[0] and cur: calls.append(cur); cur = []
    cur.append(k)
calls.append(cur)
for c in calls[-3:]:
    if not any("msm_digits" in k[0] for k in c): continue
    span = (c[-1][2] - c[0][1]) / 1e6; busy = sum(e - s for _, s, e in c) / 1e6
    print(f"call: {len(c)} kernels  span {span:.3f} ms  busy {busy:.3f} ms  idle {span-busy:.3f} ms")
    prev = None
    for nm, s, e in c:
        gap = (s - prev) / 1e3 if prev else 0.0
        print(f"   {nm:30s} dur {1e-3*(e-s):10.1f} us   gap before {gap:8.1f} us")
        prev = e
