#!/usr/bin/env python
"""Largest idle gaps between consecutive kernels of one step of a kernel trace: tools/gaps.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]][1::2]
for which in (-4, -3, -2):
    seg = rows[ends[which] + 1:ends[which + 1] + 1]
    span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e6
    gaps = sorted(((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3, a["Kernel_Name"][:60], b["Kernel_Name"][:60]) for a, b in zip(seg, seg[1:]))
    print("step %d: %d kernels, span %.2f ms, busy %.2f ms; largest gaps (us):" % (which, len(seg), span, busy))
    for g in gaps[-4:]:
        print("   %8.1f  after %s  before %s" % g)
