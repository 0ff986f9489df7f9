#!/usr/bin/env python
"""Kernels of one step that run with few waves on the chip: tools/lowocc.py <kernel_trace.csv> - per kernel name and grid: launches per
step, average us, waves launched, waves per SIMD (1 024 SIMDs).  Long launches at <= 2 waves per SIMD are the latency-bound candidates."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
seg = rows[ends[-3] + 1:ends[-1] + 1] if len(ends) >= 3 else rows        # one step = up to the last of its adamw launches
agg = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    k = (r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:70], g // 64, wg, int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"]), int(r["LDS_Block_Size"]))
    agg[k][0] += 1
    agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("%-72s %5s %8s %8s %6s %5s %6s %7s" % ("kernel", "n", "us each", "us total", "waves", "w/SIMD", "regs", "LDS"))
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if k[1] <= 3072 and t / n >= 12:
        print("%-72s %5d %8.1f %8.1f %6d %5.2f %6d %7d" % (k[0], n, t / n, t, k[1], k[1] / 1024.0, k[3], k[4]))
