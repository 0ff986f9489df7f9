#!/usr/bin/env python
"""GPU clock seen by each kernel family: GRBM_GUI_ACTIVE (cycles the GPU was busy) of a dispatch over its duration.
   rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d D -- python3 <prog>
   tools/clock_by_kernel.py D [substring ...]"""
import collections, csv, glob, os, sys
d = sys.argv[1]
ct = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
agg = collections.defaultdict(lambda: [0, 0, 0])
for r in csv.DictReader(open(ct)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or r["Dispatch_Id"] not in dur:
        continue
    ns, name = dur[r["Dispatch_Id"]]
    k = name.replace("void ", "")[:70]
    a = agg[k]
    a[0] += float(r["Counter_Value"]); a[1] += ns; a[2] += 1
keys = sys.argv[2:]
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print("%-72s %6s %9s %8s" % ("kernel", "n", "avg us", "GHz"))
for k, (cyc, ns, n) in rows[:40]:
    if keys and not any(s in k for s in keys):
        continue
    print("%-72s %6d %9.1f %8.3f" % (k, n, ns / n / 1e3, cyc / ns))
tot_c, tot_ns = sum(v[0] for v in agg.values()), sum(v[1] for v in agg.values())
print("all kernels: %.3f GHz over %.1f ms of kernel time" % (tot_c / tot_ns, tot_ns / 1e6))
