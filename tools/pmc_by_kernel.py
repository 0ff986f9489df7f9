#!/usr/bin/env python
"""Per-kernel averages of the counters of one rocprofv3 --pmc pass (+ --kernel-trace for the durations):
   tools/pmc_by_kernel.py <dir> [kernel-name substring ...]"""
import collections, csv, glob, os, sys
d = sys.argv[1]
ct = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
dur = {}
if kt:
    for r in csv.DictReader(open(kt[0])):
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
names = set()
for r in csv.DictReader(open(ct)):
    k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:64]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    names.add(r["Counter_Name"])
    if r["Dispatch_Id"] not in cnt[k]:
        cnt[k].add(r["Dispatch_Id"])
        agg[k]["_ns"] += dur.get(r["Dispatch_Id"], 0)
names = sorted(names)
keys = sys.argv[2:]
print("%-66s %6s %9s " % ("kernel", "n", "avg us") + " ".join("%16s" % n[:16] for n in names))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["_ns"])[:60]:
    if keys and not any(s in k for s in keys):
        continue
    n = len(cnt[k])
    print("%-66s %6d %9.1f " % (k, n, v["_ns"] / n / 1e3) + " ".join("%16.0f" % (v[c] / n) for c in names))
