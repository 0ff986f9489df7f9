#!/usr/bin/env python
"""Folds the rocprofv3 --pmc passes over tools/roofline_kernel.py into profiles/pmc_dominant_kernel.json (read by bench.py).
usage: tools/pmc_dominant.py <dir FETCH_SIZE> <dir WRITE_SIZE> <dir TCC_HIT_sum TCC_MISS_sum> <out.json>"""
import csv
import datetime
import glob
import json
import os
import sys


def mean(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and "igemm_dma_kernel" in r["Kernel_Name"]]
    v = v[3:] if len(v) > 6 else v            # drop the cold launches
    return sum(v) / len(v), len(v)


fetch, n = mean(sys.argv[1], "FETCH_SIZE")
write, _ = mean(sys.argv[2], "WRITE_SIZE")
hit, _ = mean(sys.argv[3], "TCC_HIT_sum")
miss, _ = mean(sys.argv[3], "TCC_MISS_sum")
out = {
    "kernel": "igemm_dma_kernel<256,160,8,1,3,GM=1,...,HALO> (bf16) conv3x3 160->160 @ 8x120x160, data-gradient launch, halo-patch variant (tools/roofline_kernel.py, %d launches averaged)" % n,
    "measured": datetime.datetime.now().strftime("%Y-%m-%d %H:%M round 3"),
    "commit": (sys.argv[5] if len(sys.argv) > 5 else "?"),
    "how": "rocprofv3 --pmc <counters> --kernel-trace --output-format csv, one pass per counter group (FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum); "
           "MI355X_MICROARCH.md HBM section: bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (FETCH_SIZE counts half of a 16 B/lane stream on gfx950; Infinity-Cache hits are counted)",
    "FETCH_SIZE_KB_mean_per_launch": fetch, "WRITE_SIZE_KB_mean_per_launch": write,
    "TCC_HIT_sum_mean_per_launch": hit, "TCC_MISS_sum_mean_per_launch": miss, "l2_hit_rate": round(hit / (hit + miss), 4),
    "traffic_bytes_per_launch": (2 * fetch + write) * 1024, "algorithmic_bytes_per_launch": 98764800,
}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out, indent=1))
