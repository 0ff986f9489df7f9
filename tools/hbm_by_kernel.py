#!/usr/bin/env python
"""HBM-side traffic per kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over bench.py.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d A -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d B -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph
    python tools/hbm_by_kernel.py A B > profiles/r02_hbm_by_kernel.txt

Bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: the counters are in KiB, and on gfx950 FETCH_SIZE reports half of the bytes of
wide coalesced reads (MI355X_MICROARCH.md, "HBM").  Infinity-Cache hits are INCLUDED in these counters (they sit on the L2's
memory side), so a figure above the 8 TB/s HBM peak means the operand was still resident in the 256 MiB on-die cache.
Durations are those of the counter run itself (dispatches are serialised there).  Every launch of the run is counted (warm-up
and timed steps alike); the table is per launch.
"""
import collections
import csv
import glob
import os
import re
import sys


def load(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        raise SystemExit("no counter_collection.csv under " + d)
    out = collections.defaultdict(lambda: [0, 0.0, 0.0])            # name -> launches, counter sum, ns
    for r in csv.DictReader(open(f[0])):
        if r.get("Counter_Name") != counter:
            continue
        n = short(r["Kernel_Name"])
        o = out[n]
        o[0] += 1
        o[1] += float(r["Counter_Value"])
        if r.get("End_Timestamp") and r.get("Start_Timestamp"):
            o[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return out


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"_ZN?\d*[A-Za-z_0-9]*?(\d+)([a-z_0-9]+_kernel)", n)
    if m:
        n = m.group(2) + n[m.end():][:24]
    return n[:70]


def main():
    rd, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    rows = []
    for k in set(rd) | set(wr):
        n = max(rd[k][0], wr[k][0])
        by = (2 * rd[k][1] + wr[k][1]) * 1024.0
        ns = rd[k][2] or wr[k][2]
        if n and ns:
            rows.append((by, n, ns, k))
    rows.sort(reverse=True)
    tot_b, tot_ns = sum(r[0] for r in rows), sum(r[2] for r in rows)
    print("all kernels: %.2f GB in %.2f ms of kernel time = %.0f GB/s (%.0f %% of the 8 TB/s HBM peak)" %
          (tot_b / 1e9, tot_ns / 1e6, tot_b / tot_ns, 100 * tot_b / tot_ns / 8000))
    print("%-70s %6s %10s %9s %8s %7s %7s" % ("kernel", "n", "MB/launch", "us/launch", "GB/s", "%8TB/s", "%6.3TB/s"))
    for by, n, ns, k in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 80]:
        gbs = by / ns
        print("%-70s %6d %10.2f %9.1f %8.0f %7.0f %7.0f" % (k, n, by / n / 1e6, ns / n / 1e3, gbs, 100 * gbs / 8000, 100 * gbs / 6300))


if __name__ == "__main__":
    main()
