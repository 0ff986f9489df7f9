#!/usr/bin/env python
"""Per-step summary of a rocprofv3 --kernel-trace CSV of bench.py: kernels of ONE graph replay (between two AdamW
launches) by category and by name.  usage: tools/trace_step.py <kernel_trace.csv> [top_n]"""
import collections
import csv
import sys


def short(n):
    return n.replace("void ", "").replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")[:78]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]][1::2]
    seg = rows[ends[-4] + 1:ends[-3] + 1]
    span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
    agg = collections.defaultdict(lambda: [0, 0])
    for r in seg:
        k = short(r["Kernel_Name"])
        agg[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        agg[k][1] += 1
    tot = sum(v[0] for v in agg.values())
    cats = collections.defaultdict(lambda: [0, 0])
    for k, (d, c) in agg.items():
        if "wgrad" in k:
            g = "conv / linear weight gradient (MFMA)"
        elif "igemm" in k:
            g = "conv / linear forward + data gradient (MFMA)"
        elif k.startswith("at::") or "rocclr" in k:
            g = "ATen elementwise / copy / fill / reduce"
        elif k.startswith("Cijk"):
            g = "rocBLAS batched GEMM"
        else:
            g = "own non-GEMM kernels"
        cats[g][0] += d
        cats[g][1] += c
    print("one step: %d kernels, span %.2f ms, busy %.2f ms" % (len(seg), span, tot / 1e6))
    for g, (d, c) in sorted(cats.items(), key=lambda x: -x[1][0]):
        print("  %-46s %6.2f ms %5.1f%% %5d launches" % (g, d / 1e6, 100 * d / tot, c))
    print()
    for k, (d, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:top]:
        print("%6.2f ms %5d x %7.1f us  %s" % (d / 1e6, c, d / c / 1e3, k))


if __name__ == "__main__":
    main()
