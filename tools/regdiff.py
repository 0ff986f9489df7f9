#!/usr/bin/env python
"""Per-kernel register / scratch / occupancy differences between two builds of one source file:
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Rpass-analysis=kernel-resource-usage -c x.hip -o /dev/null 2> before.txt   (and after.txt)
    tools/regdiff.py before.txt after.txt
Worth running after every epilogue change: a run-time branch that 'costs nothing' has moved whole kernels to another occupancy step
more than once (thin data gradient 64 -> 256 VGPRs, 160-wide tiles into scratch)."""
import re, sys


def parse(f):
    out, name = {}, None
    for l in open(f):
        m = re.search(r"Function Name: (\S+)", l)
        if m:
            name = m.group(1)
            out[name] = {}
        for k in ("VGPRs:", "AGPRs:", "ScratchSize", "Occupancy"):
            m = re.search(r"remark: .*?" + re.escape(k) + r".*?(\d+)", l)
            if m and name and k in l:
                out[name][k.rstrip(":")] = int(m.group(1))
    return out


a, b = parse(sys.argv[1]), parse(sys.argv[2])
for k in b:
    if k in a and a[k] != b[k]:
        print(k[:100], a[k], "->", b[k])
print(len(a), "kernels before,", len(b), "after,", len(set(a) ^ set(b)), "renamed / new")
