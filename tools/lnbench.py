#!/usr/bin/env python
"""LayerNorm forward / backward at the step's shapes on operand sets that do NOT stay in the 256 MiB on-die cache: every call works on
the next of a ring of buffer sets totalling > 600 MB, so the GB/s are HBM figures (a single warm set overstates)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gw_depth_amd import hip

lib = hip.library()


def timeit(fns, n=60):
    for f in fns:
        f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for i in range(n):
        fns[i % len(fns)]()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1000


print("library", os.environ.get("GWD_LIB", "(in tree)"))
for rows, C, gelu in [(153600, 160, True), (153600, 160, False), (153600, 64, False), (153600, 128, False), (38400, 64, True), (38400, 128, False),
                      (162288, 64, False), (42336, 128, False), (11760, 256, False), (2400, 256, False), (800, 256, False)]:
    per_set = rows * C * 2 * 3
    sets = max(2, int(640e6 // per_set) + 1)
    bufs = []
    for _ in range(sets):
        x = torch.randn(rows, C, device="cuda").bfloat16()
        bufs.append((x, torch.randn_like(x), torch.empty_like(x), torch.empty_like(x), torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")))
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    tf = timeit([(lambda t=t: lib.layernorm_forward(t[0], g, b, t[2], t[4], t[5], rows, C, gelu)) for t in bufs])
    tb = timeit([(lambda t=t: lib.layernorm_backward(t[1], t[0], g, b, t[4], t[5], t[3], dg, db, rows, C, gelu)) for t in bufs])
    mb = rows * C * 2 / 1e6
    print("rows %6d C %4d %-4s (%5.1f MB / map, %2d sets): fwd %6.1f us %5.0f GB/s | bwd %6.1f us %5.0f GB/s" %
          (rows, C, "gelu" if gelu else "", mb, sets, tf, 2 * mb / tf * 1e3, tb, 3 * mb / tb * 1e3), flush=True)
    del bufs
