#!/usr/bin/env python
"""LayerNorm forward/backward timing at the step's shapes, with and without the affine-gradient reduction."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gw_depth_amd import hip
lib = hip.library()
def timeit(fn, n=50):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1000
for rows, C in [(2400, 256), (800, 256), (2400, 512), (9600, 512), (38400, 128), (153600, 64), (38400, 64), (153600, 160), (38400, 60)]:
    x = torch.randn(rows, C, device="cuda").bfloat16(); gy = torch.randn_like(x)
    g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
    y = torch.empty_like(x); gx = torch.empty_like(x)
    mean = torch.empty(rows, device="cuda"); rstd = torch.empty(rows, device="cuda")
    dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    tf = timeit(lambda: lib.layernorm_forward(x, g, b, y, mean, rstd, rows, C, False))
    tb = timeit(lambda: lib.layernorm_backward(gy, x, g, b, mean, rstd, gx, dg, db, rows, C, False))
    tb0 = timeit(lambda: lib.layernorm_backward(gy, x, g, b, mean, rstd, gx, None, None, rows, C, False))
    mb = rows * C * 2 / 1e6
    print("rows %6d C %4d (%.1f MB): fwd %6.1f us  bwd %6.1f us  bwd without dgamma/dbeta %6.1f us" % (rows, C, mb, tf, tb, tb0))
