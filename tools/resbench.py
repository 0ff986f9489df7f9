#!/usr/bin/env python
"""resample forward/backward timing at the step's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gw_depth_amd import hip
lib = hip.library()
def timeit(fn, n=30):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1000
for (B, Hs, Ws, Ho, Wo, C, mode) in [(8, 7, 10, 120, 160, 160, 0), (8, 15, 20, 120, 160, 160, 0), (8, 30, 40, 120, 160, 160, 0), (8, 60, 80, 120, 160, 160, 0),
                                    (8, 3, 5, 60, 80, 60, 0), (8, 7, 10, 60, 80, 60, 0), (8, 15, 20, 60, 80, 60, 0), (8, 30, 40, 60, 80, 60, 0),
                                    (8, 15, 20, 30, 40, 64, 1), (8, 30, 40, 60, 80, 64, 1), (8, 60, 80, 120, 160, 64, 1), (8, 15, 20, 120, 160, 1, 0)]:
    x = torch.randn(B, Hs, Ws, C, device="cuda").bfloat16(); y = torch.empty(B, Ho, Wo, C, device="cuda", dtype=torch.bfloat16)
    gy = torch.randn_like(y); gx = torch.empty_like(x)
    tf = timeit(lambda: lib.resample_forward(x, y, B, Hs, Ws, Ho, Wo, C, mode))
    tb = timeit(lambda: lib.resample_backward(gy, gx, B, Hs, Ws, Ho, Wo, C, mode))
    print("%-34s fwd %7.1f us  bwd %7.1f us   (out %.1f MB)" % (str((B, Hs, Ws, Ho, Wo, C, mode)), tf, tb, y.numel() * 2 / 1e6))
