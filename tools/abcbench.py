#!/usr/bin/env python
"""act_backward_colsum on the step's shapes, cache-cold (a ring of operand sets > 256 MB): us per launch and GB/s of algorithmic bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gw_depth_amd import hip
lib = hip.library()
for rows, C, act in [(2400, 2048, hip.ACT_RELU), (800, 2048, hip.ACT_RELU), (19200, 512, hip.ACT_GELU), (153600, 64, hip.ACT_RELU), (38400, 256, hip.ACT_RELU), (9600, 1024, hip.ACT_RELU)]:
    nbytes = rows * C * 2 * 3
    n = max(2, int(600e6 // nbytes))
    sets = [(torch.randn(rows, C, device="cuda").bfloat16(), torch.randn(rows, C, device="cuda").bfloat16(), torch.empty(rows, C, device="cuda", dtype=torch.bfloat16)) for _ in range(n)]
    db = torch.zeros(C, device="cuda")
    for s in sets:
        lib.act_backward_colsum(s[0], s[1], s[2], db, rows, C, act, 1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        for s in sets:
            lib.act_backward_colsum(s[0], s[1], s[2], db, rows, C, act, 1.0)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (3 * n)
    print("%7d x %4d act %d: %6.1f us  %6.0f GB/s" % (rows, C, act, us, nbytes / us / 1e3), flush=True)
