#!/usr/bin/env python
"""Small row-wise launches of the step (LayerNorm forward / backward of the DETR and Swin token shapes), run under
`rocprofv3 --kernel-trace --stats` to read kernel times (tools/rowbench.sh); 30 launches per shape, operands reused (L2-warm, as in the step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gw_depth_amd import hip
lib = hip.library()
for rows, C in [(2400, 256), (800, 256), (11760, 384), (42336, 128), (162288, 64), (19200, 512)]:
    x = torch.randn(rows, C, device="cuda").bfloat16()
    gy = torch.randn(rows, C, device="cuda").bfloat16()
    ga, be = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    y, gx = torch.empty_like(x), torch.empty_like(x)
    mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    for _ in range(30):
        lib.layernorm_forward(x, ga, be, y, mean, rstd, rows, C, False)
        lib.layernorm_backward(gy, x, ga, be, mean, rstd, gx, dg, db, rows, C, False)
    torch.cuda.synchronize()
    print("done", rows, C, flush=True)
