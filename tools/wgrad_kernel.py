#!/usr/bin/env python
"""Runs one weight-gradient launch shape repeatedly (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gw_depth_amd import hip
lib = hip.library()
B, H, W, Ci, Co, K = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "8,120,160,160,160,3").split(",")]
x = torch.randn(B, H, W, Ci, device="cuda").to(torch.bfloat16)
gy = torch.randn(B, H, W, Co, device="cuda").to(torch.bfloat16)
dw = torch.zeros(Co, K, K, Ci, device="cuda")
for _ in range(20):
    lib.conv_wgrad(x, gy, dw, (B, H, W, Ci, H, W, Co, K, K), stride=1, pad=K // 2)
torch.cuda.synchronize()
print("done")
