#!/usr/bin/env python
"""Weight gradient of the two full-resolution heads (32 -> 1 / 2 channels at 8x480x640): run under rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gw_depth_amd import hip
lib = hip.library()
B, H, W, C = 8, 480, 640, 32
x = torch.randn(B, H, W, C, device="cuda").bfloat16()
for N in (1, 2):
    gy = torch.randn(B, H, W, N, device="cuda").bfloat16()
    dw = torch.zeros(N, 3, 3, C, device="cuda")
    for _ in range(10):
        lib.conv_wgrad(x, gy, dw, (B, H, W, C, H, W, N, 3, 3), stride=1, pad=1)
torch.cuda.synchronize()
