#!/usr/bin/env python
"""Data gradient of the three stride-2 3x3 ResNet layers (layer2-4, block 0): us per launch, back to back."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gw_depth_amd import hip
lib = hip.library()
for (B, H, W, C) in [(8, 120, 160, 128), (8, 60, 80, 256), (8, 30, 40, 512)]:
    Ho, Wo = H // 2, W // 2
    gy = torch.randn(B, Ho, Wo, C, device="cuda").bfloat16()
    wt = (torch.randn(C, 3, 3, C, device="cuda") * (9 * C) ** -0.5).bfloat16()
    gx = torch.empty(B, H, W, C, device="cuda", dtype=torch.bfloat16)
    gate = torch.randn(B, H, W, C, device="cuda").bfloat16()
    f = lambda: lib.conv_forward(gy, wt, gx, (B, Ho, Wo, C, H, W, C, 3, 3), stride=2, pad=1, gather=hip.GATHER_TRANSPOSED, gate=gate, gate_act=hip.ACT_RELU)
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(30):
        f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    print("%s: %6.1f us, %6.1f TF/s of real arithmetic (2.25 taps per pixel)" % ((B, H, W, C), us, 2.0 * B * H * W * C * C * 2.25 / us / 1e6), flush=True)
