#!/usr/bin/env python
"""Runs only the dominant kernel of the step (conv3x3 160->160 at B=8, 120x160, bf16) so that rocprofv3 --pmc
passes (FETCH_SIZE, WRITE_SIZE) can be attributed to it.  Same launch as bench.py's roofline leg."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gw_depth_amd import hip

lib = hip.library()
B, H, W, C = 8, 120, 160, 160
x = torch.randn(B, H, W, C, device="cuda").to(torch.bfloat16)
w = (torch.randn(C, 3, 3, C, device="cuda") * (9 * C) ** -0.5).to(torch.bfloat16)
y = torch.empty(B, H, W, C, device="cuda", dtype=torch.bfloat16)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    lib.conv_forward(x, w, y, (B, H, W, C, H, W, C, 3, 3), stride=1, pad=1, gather=hip.GATHER_TRANSPOSED)    # the step's data-gradient launch
torch.cuda.synchronize()
print("algorithmic bytes per launch: in %d + w %d + out %d" % (x.numel() * 2, w.numel() * 2, y.numel() * 2))
