#!/usr/bin/env python
"""Correctness of the big-M igemm variants against the fp32 kernel path (same library), forward and data gradient."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gw_depth_amd import hip
lib = hip.library()
torch.manual_seed(0)
for (B, H, W, Ci, Co, K) in [(8, 128, 128, 160, 160, 3), (8, 120, 160, 320, 160, 1), (9, 120, 130, 96, 320, 3)]:
    p = K // 2
    x = torch.randn(B, H, W, Ci, device="cuda")
    w = torch.randn(Co, K, K, Ci, device="cuda") * (K * K * Ci) ** -0.5
    sh = torch.randn(Co, device="cuda")
    dims = (B, H, W, Ci, H, W, Co, K, K)
    y32 = torch.empty(B, H, W, Co, device="cuda")
    lib.conv_forward(x, w, y32, dims, shift=sh, stride=1, pad=p, act=hip.ACT_RELU)
    y16 = torch.empty(B, H, W, Co, device="cuda", dtype=torch.bfloat16)
    lib.conv_forward(x.bfloat16(), w.bfloat16(), y16, dims, shift=sh, stride=1, pad=p, act=hip.ACT_RELU)
    e1 = float((y16.float() - y32).norm() / y32.norm())
    # data gradient: transposed gather with [Cin][taps][Cout] weights
    gy = torch.randn(B, H, W, Co, device="cuda")
    wt = w.permute(3, 1, 2, 0).contiguous()
    gx32 = torch.empty(B, H, W, Ci, device="cuda")
    lib.conv_forward(gy, wt, gx32, (B, H, W, Co, H, W, Ci, K, K), stride=1, pad=p, gather=hip.GATHER_TRANSPOSED)
    gx16 = torch.empty(B, H, W, Ci, device="cuda", dtype=torch.bfloat16)
    lib.conv_forward(gy.bfloat16(), wt.bfloat16(), gx16, (B, H, W, Co, H, W, Ci, K, K), stride=1, pad=p, gather=hip.GATHER_TRANSPOSED)
    e2 = float((gx16.float() - gx32).norm() / gx32.norm())
    print((B, H, W, Ci, Co, K), "fwd rel %.3e dgrad rel %.3e" % (e1, e2), "OK" if max(e1, e2) < 1e-2 else "MISMATCH")
