#!/usr/bin/env python
"""ConvLn forward: one fused launch (gwd_conv_desc.ln_*) against convolution + gwd_layernorm_forward, back to back, same box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gw_depth_amd import hip

SHAPES = [(8, 120, 160, 160, 160, 160, True, False), (8, 120, 160, 160, 160, 160, False, True), (8, 120, 160, 80, 160, 160, True, False),
          (8, 60, 80, 64, 64, 60, True, False), (8, 60, 80, 64, 64, 60, False, True), (8, 30, 40, 160, 160, 160, True, False)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1000


lib = hip.library()
for (B, H, W, Ci, Np, C, gelu, with_res) in SHAPES:
    dt = torch.bfloat16
    x = torch.randn(B, H, W, Ci, device="cuda").to(dt)
    w = (torch.randn(Np, 3, 3, Ci, device="cuda") * (9 * Ci) ** -0.5).to(dt)
    g, b = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
    res = torch.randn(B, H, W, Np, device="cuda").to(dt) if with_res else None
    y, z = torch.empty(B, H, W, Np, device="cuda", dtype=dt), torch.empty(B, H, W, Np, device="cuda", dtype=dt)
    rows = B * H * W
    mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    dims = (B, H, W, Ci, H, W, Np, 3, 3)
    act = hip.ACT_GELU if gelu else hip.ACT_NONE

    def fused():
        lib.conv_forward(x, w, y, dims, z=z, scale=g, shift=b, residual=res, stride=1, pad=1, act=act, ln=(mean, rstd, C))

    def fused_infer():
        lib.conv_forward(x, w, y, dims, scale=g, shift=b, residual=res, stride=1, pad=1, act=act, ln=(mean, rstd, C))

    def conv_only():
        lib.conv_forward(x, w, z, dims, stride=1, pad=1)

    def ln_only():
        lib.layernorm_forward(z, g, b, y, mean, rstd, rows, C, gelu, residual=res, ld=0 if Np == C else Np)

    tc, tl, tf, ti = timeit(conv_only), timeit(ln_only), timeit(fused), timeit(fused_infer)
    print("%-40s conv %6.1f us + LN %6.1f us = %6.1f | fused %6.1f us | fused, no conv copy %6.1f us" %
          (str((B, H, W, Ci, Np, C, "gelu" if gelu else "", "res" if with_res else "")), tc, tl, tc + tl, tf, ti), flush=True)
