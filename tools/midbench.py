#!/usr/bin/env python
"""Mid-size GEMM / conv shapes of the step (ResNet layer2-4, DETR / Swin linears): forward, data gradient, weight gradient, us per launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gw_depth_amd import hip

SHAPES = [  # B, H, W, Cin, Cout, K
    (8, 30, 40, 1024, 256, 1), (8, 30, 40, 256, 256, 3), (8, 30, 40, 256, 1024, 1), (8, 15, 20, 2048, 512, 1), (8, 15, 20, 512, 512, 3),
    (8, 15, 20, 512, 2048, 1), (8, 60, 80, 512, 128, 1), (8, 60, 80, 128, 128, 3), (8, 60, 80, 128, 512, 1), (2400, 1, 1, 256, 256, 1),
    (800, 1, 1, 256, 256, 1), (2400, 1, 1, 256, 2048, 1), (3528, 1, 1, 512, 1536, 1), (3528, 1, 1, 512, 512, 1), (11760, 1, 1, 256, 256, 1),
    (42336, 1, 1, 128, 128, 1), (11760, 1, 1, 384, 384, 1), (8, 120, 160, 64, 64, 3), (8, 120, 160, 64, 256, 1), (8, 120, 160, 256, 64, 1),
]


def timeit(fn, n=30):
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
dt = torch.bfloat16
tot = [0.0, 0.0, 0.0, 0.0]
for (B, H, W, Ci, Co, K) in SHAPES:
    p = K // 2
    x = torch.randn(B, H, W, Ci, device="cuda").to(dt)
    w = (torch.randn(Co, K, K, Ci, device="cuda") * (K * K * Ci) ** -0.5).to(dt)
    wt = w.permute(3, 1, 2, 0).contiguous()
    y = torch.empty(B, H, W, Co, device="cuda", dtype=dt)
    gx = torch.empty(B, H, W, Ci, device="cuda", dtype=dt)
    dims = (B, H, W, Ci, H, W, Co, K, K)
    fl = 2.0 * B * H * W * Co * K * K * Ci
    tf = timeit(lambda: lib.conv_forward(x, w, y, dims, stride=1, pad=p, act=hip.ACT_RELU))
    td = timeit(lambda: lib.conv_forward(y, wt, gx, (B, H, W, Co, H, W, Ci, K, K), stride=1, pad=p, gather=hip.GATHER_TRANSPOSED))
    dw = torch.zeros(Co, K, K, Ci, device="cuda")
    sc = torch.rand(Co, device="cuda") + 0.5
    tw = timeit(lambda: lib.conv_wgrad(x, y, dw, dims, stride=1, pad=p))
    tws = timeit(lambda: lib.conv_wgrad(x, y, dw, dims, stride=1, pad=p, scale=sc))
    tot[0] += tf
    tot[1] += td
    tot[2] += tw
    tot[3] += tws
    print("%-30s fwd %6.1f us %6.1f TF/s | dgrad %6.1f us %6.1f TF/s | wgrad %6.1f us, with a FrozenBN scale %6.1f us" %
          (str((B, H, W, Ci, Co, K)), tf, fl / tf / 1e6, td, fl / td / 1e6, tw, tws), flush=True)
print("sum fwd %.1f us, dgrad %.1f us, wgrad %.1f us, wgrad (scale) %.1f us" % tuple(tot))
