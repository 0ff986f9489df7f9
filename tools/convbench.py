#!/usr/bin/env python
"""Micro-benchmark of the implicit-GEMM kernels on the shapes that dominate the train step."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gw_depth_amd import hip

SHAPES = [  # B, H, W, Cin, Cout, K
    (8, 120, 160, 160, 160, 3), (8, 120, 160, 800, 320, 3), (8, 120, 160, 64, 256, 1), (8, 60, 80, 128, 128, 3),
    (8, 30, 40, 256, 256, 3), (8, 240, 320, 64, 64, 3), (8, 480, 640, 32, 32, 3), (8, 60, 80, 512, 128, 1),
    (2400, 1, 1, 256, 256, 1), (153600, 1, 1, 64, 128, 1),
]


if os.environ.get("SHAPES"):
    SHAPES = [tuple(int(v) for v in t.split(",")) for t in os.environ["SHAPES"].split(";")]


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
    return s.elapsed_time(e) / n


def main():
    lib = hip.library()
    dt = torch.bfloat16
    for (B, H, W, Ci, Co, K) in SHAPES:
        p = K // 2
        x = torch.randn(B, H, W, Ci, device="cuda").to(dt)
        w = (torch.randn(Co, K, K, Ci, device="cuda") * (K * K * Ci) ** -0.5).to(dt)
        y = torch.empty(B, H, W, Co, device="cuda", dtype=dt)
        dims = (B, H, W, Ci, H, W, Co, K, K)
        dw = torch.zeros(Co, K, K, Ci, device="cuda")
        fl = 2.0 * B * H * W * Co * K * K * Ci
        tf = timeit(lambda: lib.conv_forward(x, w, y, dims, stride=1, pad=p))
        tw = timeit(lambda: lib.conv_wgrad(x, y, dw, dims, stride=1, pad=p))
        print("%-28s fwd %7.3f ms %7.1f TF/s | wgrad %7.3f ms %7.1f TF/s" % (str((B, H, W, Ci, Co, K)), tf, fl / tf / 1e9, tw, fl / tw / 1e9))


if __name__ == "__main__":
    main()
