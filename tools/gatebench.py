#!/usr/bin/env python
"""Cost of gwd_conv_desc.gate per kernel family: the data-gradient launches of the train step that carry an activation gate, timed
with and without it (and the separate activation-backward pass the gate replaces)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gw_depth_amd import hip

CASES = [  # name, B, H, W, C of gy, C of gx, K, stride
    ("thin head seg   2 -> 32 @480x640", 8, 480, 640, 2, 32, 3, 1),
    ("thin head depth 1 -> 32 @480x640", 8, 480, 640, 1, 32, 3, 1),
    ("tile conv2     32 -> 32 @480x640", 8, 480, 640, 32, 32, 3, 1),
    ("resnet conv3  512 -> 128 @60x80 ", 8, 60, 80, 512, 128, 1, 1),
    ("resnet conv2  128 -> 128 @60x80 ", 8, 60, 80, 128, 128, 3, 1),
    ("resnet conv1  128 -> 512 @60x80 ", 8, 60, 80, 128, 512, 1, 1),
    ("resnet conv3 1024 -> 256 @30x40 ", 8, 30, 40, 1024, 256, 1, 1),
]


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
    return s.elapsed_time(e) / n * 1e3


lib = hip.library()
dt = torch.bfloat16
for name, B, H, W, Cg, Cx, K, stride in CASES:
    gy = torch.randn(B, H, W, Cg, device="cuda", dtype=dt)
    wt = torch.randn(Cx, K, K, Cg, device="cuda", dtype=dt) * 0.05
    x = torch.randn(B, H, W, Cx, device="cuda", dtype=dt).clamp_min(0)
    gx, dv = torch.empty_like(x), torch.empty_like(x)
    dims = (B, H, W, Cg, H, W, Cx, K, K)
    kw = dict(stride=stride, pad=K // 2, gather=hip.GATHER_TRANSPOSED)
    t0 = timeit(lambda: lib.conv_forward(gy, wt, gx, dims, **kw))
    t1 = timeit(lambda: lib.conv_forward(gy, wt, gx, dims, gate=x, gate_act=hip.ACT_RELU, **kw))
    t2 = timeit(lambda: lib.act_backward(gx, x, dv, None, B * H * W, Cx, hip.ACT_RELU, 1.0))
    print("%s  plain %6.1f us   gated %6.1f us (+%5.1f)   separate pass %6.1f us" % (name, t0, t1, t1 - t0, t2), flush=True)
