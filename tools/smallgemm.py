#!/usr/bin/env python
"""The small GEMMs of the DETR encoder / decoder (M = 800 / 2400 rows): launch time by epilogue operands, back to back and in a HIP
graph (what the step replays), against an empty-kernel floor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gw_depth_amd import hip

lib = hip.library()
dt = torch.bfloat16


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def graphed(fn, n=50):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    return timeit(g.replay, 10) / n


SH = ((2400, 256, 256), (800, 256, 256), (2400, 256, 512), (2400, 256, 2048), (2400, 2048, 256), (800, 2048, 256), (9600, 256, 256), (38400, 128, 128))
if os.environ.get('SHAPES'):
    SH = [tuple(int(v) for v in t.split(',')) for t in os.environ['SHAPES'].split(';')]
for M, K, N in SH:
    x = torch.randn(M, 1, 1, K, device="cuda", dtype=dt)
    w = (torch.randn(N, 1, 1, K, device="cuda") * K ** -0.5).to(dt)
    y = torch.empty(M, 1, 1, N, device="cuda", dtype=dt)
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, 1, 1, N, device="cuda", dtype=dt)
    dims = (M, 1, 1, K, 1, 1, N, 1, 1)
    row = []
    for name, kw in (("plain", {}), ("bias", dict(shift=b)), ("bias+res", dict(shift=b, residual=r)), ("bias+relu", dict(shift=b, act=hip.ACT_RELU)),
                     ("bias+mult+res", dict(shift=b, residual=r, mult=r)), ("dgrad", dict(gather=hip.GATHER_TRANSPOSED))):
        f = lambda: lib.conv_forward(x, w, y, dims, **kw)
        row.append("%s %.1f/%.1f" % (name, timeit(f), graphed(f)))
    print("M=%d K=%d N=%d  (eager/graph us)  " % (M, K, N) + "  ".join(row), flush=True)
