#!/usr/bin/env python
"""The dominant conv / weight-gradient launches with WARM operands (one buffer set, as tools/convbench.py) and with COLD ones
(a ring of buffer sets larger than the 256 MB on-die cache, as inside the train step): what does first-touch HBM latency cost?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gw_depth_amd import hip

lib = hip.library()
SHAPES = [(8, 120, 160, 160, 160, 3), (8, 120, 160, 160, 320, 3)]
if os.environ.get("SHAPES"):
    SHAPES = [tuple(int(v) for v in t.split(",")) for t in os.environ["SHAPES"].split(";")]
NSETS = int(os.environ.get("NSETS", "8"))


def run(fns, n=40):
    for f in fns:
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for i in range(n):
        fns[i % len(fns)]()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (B, H, W, Ci, Co, K) in SHAPES:
    dims = (B, H, W, Ci, H, W, Co, K, K)
    w = (torch.randn(Co, K, K, Ci, device="cuda") * 0.02).bfloat16()
    sets = [(torch.randn(B, H, W, Ci, device="cuda").bfloat16(), torch.empty(B, H, W, Co, device="cuda", dtype=torch.bfloat16),
             torch.zeros(Co, K, K, Ci, device="cuda")) for _ in range(NSETS)]
    fwd = [lambda s=s: lib.conv_forward(s[0], w, s[1], dims, stride=1, pad=K // 2) for s in sets]
    wg = [lambda s=s: lib.conv_wgrad(s[0], s[1], s[2], dims, stride=1, pad=K // 2) for s in sets]
    fl = 2.0 * B * H * W * Co * K * K * Ci
    for name, fns in (("fwd", fwd), ("wgrad", wg)):
        warm, cold = run(fns[:1]), run(fns)
        print("%-26s %-5s warm %7.1f us %6.1f TF/s | cold (%d sets) %7.1f us %6.1f TF/s" % ((B, H, W, Ci, Co, K), name, warm, fl / warm / 1e6, NSETS, cold, fl / cold / 1e6))

# sustained load: does the launch time drift when the same launch runs for ~0.5 s (clock / power management)?
B, H, W, Ci, Co, K = SHAPES[0]
dims = (B, H, W, Ci, H, W, Co, K, K)
w = (torch.randn(Co, K, K, Ci, device="cuda") * 0.02).bfloat16()
sets = [(torch.randn(B, H, W, Ci, device="cuda").bfloat16(), torch.empty(B, H, W, Co, device="cuda", dtype=torch.bfloat16)) for _ in range(NSETS)]
fwd = [lambda s=s: lib.conv_forward(s[0], w, s[1], dims, stride=1, pad=K // 2) for s in sets]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
torch.cuda.synchronize()
ev[0].record()
for blk in range(40):
    for i in range(100):
        fwd[i % NSETS]()
    ev[blk + 1].record()
torch.cuda.synchronize()
print("sustained fwd, us per launch in blocks of 100:", " ".join("%.0f" % (ev[i].elapsed_time(ev[i + 1]) * 10) for i in range(40)))
