#!/usr/bin/env python
"""Does the dominant conv launch slow down when it runs back to back between bandwidth-bound launches (as inside the captured step)?
HIP events around every conv launch: (a) conv only, (b) conv alternating with a streaming kernel, (c) the same inside a HIP graph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gw_depth_amd import hip
lib = hip.library()
B, H, W, Ci, Co, K = 8, 120, 160, 160, 160, 3
dims = (B, H, W, Ci, H, W, Co, K, K)
w = (torch.randn(Co, K, K, Ci, device="cuda") * 0.02).bfloat16()
x = torch.randn(B, H, W, Ci, device="cuda").bfloat16()
y = torch.empty(B, H, W, Co, device="cuda", dtype=torch.bfloat16)
big_a = torch.randn(int(os.environ.get("STREAM_MB", "200")) * 1024 * 1024 // 2, device="cuda").bfloat16()
big_b = torch.empty_like(big_a)
ga, be = torch.ones(Co, device="cuda"), torch.zeros(Co, device="cuda")
mean, rstd = torch.empty(B * H * W, device="cuda"), torch.empty(B * H * W, device="cuda")
y2 = torch.empty_like(y)


def conv():
    lib.conv_forward(x, w, y, dims, stride=1, pad=1)


def stream():
    torch.add(big_a, 1.0, out=big_b)


def ln():
    lib.layernorm_forward(y, ga, be, y2, mean, rstd, B * H * W, Co, True)


def timed(seq, n=30):
    evs = []
    for _ in range(3):
        for f in seq:
            f()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        for f in seq:
            if f is conv:
                a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); f(); e.record()
                evs.append((a, e))
            else:
                f()
    t1.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(e) * 1e3 for a, e in evs)
    return ts[len(ts) // 2], t0.elapsed_time(t1) / n * 1e3


for name, seq in (("conv only", [conv]), ("conv + LN", [conv, ln]), ("conv + stream", [conv, stream]), ("conv + stream + LN + stream", [conv, stream, ln, stream])):
    med, tot = timed(seq)
    print("%-30s conv median %6.1f us   sequence %7.1f us" % (name, med, tot))

# the same sequences captured in a HIP graph: total time per replay minus the other launches' own graph time
def graph_time(seq, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for f in seq:
            f()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(10):
                for f in seq:
                    f()
        g.replay()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            g.replay()
        e.record()
        torch.cuda.synchronize()
    return a.elapsed_time(e) / reps / 10 * 1e3


tc, ts_, tl = graph_time([conv]), graph_time([stream]), graph_time([ln])
print("graph: conv %.1f us, stream %.1f us, LN %.1f us" % (tc, ts_, tl))
print("graph: conv + stream = %.1f us (sum of parts %.1f)" % (graph_time([conv, stream]), tc + ts_))
print("graph: conv + LN = %.1f us (sum of parts %.1f)" % (graph_time([conv, ln]), tc + tl))
print("graph: conv + stream + LN + stream = %.1f us (sum of parts %.1f)" % (graph_time([conv, stream, ln, stream]), tc + 2 * ts_ + tl))

# producer -> consumer through memory: the conv reads what the previous launch has just written (as in the step)
def ln_into_x():
    lib.layernorm_forward(y, ga, be, x, mean, rstd, B * H * W, Co, True)      # y -> x, then conv x -> y


def copy_into_x():
    x.copy_(y2)


tl2, tcp = graph_time([ln_into_x]), graph_time([copy_into_x])
print("graph: LN(y->x) %.1f us, copy(y2->x) %.1f us" % (tl2, tcp))
print("graph: LN(y->x) + conv(x->y) = %.1f us (sum of parts %.1f)" % (graph_time([ln_into_x, conv]), tc + tl2))
print("graph: copy(y2->x) + conv(x->y) = %.1f us (sum of parts %.1f)" % (graph_time([copy_into_x, conv]), tc + tcp))
med, tot = timed([ln_into_x, conv])
print("eager events: LN(y->x) + conv: conv median %.1f us, sequence %.1f" % (med, tot))
