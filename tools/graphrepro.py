#!/usr/bin/env python
"""Minimal HIP-graph repro candidates: replay twice, compare with eager."""
import torch
torch.manual_seed(0)
dev = "cuda"
def run(name, fn, *args):
    ref = fn(*args)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): fn(*args)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        out = fn(*args)
    res = []
    for r in range(3):
        g.replay(); torch.cuda.synchronize()
        o = out if torch.is_tensor(out) else out[0]; rf = ref if torch.is_tensor(ref) else ref[0]
        res.append((bool(torch.isfinite(o.float()).all()), float((o.float() - rf.float()).abs().max())))
    print("%-28s" % name, res, flush=True)

ra = torch.randn(8, 441, 40, 16, device=dev, dtype=torch.bfloat16)
run("mean(1,2) f32", lambda t: t.float().mean(dim=(1, 2), keepdim=True), ra)
run("var(1,2) f32", lambda t: t.float().var(dim=(1, 2), keepdim=True, unbiased=False), ra)
qs = torch.randn(8, 9, 16, 49, 32, device=dev, dtype=torch.bfloat16); rk = torch.randn(8, 16, 40, 32, device=dev, dtype=torch.bfloat16)
run("einsum bwhnd,bhrd->bwnrh", lambda a, b: torch.einsum("bwhnd,bhrd->bwnrh", a, b).contiguous(), qs, rk)
att = torch.randn(8, 9, 16, 49, 40, device=dev, dtype=torch.bfloat16)
run("einsum bwhnr,bhrd->bwnhd", lambda a, b: torch.einsum("bwhnr,bhrd->bwnhd", a, b).contiguous(), att, rk)
def chain(t):
    uf = t.float(); mu = uf.mean(dim=(1, 2), keepdim=True); var = uf.var(dim=(1, 2), keepdim=True, unbiased=False)
    return t + torch.nn.functional.gelu((uf - mu) * torch.rsqrt(var + 1e-5)).to(t.dtype)
run("norm-gelu chain", chain, ra)
big = torch.randn(8, 160, 120, 160, device=dev)
run("sum all", lambda t: t.sum(), big)
run("sum dims", lambda t: t.sum(dim=(0, 2, 3)), big)
def chain3(t):
    for _ in range(3):
        t = chain(t)
    return t
run("norm-gelu chain x3", chain3, ra)
def many(t):
    outs = []
    for i in range(40):
        u = (t.float() * (1.0 + i)).contiguous()
        outs.append(u.mean(dim=(1, 2)))
        outs.append(u.var(dim=(1, 2), unbiased=False))
        del u
    return torch.stack(outs)
run("40x mean/var with reuse", many, ra)
import ctypes
hipl = ctypes.CDLL("libamdhip64.so")
def memset_chain(t):
    # raw hipMemsetAsync nodes interleaved with kernels on recycled small blocks
    outs = []
    st = torch.cuda.current_stream().cuda_stream
    for i in range(64):
        s = torch.empty(64, dtype=torch.int32, device=dev)
        hipl.hipMemsetAsync(ctypes.c_void_p(s.data_ptr()), 0, ctypes.c_size_t(256), ctypes.c_void_p(st))
        s.add_(i + 1)
        outs.append(s[:1].clone())
        del s
    return torch.cat(outs).float()
run("memset+add on recycled block", memset_chain, ra)
