#!/usr/bin/env python
"""Window attention forward + backward at the step's stage shapes: us per launch (HIP events around each direction, back to back)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gw_depth_amd import ops
torch.manual_seed(0)
# (windows, heads, head_dim, windows per image): dense transformer 1/32, class transformers 1/16, 1/8, 1/4
for W_, H_, D_, wpi in [(72, 16, 32, 9), (240, 16, 16, 30), (864, 16, 8, 108), (3312, 16, 4, 414)]:
    qkv = (torch.randn(W_, 49, 3, H_, D_, device="cuda") * 0.5).bfloat16().requires_grad_(True)
    table = torch.randn(169, H_, device="cuda").requires_grad_(True)
    coords = torch.stack(torch.meshgrid(torch.arange(7), torch.arange(7), indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0)
    rel = ((rel[..., 0] + 6) * 13 + rel[..., 1] + 6).flatten().int().cuda()
    region = None
    for name, reg in (("plain", None),):
        out = ops.window_attention_packed(qkv, table, rel, reg, wpi, D_ ** -0.5)
        g = torch.randn_like(out)
        def fwd():
            return ops.window_attention_packed(qkv, table, rel, reg, wpi, D_ ** -0.5)
        for _ in range(3):
            fwd().backward(g)
        torch.cuda.synchronize()
        tf = tb = 0.0
        n = 20
        for _ in range(n):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record(); o = fwd(); e1.record(); o.backward(g); e2.record()
            torch.cuda.synchronize()
            tf += e0.elapsed_time(e1); tb += e1.elapsed_time(e2)
        print("windows %5d heads %d head_dim %2d: forward %6.1f us, backward %6.1f us (events around the autograd node, launch overhead included)" % (W_, H_, D_, tf / n * 1e3, tb / n * 1e3), flush=True)
