#!/usr/bin/env python
"""Throughput of the device-side geometric transforms against Pillow on the host: one 720 x 1280 frame (RGB + depth + labels) through
flip -> resize(480) and through resize(500) -> crop -> resize(640), as the reference's training chain does."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image
from gw_depth_amd import data

rng = np.random.default_rng(0)
h, w = 720, 1280
rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
dep = rng.integers(0, 12000, (h, w)).astype(np.int32)
lab = rng.integers(0, 3, (h, w)).astype(np.uint8)
lines = torch.rand(12, 4) * torch.tensor([w, h, w, h])
chains = {
    "hflip -> resize 480": {"flip": "h", "steps": [("resize", 480, 1024)]},
    "resize 500 -> crop 450x500 -> resize 640": {"flip": None, "steps": [("resize", 500, None), ("crop", (10, 20, 450, 500)), ("resize", 640, 1024)]},
}
t_rgb, t_dep, t_lab = torch.from_numpy(rgb).cuda(), torch.from_numpy(dep).cuda(), torch.from_numpy(lab).cuda()
for name, p in chains.items():
    for _ in range(3):
        data.DeviceAugment.apply(t_rgb, t_dep, t_lab, lines, p)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        data.DeviceAugment.apply(t_rgb, t_dep, t_lab, lines, p)
    torch.cuda.synchronize()
    dev_ms = (time.perf_counter() - t0) / n * 1e3

    def pil_chain():
        a, b, c = Image.fromarray(rgb), Image.fromarray(dep, mode="I"), Image.fromarray(lab, mode="L")
        if p["flip"] == "h":
            a, b, c = (im.transpose(Image.FLIP_LEFT_RIGHT) for im in (a, b, c))
        for s in p["steps"]:
            if s[0] == "resize":
                oh, ow = data.resized_shape(a.size[0], a.size[1], s[1], s[2])
                a, b, c = a.resize((ow, oh), Image.BILINEAR), b.resize((ow, oh), Image.NEAREST), c.resize((ow, oh), Image.NEAREST)
            else:
                i, j, ch, cw = s[1]
                a, b, c = (im.crop((j, i, j + cw, i + ch)) for im in (a, b, c))
        return a

    pil_chain()
    t0 = time.perf_counter()
    for _ in range(10):
        pil_chain()
    pil_ms = (time.perf_counter() - t0) / 10 * 1e3
    print("%-46s device %.3f ms / frame (host-launched, tables built per call)   Pillow on one host core %.2f ms" % (name, dev_ms, pil_ms))
