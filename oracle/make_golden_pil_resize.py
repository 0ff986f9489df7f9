#!/usr/bin/env python
"""Golden vectors for oracle/pil_resize_ref.py and the device kernels: small images resized by the INSTALLED Pillow (the library the
reference's transforms call through torchvision: src/datasets/transforms_depth.py:315-372), RGB BILINEAR and 32-bit / 8-bit NEAREST,
up- and down-scaling, plus Pillow's own flips.  Writes tests/golden/pil_resize.npz (inputs + outputs).  Run in the build container."""
import os

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rng = np.random.default_rng(20260704)
cases = [(48, 64, 30, 41), (37, 53, 74, 91), (60, 80, 60, 33), (45, 45, 17, 45), (96, 128, 75, 100), (20, 31, 64, 99), (90, 120, 32, 43)]
out = {"pillow_version": np.array(Image.__version__ if hasattr(Image, "__version__") else __import__("PIL").__version__)}
for n, (h, w, oh, ow) in enumerate(cases):
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    dep = rng.integers(0, 12000, (h, w)).astype(np.int32)
    lab = rng.integers(0, 3, (h, w)).astype(np.uint8)
    out[f"rgb{n}"], out[f"dep{n}"], out[f"lab{n}"], out[f"size{n}"] = rgb, dep, lab, np.array([oh, ow])
    out[f"rgb_out{n}"] = np.asarray(Image.fromarray(rgb).resize((ow, oh), Image.BILINEAR))
    out[f"dep_out{n}"] = np.asarray(Image.fromarray(dep, mode="I").resize((ow, oh), Image.NEAREST))
    out[f"lab_out{n}"] = np.asarray(Image.fromarray(lab, mode="L").resize((ow, oh), Image.NEAREST))
    flipped = Image.fromarray(rgb).transpose(Image.FLIP_LEFT_RIGHT if n % 2 == 0 else Image.FLIP_TOP_BOTTOM)
    out[f"rgb_flip_out{n}"] = np.asarray(flipped.resize((ow, oh), Image.BILINEAR))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "pil_resize.npz"), **out)
print("wrote", len(out), "arrays")
