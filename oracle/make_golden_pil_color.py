#!/usr/bin/env python
"""Golden vectors for oracle/pil_color_ref.py and csrc/color.hip: one small RGB image (random pixels + grey runs + black / white) through
the Pillow calls torchvision's PIL colour adjustments make - ImageEnhance.Brightness / Contrast / Color at several factors (inside and
outside [0, 1]) and the HSV round trip with a uint8 hue shift.  Writes tests/golden/pil_color.npz.  Run in the build container."""
import os

import numpy as np
from PIL import Image, ImageEnhance

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rng = np.random.default_rng(20260705)
img = rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)
img[0] = img[0, :, :1]
img[1, :16], img[1, 16:32] = 0, 255
im = Image.fromarray(img)
out = {"rgb": img, "factors": np.array([0.0, 0.6, 0.83, 1.0, 1.27, 1.4]), "hue_factors": np.array([-0.4, -0.13, 0.0, 0.2, 0.4])}
for f in out["factors"]:
    out["brightness_%.2f" % f] = np.asarray(ImageEnhance.Brightness(im).enhance(float(f)))
    out["contrast_%.2f" % f] = np.asarray(ImageEnhance.Contrast(im).enhance(float(f)))
    out["saturation_%.2f" % f] = np.asarray(ImageEnhance.Color(im).enhance(float(f)))
out["hsv"] = np.asarray(im.convert("HSV"))
for f in out["hue_factors"]:
    h, s, v = im.convert("HSV").split()
    nh = ((np.array(h, dtype=np.int32) + (int(f * 255) & 255)) & 255).astype(np.uint8)            # np_h += np.uint8(hue_factor * 255), wrapping
    out["hue_%.2f" % f] = np.asarray(Image.merge("HSV", (Image.fromarray(nh, "L"), s, v)).convert("RGB"))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "pil_color.npz"), **out)
print("wrote", len(out), "arrays")
