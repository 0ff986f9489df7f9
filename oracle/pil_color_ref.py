"""TEST INFRASTRUCTURE (oracle): numpy restatement of the colour jitter primitives the reference's ColorJitter
(/root/reference/src/datasets/transforms_depth.py:551-600) reaches through torchvision.transforms.functional on PIL images:
adjust_brightness / adjust_contrast / adjust_saturation = Pillow's ImageEnhance (Image.blend with a black / mean-grey / greyscale
"degenerate" image), adjust_hue = RGB -> HSV (Pillow Convert.c), uint8 wrap-around shift of H, HSV -> RGB.  The arithmetic lives in
third-party dependencies (Pillow 12.2.0 here; torchvision - not installed, its PIL glue is restated from its published source:
enhance(factor) per property, and `np_h += np.uint8(hue_factor * 255)`); pinned against Pillow itself by tests/golden/pil_color.npz
(oracle/make_golden_pil_color.py) and by direct comparison wherever Pillow is importable.  Only tests/ may import this module."""
import numpy as np


def to_gray(rgb):
    """Image.convert('L') of RGB: ITU-R 601-2 luma in 16.16 fixed point, rounded."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend(deg, img, alpha):
    """Image.blend(degenerate, image, alpha) on uint8 data: single-precision arithmetic, truncation; outside [0, 1] the result is
    clipped to 0..255 first."""
    a = np.float32(alpha)
    d, i = deg.astype(np.int32), img.astype(np.int32)
    t = (d.astype(np.float32) + a * (i - d).astype(np.float32)).astype(np.float32)
    if np.float32(0.0) <= a <= np.float32(1.0):            # the C function takes a float: the test is on the rounded value
        return t.astype(np.int32).astype(np.uint8)
    out = np.where(t <= 0.0, 0, np.where(t >= 255.0, 255, t.astype(np.int32)))
    return out.astype(np.uint8)


def adjust_brightness(rgb, f):
    return blend(np.zeros_like(rgb), rgb, f)


def adjust_contrast(rgb, f):
    g = to_gray(rgb)
    mean = int(float(g.astype(np.float64).sum()) / g.size + 0.5)          # ImageStat.Stat(L).mean[0] + 0.5, truncated
    return blend(np.full_like(rgb, mean), rgb, f)


def adjust_saturation(rgb, f):
    g = to_gray(rgb)
    return blend(np.repeat(g[..., None], 3, axis=2), rgb, f)


def rgb_to_hsv(rgb):
    r, g, b = (rgb[..., i].astype(np.int32) for i in range(3))
    maxc, minc = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
    grey = maxc == minc
    cr = (maxc - minc).astype(np.float32)
    cr_safe = np.where(grey, np.float32(1), cr)
    s = cr / np.where(grey, np.float32(1), maxc.astype(np.float32))
    rc, gc, bc = ((maxc - c).astype(np.float32) / cr_safe for c in (r, g, b))
    # float h; the literals 2.0 / 4.0 / 6.0 / 1.0 are doubles in the C source: every expression below is evaluated in double and
    # stored to float where the C code assigns to `h`
    h = np.where(r == maxc, (bc - gc).astype(np.float32),
                 np.where(g == maxc, (2.0 + rc.astype(np.float64) - bc.astype(np.float64)).astype(np.float32),
                          (4.0 + gc.astype(np.float64) - rc.astype(np.float64)).astype(np.float32)))
    h = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(np.float32)
    uh = np.clip((h.astype(np.float64) * 255.0).astype(np.int64), 0, 255)
    us = np.clip((s.astype(np.float64) * 255.0).astype(np.int64), 0, 255)
    out = np.stack([np.where(grey, 0, uh), np.where(grey, 0, us), maxc], axis=-1)
    return out.astype(np.uint8)


def hsv_to_rgb(hsv):
    h, s, v = (hsv[..., i].astype(np.int32) for i in range(3))
    hf = h.astype(np.float32).astype(np.float64) * 6.0 / 255.0
    i = np.floor(hf).astype(np.int32)
    f = (hf - i.astype(np.float32).astype(np.float64)).astype(np.float32)
    fs = (s.astype(np.float32).astype(np.float64) / 255.0).astype(np.float32)
    vf = v.astype(np.float32).astype(np.float64)
    rnd = lambda x: np.where(x >= 0, np.floor(x + 0.5), -np.floor(-x + 0.5)).astype(np.int64)     # C round(): half away from zero
    p = np.clip(rnd(vf * (1.0 - fs.astype(np.float64))), 0, 255)
    q = np.clip(rnd(vf * (1.0 - fs.astype(np.float64) * f.astype(np.float64))), 0, 255)
    t = np.clip(rnd(vf * (1.0 - fs.astype(np.float64) * (1.0 - f.astype(np.float64)))), 0, 255)
    k = i % 6
    r = np.select([k == 0, k == 1, k == 2, k == 3, k == 4, k == 5], [v, q, p, p, t, v])
    g = np.select([k == 0, k == 1, k == 2, k == 3, k == 4, k == 5], [t, v, v, q, p, p])
    b = np.select([k == 0, k == 1, k == 2, k == 3, k == 4, k == 5], [p, p, t, v, v, q])
    grey = s == 0
    out = np.stack([np.where(grey, v, r), np.where(grey, v, g), np.where(grey, v, b)], axis=-1)
    return out.astype(np.uint8)


def adjust_hue(rgb, hue_factor):
    hsv = rgb_to_hsv(rgb)
    shift = int(hue_factor * 255) & 255          # np.uint8(hue_factor * 255) of the numpy versions torchvision's PIL path was written for
    hsv[..., 0] = ((hsv[..., 0].astype(np.int32) + shift) & 255).astype(np.uint8)                        # uint8 wrap-around
    return hsv_to_rgb(hsv)
