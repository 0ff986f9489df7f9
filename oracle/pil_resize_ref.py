"""TEST INFRASTRUCTURE (oracle): numpy restatement of the two Pillow resizes the reference's input pipeline calls through torchvision
(/root/reference/src/datasets/transforms_depth.py:315-372: `F.resize(image, size)` = PIL BILINEAR with its built-in box scaling of
the filter support, and `F.resize(mat, size, interpolation=NEAREST)` for the depth / label maps), plus the flips and the crop of
:59-262.  The algorithms live in a third-party dependency, Pillow (Resample.c / Geometry.c; the container has Pillow 12.2.0); they
are restated here from their published behaviour and PINNED against Pillow itself: tests/golden/pil_resize.npz holds inputs and the
outputs Pillow produced (oracle/make_golden_pil_resize.py), and tests/test_augment.py also compares against the installed Pillow
directly wherever it is importable.  Only tests/ may import this module."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def bilinear_tables(in_size, out_size):
    """Per output index: first source index, tap count, integer coefficients (ksize per output) - Pillow's precompute_coeffs +
    normalize_coeffs_8bpc for the BILINEAR filter (support 1, scaled by max(1, in/out))."""
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 1.0 * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / fscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size) - xmin
        w = np.zeros(ksize, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - a if a < 1.0 else 0.0
            ww += w[x]
        if ww != 0.0:
            w[:xmax] /= ww
        for x in range(xmax):
            kk[xx, x] = int(-0.5 + w[x] * (1 << PRECISION_BITS)) if w[x] < 0 else int(0.5 + w[x] * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    """One separable pass over uint8 data (H, W, C) along `axis` (0 = vertical, 1 = horizontal)."""
    src = img.astype(np.int64)
    n = bounds.shape[0]
    shape = list(img.shape)
    shape[axis] = n
    out = np.empty(shape, dtype=np.uint8)
    for o in range(n):
        x0, cnt = bounds[o]
        acc = np.full(shape[:axis] + shape[axis + 1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for k in range(cnt):
            acc = acc + np.take(src, x0 + k, axis=axis) * int(kk[o, k])
        v = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
        if axis == 0:
            out[o] = v
        else:
            out[:, o] = v
    return out


def resize_bilinear_u8(img, oh, ow):
    """Pillow Image.resize((ow, oh), BILINEAR) of an (H, W, C) uint8 image: horizontal pass (only the rows the vertical pass will
    read), then vertical pass, each rounding to uint8."""
    H, W = img.shape[:2]
    need_h, need_v = ow != W, oh != H
    bh, kh = bilinear_tables(W, ow)
    bv, kv = bilinear_tables(H, oh)
    out = img
    if need_h:
        if need_v:
            first = int(bv[0, 0])
            last = int(bv[-1, 0] + bv[-1, 1])
            out = _pass(out[first:last], bh, kh, 1)
            bv = bv.copy()
            bv[:, 0] -= first
        else:
            out = _pass(out, bh, kh, 1)
    if need_v:
        out = _pass(out, bv, kv, 0)
    return out if (need_h or need_v) else img.copy()


def nearest_table(in_size, out_size):
    """Source index per output index of Pillow's NEAREST resize (scale-only affine transform: the coordinate is accumulated in
    double precision, one addition per output pixel, then truncated)."""
    a = in_size / out_size
    xo = a * 0.5
    tab = np.empty(out_size, dtype=np.int32)
    for x in range(out_size):
        tab[x] = int(xo)
        xo += a
    return np.clip(tab, 0, in_size - 1)


def resize_nearest(mat, oh, ow):
    H, W = mat.shape[:2]
    if (oh, ow) == (H, W):
        return mat.copy()
    return mat[nearest_table(H, oh)][:, nearest_table(W, ow)]
