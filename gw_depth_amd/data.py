"""Device-side batch assembly: the tail of the reference's input pipeline (first slice).

The reference normalises every image on the host (ToTensor + Normalize, src/datasets/transforms_depth.py:618-660,
src/datasets/coco.py:76-79), converts depth / segmentation there (src/datasets/glassrgbd_norhint.py:277-281), pads and
masks in collate_fn_aux (src/util/misc.py:273-313) and ships 24 bytes per pixel of fp32 / int64 to the GPU.  Here the host
ships what the decoder produced - uint8 RGB, 16-bit depth in millimetres, uint8 labels: 5 bytes per pixel (8 with depth
widened to int32) - and ONE kernel (gwd_collate) writes the normalised pixel-major image batch, the padding mask, metric depth
and {0,1} labels.  The geometric and photometric augmentation in front of it follows below (DeviceAugment).
"""
import torch

from . import hip

MEAN, STD = (0.538, 0.494, 0.453), (0.257, 0.263, 0.273)          # src/datasets/coco.py:78


def device_collate(samples, device="cuda", dtype=torch.float32, mean=MEAN, std=STD):
    """samples: list (<= 16) of (rgb uint8 (h,w,3), depth_mm integer (h,w), labels uint8 (h,w)) host or device tensors, as
    decoded (any element but rgb may be None for the whole batch).  Returns the batch dict TrainStep / evaluate take:
    images (B,3,H,W) [a view of the pixel-major buffer the model reads in place], pad_mask (B,H,W) bool, depth (B,1,H,W) fp32
    metres, seg (B,1,H,W) int64."""
    lib = hip.library()
    if not 0 < len(samples) <= hip.COLLATE_BATCH:
        raise ValueError("1..%d images per call" % hip.COLLATE_BATCH)
    dev = torch.device(device)

    def up(t, dt):
        if t is None:
            return None
        t = torch.as_tensor(t)
        if t.device != dev and t.device.type == "cpu" and dev.type == "cuda":
            t = t.pin_memory()
        return t.to(dev, dtype=dt, non_blocking=True).contiguous()

    dev_samples = [(up(r, torch.uint8), up(d, torch.int32), up(l, torch.uint8)) for r, d, l in samples]
    for r, d, l in dev_samples:
        if r.dim() != 3 or r.shape[2] != 3 or (d is not None and d.shape != r.shape[:2]) or (l is not None and l.shape != r.shape[:2]):
            raise ValueError("rgb must be (h,w,3) with depth / labels of the same (h,w)")
    B = len(dev_samples)
    H, W = max(s[0].shape[0] for s in dev_samples), max(s[0].shape[1] for s in dev_samples)
    have_d, have_l = dev_samples[0][1] is not None, dev_samples[0][2] is not None
    images = torch.empty((B, H, W, 3), dtype=dtype, device=dev)
    mask = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    depth = torch.empty((B, H, W), dtype=torch.float32, device=dev) if have_d else None
    seg = torch.empty((B, H, W), dtype=torch.int64, device=dev) if have_l else None
    lib.collate(dev_samples, H, W, mean, std, images, mask, depth, seg)
    out = {"images": images.permute(0, 3, 1, 2), "pad_mask": mask.to(torch.bool)}
    if have_d:
        out["depth"] = depth.view(B, 1, H, W)
    if have_l:
        out["seg"] = seg.view(B, 1, H, W)
    return out


# ----------------------------------------------------------------------------------------------------------------------------------
# Second slice of the input pipeline: the GEOMETRIC transforms (flip, crop, resize) on the device, bit-exact with the Pillow calls
# the reference makes (src/datasets/transforms_depth.py:59-372 through torchvision.transforms.functional on PIL images), and the
# matching arithmetic on the line targets.  Decoding (PNG / JSON) and the colour jitter stay on the host.
import math
import random

import numpy as np

_PRECISION_BITS = 32 - 8 - 2


def bilinear_tables(in_size, out_size):
    """(bounds (out,2) int32, coefficients (out,ksize) int32) of Pillow's BILINEAR resize along one axis: triangle filter whose support
    is scaled by max(1, in/out), normalised in double precision, rounded to 22 fractional bits (Resample.c: precompute_coeffs,
    normalize_coeffs_8bpc).  Sums run tap by tap, as Pillow's loops do - the order matters for bit-exact coefficients."""
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 1.0 * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)              # (int) truncation of non-negative values / clamp at 0
    xmin = np.where(center - support + 0.5 < 0, 0, xmin)
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    ss = 1.0 / fscale
    w = np.zeros((out_size, ksize), dtype=np.float64)
    ww = np.zeros(out_size, dtype=np.float64)
    for t in range(ksize):
        a = np.abs((t + xmin - center + 0.5) * ss)
        wt = np.where((a < 1.0) & (t < xmax), 1.0 - a, 0.0)
        w[:, t] = wt
        ww = ww + wt
    nz = ww != 0.0
    w[nz] = w[nz] / ww[nz, None]
    kk = (0.5 + w * (1 << _PRECISION_BITS)).astype(np.int64)                     # weights are >= 0: (int)(0.5 + w * 2^22)
    kk[np.arange(ksize)[None, :] >= xmax[:, None]] = 0
    return np.stack([xmin, xmax], axis=1).astype(np.int32), kk.astype(np.int32)


def nearest_table(in_size, out_size):
    """Source index per output index of Pillow's NEAREST resize: the coordinate a/2 + a + a + ... accumulated in double precision, one
    addition per pixel, truncated (Geometry.c, scale-only affine transform)."""
    a = in_size / out_size
    steps = np.full(out_size, a, dtype=np.float64)
    steps[0] = a * 0.5
    return np.clip(np.add.accumulate(steps).astype(np.int64), 0, in_size - 1).astype(np.int32)


def _flip_map(n, flipped):
    return (n - 1, -1) if flipped else (0, 1)


def device_resize_rgb(img, size, hflip=False, vflip=False):
    """Image.resize(size[::-1], BILINEAR) of a uint8 (h,w,3) device image - which may be a crop window view img[i:i+h, j:j+w] of a
    larger one - optionally flipped first (F.hflip / F.vflip): flip, crop and resize are one read of the source."""
    lib = hip.library()
    h, w, C = img.shape
    oh, ow = int(size[0]), int(size[1])
    if img.dtype != torch.uint8 or img.stride(2) != 1 or img.stride(1) != C:
        raise ValueError("device_resize_rgb: a uint8 (h,w,C) image or a row-window view of one expected")
    dev = img.device
    rs = img.stride(0)
    need_h, need_v = ow != w, oh != h
    xb, xs = _flip_map(w, hflip)
    yb, ys = _flip_map(h, vflip)
    if not need_h and not need_v:
        out = torch.empty((h, w, C), dtype=torch.uint8, device=dev)
        yt = torch.from_numpy(np.arange(h, dtype=np.int32)[::-1].copy() if vflip else np.arange(h, dtype=np.int32)).to(dev)
        xt = torch.from_numpy(np.arange(w, dtype=np.int32)[::-1].copy() if hflip else np.arange(w, dtype=np.int32)).to(dev)
        lib.gather2d(img, out, yt, xt, rs, C)
        return out
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    cur, cur_rs, cur_h = img, rs, h
    cyb, cys = yb, ys
    bv = kv = None
    if need_v:
        bv, kv = bilinear_tables(h, oh)
    if need_h:
        bh, kh = bilinear_tables(w, ow)
        first, last = (int(bv[0, 0]), int(bv[-1, 0] + bv[-1, 1])) if need_v else (0, h)      # only the rows the vertical pass reads
        tmp = torch.empty((last - first, ow, C), dtype=torch.uint8, device=dev)
        # row j of tmp = (flipped) source row first + j
        lib.resample_u8_pass(img, tmp, to_dev(bh), to_dev(kh), 1, rs, xb, xs, yb + ys * first, ys)
        if need_v:
            bv = bv.copy()
            bv[:, 0] -= first
        cur, cur_rs, cur_h, cyb, cys = tmp, ow * C, last - first, 0, 1
        xb, xs = 0, 1
    if need_v:
        out = torch.empty((oh, cur.shape[1], C), dtype=torch.uint8, device=dev)
        lib.resample_u8_pass(cur, out, to_dev(bv), to_dev(kv), 0, cur_rs, cyb, cys, xb, xs)
        return out
    return cur


def device_resize_nearest(mat, size, hflip=False, vflip=False):
    """Image.resize(size[::-1], NEAREST) of an (h,w) device map of 1-, 2- or 4-byte elements (depth in mm, labels), optionally
    flipped first; `mat` may be a crop window view of a larger map."""
    lib = hip.library()
    h, w = mat.shape
    oh, ow = int(size[0]), int(size[1])
    if mat.stride(1) != 1:
        raise ValueError("device_resize_nearest: an (h,w) map or a row-window view of one expected")
    yt = nearest_table(h, oh) if oh != h else np.arange(h, dtype=np.int32)
    xt = nearest_table(w, ow) if ow != w else np.arange(w, dtype=np.int32)
    if vflip:
        yt = (h - 1 - yt).astype(np.int32)
    if hflip:
        xt = (w - 1 - xt).astype(np.int32)
    out = torch.empty((oh, ow), dtype=mat.dtype, device=mat.device)
    eb = mat.element_size()
    lib.gather2d(mat, out, torch.from_numpy(np.ascontiguousarray(yt)).to(mat.device), torch.from_numpy(np.ascontiguousarray(xt)).to(mat.device),
                 mat.stride(0) * eb, eb)
    return out


# --- the same transforms on the line targets (host tensors, a few dozen rows); pinned by tests/golden/line_transforms.npz, which
# oracle/make_golden_lines.py produces with the reference's own crop / hflip / vflip / resize / Normalize ---------------------------
def hflip_lines(lines, w):
    """transforms_depth.py:218-222: end points swapped, x -> w - x."""
    return lines[:, [2, 3, 0, 1]] * torch.as_tensor([-1.0, 1.0, -1.0, 1.0]) + torch.as_tensor([float(w), 0.0, float(w), 0.0])


def vflip_lines(lines, h):
    """transforms_depth.py:242-250: y -> h - y; vertical lines keep their upper point first."""
    lines = lines * torch.as_tensor([1.0, -1.0, 1.0, -1.0]) + torch.as_tensor([0.0, float(h), 0.0, float(h)])
    vert = lines[:, 0] == lines[:, 2]
    lines[vert] = lines[vert][:, [2, 3, 0, 1]]
    return lines


def resized_shape(w, h, size, max_size=None):
    """(oh, ow) of RandomResize's aspect-preserving resize (transforms_depth.py:318-343)."""
    if isinstance(size, (list, tuple)):
        return int(size[1]), int(size[0])
    if max_size is not None:
        mn, mx = float(min(w, h)), float(max(w, h))
        if mx / mn * size > max_size:
            size = int(round(max_size * mn / mx))
    if (w <= h and w == size) or (h <= w and h == size):
        return h, w
    if w < h:
        return int(size * h / w), size
    return size, int(size * w / h)


def resize_lines(lines, w, h, ow, oh):
    """transforms_depth.py:350-354."""
    rw, rh = float(ow) / float(w), float(oh) / float(h)
    return lines * torch.as_tensor([rw, rh, rw, rh])


# crop(): the eight clipping rules in the order the reference applies them (transforms_depth.py:95-121).  Each rule moves ONE end
# point P of a line onto a window edge and slides it along the line through the other end point Q:
#   (P, axis, edge): when P[axis] is beyond that edge, P[axis] = edge and the other coordinate of P follows from Q and the slope
#   (x-rule: P.y = Q.y + (P.x - Q.x) * slope;  y-rule: P.x = Q.x + (P.y - Q.y) / slope - the reference writes rules 2 and 6 as
#   Q.x - (Q.y - P.y) / slope, the same fp32 value: negation is exact and commutes with the division).
_CLIP_RULES = ((0, 0, "lo"), (0, 1, "lo"), (1, 0, "hi"), (1, 1, "hi"), (1, 0, "lo"), (1, 1, "lo"), (0, 0, "hi"), (0, 1, "hi"))


def crop_lines(lines, region):
    """Line targets under a crop window (i, j, h, w), transforms_depth.py:59-128: shift into the window, drop the lines that lie
    wholly beyond one edge, clip the others edge by edge along their slope (the rule table above, all lines at once), clamp.
    Returns (lines (m,4), keep mask over the input rows).  Pinned by tests/golden/line_transforms.npz (the reference's own crop)."""
    i, j, h, w = region
    shifted = lines - torch.as_tensor([j, i, j, i], dtype=lines.dtype)
    pts = shifted.view(-1, 2, 2)                                     # (line, end point, axis)
    hi = torch.as_tensor([w, h], dtype=lines.dtype)
    beyond = ((pts < 0).all(dim=1) | (pts > hi).all(dim=1)).any(dim=1)         # both end points past the same edge, on either axis
    keep = ~beyond
    p = pts[keep].clone()
    slope = (p[:, 1, 1] - p[:, 0, 1]) / (p[:, 1, 0] - p[:, 0, 0] + 1e-12)
    # The reference assigns the window edge as a Python int; where BOTH coordinates of a y-rule's difference are such edge values the
    # quotient is `int / tensor`, which torch evaluates as tensor.reciprocal() * int - one rounding more than a division.  `on_edge`
    # tracks which coordinates are edge values so that the result is the reference's to the last bit.
    on_edge = torch.zeros(p.shape, dtype=torch.bool)
    for end, axis, side in _CLIP_RULES:
        edge = 0.0 if side == "lo" else float(hi[axis])
        coord = p[:, end, axis]
        hit = coord < 0 if side == "lo" else coord > edge
        moved = torch.where(hit, torch.full_like(coord, edge), coord)
        delta = moved - p[:, 1 - end, axis]
        if axis == 0:
            step = delta * slope
        else:
            step = torch.where(on_edge[:, 1 - end, axis], delta * (1.0 / slope), delta / slope)
        p[:, end, 1 - axis] = torch.where(hit, p[:, 1 - end, 1 - axis] + step, p[:, end, 1 - axis])
        p[:, end, axis] = moved
        on_edge[:, end, axis] |= hit
        on_edge[:, end, 1 - axis] &= ~hit                           # re-computed along the slope: a tensor value again
    p = torch.minimum(torch.maximum(p, torch.zeros_like(hi)), hi)
    return p.reshape(-1, 4), keep


def chain_points(poly_lines):
    """The vertex list the reference averages for a polygon's centre: both end points of the first line, then the END point of
    every further line - for a closed polygon the first vertex therefore counts twice (transforms_depth.py:150,
    glassrgbd_norhint.py:180-181).  (m,4) -> (m+1, 2)."""
    return torch.cat([poly_lines[0].view(2, 2), poly_lines[1:, 2:]], 0)


def _clip_polygon(pts, x0, y0, x1, y1):
    """Sutherland-Hodgman clip of a polygon (list of (x, y)) against an axis-aligned window."""
    def clip(points, inside, cross):
        out = []
        for k, cur in enumerate(points):
            prev = points[k - 1]
            if inside(cur):
                if not inside(prev):
                    out.append(cross(prev, cur))
                out.append(cur)
            elif inside(prev):
                out.append(cross(prev, cur))
        return out

    def at_x(x):
        return lambda a, b: (x, a[1] + (b[1] - a[1]) * (x - a[0]) / (b[0] - a[0]))

    def at_y(y):
        return lambda a, b: (a[0] + (b[0] - a[0]) * (y - a[1]) / (b[1] - a[1]), y)

    for inside, cross in ((lambda q: q[0] >= x0, at_x(x0)), (lambda q: q[0] <= x1, at_x(x1)),
                          (lambda q: q[1] >= y0, at_y(y0)), (lambda q: q[1] <= y1, at_y(y1))):
        if not pts:
            break
        pts = clip(pts, inside, cross)
    return pts


def crop_targets(lines, poly_ids, centres, region):
    """crop() on the whole target (transforms_depth.py:59-186): clipped lines, surviving polygon ids and the re-computed polygon
    centres.  A polygon that keeps more than three lines gets the mean of its clipped chain's vertices (chain_points; pinned by the
    reference's own crop through tests/golden/line_transforms.npz).  One that keeps three or fewer gets the vertex mean of
    (window INTERSECT original polygon), for which the reference calls shapely (absent here and unpinned by the reference): the
    intersection is restated as a Sutherland-Hodgman clip with the ring closed the way shapely's `exterior.coords` closes it
    (first vertex repeated), the vertex ORDER GEOS would emit is not reproducible without it - that branch is parity-unpinned."""
    i, j, h, w = region
    out, keep = crop_lines(lines, region)
    ids = poly_ids[keep]
    if centres is None:
        return out, ids, None, keep
    flipped = bool(lines.shape[0] > 1 and lines[0, 0] == lines[1, 2] and lines[0, 1] == lines[1, 3])      # :139-141
    new_c = torch.zeros((out.shape[0], 2), dtype=centres.dtype)

    def chain(pl):
        return chain_points(pl.view(-1, 2, 2).flip(1).reshape(-1, 4) if flipped else pl)

    for pid in torch.unique(ids):
        sel = ids == pid
        if int(sel.sum()) > 3:
            new_c[sel] = chain(out[sel]).mean(0)
            continue
        ring = _clip_polygon([tuple(q) for q in chain(lines[poly_ids == pid]).tolist()], j, i, j + w - 1, i + h - 1)
        if len(ring) >= 3:
            ring = ring + [ring[0]]
            c = torch.tensor([sum(q[0] for q in ring) / len(ring) - j, sum(q[1] for q in ring) / len(ring) - i], dtype=centres.dtype)
            new_c[sel] = torch.minimum(torch.maximum(c, torch.zeros(2)), torch.tensor([float(w), float(h)]))
        else:
            pl = out[sel]                                            # the reference flips an already flipped chain back here (:165-167)
            new_c[sel] = chain_points(pl).mean(0)
    return out, ids, new_c, keep


def normalize_lines(lines, w, h, centres=None):
    """Normalize.__call__ on the targets (transforms_depth.py:632-641): pixel coordinates -> fractions of the final image size."""
    lines = lines / torch.tensor([w, h, w, h], dtype=torch.float32)
    if centres is None:
        return lines
    return lines, centres / torch.tensor([w, h], dtype=torch.float32)


def polygon_lines(shapes):
    """generate_line_labels (glassrgbd_norhint.py:161-193): every labelled polygon (a dict with 'points' and 'poly_id') becomes the
    closed chain of its edges; returns (lines (n,4), poly_ids (n,), centres (n,2)) as float64 / int64 numpy arrays, the centre of
    a polygon being the mean of chain_points (first vertex counted twice)."""
    import numpy as np
    lines, ids, centres = [], [], []
    for poly in shapes:
        pts = np.asarray(poly["points"], dtype=np.float64)
        if len(pts) == 0:
            continue
        edges = np.concatenate([pts, np.roll(pts, -1, axis=0)], axis=1)           # (x1, y1, x2, y2), the last edge closes the ring
        chain = np.concatenate([pts, pts[:1]], axis=0)
        c = (sum(chain[:, 0].tolist()) / len(chain), sum(chain[:, 1].tolist()) / len(chain))
        lines += edges.tolist()
        ids += [poly["poly_id"]] * len(edges)
        centres += [c] * len(edges)
    return np.array(lines), np.array(ids), centres


def assemble_item(rgb, depth_mm, labels, shapes, image_id, with_center=True, params=None, mean=MEAN, std=STD):
    """One dataset item from DECODED arrays - DataLoadPreprocess.__getitem__ without the file handling
    (glassrgbd_norhint.py:236-299 with ConvertLinePolysToMask :121-148): polygon JSON -> line targets (clamped to the frame),
    the transform chain on the device (DeviceAugment.apply with `params`, None = no geometric / photometric step), then the
    per-item tail (lines / centres as fractions of the final size, centres appended to the lines under --with_center).  The
    pixel tail (ToTensor + Normalize, depth / 1000, label > 0) is NOT applied here: device_collate does it for the whole batch in
    one launch.  rgb uint8 (h,w,3), depth_mm integer (h,w), labels uint8 (h,w): device tensors.
    Returns (rgb, depth_mm, labels, target) with target = {lines, labels, poly_ids, image_id, orig_size, size}."""
    h, w = int(rgb.shape[0]), int(rgb.shape[1])
    ln, ids, cs = polygon_lines(shapes)
    lines = torch.as_tensor(ln, dtype=torch.float32).reshape(-1, 4)
    centres = torch.as_tensor(cs, dtype=torch.float32).reshape(-1, 2)
    ids = torch.as_tensor(ids, dtype=torch.int64).reshape(-1)
    if len(lines) > 0:
        lines = torch.minimum(torch.maximum(lines, torch.zeros(4)), torch.tensor([w, h, w, h], dtype=torch.float32))
        centres = torch.minimum(torch.maximum(centres, torch.zeros(2)), torch.tensor([w, h], dtype=torch.float32))
    if params is not None:
        rgb, depth_mm, labels, lines, ids, centres = DeviceAugment.apply(rgb, depth_mm, labels, lines, params, poly_ids=ids, centres=centres)
    fh, fw = int(rgb.shape[0]), int(rgb.shape[1])
    lines, centres = normalize_lines(lines, fw, fh, centres)
    target = {"lines": torch.cat([lines, centres], dim=1) if with_center else lines,
              "labels": torch.zeros(lines.shape[0], dtype=torch.int64), "poly_ids": ids,
              "image_id": torch.tensor([image_id]), "orig_size": torch.as_tensor([h, w]), "size": torch.as_tensor([fh, fw])}
    return rgb, depth_mm, labels, target


class DeviceAugment:
    """The reference's training / validation transform chain (src/datasets/coco.py:74-117) over DEVICE images: random flip, random
    resize (optionally resize -> random crop -> resize), colour jitter, then device_collate normalises and pads.  `params()` draws
    the random choices (the same choices the reference makes, from this object's own generator); `apply()` is deterministic
    given them."""
    SCALES = [480, 512, 544, 576, 608, 640, 672, 680, 690, 704, 736, 768, 788, 800]

    def __init__(self, train=True, max_size=1024, test_size=1024, seed=None):
        self.train, self.max_size, self.test_size = train, max_size, test_size
        self.rng = random.Random(seed)

    def params(self, w, h):
        r = self.rng
        if not self.train:
            return {"flip": None, "steps": [("resize", self.test_size, self.max_size)]}
        flip = "h" if r.random() < 0.5 else "v"                     # RandomSelect(HFlip, VFlip); each flip itself has p = 0.5
        if r.random() >= 0.5:
            flip = None
        if r.random() < 0.5:
            steps = [("resize", r.choice(self.SCALES), self.max_size)]
        else:
            s1 = r.choice([400, 500, 600])
            oh, ow = resized_shape(w, h, s1)
            cw, ch = r.randint(384, min(ow, 600)), r.randint(384, min(oh, 600))
            i, j = r.randint(0, oh - ch), r.randint(0, ow - cw)
            steps = [("resize", s1, None), ("crop", (i, j, ch, cw)), ("resize", r.choice(self.SCALES), self.max_size)]
        return {"flip": flip, "steps": steps, "jitter": jitter_params(r)}          # T.ColorJitter() with its defaults (coco.py:107)

    @staticmethod
    def apply(rgb, depth_mm, labels, lines, p, poly_ids=None, centres=None):
        """rgb uint8 (h,w,3), depth_mm int32 (h,w), labels uint8 (h,w) device tensors, lines (n,4) host fp32 in pixels.
        Returns the transformed (rgb, depth_mm, labels, lines [still in pixels], keep mask over the input lines); with poly_ids
        (n,) and centres (n,2) - the polygon bookkeeping of the dataset's targets - (rgb, depth_mm, labels, lines, poly_ids, centres)."""
        h, w = rgb.shape[:2]
        hf, vf = p["flip"] == "h", p["flip"] == "v"
        lines = lines.clone().float()
        keep = torch.ones(lines.shape[0], dtype=torch.bool)
        full = poly_ids is not None
        if full:
            poly_ids, centres = poly_ids.clone(), centres.clone().float()
        if hf:
            lines = hflip_lines(lines, w)
            if full:                                                # transforms_depth.py:223-225
                centres = centres * torch.as_tensor([-1.0, 1.0]) + torch.as_tensor([float(w), 0.0])
        if vf:
            lines = vflip_lines(lines, h)
            if full:                                                # :251-253
                centres = centres * torch.as_tensor([1.0, -1.0]) + torch.as_tensor([0.0, float(h)])
        first = True
        for step in p["steps"]:
            if step[0] == "resize":
                oh, ow = resized_shape(w, h, step[1], step[2])
                f = (hf, vf) if first else (False, False)           # the flip rides in the first resize's read
                rgb = device_resize_rgb(rgb, (oh, ow), *f)
                depth_mm = device_resize_nearest(depth_mm, (oh, ow), *f) if depth_mm is not None else None
                labels = device_resize_nearest(labels, (oh, ow), *f) if labels is not None else None
                lines = resize_lines(lines, w, h, ow, oh)
                if full:                                            # :356-357
                    centres = centres * torch.as_tensor([float(ow) / float(w), float(oh) / float(h)])
                h, w = oh, ow
                first = False
            else:
                i, j, ch, cw = step[1]
                rgb = rgb[i:i + ch, j:j + cw]                       # window views: the next resize reads them in place
                depth_mm = depth_mm[i:i + ch, j:j + cw] if depth_mm is not None else None
                labels = labels[i:i + ch, j:j + cw] if labels is not None else None
                if full:
                    lines, poly_ids, centres, k = crop_targets(lines, poly_ids, centres, step[1])
                else:
                    lines, k = crop_lines(lines, step[1])
                idx = torch.nonzero(keep).flatten()
                keep = torch.zeros_like(keep)
                keep[idx[k]] = True
                h, w = ch, cw
        if first and (hf or vf):                                    # a flip with no resize behind it
            rgb = device_resize_rgb(rgb, (h, w), hf, vf)
            depth_mm = device_resize_nearest(depth_mm, (h, w), hf, vf) if depth_mm is not None else None
            labels = device_resize_nearest(labels, (h, w), hf, vf) if labels is not None else None
        if p.get("jitter"):
            rgb = device_color_jitter(rgb, p["jitter"])
        rgb = rgb.contiguous()
        depth_mm = None if depth_mm is None else depth_mm.contiguous()
        labels = None if labels is None else labels.contiguous()
        if full:
            return rgb, depth_mm, labels, lines, poly_ids, centres
        return rgb, depth_mm, labels, lines, keep


# --- third slice: the photometric jitter (transforms_depth.py:551-600) ------------------------------------------------------------
def hue_shift(hue_factor):
    """uint8 shift of the H channel for adjust_hue(hue_factor): np.uint8(hue_factor * 255) as a C cast wraps it."""
    return int(hue_factor * 255) & 255


def device_color_jitter(rgb, ops):
    """rgb uint8 (h,w,3) device image; ops: sequence of ('brightness' | 'contrast' | 'saturation' | 'hue', factor) applied in order,
    each bit-exact with torchvision.transforms.functional.adjust_* on the PIL image (Pillow ImageEnhance / HSV conversion)."""
    lib = hip.library()
    cur = rgb.contiguous()
    scratch = None
    for name, f in ops:
        out = torch.empty_like(cur)
        if name == "contrast" and scratch is None:
            scratch = torch.zeros(1, dtype=torch.int64, device=cur.device)
        lib.color_adjust(cur, out, name, float(hue_shift(f)) if name == "hue" else float(f), scratch if name == "contrast" else None)
        cur = out
    return cur


def jitter_params(rng, brightness=0.4, contrast=0.4, saturation=0.4, hue=0.4):
    """The random choices of ColorJitter.__call__ (a random order of the four adjustments, one uniform factor each) from a
    random.Random; the reference draws them from torch's generator (randperm(4), uniform_)."""
    order = list(range(4))
    rng.shuffle(order)
    names = ["brightness", "contrast", "saturation", "hue"]
    lo_hi = [(max(0.0, 1 - brightness), 1 + brightness), (max(0.0, 1 - contrast), 1 + contrast), (max(0.0, 1 - saturation), 1 + saturation),
             (-hue, hue)]
    return [(names[i], rng.uniform(*lo_hi[i])) for i in order]
