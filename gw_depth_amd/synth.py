"""Deterministic synthetic batches and deterministic weights for the GW-Depth train step.

Shapes and statistics follow SURVEY.md §8(d): images N(0,1) (post-normalisation statistics of
/root/reference/src/datasets/coco.py:76-79), depth GT U(0.5, 9.5) m with 10 % of pixels zeroed
so the validity mask [0.2, 10) of /root/reference/src/engine_glassrgbd.py:65 is exercised, seg
GT Bernoulli(0.5) int64, T target lines U(0,1) (T,6) with label 0
(/root/reference/src/datasets/glassrgbd_norhint.py:279-295).  Everything is generated on the
CPU generator so the GPU run, the CPU baseline and the golden fixtures see the same bytes.
"""
import zlib

import torch


def synth_batch(batch, height, width, seed=1, n_lines=7, sizes=None):
    """Returns dict(images (B,3,H,W) f32, pad_mask (B,H,W) bool [True = padding],
    depth (B,1,H,W) f32, seg (B,1,H,W) i64, targets [ {lines (T,6), labels (T,)} ]).
    `sizes` = optional per-image (h, w) <= (height, width): ragged batch, zero padded bottom/right
    exactly like nested_tensor_from_tensor_list (/root/reference/src/util/misc.py:291-313)."""
    g = torch.Generator().manual_seed(int(seed))
    images = torch.randn(batch, 3, height, width, generator=g)
    depth = torch.rand(batch, 1, height, width, generator=g) * 9.0 + 0.5
    hole = torch.rand(batch, 1, height, width, generator=g) < 0.1
    depth = depth.masked_fill(hole, 0.0)
    seg = (torch.rand(batch, 1, height, width, generator=g) < 0.5).to(torch.int64)
    pad_mask = torch.zeros(batch, height, width, dtype=torch.bool)
    if sizes is not None:
        for b, (h, w) in enumerate(sizes):
            pad_mask[b, h:, :] = True
            pad_mask[b, :, w:] = True
        keep = (~pad_mask)[:, None]
        images = images * keep
        depth = depth * keep
        seg = seg * keep
    targets = []
    for b in range(batch):
        t = n_lines if isinstance(n_lines, int) else n_lines[b]
        targets.append({"lines": torch.rand(t, 6, generator=g),
                        "labels": torch.zeros(t, dtype=torch.int64)})
    return {"images": images, "pad_mask": pad_mask, "depth": depth, "seg": seg, "targets": targets}


def det_fill_(state, seed=0):
    """Overwrite every floating tensor of a GW-Depth state dict, in place, with a value stream that
    depends only on (seed, key name, shape).  Integer buffers (relative_position_index) are left
    alone.  Scales are chosen so activations stay O(1) through the 100+ layer path."""
    for name, t in state.items():
        if not t.is_floating_point():
            continue
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
        leaf = name.rsplit(".", 1)[-1]
        shape = tuple(t.shape)
        if leaf == "running_var":
            v = torch.rand(shape, generator=g) + 0.5
        elif leaf == "running_mean":
            v = torch.randn(shape, generator=g) * 0.1
        elif t.dim() >= 2 and leaf not in ("depth_token", "seg_token", "diff_mu", "diff_logsigma",
                                           "border_mu", "border_logsigma"):
            fan_in = t[0].numel()
            fan_out = t.shape[0] * (t[0, 0].numel() if t.dim() > 2 else 1)
            if leaf == "relative_position_bias_table":
                std = 0.2
            elif "query_embed" in name:
                std = 1.0
            else:
                std = (2.0 / (fan_in + fan_out)) ** 0.5
            v = torch.randn(shape, generator=g) * std
        elif leaf == "weight":       # LayerNorm / FrozenBN scale
            v = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif leaf == "bias":
            v = 0.02 * torch.randn(shape, generator=g)
        elif leaf in ("diff_mu", "border_mu"):
            v = torch.randn(shape, generator=g)
        else:                         # tokens, logsigma
            v = 0.1 * torch.randn(shape, generator=g)
        t.copy_(v.to(t.dtype))
    return state
