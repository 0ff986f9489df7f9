"""Generate tests/golden/plane_loss.npz by running the REAL reference's PlaneLoss on CPU (build container only).

TEST INFRASTRUCTURE ONLY.  Usage:  python -m oracle.make_golden_plane

What is executed is the reference's own, unmodified models.glassrgbd.PlaneLoss (src/models/glassrgbd.py:385-450:
Sobel normals src/models/losses/sobel.py:5-27, matplotlib.path.Path.contains_points, per-plane variances), imported
from /root/reference under the stand-ins of oracle/ref_stubs.py, forward AND backward (gradient w.r.t. the predicted
depth).  `Tensor.cuda` / `Module.cuda` are the identity for the duration of the call (PlaneLoss calls .cuda() on its
constants; this container has no GPU).  matplotlib 3.10.8 is installed here and is used as is.  Only vectors travel.
"""
import os

import numpy as np
import torch

from . import ref_stubs

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def canned_inputs(case):
    H, W, seed = case["H"], case["W"], case["seed"]
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    depth = 2.0 + 0.01 * xx + 0.02 * yy + 0.3 * torch.sin(xx / 7.0) * torch.cos(yy / 5.0) + 0.05 * torch.randn(H, W, generator=g)
    depth = depth.view(1, 1, H, W)
    gt = (depth + 0.1 * torch.randn(1, 1, H, W, generator=g)).clamp(0.0, 12.0)
    valid = (gt >= 0.2) & (gt < 10.0) & (torch.rand(1, 1, H, W, generator=g) > 0.1)
    lines = torch.rand(1, 100, 6, generator=g)
    # hand-made triangles: axis-aligned edges through pixel centres (boundary rule), a degenerate one, a tiny one (< min area)
    lines[0, 0] = torch.tensor([0.1, 0.1, 0.7, 0.1, 0.1, 0.8])
    lines[0, 1] = torch.tensor([0.5, 0.5, 0.5, 0.5, 0.9, 0.9])
    lines[0, 2] = torch.tensor([0.30, 0.30, 0.33, 0.30, 0.30, 0.34])
    lines[0, 3] = torch.tensor([0.0, 0.0, 1.0, 0.0, 1.0, 1.0])          # clamps to the image corners
    scores = torch.randn(1, 100, 2, generator=g)
    scores[0, :case["confident"], 0] += 4.0                                # softmax > 0.6 for these
    scores[0, case["confident"]:, 0] -= 2.0
    return depth, gt, lines, scores, valid


CASES = {"p40_96x128": dict(H=96, W=128, seed=31, confident=40), "p5_60x80": dict(H=60, W=80, seed=32, confident=5),
         "p0_48x64": dict(H=48, W=64, seed=33, confident=0)}


def main():
    ref_stubs.install()
    orig_t, orig_m = torch.Tensor.cuda, torch.nn.Module.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    out = {}
    try:
        from models.glassrgbd import PlaneLoss           # /root/reference/src/models/glassrgbd.py:385
        crit = PlaneLoss(28, line_score_thresh=0.6, min_plane_area=100)          # as build() constructs it (:575)
        for name, case in CASES.items():
            depth, gt, lines, scores, valid = canned_inputs(case)
            depth = depth.clone().requires_grad_(True)
            loss = crit(depth, gt, lines, scores, valid)
            if loss.requires_grad:
                loss.backward()
                grad = depth.grad.detach()
            else:
                grad = torch.zeros_like(depth)                                    # no plane survived: constant 0
            out[name + "/depth"] = depth.detach().numpy()
            out[name + "/gt"] = gt.numpy()
            out[name + "/lines"] = lines.numpy()
            out[name + "/scores"] = scores.numpy()
            out[name + "/valid"] = valid.numpy()
            out[name + "/loss"] = np.float64(float(loss))
            out[name + "/grad"] = grad.numpy()
            print(name, "loss", float(loss), "grad l2", float(grad.norm()))
    finally:
        torch.Tensor.cuda, torch.nn.Module.cuda = orig_t, orig_m
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    path = os.path.join(GOLDEN_DIR, "plane_loss.npz")
    np.savez_compressed(path, **out)
    print(path, "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
