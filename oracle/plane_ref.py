"""CPU restatement (torch, autograd for the gradient) of the reference's PlaneLoss.  TEST INFRASTRUCTURE ONLY.

Follows src/models/glassrgbd.py:385-450 line by line; the one third-party piece, matplotlib.path.Path.contains_points
(matplotlib 3.10.8, `point_in_path_impl` of src/_path.h: the crossing test of Graphics Gems IV with the closing edge
appended), is restated in `points_in_triangle` - all operands are integers held in doubles, so it is exact.
Pinned by tests/golden/plane_loss.npz, produced by oracle/make_golden_plane.py from the reference's own PlaneLoss
(forward and backward): tests/test_plane_loss.py::test_oracle_*.
"""
import torch
import torch.nn.functional as F

SOBEL = torch.tensor([[[1., 0., -1.], [2., 0., -2.], [1., 0., -1.]],
                      [[1., 2., 1.], [0., 0., 0.], [-1., -2., -1.]]]).view(2, 1, 3, 3)       # src/models/losses/sobel.py:9-13


def points_in_triangle(tri, px, py):
    """Path(tri).contains_points(points): tri (3,2) integer vertices (x, y), px/py float64 arrays.  For every edge a->b of
    the implicitly closed path: if (a.y >= ty) != (b.y >= ty) and ((b.y-ty)*(a.x-b.x) >= (b.x-tx)*(a.y-b.y)) == (b.y >= ty): flip."""
    inside = torch.zeros_like(px, dtype=torch.bool)
    v = tri.double()
    for a, b in ((0, 1), (1, 2), (2, 0)):
        ax, ay, bx, by = v[a, 0], v[a, 1], v[b, 0], v[b, 1]
        f0, f1 = ay >= py, by >= py
        hit = ((by - py) * (ax - bx) >= (bx - px) * (ay - by)) == f1
        inside ^= (f0 != f1) & hit
    return inside


def choose_triangles(line_pred, line_score, H, W, num_ref=28, thresh=0.6):
    """:397-418 -> (top_num, (top_num,3,2) int64 vertices).  NB the reference COUNTS the lines whose softmax score passes
    the threshold but then takes the top `count` lines by RAW class-0 logit."""
    keep = torch.softmax(line_score, dim=-1)[:, :, 0] > thresh
    top_num = min(int(keep.sum()), num_ref)
    _, ids = torch.topk(line_score[:, :, 0], top_num, dim=-1)
    lines = line_pred[0][ids[0]] * torch.tensor([W, H, W, H, W, H], dtype=torch.float32)
    lines = torch.round(lines)
    lines[:, 0::2].clamp_(min=0, max=W - 1)
    lines[:, 1::2].clamp_(min=0, max=H - 1)
    return top_num, lines.reshape(-1, 3, 2).long()


def plane_loss(depth_pred, line_pred, line_score, valid_mask, num_ref=28, thresh=0.6, min_area=100):
    """PlaneLoss.forward (:393-450) for one image: depth_pred (1,1,H,W) (may require grad), valid_mask (1,1,H,W) bool."""
    assert line_score.shape[0] == 1, "one image each iter"
    H, W = depth_pred.shape[-2:]
    g = F.conv2d(depth_pred, SOBEL, padding=1)                       # :405 (cross-correlation, zero padding)
    nx, ny = -g[0, 0].flatten(), -g[0, 1].flatten()                 # :406-408
    top_num, tris = choose_triangles(line_pred, line_score, H, W, num_ref, thresh)
    valid = valid_mask[0, 0].flatten()
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    px, py = xs.flatten()[valid], ys.flatten()[valid]               # :420-424 (x = column, y = row), valid pixels only
    nxv, nyv = nx[valid], ny[valid]
    total, planes = depth_pred.new_zeros(()), 0
    for j in range(top_num):
        m = points_in_triangle(tris[j], px, py)
        if int(m.sum()) < min_area:                                  # :437-440
            continue
        total = total + torch.var(nxv[m], unbiased=False) + torch.var(nyv[m], unbiased=False)
        planes += 1
    return total / max(1, planes)
