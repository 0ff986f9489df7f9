"""PlaneLoss (--with_plane_norm_loss): reference PlaneLoss -> golden vectors -> oracle -> device kernels.

tests/golden/plane_loss.npz holds three cases run through the reference's OWN PlaneLoss (forward + backward,
oracle/make_golden_plane.py; matplotlib's contains_points included).  The triangle masks are integer work: the oracle
reproduces the reference's gradient bit for bit; the kernels sum in f64 / multiply in fp32, tolerance 2e-5 relative.
"""
import os

import numpy as np
import pytest
import torch

from gw_depth_amd import hip
from gw_depth_amd.criteria import PlaneLoss
from oracle import plane_ref
from tests.fake_device import FakeDevice

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "plane_loss.npz")
CASES = ["p40_96x128", "p5_60x80", "p0_48x64"]


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(GOLDEN))


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-20))


def inputs(gold, name, device="cpu"):
    t = lambda k: torch.from_numpy(gold[name + "/" + k]).to(device)
    return t("depth"), t("gt"), t("lines"), t("scores"), t("valid")


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_plane_loss(gold, name):
    depth, gt, lines, scores, valid = inputs(gold, name)
    depth.requires_grad_(True)
    loss = plane_ref.plane_loss(depth, lines, scores, valid)
    loss.backward()
    assert float(loss.detach()) == pytest.approx(float(gold[name + "/loss"]), rel=1e-6)
    assert torch.equal(depth.grad, torch.from_numpy(gold[name + "/grad"]))          # identical masks, identical arithmetic


def test_point_in_triangle_boundary_rule():
    """Edges through pixel centres: matplotlib's crossing test includes the left / upper boundary and excludes the right /
    lower one (right triangle (1,1),(7,1),(1,8)): checked pixel by pixel against hand-derived membership."""
    tri = torch.tensor([[1, 1], [7, 1], [1, 8]])
    ys, xs = torch.meshgrid(torch.arange(10, dtype=torch.float64), torch.arange(10, dtype=torch.float64), indexing="ij")
    m = plane_ref.points_in_triangle(tri, xs.flatten(), ys.flatten()).view(10, 10)
    assert not m[1].any() and m[2, 1] and m[2, 6] and not m[2, 7]                  # y = 1 edge excluded; x = 1 edge included
    assert not m[:, 0].any() and not m[9].any() and int(m.sum()) > 0


@pytest.fixture()
def fake():
    hip.set_library(FakeDevice())
    yield
    hip.set_library(None)


@pytest.mark.parametrize("name", CASES)
def test_host_logic_matches_reference_plane_loss(fake, gold, name):
    depth, gt, lines, scores, valid = inputs(gold, name)
    depth.requires_grad_(True)
    loss = PlaneLoss(28, 0.6, 100)(depth, gt, lines, scores, valid)
    loss.backward()
    assert float(loss.detach()) == pytest.approx(float(gold[name + "/loss"]), rel=1e-5)
    assert rel(depth.grad, gold[name + "/grad"]) < 1e-5
    with pytest.raises(AssertionError):
        PlaneLoss()(depth, gt, lines.repeat(2, 1, 1), scores.repeat(2, 1, 1), valid)


def test_build_model_returns_plane_criterion():
    from gw_depth_amd import Config, build_model
    _, crits, _ = build_model(Config(device="cpu", with_plane_norm_loss=True))
    assert isinstance(crits[3], PlaneLoss) and crits[3].num_ref == 28
    _, crits, _ = build_model(Config(device="cpu"))
    assert crits[3] is None


@pytest.fixture()
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    hip.set_library(None)
    return hip.library()


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_kernels_match_reference_plane_loss(dev, gold, name):
    depth, gt, lines, scores, valid = inputs(gold, name, "cuda")
    depth.requires_grad_(True)
    loss = PlaneLoss(28, 0.6, 100)(depth, gt, lines, scores, valid)
    (3.0 * loss).backward()
    torch.cuda.synchronize()
    assert float(loss.detach()) == pytest.approx(float(gold[name + "/loss"]), rel=2e-5)
    assert rel(depth.grad / 3.0, gold[name + "/grad"]) < 2e-5


@pytest.mark.gpu
def test_kernels_full_size_vs_oracle_and_no_plane_case(dev):
    """480x640 (BASELINE's frame): kernels == oracle; with no confident line the loss is exactly 0 with a zero gradient."""
    H, W = 480, 640
    g = torch.Generator().manual_seed(9)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    depth = (3.0 + 0.002 * xx + 0.004 * yy + 0.2 * torch.sin(xx / 30) + 0.02 * torch.randn(H, W, generator=g)).view(1, 1, H, W)
    gt = depth + 0.1 * torch.randn(1, 1, H, W, generator=g)
    valid = (gt >= 0.2) & (gt < 10.0) & (torch.rand(1, 1, H, W, generator=g) > 0.1)
    lines = torch.rand(1, 100, 6, generator=g)
    scores = torch.randn(1, 100, 2, generator=g)
    scores[0, :35, 0] += 4.0
    d0 = depth.clone().requires_grad_(True)
    ref = plane_ref.plane_loss(d0, lines, scores, valid)
    ref.backward()
    d1 = depth.clone().cuda().requires_grad_(True)
    loss = PlaneLoss()(d1, gt.cuda(), lines.cuda(), scores.cuda(), valid.cuda())
    loss.backward()
    assert float(loss.detach()) == pytest.approx(float(ref.detach()), rel=2e-5)
    assert rel(d1.grad, d0.grad) < 2e-5
    scores[0, :, 0] = -5.0
    d2 = depth.clone().cuda().requires_grad_(True)
    loss = PlaneLoss()(d2, gt.cuda(), lines.cuda(), scores.cuda(), valid.cuda())
    loss.backward()
    assert float(loss.detach()) == 0.0 and float(d2.grad.abs().max()) == 0.0
