"""CPU: the product's host logic (module wiring, pixel-major layouts, autograd plumbing, state-dict
mapping, flat optimizer buffers) against the golden vectors of the real reference, with the device
library replaced by tests/fake_device.py (plain torch math).  The kernels themselves are checked on
the GPU by tests/test_hip_kernels.py and tests/test_gpu_parity.py."""
import os

import numpy as np
import pytest
import torch

from gw_depth_amd import hip
from gw_depth_amd.synth import det_fill_, synth_batch
from tests.fake_device import FakeDevice
from tests.golden_check import build, check_train_step
from tests.helpers import reference_state_shapes

FP_TOL = 3e-4


@pytest.fixture()
def fake():
    hip.set_library(FakeDevice())
    yield
    hip.set_library(None)


def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + 1e-12))


def test_state_dict_roundtrip_is_reference_layout():
    _, model, _ = build()
    ref = det_fill_(reference_state_shapes(), seed=0)
    sd = model.state_dict()
    assert list(sd) and set(sd) == set(ref)
    for k in ref:
        assert sd[k].shape == ref[k].shape and torch.equal(sd[k], ref[k]), k
    # conv weights are stored kernel-native (Cout,KH,KW,Cin) but exported as (Cout,Cin,KH,KW)
    w = model.backbone._modules["0"].body.conv1.weight
    assert tuple(w.shape) == (64, 7, 7, 3) and tuple(sd["backbone.0.body.conv1.weight"].shape) == (64, 3, 7, 7)


def test_no_cpu_path_without_device_library():
    hip.set_library(None)
    _, model, _ = build()
    b = synth_batch(1, 64, 64, seed=3)
    with pytest.raises((hip.HipUnavailable, RuntimeError)):
        model([b["images"][0]])


@pytest.mark.parametrize("case", ["tiny_b2_96x128", "ragged_b2_96x128", "plane_b1_96x128"])
def test_train_step_wiring_matches_reference(fake, golden_dir, case):
    check_train_step(case, golden_dir, "cpu", tol=FP_TOL, grad_tol=3e-3)


def test_matcher_prefetch_equals_per_layer_matching(fake):
    """The single early Hungarian hand-off (HungarianMatcherLine.prefetch) returns, layer by layer, exactly what the
    reference's per-call matcher returns."""
    from gw_depth_amd.model import NestedTensor
    cfg, model, crits = build()
    b = synth_batch(2, 96, 128, seed=31, n_lines=[4, 2])
    model.train()
    matcher = crits[0].matcher
    with torch.no_grad():
        out = model(NestedTensor(b["images"], b["pad_mask"]), match=(matcher, b["targets"]))
    handle = out["_match_prefetch"]
    layers = [out] + out["aux_outputs"]
    for lay in layers:
        got = matcher.from_prefetch(handle)
        want = matcher(lay, b["targets"])
        for (gi, gj), (wi, wj) in zip(got, want):
            assert torch.equal(gi, wi) and torch.equal(gj, wj)


def test_sync_free_criterion_equals_reference_criterion(fake):
    """SetCriterion.forward_packed (device LSAP, static shapes, graph-capturable) == SetCriterion.forward."""
    from gw_depth_amd.criteria import pack_targets
    from gw_depth_amd.model import NestedTensor
    cfg, model, crits = build()
    b = synth_batch(2, 96, 128, seed=33, n_lines=[5, 2])
    model.train()
    with torch.no_grad():
        out = model(NestedTensor(b["images"], b["pad_mask"]))
        want = crits[0](out, b["targets"])
        got = crits[0].forward_packed(out, pack_targets(b["targets"], "cpu"))
    assert set(want) == set(got)
    for k in want:
        assert abs(float(want[k]) - float(got[k])) <= 1e-5 * max(1.0, abs(float(want[k]))), k


def test_padded_pyramid_equals_the_plain_one(fake):
    """PointBasedPred with the 30-point pyramid on zero-padded channel counts (32 / 64 / 320 / 128: ops._PadConvFn, LayerNorm with a
    row pitch, weight gradients folded back by unpad_add_batch) == the same module on its own widths: output and every gradient."""
    from gw_depth_amd.model import PointBasedPred, PyramidLayer
    torch.manual_seed(0)
    m = PointBasedPred(32, 16, 30)
    for p in m.parameters():
        torch.nn.init.normal_(p, std=0.2)
    B, H, W = 2, 18, 20
    x = torch.randn(B, H * W, 32, requires_grad=True)
    dtok = torch.randn(B, H * W, 16, requires_grad=True)
    pre = torch.rand(B, 1, 9, 10)
    coords = torch.rand(B, 30, 1, 2) * 2 - 1
    pos = torch.randn(B, H, W, 32)

    def run(force):
        PyramidLayer.FORCE_PAD = force
        try:
            for p in m.parameters():
                p.grad = None
            x.grad = dtok.grad = None
            out = m(x, dtok, pre, coords, H, W, pos)
            (out * torch.linspace(-1, 1, out.numel()).view_as(out)).sum().backward()
            return out.detach().clone(), x.grad.clone(), dtok.grad.clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        finally:
            PyramidLayer.FORCE_PAD = False

    o0, gx0, gd0, gp0 = run(False)
    o1, gx1, gd1, gp1 = run(True)
    assert rel(o1, o0) < 1e-5 and rel(gx1, gx0) < 1e-4 and rel(gd1, gd0) < 1e-4
    assert set(gp0) == set(gp1)
    for n in gp0:
        assert gp1[n].shape == gp0[n].shape and rel(gp1[n], gp0[n]) < 1e-4, n
