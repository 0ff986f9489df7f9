"""Checkpoint interchange: the key remaps of main_glassrgbd.py:104-157, and the optimizer state moving
both ways between TrainStep's flat buffers and a REAL torch.optim.AdamW built the way main_glassrgbd.py:59-66 builds it.
CPU only: the device library is tests/fake_device.py (host logic under test, not kernels)."""
import os

import pytest
import torch

from gw_depth_amd import hip
from gw_depth_amd.checkpoint import (FlatAdamW, filter_detr_state_dict, load_checkpoint, new_parameters,
                                     remap_resume_state_dict, save_checkpoint)
from gw_depth_amd.engine import TrainStep
from tests.fake_device import FakeDevice
from tests.golden_check import build


@pytest.fixture()
def fake():
    hip.set_library(FakeDevice())
    yield
    hip.set_library(None)


def test_resume_key_remap_follows_the_reference_quirks():
    t = torch.zeros(1)
    sd = {"module.backbone.0.body.conv1.weight": t, "transformer.encoder.layers.0.norm1.weight": t,
          "bbox_embed.layers.0.weight": t, "module.bbox_embed.layers.1.bias": t, "modulex.foo": t}
    msgs = []
    out = remap_resume_state_dict(sd, lambda *a: msgs.append(a))
    assert set(out) == {"backbone.0.body.conv1.weight", "transformer.encoder.layers.0.norm1.weight",
                        "lines_embed.layers.0.weight",
                        "lines_embed.bbox_embed.layers.1.bias",      # :139 splits the ORIGINAL key: a DataParallel-era bbox_embed never loads
                        ".foo"}                                      # re.compile('module.'): the dot matches ANY character (:131), so "modulex" goes
    assert len(msgs) == 2


def test_detr_r50_partial_load_filter():
    t = torch.zeros(1)
    sd = {"class_embed.weight": t, "bbox_embed.layers.0.weight": t, "query_embed.weight": t, "input_proj.weight": t,
          "backbone.0.body.layer1.0.conv1.weight": t, "transformer.decoder.norm.weight": t}
    assert set(filter_detr_state_dict(sd, layer1_num=3)) == {"input_proj.weight", "backbone.0.body.layer1.0.conv1.weight",
                                                             "transformer.decoder.norm.weight"}
    assert "input_proj.weight" not in filter_detr_state_dict(sd, layer1_num=4)


def _fake_grads(step, seed):
    g = torch.Generator().manual_seed(seed)
    r = torch.randn(step.flat_g.shape, generator=g) * 1e-3
    step.flat_g.zero_()                                                  # alignment padding between parameters stays zero, as in a real step
    for n, p in step.params.items():
        o = step.offsets[n]
        step.flat_g[o:o + p.numel()].copy_(r[o:o + p.numel()])


def reference_parameter_order():
    """named_parameters() order of the REFERENCE model: its state-dict key order (tests/golden/state_dict_spec.json, dumped from
    the reference's build_model) restricted to parameters (tests/golden/param_sets.json)."""
    import json
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = json.load(open(os.path.join(here, "state_dict_spec.json")))
    sets = json.load(open(os.path.join(here, "param_sets.json")))
    params, trainable = set(sets["parameters"]), set(sets["trainable"])
    return [k for k in spec if k in params], [k for k in spec if k in trainable]


def test_parameter_registration_order_is_the_references():
    """torch.optim state dicts are positional: optimizer checkpoints interchange only if named_parameters() enumerates exactly
    as the reference's model does (border_* before relative_position_bias_table, get_depth before the *_seg modules)."""
    _, model, _ = build()
    every, trainable = reference_parameter_order()
    assert [n for n, _ in model.named_parameters()] == every
    assert [n for n, p in model.named_parameters() if p.requires_grad] == trainable
    import json
    spec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "state_dict_spec.json")))
    assert list(model.state_dict().keys()) == list(spec.keys())


def _reference_adamw(model, cfg):
    """A real torch.optim.AdamW over reference-layout copies of the parameters, grouped as main_glassrgbd.py:59-66 and
    enumerated in the REFERENCE's parameter order (not the product model's own)."""
    sd = model.state_dict()                                              # reference layout (Cout,Cin,KH,KW)
    named = [(n, torch.nn.Parameter(sd[n].detach().clone())) for n in reference_parameter_order()[1]]
    groups = [{"params": [p for n, p in named if "backbone" not in n]},
              {"params": [p for n, p in named if "backbone" in n], "lr": cfg.lr_backbone}]
    return named, torch.optim.AdamW(groups, lr=cfg.lr, weight_decay=cfg.weight_decay)


def _to_reference_layout(model, name, flat_view):
    from gw_depth_amd.layers import Conv
    native = {id(m.weight) for m in model.modules() if isinstance(m, Conv)}
    p = model.get_parameter(name)
    t = flat_view.view(p.shape)
    return t.permute(0, 3, 1, 2).clone(memory_format=torch.contiguous_format) if id(p) in native else t.clone()   # never a view


def test_optimizer_state_roundtrip_with_a_real_adamw(fake, tmp_path):
    cfg, model, crits = build()
    step = TrainStep(model, crits, cfg, compute_dtype=torch.float32)
    opt = FlatAdamW(step)
    sched = torch.optim.lr_scheduler.StepLR(opt, 2)                      # main_glassrgbd.py:67 with --lr_drop 2
    for epoch in range(2):
        _fake_grads(step, 100 + epoch)
        opt.step()
        sched.step()
    assert step.step_count == 2 and opt.lrs() == pytest.approx((cfg.lr * 0.1, cfg.lr_backbone * 0.1))
    path = os.path.join(tmp_path, "checkpoint.pth")
    save_checkpoint(path, model, opt, sched, epoch=1, args=None)
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ckpt) == {"model", "optimizer", "lr_scheduler", "epoch", "args"}                     # :216-222
    assert set(ckpt["model"]) == set(model.state_dict())

    # (a) the reference side: a real AdamW + StepLR accept the file and continue from it
    named, ref_opt = _reference_adamw(model, cfg)
    ref_sched = torch.optim.lr_scheduler.StepLR(ref_opt, 2)
    ref_opt.load_state_dict(ckpt["optimizer"])
    ref_sched.load_state_dict(ckpt["lr_scheduler"])
    for n, p in named:
        st = ref_opt.state[p]
        assert st["exp_avg"].shape == p.shape and float(st["step"]) == 2.0, n
        assert torch.equal(st["exp_avg"], _to_reference_layout(model, n, step.flat_m[step.offsets[n]:step.offsets[n] + p.numel()])), n
    assert [g["lr"] for g in ref_opt.param_groups] == pytest.approx([cfg.lr * 0.1, cfg.lr_backbone * 0.1])

    # one more step on both sides from identical gradients: clip_grad_norm_(0.1) + AdamW (engine_glassrgbd.py:157-159)
    _fake_grads(step, 777)
    for n, p in named:
        p.grad = _to_reference_layout(model, n, step.flat_g[step.offsets[n]:step.offsets[n] + p.numel()])
    torch.nn.utils.clip_grad_norm_([p for _, p in named], cfg.clip_max_norm)
    ref_opt.step()
    opt.step()
    after = model.state_dict()
    worst = max(float((after[n] - p.detach()).abs().max() / (p.detach().abs().max() + 1e-12)) for n, p in named)
    assert worst < 1e-6, worst

    # (b) our side: a fresh model + TrainStep resumes from the file bit for bit
    cfg2, model2, crits2 = build()
    with torch.no_grad():
        for p in model2.parameters():
            p.add_(1.0)                                                  # make sure the load is what sets the values
    step2 = TrainStep(model2, crits2, cfg2, compute_dtype=torch.float32)
    opt2 = FlatAdamW(step2)
    sched2 = torch.optim.lr_scheduler.StepLR(opt2, 7)
    args = type("A", (), {"lr_drop": 2, "no_opt": False, "eval": False})()
    assert load_checkpoint(path, model2, opt2, sched2, args, log=None) == 2                         # :161 start_epoch = epoch + 1
    assert step2.step_count == 2 and sched2.step_size == 2 and opt2.lrs() == pytest.approx((cfg.lr * 0.1, cfg.lr_backbone * 0.1))
    _fake_grads(step2, 777)
    opt2.step()
    assert torch.equal(step2.flat_p, step.flat_p) and torch.equal(step2.flat_m, step.flat_m) and torch.equal(step2.flat_v, step.flat_v)

    # --eval / --no_opt: weights only (:157)
    cfg3, model3, crits3 = build()
    assert load_checkpoint(ckpt, model3, None, None, type("A", (), {"eval": True})(), log=None) is None
    assert all(torch.equal(a, b) for a, b in zip(model3.state_dict().values(), ckpt["model"].values()))
    assert new_parameters(model3, {}) == [n for n, _ in model3.named_parameters()]
