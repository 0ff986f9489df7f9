"""GPU parity proper: one full fp32 train step of the product, through the C ABI on the MI355X, against
the golden vectors of the real reference (outputs, 17 loss terms, per-parameter gradient norms, global
norm, AdamW update).  north_star tolerance: 1e-3 relative fp32; index outputs bit-exact."""
import numpy as np
import pytest
import torch

from gw_depth_amd import hip
from tests.golden_check import build, check_train_step, rel, to_device

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def real_library():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    hip.set_library(None)
    assert not getattr(hip.library(), "is_fake", False)
    yield


@pytest.mark.parametrize("case", ["tiny_b2_96x128", "ragged_b2_96x128", "mid_b1_224x288", "plane_b1_96x128"])
def test_fp32_train_step_matches_reference(golden_dir, case):
    check_train_step(case, golden_dir, "cuda", tol=1e-3, grad_tol=5e-3)


def test_bf16_step_runs_and_stays_close(golden_dir):
    """bf16 storage / fp32 accumulate is the bench mode; index ops make end-to-end equality meaningless
    (BASELINE.md §3), so check the parts in front of the first index op and that the step is finite."""
    import os
    from gw_depth_amd.engine import TrainStep
    from gw_depth_amd.synth import synth_batch
    from oracle.make_golden import CASES
    g = np.load(os.path.join(golden_dir, "tiny_b2_96x128.npz"))
    c = CASES["tiny_b2_96x128"]
    cfg, model, crits = build(device="cuda")
    b = to_device(synth_batch(c["batch"], c["height"], c["width"], seed=c["seed"], n_lines=c["n_lines"]), "cuda")
    step = TrainStep(model, crits, cfg, compute_dtype=torch.bfloat16)
    out, total, terms = step(b)
    assert torch.isfinite(total)
    assert rel(out["pred_lines"].detach(), g["pred_lines"]) < 5e-2
    assert rel(out["pred_logits"].detach(), g["pred_logits"]) < 1e-1
    assert abs(float(total) - float(g["stat/loss"])) / float(g["stat/loss"]) < 0.1


def test_index_ops_bit_exact_on_identical_operands():
    """top-k line selection and CertainSample are integer-valued: given bit-identical operands the GPU
    path must return exactly the oracle's indices / coordinates (SURVEY.md §8a a12, §7 'index-op chaos')."""
    from gw_depth_amd.model import certain_sample
    from oracle import gwdepth_ref as R
    g = torch.Generator().manual_seed(3)
    for (hs, ws, hl, wl, S) in ((15, 20, 30, 40, 30), (30, 40, 60, 80, 80), (6, 8, 12, 16, 30), (3, 4, 6, 8, 30)):
        small = torch.zeros(4, 1, hs, ws)       # bilinear(0) == 0 on every device: variance = large^2, one exact multiply
        large = torch.rand(4, 1, hl, wl, generator=g) * 0.98 + 0.01
        large[1] = large[1] * 0.05            # everything in the lowest interval
        large[2] = 0.95 + large[2] * 0.04     # everything in the highest interval
        big = torch.rand(1, hl, wl, generator=g) * 0.5 + 1.2      # outside every interval ...
        keep = torch.rand(1, hl, wl, generator=g) < 0.06          # ... except ~6 % of the pixels: repeat/complement rules
        large[3] = torch.where(keep, large[3], big)
        if hs == 6:
            large[3] = big                    # no pixel in any interval: the global top-S branch
        ref = R.certain_sample(small, large, (0.1, 0.3, 0.5, 0.7, 0.9), S, 1e-4)
        got = certain_sample(small.cuda(), large.cuda(), (0.1, 0.3, 0.5, 0.7, 0.9), S, 1e-4)
        assert torch.equal(got.cpu(), ref), (hs, ws)
    logits = torch.randn(8, 100, 2, generator=g)
    assert torch.equal(torch.topk(logits[:, :, 0].cuda(), 20, dim=-1)[1].cpu(), torch.topk(logits[:, :, 0], 20, dim=-1)[1])


def test_hip_graph_step_equals_eager_step():
    """The captured step (zero_grad + forward + losses + backward in one HIP graph, device LSAP / CertainSample, no host
    sync) must reproduce the eager step: same loss terms, same flat gradient, same parameters after AdamW.  Step 1 is
    compared tightly; steps 2-3 (replays on memory the previous replay left behind) loosely, because AdamW's first
    updates are ~lr*sign(g) and amplify atomic-order noise in near-zero gradients."""
    from gw_depth_amd.engine import TrainStep
    from gw_depth_amd.synth import synth_batch
    b = to_device(synth_batch(2, 96, 128, seed=41, n_lines=[4, 6]), "cuda")
    res = []
    for graph in (False, True):
        cfg, model, crits = build(device="cuda")
        step = TrainStep(model, crits, cfg, compute_dtype=torch.float32, graph=graph)
        snaps = []
        for _ in range(3):
            out, total, terms = step(b)
            torch.cuda.synchronize()
            snaps.append((float(total), {k: float(v) for k, v in terms.items()}, step.flat_g.clone(), step.flat_p.clone(),
                          out["pred_depth"][-1].clone()))
        if graph:
            assert all(e["graph"] is not None for e in step._graphs.values()), "capture was refused"
        res.append(snaps)
    for i, tol in ((0, 2e-5), (2, 1e-3)):
        (l0, t0, g0, p0, d0), (l1, t1, g1, p1, d1) = res[0][i], res[1][i]
        assert abs(l0 - l1) <= tol * abs(l0), i
        for k in t0:
            assert abs(t0[k] - t1[k]) <= tol * max(1.0, abs(t0[k])), (i, k)
        assert rel(d1, d0) < tol
        assert rel(g1, g0) < 50 * tol and rel(p1, p0) < 1e-4       # AdamW: lr * sign(noise-level gradient)


def test_device_matcher_step_equals_host_matcher_step():
    """Default eager step (device LSAP, no host sync in the criterion) == the same step with the host matcher."""
    from gw_depth_amd.engine import TrainStep
    from gw_depth_amd.synth import synth_batch
    b = to_device(synth_batch(2, 96, 128, seed=43, n_lines=[6, 3]), "cuda")
    res = []
    for dev_match in (False, True):
        cfg, model, crits = build(device="cuda")
        step = TrainStep(model, crits, cfg, compute_dtype=torch.float32)
        step.device_matcher = dev_match
        out, total, terms = step(b)
        torch.cuda.synchronize()
        res.append((float(total), {k: float(v) for k, v in terms.items()}, step.flat_p.clone()))
    (l0, t0, p0), (l1, t1, p1) = res
    assert abs(l0 - l1) <= 2e-5 * abs(l0)
    for k in t0:
        assert abs(t0[k] - t1[k]) <= 2e-5 * max(1.0, abs(t0[k])), k
    assert rel(p1, p0) < 1e-6


def test_graph_capture_refused_when_a_memset_is_seen(monkeypatch):
    """The capture audit: a step that issues hipMemsetAsync (nodes that do not replay faithfully) must NOT be captured;
    the step then runs eagerly - same numbers, with a warning."""
    import warnings
    from gw_depth_amd.engine import TrainStep
    from gw_depth_amd.synth import synth_batch
    b = to_device(synth_batch(2, 96, 128, seed=45, n_lines=[3, 4]), "cuda")
    cfg, model, crits = build(device="cuda")
    ref = TrainStep(model, crits, cfg, compute_dtype=torch.float32)
    _, total_ref, _ = ref(b)
    cfg2, model2, crits2 = build(device="cuda")
    step = TrainStep(model2, crits2, cfg2, compute_dtype=torch.float32, graph=True)
    monkeypatch.setattr(step, "_count_memsets", lambda st: (step._sync_free_fb(st), 3)[1])
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        _, total, _ = step(b)
        torch.cuda.synchronize()
    assert any("capture refused" in str(x.message) for x in w)
    assert all(e["graph"] is None for e in step._graphs.values())
    assert abs(float(total) - float(total_ref)) <= 2e-5 * abs(float(total_ref))


def test_eval_forward_480x640_matches_oracle():
    """BASELINE config C1: eval-mode forward of ONE 480x640 image (fp32, no_grad) through the HIP kernels vs the oracle
    (the CPU restatement pinned by the reference's golden vectors).  Near-ties of the top-k / CertainSample index ops
    are teacher-forced from the oracle's taps so that both sides gather identical points."""
    from gw_depth_amd.synth import synth_batch
    from oracle import gwdepth_ref as R
    cfg, model, crits = build(device="cuda")
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    b = synth_batch(1, 480, 640, seed=51, n_lines=[7])
    otaps = {}
    with torch.no_grad():
        ref = R.forward(sd, b["images"], b["pad_mask"], R.Cfg(dropout=0.1, log_depth_error=True), training=False, taps=otaps)
    model.eval()
    taps = {"force_points1": otaps["points1"].cuda(), "force_points2": otaps["points2"].cuda()}
    with torch.no_grad():
        from gw_depth_amd.model import NestedTensor
        out = model(NestedTensor(b["images"].cuda(), b["pad_mask"].cuda()), taps=taps)
    torch.cuda.synchronize()
    assert torch.equal(taps["topk_ids"].cpu(), otaps["topk_ids"])
    assert out["pred_depth"][-1].shape == (1, 1, 480, 640) and out["pred_seg"].shape == (1, 2, 480, 640)
    assert rel(out["pred_logits"], ref["pred_logits"]) < 1e-3 and rel(out["pred_lines"], ref["pred_lines"]) < 1e-3
    for a, r in zip(out["pred_depth"], ref["pred_depth"]):
        assert rel(a, r) < 1e-3
    assert rel(out["pred_seg"], ref["pred_seg"]) < 1e-3


def test_inference_960x1280_bf16_batch_invariance():
    """BASELINE config C5 (inference at 960x1280, half precision), through a size-independent property: the two copies
    of one image in a batch of 2 must produce bit-identical outputs, all finite, with the documented shapes."""
    from gw_depth_amd.model import NestedTensor
    from gw_depth_amd.synth import synth_batch
    cfg, model, crits = build(device="cuda")
    model.compute_dtype = torch.bfloat16
    model.eval()
    b = synth_batch(1, 960, 1280, seed=52, n_lines=[7])
    img = b["images"].cuda().repeat(2, 1, 1, 1)
    msk = b["pad_mask"].cuda().repeat(2, 1, 1)
    with torch.no_grad():
        out = model(NestedTensor(img, msk))
    torch.cuda.synchronize()
    assert out["pred_depth"][-1].shape == (2, 1, 960, 1280) and out["pred_seg"].shape == (2, 2, 960, 1280)
    assert [tuple(d.shape[-2:]) for d in out["pred_depth"]] == [(60, 80), (120, 160), (240, 320), (960, 1280)]
    for k in ("pred_logits", "pred_lines", "pred_seg"):
        assert torch.isfinite(out[k].float()).all() and torch.equal(out[k][0], out[k][1]), k
    for d in out["pred_depth"]:
        assert torch.isfinite(d.float()).all() and torch.equal(d[0], d[1])
    assert float(out["pred_depth"][-1].min()) >= 0.0 and float(out["pred_depth"][-1].max()) <= 10.0


def test_graph_mode_reports_non_finite_loss_one_step_late():
    """engine_glassrgbd.py:150-153 stops on a non-finite loss.  Graph mode reads the loss asynchronously (no device idle
    time): the step that produced it returns, the NEXT step - or flush() - raises."""
    from gw_depth_amd.engine import TrainStep
    from gw_depth_amd.synth import synth_batch
    b = to_device(synth_batch(1, 96, 128, seed=47, n_lines=[3]), "cuda")
    cfg, model, crits = build(device="cuda")
    step = TrainStep(model, crits, cfg, compute_dtype=torch.float32, graph=True)
    step(b)
    step.flush()
    step.flat_p[:64].fill_(float("nan"))
    step(b)                                      # produces the NaN loss, does not raise yet
    with pytest.raises(FloatingPointError):
        step.flush()
    eager = TrainStep(*((lambda c: (c[1], c[2], c[0]))(build(device="cuda"))), compute_dtype=torch.float32)
    eager.flat_p[:64].fill_(float("nan"))
    with pytest.raises(FloatingPointError):      # eager mode: immediately, as the reference
        eager(b)


def test_evaluate_end_to_end_matches_oracle():
    """§8f-1 end to end on the GPU: evaluate() (eval-mode HIP forward, fp32, batches of 2 with ragged sizes, metrics by
    gwd_eval_accumulate) against the oracle forward + oracle/eval_ref.py on the same weights and inputs - depth RMSE and
    the other eight measures within 1e-3 relative (BASELINE's bar), confusion-derived scores within 0.05 points (a
    handful of pixels whose two logits tie to 1e-6 may flip)."""
    import numpy as np
    from gw_depth_amd.evaluate import METRIC_NAMES, evaluate
    from gw_depth_amd.model import NestedTensor
    from gw_depth_amd.synth import synth_batch
    from oracle import eval_ref
    from oracle import gwdepth_ref as R
    cfg, model, crits = build(device="cuda")
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    batches = [synth_batch(2, 96, 128, seed=61, n_lines=[3, 4], sizes=[(96, 128), (80, 104)]),
               synth_batch(2, 96, 128, seed=62, n_lines=[2, 5])]
    per_image, conf = [], np.zeros((2, 2))
    loader = []
    for b in batches:
        with torch.no_grad():
            ref = R.forward(sd, b["images"], b["pad_mask"], R.Cfg(dropout=0.0, log_depth_error=True), training=False)
        g = b["depth"].clone()
        s = b["seg"].clone()
        pad = b["pad_mask"].unsqueeze(1)
        g[pad] = 0.0                                                      # padding is not evaluated
        s[pad] = 255
        rec, _ = eval_ref.evaluate_dense(ref["pred_depth"][-1].numpy(), g.numpy(), ref["pred_seg"].numpy(), s.numpy())
        per_image.append(rec)
        for i in range(g.shape[0]):
            conf += eval_ref.confusion_matrix(s[i, 0].numpy(), ref["pred_seg"][i].argmax(0).numpy())
        loader.append((NestedTensor(b["images"], b["pad_mask"]), NestedTensor(b["depth"], b["pad_mask"]),
                       NestedTensor(b["seg"], b["pad_mask"]), b["targets"], ["synthetic\n"]))
    want = eval_ref.seg_scores(conf)
    want.update({k: float(v) for k, v in zip(METRIC_NAMES, np.concatenate(per_image).mean(0))})
    args = type("A", (), {"with_line": False, "with_dense": True, "min_depth_eval": 1e-3, "max_depth_eval": 10.0})()
    stats = evaluate(model, crits, None, loader, None, "cuda", None, args)
    for k in METRIC_NAMES:
        assert abs(stats[k] - want[k]) <= 1e-3 * max(abs(want[k]), 1e-6), (k, stats[k], want[k])
    for k in ("Background", "Glass", "Pixel accuracy", "Mean accuracy", "Mean IU"):
        assert abs(stats[k] - want[k]) <= 0.05, (k, stats[k], want[k])


def test_flat_adamw_on_device_matches_torch_adamw():
    """§8f-3 on the GPU: FlatAdamW.step (fused device clip + AdamW) under a StepLR == torch.optim.AdamW +
    clip_grad_norm_ on the same gradients, two epochs with an LR drop in between."""
    from gw_depth_amd.checkpoint import FlatAdamW
    from gw_depth_amd.engine import TrainStep
    cfg, model, crits = build(device="cuda")
    step = TrainStep(model, crits, cfg, compute_dtype=torch.float32)
    opt = FlatAdamW(step)
    sched = torch.optim.lr_scheduler.StepLR(opt, 1)
    named = [(n, torch.nn.Parameter(p.detach().clone())) for n, p in model.named_parameters() if p.requires_grad]
    ref = torch.optim.AdamW([{"params": [p for n, p in named if "backbone" not in n]},
                             {"params": [p for n, p in named if "backbone" in n], "lr": cfg.lr_backbone}],
                            lr=cfg.lr, weight_decay=cfg.weight_decay)
    ref_sched = torch.optim.lr_scheduler.StepLR(ref, 1)
    gen = torch.Generator(device="cuda").manual_seed(3)
    for epoch in range(2):
        step.flat_g.zero_()
        for n, p in named:
            g = torch.randn(p.shape, generator=gen, device="cuda") * 1e-3
            p.grad = g
            o = step.offsets[n]
            step.flat_g[o:o + p.numel()].copy_(g.reshape(-1))
        torch.nn.utils.clip_grad_norm_([p for _, p in named], cfg.clip_max_norm)
        ref.step()
        ref_sched.step()
        opt.step()
        sched.step()
    torch.cuda.synchronize()
    assert opt.lrs() == tuple(g["lr"] for g in ref.param_groups)
    worst = max(float((model.get_parameter(n).detach() - p.detach()).abs().max() / (p.detach().abs().max() + 1e-12)) for n, p in named)
    assert worst < 1e-6, worst


# (test_bf16_attention_kernels_match_the_unfused_path lived here until round 3: it compared the matrix-core attention kernels with the
# library's OWN lane-per-row kernels through two environment switches.  tests/test_bf16_pinning.py compares the whole bf16 step - those
# kernels included - with the fp32 parity mode, which runs different code (VALU window attention, gwd_bmm + softmax for the DETR
# attention) and is itself pinned to the reference's fixtures; the switches are gone.)


def test_hip_graph_step_with_plane_loss_equals_eager_step(golden_dir):
    """--with_plane_norm_loss inside the captured step (the triangle count stays a device scalar, no host sync): the
    graph is captured, its logged loss_plane equals the eager one, and step 1 gives the reference's number (golden case
    plane_b1_96x128)."""
    import numpy as np
    from gw_depth_amd.engine import TrainStep
    from gw_depth_amd.synth import synth_batch
    from oracle.make_golden import CASES
    c = CASES["plane_b1_96x128"]
    g = np.load(golden_dir + "/plane_b1_96x128.npz")
    b = to_device(synth_batch(c["batch"], c["height"], c["width"], seed=c["seed"], n_lines=c["n_lines"], sizes=c["sizes"]), "cuda")
    vals = []
    for graph in (False, True):
        cfg, model, crits = build(device="cuda", case=c)
        step = TrainStep(model, crits, cfg, compute_dtype=torch.float32, graph=graph)
        per_step = []
        for _ in range(2):
            out, total, terms = step(b)
            torch.cuda.synchronize()
            per_step.append((float(total), float(terms["loss_plane"])))
        if graph:
            assert all(e["graph"] is not None for e in step._graphs.values()), "capture was refused"
        vals.append(per_step)
    for (l0, p0), (l1, p1) in zip(vals[0], vals[1]):
        assert abs(p0 - p1) <= 1e-3 * max(1.0, abs(p0)) and abs(l0 - l1) <= 1e-3 * abs(l0)
    assert abs(vals[1][0][1] - float(g["stat/loss_plane"])) <= 1e-3 * float(g["stat/loss_plane"])
