"""BASELINE.json's configurations at their FULL sizes on the MI355X ("configs untested").

C2  fwd+bwd 8x480x640 bf16 in HIP-graph mode (the bench line's configuration): the fp32 mode against the oracle at that
    size (every output and the 17 loss terms, north_star's 1e-3), the bf16 graph step finite, captured, and close to it.
C4  the same model at 16 images per GPU.
C5  inference at 960x1280 at batch 32, half precision (bf16 storage - DESIGN.md records why not fp16): finite, documented
    shapes, and every image of the batch equal to what a batch-1 run of that image gives (no cross-image leakage, no
    index overflow at 39 M pixels per map).
depth RMSE: the fp32 mode within north_star's 1e-3 of the reference's, the bf16 (timed) mode within its stated bound, and
the measurement that shows 1e-3 to be out of reach of ANY bf16 weight storage on this sample.

The oracle (oracle/gwdepth_ref.py, pinned by the reference's golden vectors) is the checker only.
"""
import numpy as np
import pytest
import torch

from gw_depth_amd import hip
from tests.golden_check import build, rel, to_device

pytestmark = pytest.mark.gpu
LINE_TERMS = ("loss_ce", "loss_line")


@pytest.fixture(autouse=True)
def real_library():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    hip.set_library(None)
    torch.set_num_threads(16)
    yield


def oracle_forward_and_losses(sd, b, losses=True):
    from oracle import gwdepth_ref as R
    cfg = R.Cfg(dropout=0.0, log_depth_error=True)
    taps = {}
    with torch.no_grad():
        out = R.forward(sd, b["images"], b["pad_mask"], cfg, training=True, taps=taps)
        terms = R.step_losses(out, b["depth"], b["seg"], b["targets"], cfg)[1] if losses else None
    return out, terms, taps


def check_outputs(out, ref, tol):
    assert rel(out["pred_logits"].detach(), ref["pred_logits"]) < tol and rel(out["pred_lines"].detach(), ref["pred_lines"]) < tol
    for a, r in zip(out["aux_outputs"], ref["aux_outputs"]):
        assert rel(a["pred_logits"].detach(), r["pred_logits"]) < tol and rel(a["pred_lines"].detach(), r["pred_lines"]) < tol
    for i, (a, r) in enumerate(zip(out["pred_depth"], ref["pred_depth"])):
        assert a.shape == r.shape and rel(a.detach(), r) < tol, i
    assert out["pred_seg"].shape == ref["pred_seg"].shape and rel(out["pred_seg"].detach(), ref["pred_seg"]) < tol


def run_config(batch_size, seed):
    from gw_depth_amd.engine import TrainStep
    from gw_depth_amd.synth import synth_batch
    b_cpu = synth_batch(batch_size, 480, 640, seed=seed)
    b = to_device(b_cpu, "cuda")
    cfg, model, crits = build(device="cuda")
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    ref, ref_terms, otaps = oracle_forward_and_losses(sd, b_cpu)

    # fp32 parity mode, eager, sample points teacher-forced from the oracle (identical index operands downstream)
    step = TrainStep(model, crits, cfg, compute_dtype=torch.float32)
    taps = {"force_points1": otaps["points1"].cuda(), "force_points2": otaps["points2"].cuda(), "force_topk_ids": otaps["topk_ids"].cuda()}
    out, total, terms = step(b, taps=taps)
    torch.cuda.synchronize()
    # the product's own top-k of the line logits: identical to the oracle's except where two logits tie to float noise
    own, want, lg = taps["own_topk_ids"].cpu(), otaps["topk_ids"], ref["pred_logits"][:, :, 0]
    for bi, pos in (own != want).nonzero().tolist():
        a, r = float(lg[bi, own[bi, pos]]), float(lg[bi, want[bi, pos]])
        assert abs(a - r) <= 2e-5 * max(1.0, abs(r)), (bi, pos, a, r)
    check_outputs(out, ref, 1e-3)
    for k, v in terms.items():
        want = float(ref_terms[k])
        tol = 2e-3 if k.startswith(LINE_TERMS) else 1e-3      # an assignment may flip between cost-equal matches (<= 1e-4 of the cost)
        assert abs(float(v) - want) <= tol * max(1.0, abs(want)), (k, float(v), want)
    fp32_total = float(total)
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.requires_grad)
    del step, out, terms

    # the timed mode: bf16 storage, HIP-graph launch, three steps (two replays)
    cfg2, model2, crits2 = build(device="cuda")
    gstep = TrainStep(model2, crits2, cfg2, compute_dtype=torch.bfloat16, graph=True)
    losses, first = [], None
    for _ in range(3):
        gout, gtotal, gterms = gstep(b)
        losses.append(float(gtotal))
        if first is None:                                                   # static output tensors: the next replay overwrites them
            first = {"pred_lines": gout["pred_lines"].float().clone(), "depth": gout["pred_depth"][-1].float().clone()}
    gstep.flush()
    torch.cuda.synchronize()
    assert gstep._graphs and all(e["graph"] is not None for e in gstep._graphs.values()), "capture was refused"
    assert all(l == l and abs(l) < 1e6 for l in losses)
    assert abs(losses[0] - fp32_total) <= 0.03 * abs(fp32_total), (losses, fp32_total)
    assert losses[2] < losses[0]                                            # AdamW on a fixed batch: the loss goes down
    # tensors in front of the first index op (top-k of the line logits): bf16 storage through 6 + 6 transformer layers
    assert rel(first["pred_lines"], ref["pred_lines"]) < 3e-2 and torch.isfinite(first["depth"]).all()
    assert rel(first["depth"], ref["pred_depth"][-1]) < 1e-1
    assert torch.isfinite(gstep.flat_p).all()


def test_c2_train_step_8x480x640_fp32_vs_oracle_and_bf16_graph():
    run_config(8, seed=1)


def test_c4_train_step_16x480x640_fp32_vs_oracle_and_bf16_graph():
    run_config(16, seed=2)


def test_c5_inference_960x1280_batch32_bf16():
    from gw_depth_amd.model import NestedTensor
    from gw_depth_amd.synth import synth_batch
    cfg, model, crits = build(device="cuda")
    model.compute_dtype = torch.bfloat16
    model.eval()
    B = 32
    b = synth_batch(4, 960, 1280, seed=53)
    img = b["images"].cuda().repeat(B // 4, 1, 1, 1)                 # images 0..3 repeated: image i == image i % 4
    msk = b["pad_mask"].cuda().repeat(B // 4, 1, 1)
    with torch.no_grad():
        out = model(NestedTensor(img, msk))
        one = model(NestedTensor(img[B - 1:], msk[B - 1:]))           # the LAST image alone: highest addresses of the batch run
    torch.cuda.synchronize()
    assert out["pred_depth"][-1].shape == (B, 1, 960, 1280) and out["pred_seg"].shape == (B, 2, 960, 1280)
    assert [tuple(d.shape[-2:]) for d in out["pred_depth"]] == [(60, 80), (120, 160), (240, 320), (960, 1280)]
    for k in ("pred_logits", "pred_lines"):
        assert torch.isfinite(out[k].float()).all()
        for i in range(4, B):
            assert torch.equal(out[k][i], out[k][i % 4]), (k, i)
    for d in out["pred_depth"] + [out["pred_seg"]]:
        assert torch.isfinite(d.float()).all()
        for i in range(4, B):
            assert torch.equal(d[i], d[i % 4]), i
    assert float(out["pred_depth"][-1].min()) >= 0.0 and float(out["pred_depth"][-1].max()) <= 10.0
    # tile selection depends on the batch (M), so batch-1 vs batch-32 agree to rounding, not bit for bit
    assert rel(one["pred_depth"][-1].float(), out["pred_depth"][-1][B - 1:].float()) < 2e-2
    assert rel(one["pred_seg"].float(), out["pred_seg"][B - 1:].float()) < 5e-2


def test_depth_rmse_fp32_mode_within_1e3_and_bf16_mode_within_its_stated_bound():
    """north_star: "depth RMSE within 1e-3 of reference".  Same sample as bench.py's depth_rmse leg (one 480x640 image, weight
    seed 0, data seed 1, eval mode), `rms` of evaluate() (src/util/metrics.py:203-204).

    fp32 (parity) mode: |RMSE - reference| <= 1e-3 (observed 1e-6).
    bf16 (timed) mode: the stated bound is 3e-2 absolute = 1 % of the RMSE (observed 1.6e-2), and the test shows WHY 1e-3 is not a
    property any bf16 mode can have on this sample: the reference's own fp32 arithmetic (the CPU oracle) with nothing but the weight
    matrices rounded to bf16 already moves the RMSE by more than 1e-3 (observed 9e-3; the synthetic ground truth is independent
    of the prediction, so the RMSE follows the MEAN of the predicted depth with slope 0.32, and weight rounding is coherent over
    all pixels - profiles/r02_bf16_precision_experiment.txt has the stage-by-stage breakdown)."""
    from gw_depth_amd import Config, build_model
    from gw_depth_amd.evaluate import DenseMetrics
    from gw_depth_amd.model import NestedTensor
    from gw_depth_amd.synth import det_fill_, synth_batch
    from oracle import eval_ref
    from oracle import gwdepth_ref as R
    cfg = Config(device="cuda", dropout=0.1, log_depth_error=True)
    model, _, _ = build_model(cfg)
    sd = det_fill_({k: v.detach().clone() for k, v in model.state_dict().items()}, seed=0)
    model.load_state_dict(sd)
    model.cuda().eval()
    b = synth_batch(1, 480, 640, seed=1)
    ocfg = R.Cfg(dropout=0.1, log_depth_error=True)

    def oracle_rms(weights):
        with torch.no_grad():
            ref = R.forward(weights, b["images"], b["pad_mask"], ocfg, training=False)
        per_image, _ = eval_ref.evaluate_dense(ref["pred_depth"][-1].numpy(), b["depth"].numpy(), ref["pred_seg"].numpy(), b["seg"].numpy())
        return float(per_image[0, 3])

    want = oracle_rms({k: v.clone() for k, v in sd.items()})
    floor = oracle_rms({k: (v.bfloat16().float() if (v.is_floating_point() and v.dim() >= 2) else v.clone()) for k, v in sd.items()})
    got = {}
    for dt in (torch.float32, torch.bfloat16):
        model.compute_dtype = dt
        with torch.no_grad():
            o = model(NestedTensor(b["images"].cuda(), b["pad_mask"].cuda()))
        dm = DenseMetrics("cuda")
        dm.update(o["pred_depth"][-1], b["depth"].cuda(), o["pred_seg"], b["seg"].cuda())
        got[dt] = dm.compute()["rms"]
    assert abs(got[torch.float32] - want) <= 1e-3, (got, want)
    assert abs(got[torch.bfloat16] - want) <= 3e-2, (got, want)
    assert abs(floor - want) > 1e-3, (floor, want)          # bf16 weights + exact fp32 arithmetic: already outside 1e-3
    assert abs(got[torch.bfloat16] - want) <= 4 * abs(floor - want), (got, floor, want)   # the kernels add no more than that floor's order


RMSE_PAIRS = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 8)]          # (weight seed, data seed)


def rmse_distribution(pairs=RMSE_PAIRS, height=480, width=640):
    """|RMSE - reference| over several (weight seed, data seed) pairs: product fp32, product bf16 and the FLOOR of any bf16 mode -
    the reference's own fp32 arithmetic (CPU oracle) with nothing but the weight matrices rounded to bf16.  One image each, eval
    mode, `rms` of evaluate() (src/util/metrics.py:203-204).  Returns a list of dicts (also used by tools/rmse_distribution.py)."""
    from gw_depth_amd import Config, build_model
    from gw_depth_amd.evaluate import DenseMetrics
    from gw_depth_amd.model import NestedTensor
    from gw_depth_amd.synth import det_fill_, synth_batch
    from oracle import eval_ref
    from oracle import gwdepth_ref as R
    cfg = Config(device="cuda", dropout=0.1, log_depth_error=True)
    ocfg = R.Cfg(dropout=0.1, log_depth_error=True)
    model, _, _ = build_model(cfg)
    shapes = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.cuda().eval()
    rows = []
    for ws, ds in pairs:
        sd = det_fill_({k: v.clone() for k, v in shapes.items()}, seed=ws)
        model.load_state_dict(sd)
        b = synth_batch(1, height, width, seed=ds)

        def oracle_rms(weights):
            with torch.no_grad():
                ref = R.forward(weights, b["images"], b["pad_mask"], ocfg, training=False)
            per_image, _ = eval_ref.evaluate_dense(ref["pred_depth"][-1].numpy(), b["depth"].numpy(), ref["pred_seg"].numpy(), b["seg"].numpy())
            return float(per_image[0, 3])

        want = oracle_rms({k: v.clone() for k, v in sd.items()})
        floor = oracle_rms({k: (v.bfloat16().float() if (v.is_floating_point() and v.dim() >= 2) else v.clone()) for k, v in sd.items()})
        row = {"weight_seed": ws, "data_seed": ds, "oracle_rms": want, "floor": abs(floor - want)}
        for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
            model.compute_dtype = dt
            with torch.no_grad():
                o = model(NestedTensor(b["images"].cuda(), b["pad_mask"].cuda()))
            dm = DenseMetrics("cuda")
            dm.update(o["pred_depth"][-1], b["depth"].cuda(), o["pred_seg"], b["seg"].cuda())
            row[name] = abs(dm.compute()["rms"] - want)
        rows.append(row)
    return rows


def test_depth_rmse_distribution_over_eight_seed_pairs():
    """The bf16 bound as a DISTRIBUTION (one sample proves nothing about a random walk).  Per pair: the fp32
    parity mode meets north_star's 1e-3; the bf16 mode stays inside the stated absolute bound 3e-2.  Over the pairs: the bf16 mode's
    mean and maximum deviation are within 4x the mean / maximum of the floor that bf16 STORAGE OF THE WEIGHTS alone imposes on the
    reference's own fp32 arithmetic - i.e. the kernels add no more than the order of what the storage format costs anyway."""
    rows = rmse_distribution()
    for r in rows:
        print("weights %d data %d: oracle rms %.6f  |fp32| %.2e  |bf16| %.2e  floor %.2e" %
              (r["weight_seed"], r["data_seed"], r["oracle_rms"], r["fp32"], r["bf16"], r["floor"]))
        assert r["fp32"] <= 1e-3, r
        assert r["bf16"] <= 3e-2, r
    bf = np.array([r["bf16"] for r in rows])
    fl = np.array([r["floor"] for r in rows])
    print("bf16: mean %.2e max %.2e; floor: mean %.2e max %.2e" % (bf.mean(), bf.max(), fl.mean(), fl.max()))
    assert bf.mean() <= 4 * fl.mean() and bf.max() <= 4 * fl.max(), (bf, fl)
