"""Pins oracle/gwdepth_ref.py (the CPU restatement) against the golden vectors that
oracle/make_golden.py produced by running the REAL reference.  CPU only."""
import os

import numpy as np
import pytest
import torch

from gw_depth_amd.synth import det_fill_, synth_batch
from oracle import gwdepth_ref as R
from oracle.make_golden import CASES, sha
from tests.helpers import reference_state_shapes

FP_TOL = 2e-4      # fp32 CPU vs fp32 CPU, different op grouping / thread counts (reference noise 4e-6)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + 1e-12))


@pytest.mark.parametrize("case", ["tiny_b2_96x128", "ragged_b2_96x128"])
def test_train_step_matches_reference(golden_dir, case):
    g = load(golden_dir, case)
    cfg_case = CASES[case]
    torch.manual_seed(0)
    sd = det_fill_(reference_state_shapes(), seed=0)
    b = synth_batch(cfg_case["batch"], cfg_case["height"], cfg_case["width"], seed=cfg_case["seed"],
                    n_lines=cfg_case["n_lines"], sizes=cfg_case["sizes"])
    assert [sha(b["images"]), sha(b["depth"]), sha(b["seg"])] == list(g["input_sha"][:3])
    cfg = R.Cfg()
    taps, opt = {}, {}
    before = {n: t.clone() for n, t in sd.items() if R.is_trainable(n)}
    out, total, terms, grads, gnorm = R.train_step(sd, b, cfg, opt_state=opt, step=1, taps=taps)

    # integer / index outputs: bit exact
    assert np.array_equal(taps["topk_ids"].numpy(), g["topk_ids"])
    assert np.array_equal(taps["points1"].numpy(), g["points1"])
    assert np.array_equal(taps["points2"].numpy(), g["points2"])
    for li, m in enumerate(taps["matches"]):
        for bi, (i, j) in enumerate(m):
            assert np.array_equal(i.numpy(), g[f"match{li}_b{bi}_src"])
            assert np.array_equal(j.numpy(), g[f"match{li}_b{bi}_tgt"])

    # every output tensor
    assert rel(out["pred_logits"].detach(), g["pred_logits"]) < FP_TOL
    assert rel(out["pred_lines"].detach(), g["pred_lines"]) < FP_TOL
    for i, a in enumerate(out["aux_outputs"]):
        assert rel(a["pred_logits"].detach(), g[f"aux{i}_pred_logits"]) < FP_TOL
        assert rel(a["pred_lines"].detach(), g[f"aux{i}_pred_lines"]) < FP_TOL
    for i, d in enumerate(out["pred_depth"]):
        assert rel(d.detach(), g[f"pred_depth{i}"]) < FP_TOL, i
    assert rel(out["pred_seg"].detach(), g["pred_seg"]) < FP_TOL

    # all 17 loss terms (the reference logs the line terms both unscaled and scaled)
    for k, v in terms.items():
        key = "stat/" + k + ("_unscaled" if k.startswith(("loss_ce", "loss_line")) else "")
        assert abs(float(v) - float(g[key])) <= FP_TOL * max(1.0, abs(float(g[key]))), k
    assert abs(float(total) - float(g["stat/loss"])) <= FP_TOL * abs(float(g["stat/loss"]))

    # gradients: same set of dead parameters, per-parameter L2 and sum, global norm
    names = list(g["grad_names"])
    have = sorted(n for n, v in grads.items() if v is not None)
    assert have == names
    assert sorted(n for n, v in grads.items() if v is None) == list(g["nograd_names"])
    l2 = np.array([float(grads[n].double().norm()) for n in names])
    big = g["grad_l2"] > 1e-6 * g["grad_l2"].max()
    assert np.max(np.abs(l2[big] - g["grad_l2"][big]) / g["grad_l2"][big]) < 2e-3
    assert abs(gnorm - float(g["grad_total_norm"])) / float(g["grad_total_norm"]) < FP_TOL

    # optimizer step (clip 0.1 + AdamW, two LR groups)
    dl2 = np.array([float((sd[n].detach() - before[n]).double().norm()) for n in names])
    # (`big` drops the 4 ref_attn_diffusion.bias tensors: a per-channel bias in front of an affine-free
    #  LayerNorm has a mathematically zero gradient, so the reference's own value is rounding noise)
    assert np.max(np.abs(dl2[big] - g["step_delta_l2"][big]) / (g["step_delta_l2"][big] + 1e-12)) < 2e-3
    pl2 = np.array([float(sd[n].detach().double().norm()) for n in names])
    assert np.max(np.abs(pl2 - g["param_l2_after"]) / (g["param_l2_after"] + 1e-12)) < 1e-5


def test_mid_multiwindow_forward_matches_reference(golden_dir):
    """224x288: 2..88 windows per map, shifted blocks with real -100 masks, no optimizer."""
    case = "mid_b1_224x288"
    g = load(golden_dir, case)
    c = CASES[case]
    sd = det_fill_(reference_state_shapes(), seed=0)
    b = synth_batch(c["batch"], c["height"], c["width"], seed=c["seed"], n_lines=c["n_lines"], sizes=c["sizes"])
    taps = {}
    with torch.no_grad():
        out = R.forward(sd, b["images"], b["pad_mask"], R.Cfg(), training=True, taps=taps)
    assert np.array_equal(taps["topk_ids"].numpy(), g["topk_ids"])
    assert np.array_equal(taps["points1"].numpy(), g["points1"])
    assert np.array_equal(taps["points2"].numpy(), g["points2"])
    for i, d in enumerate(out["pred_depth"]):
        assert rel(d, g[f"pred_depth{i}"]) < FP_TOL, i
    assert rel(out["pred_seg"], g["pred_seg"]) < FP_TOL
    assert rel(out["pred_lines"], g["pred_lines"]) < FP_TOL
