"""Generate tests/golden/*.npz by running the REAL reference on CPU (build container only).

TEST INFRASTRUCTURE ONLY.  Usage:  python -m oracle.make_golden [case ...]

What is executed is the reference's own code, unmodified, imported from /root/reference under
the stand-ins of oracle/ref_stubs.py: models.build_model (src/models/__init__.py:5-6) and
engine_glassrgbd.train_one_epoch (src/engine_glassrgbd.py:22-171) with a one-batch loader and a
real torch.optim.AdamW built exactly as src/main_glassrgbd.py:59-66 builds it.  Instrumentation
is by hooks only (forward hooks for index outputs, a wrapper around clip_grad_norm_ to read the
un-clipped gradients).  Weights come from gw_depth_amd.synth.det_fill_ (name-hashed, so the
266 MB state dict is never stored); inputs from gw_depth_amd.synth.synth_batch.
"""
import hashlib
import os
import random
import sys

import numpy as np
import torch

from gw_depth_amd.synth import det_fill_, synth_batch
from . import ref_stubs

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# name -> (B, H, W, ragged sizes or None, lines per image, store full dense outputs?)
CASES = {
    "tiny_b2_96x128": dict(batch=2, height=96, width=128, sizes=None, n_lines=[5, 3], seed=11),
    "ragged_b2_96x128": dict(batch=2, height=96, width=128, sizes=[(96, 128), (80, 104)], n_lines=[4, 6], seed=12),
    "mid_b1_224x288": dict(batch=1, height=224, width=288, sizes=None, n_lines=[7], seed=13),
    # --with_plane_norm_loss (one image per step, engine_glassrgbd.py:85-86,133-135); the class head gets a bias towards
    # "line" so that PlaneLoss finds confident lines on random-init weights
    "plane_b1_96x128": dict(batch=1, height=96, width=128, sizes=None, n_lines=[5], seed=14,
                            extra=["--with_plane_norm_loss"], class_bias=(1.5, -1.5)),
}


def sha(t):
    return hashlib.sha256(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes()).hexdigest()


def run_case(name, cfg, train=True):
    ref_stubs.install()
    import engine_glassrgbd as eng                  # /root/reference/src/engine_glassrgbd.py
    from models import build_model                   # /root/reference/src/models/__init__.py
    from util.misc import NestedTensor               # /root/reference/src/util/misc.py:347

    torch.manual_seed(0)
    random.seed(0)
    np.random.seed(0)
    args = ref_stubs.reference_args(cfg.get("extra", ()))
    orig_cuda = (torch.Tensor.cuda, torch.nn.Module.cuda)
    if cfg.get("extra"):                 # PlaneLoss calls .cuda() on its constants (glassrgbd.py:391,404,412); no GPU here
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.nn.Module.cuda = lambda self, *a, **k: self
    model, criterions, _ = build_model(args)
    det_fill_(model.state_dict(), seed=0)
    if cfg.get("class_bias"):
        with torch.no_grad():
            model.class_embed.bias.copy_(torch.tensor(cfg["class_bias"]))

    b = synth_batch(cfg["batch"], cfg["height"], cfg["width"], seed=cfg["seed"],
                    n_lines=cfg["n_lines"], sizes=cfg["sizes"])
    samples = NestedTensor(b["images"], b["pad_mask"])
    depth_gt = NestedTensor(b["depth"], b["pad_mask"])
    seg_gt = NestedTensor(b["seg"], b["pad_mask"])
    loader = [(samples, depth_gt, seg_gt, b["targets"], ["synthetic\n"])]

    out = {}
    cap = {}

    def keep(key):
        def hook(_m, _inp, res):
            cap[key] = res
        return hook

    handles = [model.register_forward_hook(keep("outputs")),
               model.dense_encoder.depth_pred32.register_forward_hook(keep("depth0_tokens")),
               model.dense_encoder.certainSample1.register_forward_hook(keep("points1")),
               model.dense_encoder.certainSample2.register_forward_hook(keep("points2"))]
    match_log = []
    handles.append(criterions[0].matcher.register_forward_hook(
        lambda _m, _i, res: match_log.append([(i.clone(), j.clone()) for i, j in res])))

    # main_glassrgbd.py:59-66
    named = list(model.named_parameters())
    groups = [{"params": [p for n, p in named if "backbone" not in n and p.requires_grad]},
              {"params": [p for n, p in named if "backbone" in n and p.requires_grad], "lr": args.lr_backbone}]
    opt = torch.optim.AdamW(groups, lr=args.lr, weight_decay=args.weight_decay)
    before = {n: p.detach().clone() for n, p in named if p.requires_grad}

    raw_grads = {}
    orig_clip = torch.nn.utils.clip_grad_norm_

    def spy_clip(params, max_norm, *a, **k):
        params = list(params)
        for (n, p) in named:
            if p.grad is not None:
                raw_grads[n] = p.grad.detach().clone()
        tn = orig_clip(params, max_norm, *a, **k)
        out["grad_total_norm"] = np.float64(float(tn))
        return tn

    torch.nn.utils.clip_grad_norm_ = spy_clip
    eng.show_labels = lambda *a, **k: None
    try:
        stats = eng.train_one_epoch(model, criterions, None, loader, opt, torch.device("cpu"), 0,
                                    args.clip_max_norm, args, save_dir=None)
    finally:
        torch.nn.utils.clip_grad_norm_ = orig_clip
        torch.Tensor.cuda, torch.nn.Module.cuda = orig_cuda
        for h in handles:
            h.remove()

    o = cap["outputs"]
    out["pred_logits"] = o["pred_logits"].detach().numpy()
    out["pred_lines"] = o["pred_lines"].detach().numpy()
    for i, a in enumerate(o["aux_outputs"]):
        out[f"aux{i}_pred_logits"] = a["pred_logits"].detach().numpy()
        out[f"aux{i}_pred_lines"] = a["pred_lines"].detach().numpy()
    for i, d in enumerate(o["pred_depth"]):
        out[f"pred_depth{i}"] = d.detach().numpy()
    out["pred_seg"] = o["pred_seg"].detach().numpy()
    out["depth0_tokens"] = cap["depth0_tokens"].detach().numpy()      # (B, H/32*W/32, 1): 1/32-scale sigmoid head
    out["points1"] = cap["points1"].detach().numpy()
    out["points2"] = cap["points2"].detach().numpy()
    # top-k line ids as the reference picks them (multiscale_transformerr.py:1166)
    out["topk_ids"] = torch.topk(o["pred_logits"][:, :, 0], args.num_ref, dim=-1)[1].numpy()
    for li, m in enumerate(match_log):          # 0 = final layer, 1..5 = aux 0..4
        for bi, (i, j) in enumerate(m):
            out[f"match{li}_b{bi}_src"] = i.numpy()
            out[f"match{li}_b{bi}_tgt"] = j.numpy()
    for k, v in stats.items():
        out["stat/" + k] = np.float64(v)

    names = sorted(raw_grads)
    out["grad_names"] = np.array(names)
    out["grad_l2"] = np.array([float(raw_grads[n].double().norm()) for n in names])
    out["grad_sum"] = np.array([float(raw_grads[n].double().sum()) for n in names])
    out["nograd_names"] = np.array(sorted(n for n, p in named if p.requires_grad and n not in raw_grads))
    after = dict(model.named_parameters())
    out["step_delta_l2"] = np.array([float((after[n].detach() - before[n]).double().norm()) for n in names])
    out["param_l2_after"] = np.array([float(after[n].detach().double().norm()) for n in names])
    out["input_sha"] = np.array([sha(b["images"]), sha(b["depth"]), sha(b["seg"]),
                                 sha(torch.cat([t["lines"] for t in b["targets"]]))])
    out["weights_sha"] = np.array([sha(torch.cat([before[n].flatten() for n in names]))])

    # eval-mode forward on the UPDATED weights is not needed; eval forward on the ORIGINAL weights is
    # the C1 plumbing case (src/eval_main_glassrgbd.py) -> regenerate with a fresh model.
    return out


def main(argv):
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    torch.set_num_threads(8)
    want = argv or list(CASES)
    for name in want:
        res = run_case(name, CASES[name])
        path = os.path.join(GOLDEN_DIR, name + ".npz")
        np.savez_compressed(path, **res)
        print(name, "->", path, "%.1f KB" % (os.path.getsize(path) / 1024),
              "loss", res["stat/loss"], "gnorm", res["grad_total_norm"])


if __name__ == "__main__":
    main(sys.argv[1:])
