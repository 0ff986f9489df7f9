"""Generate tests/golden/eval_metrics.npz by running the REAL reference's evaluate() on CPU (build container only).

TEST INFRASTRUCTURE ONLY.  Usage:  python -m oracle.make_golden_eval

What is executed is the reference's own, unmodified engine_glassrgbd.evaluate
(src/engine_glassrgbd.py:174-345) - its clamp / validity-mask steps (:249-254), util.metrics.compute_depth_errors
(src/util/metrics.py:198-218) and compute_mean_ioU (:57-99) - imported from /root/reference under the
stand-ins of oracle/ref_stubs.py.  The model is a canned-output stand-in (the network itself is pinned by
oracle/make_golden.py); the loader is a list; `Tensor.cuda` is the identity for the duration of the call
(evaluate() accumulates into `torch.zeros(10).cuda()`, :203, and this container has no GPU).  Only the vectors
(inputs and the returned stats) travel.
"""
import os
import tempfile

import numpy as np
import torch

from . import ref_stubs

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def canned_inputs(n=5, H=60, W=80, seed=21):
    """n single-image batches: predicted depth with out-of-range / inf / nan pixels, GT with holes and
    beyond-range pixels, 2-class logits, labels in {0, 1, 255 = ignore}."""
    g = torch.Generator().manual_seed(seed)
    pred = torch.rand(n, 1, H, W, generator=g) * 11.0 - 0.5            # (-0.5, 10.5): both clamps fire
    gt = torch.rand(n, 1, H, W, generator=g) * 10.5                     # some > max_depth_eval
    gt[torch.rand(n, 1, H, W, generator=g) < 0.15] = 0.0                # holes
    pred[:, :, 3, 5] = float("inf")
    pred[:, :, 4, 6] = float("-inf")
    pred[:, :, 7, 9] = float("nan")
    pred[2] = gt[2].clamp(1e-3, 10) * 1.2                               # a near-perfect image: d1 = 1 on it
    logits = torch.randn(n, 2, H, W, generator=g)
    logits[:, :, 10, :] = 0.0                                           # ties: argmax picks class 0
    seg = (torch.rand(n, 1, H, W, generator=g) < 0.4).long()
    seg[torch.rand(n, 1, H, W, generator=g) < 0.05] = 255
    seg[4] = 0                                                          # an image without class 1
    return pred, gt, logits, seg


def main():
    ref_stubs.install()
    import engine_glassrgbd as eng                   # /root/reference/src/engine_glassrgbd.py
    from util.misc import NestedTensor               # /root/reference/src/util/misc.py:347

    args = ref_stubs.reference_args()
    args.with_line = False                           # evaluate(): skip the line criterion, keep the dense metrics
    args.coco_path = None
    args.append_word = None
    pred, gt, logits, seg = canned_inputs()
    n, _, H, W = pred.shape

    class Loader(list):
        pass

    loader = Loader()
    loader.dataset = type("D", (), {"id_to_img": {i: "img%d" % i for i in range(n)}})()
    pad = torch.zeros(1, H, W, dtype=torch.bool)
    for i in range(n):
        loader.append((NestedTensor(torch.zeros(1, 3, H, W), pad), NestedTensor(gt[i:i + 1], pad),
                       NestedTensor(seg[i:i + 1], pad), [{"image_id": torch.tensor([i])}], ["img%d\n" % i]))

    class Canned(torch.nn.Module):
        k = 0

        def forward(self, samples, reflc_mat=None, img_name=None):
            i = Canned.k
            Canned.k += 1
            return {"pred_depth": [pred[i:i + 1] * 0.5, pred[i:i + 1]], "pred_seg": logits[i:i + 1]}

    per_image = []
    orig_errors = eng.compute_depth_errors           # instrumentation by wrapper only: record each image's 9 measures

    def spy_errors(g, p):
        res = orig_errors(g, p)
        per_image.append([float(v) for v in res])
        return res

    eng.compute_depth_errors = spy_errors
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        with tempfile.TemporaryDirectory() as tmp:
            stats = eng.evaluate(Canned(), (None, None, None, None), None, loader, None, torch.device("cpu"), tmp, args,
                                 save_dir=tmp, epoch=0)
    finally:
        torch.Tensor.cuda = orig_cuda
        eng.compute_depth_errors = orig_errors

    out = {"pred_depth": pred.numpy(), "gt_depth": gt.numpy(), "seg_logits": logits.numpy(), "seg_gt": seg.numpy(),
           "min_depth_eval": np.float64(args.min_depth_eval), "max_depth_eval": np.float64(args.max_depth_eval)}
    out["per_image"] = np.array(per_image, dtype=np.float64)          # (n, 9) in metric_names order (:204)
    for k, v in stats.items():
        out["stat/" + k] = np.float64(v)
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    path = os.path.join(GOLDEN_DIR, "eval_metrics.npz")
    np.savez_compressed(path, **out)
    print(path, "%.1f KB" % (os.path.getsize(path) / 1024))
    for k, v in stats.items():
        print("  %-16s %r" % (k, float(v)))


if __name__ == "__main__":
    main()
