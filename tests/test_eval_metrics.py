"""Dense evaluation metrics: reference evaluate() -> golden vectors -> oracle -> device kernel.

tests/golden/eval_metrics.npz holds the inputs and the stats the reference's OWN evaluate()
(src/engine_glassrgbd.py:174-345) returned for them (oracle/make_golden_eval.py).  Tolerances: the three threshold
accuracies d1/d2/d3 and the confusion counts are bit-exact (IEEE division, integer counts); the other measures are fp32
means in the reference (numpy pairwise sums) against f64 sums here: 2e-6 relative.
"""
import os

import numpy as np
import pytest
import torch

from gw_depth_amd import hip
from gw_depth_amd.evaluate import METRIC_NAMES, SEG_LABELS, DenseMetrics, evaluate
from gw_depth_amd.model import NestedTensor
from oracle import eval_ref
from tests.fake_device import FakeDevice

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "eval_metrics.npz")
RTOL = 2e-6
SEG_KEYS = SEG_LABELS + ["Pixel accuracy", "Mean accuracy", "Mean IU"]


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(GOLDEN))


def check_stats(stats, gold, rtol=RTOL):
    for k in SEG_KEYS:
        assert stats[k] == pytest.approx(float(gold["stat/" + k]), rel=1e-12), k          # integer counts: exact up to the last division
    for k in METRIC_NAMES:
        assert stats[k] == pytest.approx(float(gold["stat/" + k]), rel=rtol), k


def check_per_image(per_image, gold):
    per_image = np.asarray(per_image, dtype=np.float64)
    ref = gold["per_image"]
    np.testing.assert_array_equal(per_image[:, 6:], ref[:, 6:])                           # d1, d2, d3: bit-exact
    np.testing.assert_allclose(per_image[:, :6], ref[:, :6], rtol=RTOL)


# ----------------------------------------------------------------------------------------------- CPU: oracle, host logic
def test_oracle_matches_reference_evaluate(gold):
    per_image, stats = eval_ref.evaluate_dense(gold["pred_depth"], gold["gt_depth"], gold["seg_logits"], gold["seg_gt"],
                                               float(gold["min_depth_eval"]), float(gold["max_depth_eval"]))
    check_per_image(per_image, gold)
    check_stats(stats, gold, rtol=1e-6)


def test_oracle_empty_valid_mask_is_nan_like_numpy():
    p = np.full((1, 1, 4, 4), 2.0, np.float32)
    g = np.zeros((1, 1, 4, 4), np.float32)
    per_image, _ = eval_ref.evaluate_dense(p, g, np.zeros((1, 2, 4, 4), np.float32), np.full((1, 1, 4, 4), 255))
    assert np.isnan(per_image).all()


@pytest.fixture()
def fake():
    hip.set_library(FakeDevice())
    yield
    hip.set_library(None)


def _tensors(gold, device="cpu", seg_dtype=torch.float32, pixel_major=False):
    pred = torch.from_numpy(gold["pred_depth"]).to(device)
    gt = torch.from_numpy(gold["gt_depth"]).to(device)
    seg = torch.from_numpy(gold["seg_logits"]).to(device).to(seg_dtype)
    if pixel_major:                                                                       # the model's own (B,H,W,2) storage, viewed as (B,2,H,W)
        seg = seg.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    tgt = torch.from_numpy(gold["seg_gt"]).to(device)
    return pred, gt, seg, tgt


def test_host_logic_matches_reference_evaluate(fake, gold):
    pred, gt, seg, tgt = _tensors(gold, pixel_major=True)
    dm = DenseMetrics("cpu", float(gold["min_depth_eval"]), float(gold["max_depth_eval"]))
    per_image = torch.cat([dm.update(pred[i:i + 1], gt[i:i + 1], seg[i:i + 1], tgt[i:i + 1]) for i in range(pred.shape[0])])
    check_per_image(per_image.numpy(), gold)
    check_stats(dm.compute(), gold)
    dm.reset()
    dm.update(pred, gt, seg, tgt)                                                         # the same five images as ONE batch
    check_stats(dm.compute(), gold)


class _Canned(torch.nn.Module):
    def __init__(self, pred, seg, bs):
        super().__init__()
        self.pred, self.seg, self.bs, self.k = pred, seg, bs, 0

    def forward(self, samples, reflc_mat=None, img_name=None):
        i, self.k = self.k, self.k + self.bs
        return {"pred_depth": [self.pred[i:i + self.bs] * 0.5, self.pred[i:i + self.bs]], "pred_seg": self.seg[i:i + self.bs]}


def _loader(gt, tgt, bs, device="cpu"):
    n, _, H, W = gt.shape
    out = []
    for i in range(0, n, bs):
        m = torch.zeros(min(bs, n - i), H, W, dtype=torch.bool, device=device)
        out.append((NestedTensor(torch.zeros(m.shape[0], 3, H, W, device=device), m), NestedTensor(gt[i:i + bs], m),
                    NestedTensor(tgt[i:i + bs], m), [{"image_id": torch.tensor([j])} for j in range(i, min(i + bs, n))],
                    ["img%d\n" % i]))
    return out


class _Args:
    with_line, with_dense, min_depth_eval, max_depth_eval = False, True, 1e-3, 10.0


@pytest.mark.parametrize("bs", [1, 2])
def test_evaluate_mirror_returns_reference_stats(fake, gold, bs):
    pred, gt, seg, tgt = _tensors(gold)
    stats = evaluate(_Canned(pred, seg, bs), (None, None, None, None), None, _loader(gt, tgt, bs), None, "cpu", None, _Args())
    check_stats(stats, gold)
    with pytest.raises(NotImplementedError):
        evaluate(_Canned(pred, seg, bs), (None,) * 4, None, [], None, "cpu", None, _Args(), save_dense=True)


def test_update_rejects_mismatched_shapes(fake, gold):
    pred, gt, seg, tgt = _tensors(gold)
    dm = DenseMetrics("cpu")
    with pytest.raises(ValueError):
        dm.update(pred, gt[:, :, :-1], None, None)
    with pytest.raises(ValueError):
        dm.update(None, None, seg[:, :1], tgt)


# ----------------------------------------------------------------------------------------------- GPU: the kernel
@pytest.fixture()
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    hip.set_library(None)
    return hip.library()


@pytest.mark.gpu
@pytest.mark.parametrize("pixel_major", [False, True])
def test_kernel_matches_reference_evaluate(dev, gold, pixel_major):
    pred, gt, seg, tgt = _tensors(gold, "cuda", pixel_major=pixel_major)
    dm = DenseMetrics("cuda", float(gold["min_depth_eval"]), float(gold["max_depth_eval"]))
    per_image = torch.cat([dm.update(pred[i:i + 1], gt[i:i + 1], seg[i:i + 1], tgt[i:i + 1]) for i in range(pred.shape[0])])
    check_per_image(per_image.cpu().numpy(), gold)
    check_stats(dm.compute(), gold)
    dm.reset()
    batch = dm.update(pred, gt, seg, tgt)                                                 # one batch of five: same records, bit for bit
    assert torch.equal(batch, per_image)
    check_stats(dm.compute(), gold)


@pytest.mark.gpu
def test_kernel_full_size_batch_vs_oracle_and_batch_invariance(dev):
    """BASELINE's 8 x 480 x 640 evaluation batch: kernel == oracle; per-image records do not depend on batching; the
    confusion counts add up to the number of non-ignored pixels; bf16 logits read in place from the pixel-major layout."""
    B, H, W = 8, 480, 640
    g = torch.Generator().manual_seed(5)
    pred = torch.rand(B, 1, H, W, generator=g) * 12 - 1
    gt = torch.rand(B, 1, H, W, generator=g) * 11
    gt[torch.rand(B, 1, H, W, generator=g) < 0.1] = 0
    gt[3] = 0                                                                             # an image without any valid pixel
    pred[0, 0, 0, :7] = torch.tensor([float("nan"), float("inf"), -float("inf"), 0.0, 1e-3, 10.0, 20.0])
    seg = torch.randn(B, H, W, 2, generator=g).to(torch.bfloat16)
    tgt = (torch.rand(B, 1, H, W, generator=g) < 0.3).long()
    tgt[torch.rand(B, 1, H, W, generator=g) < 0.02] = 255
    tgt[5] = 255                                                                          # an image that is ignored entirely
    seg_view = seg.cuda().permute(0, 3, 1, 2)
    dm = DenseMetrics("cuda")
    rec = dm.update(pred.cuda(), gt.cuda(), seg_view, tgt.cuda()).cpu().numpy()
    conf = dm.confusion.cpu().numpy().copy()
    run = dm.running.cpu().numpy().copy()
    ref_rec, _ = eval_ref.evaluate_dense(pred.numpy(), gt.numpy(), seg.float().permute(0, 3, 1, 2).numpy(), tgt.numpy())
    assert np.isnan(rec[3]).all() and np.isnan(ref_rec[3]).all()
    keep = [i for i in range(B) if i != 3]
    np.testing.assert_array_equal(rec[keep, 6:], ref_rec[keep, 6:])
    np.testing.assert_allclose(rec[keep, :6], ref_rec[keep, :6], rtol=RTOL)
    ref_conf = sum(eval_ref.confusion_matrix(tgt[i, 0].numpy(), seg[i].float().argmax(-1).numpy()) for i in range(B))
    np.testing.assert_array_equal(conf.reshape(2, 2), ref_conf.astype(np.int64))
    assert conf.sum() == int((tgt != 255).sum())
    assert run[9] == B and np.isnan(run[:9]).all()                                        # the NaN image poisons the mean, as in the reference
    dm2 = DenseMetrics("cuda")
    one = np.concatenate([dm2.update(pred[i:i + 1].cuda(), gt[i:i + 1].cuda(), seg_view[i:i + 1], tgt[i:i + 1].cuda()).cpu().numpy()
                          for i in range(B)])
    np.testing.assert_array_equal(one, rec)
    np.testing.assert_array_equal(dm2.confusion.cpu().numpy(), conf)


@pytest.mark.gpu
def test_kernel_depth_only_and_seg_only(dev, gold):
    pred, gt, seg, tgt = _tensors(gold, "cuda")
    a, b, c = DenseMetrics("cuda"), DenseMetrics("cuda"), DenseMetrics("cuda")
    a.update(pred, gt, seg, tgt)
    b.update(pred, gt, None, None)
    c.update(None, None, seg, tgt)
    assert torch.equal(a.running, b.running) and torch.equal(a.confusion, c.confusion)
    assert int(b.confusion.sum()) == 0 and float(c.running.sum()) == 0.0
