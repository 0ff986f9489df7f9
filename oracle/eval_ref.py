"""CPU restatement (numpy) of the reference's dense evaluation arithmetic.  TEST INFRASTRUCTURE ONLY:
only tests/, __graft_entry__.smoke() and bench.py's checker legs may import this; the product never does.

Pinned by tests/golden/eval_metrics.npz, which oracle/make_golden_eval.py produced by running the reference's own
evaluate() (src/engine_glassrgbd.py:174-345) in the build container: tests/test_eval_metrics.py::test_oracle_*.
"""
import warnings

import numpy as np

METRIC_NAMES = ['silog', 'abs_rel', 'log10', 'rms', 'sq_rel', 'log_rms', 'd1', 'd2', 'd3']     # src/engine_glassrgbd.py:204
SEG_LABELS = ['Background', 'Glass']                                                            # src/util/metrics.py:10-11


def clamp_and_mask(pred, gt, min_d, max_d):
    """src/engine_glassrgbd.py:249-253: clamp the prediction (in that order), GT strictly inside (min, max) is valid."""
    pred = np.array(pred, dtype=np.float32, copy=True)
    pred[pred < min_d] = min_d
    pred[pred > max_d] = max_d
    pred[np.isinf(pred)] = max_d
    pred[np.isnan(pred)] = min_d
    valid = np.logical_and(gt > min_d, gt < max_d)
    return pred, valid


def depth_errors(gt, pred):
    """src/util/metrics.py:198-218 on the valid pixels (fp32 arrays, numpy's own fp32 means)."""
    with np.errstate(all="ignore"), warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)          # an image without valid pixels: NaN, as in the reference
        thresh = np.maximum(gt / pred, pred / gt)
        d1 = (thresh < 1.25).mean()
        d2 = (thresh < 1.25 ** 2).mean()
        d3 = (thresh < 1.25 ** 3).mean()
        rms = np.sqrt(((gt - pred) ** 2).mean())
        log_rms = np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())
        abs_rel = np.mean(np.abs(gt - pred) / gt)
        sq_rel = np.mean(((gt - pred) ** 2) / gt)
        err = np.log(pred) - np.log(gt)
        silog = np.sqrt(np.mean(err ** 2) - np.mean(err) ** 2) * 100
        log10 = np.mean(np.abs(np.log10(pred) - np.log10(gt)))
    return [silog, abs_rel, log10, rms, sq_rel, log_rms, d1, d2, d3]


def confusion_matrix(gt, pred, num_classes=2):
    """src/util/metrics.py:37-55 after the ignore-255 filter of :69-72; counts[i_gt, i_pred]."""
    keep = gt != 255
    gt, pred = gt[keep].astype(np.int64), pred[keep].astype(np.int64)
    idx = gt * num_classes + pred
    cnt = np.bincount(idx[(idx >= 0) & (idx < num_classes * num_classes)], minlength=num_classes * num_classes)
    return cnt.reshape(num_classes, num_classes).astype(np.float64)


def seg_scores(conf):
    """src/util/metrics.py:76-98: per-class IoU, pixel accuracy, mean accuracy, mean IoU (all in percent)."""
    pos, res, tp = conf.sum(1), conf.sum(0), np.diag(conf)
    with np.errstate(all="ignore"):
        pixel_accuracy = tp.sum() / pos.sum() * 100
    mean_accuracy = (tp / np.maximum(1.0, pos)).mean() * 100
    iou = tp / np.maximum(1.0, pos + res - tp) * 100
    out = {lab: float(v) for lab, v in zip(SEG_LABELS, iou)}
    out['Pixel accuracy'] = float(pixel_accuracy)
    out['Mean accuracy'] = float(mean_accuracy)
    out['Mean IU'] = float(iou.mean())
    return out


def evaluate_dense(pred_depth, gt_depth, seg_logits, seg_gt, min_d=1e-3, max_d=10.0):
    """The with_dense part of evaluate() (src/engine_glassrgbd.py:231-264, 309-326) for n images.
    pred_depth / gt_depth (n,1,H,W) fp32, seg_logits (n,2,H,W), seg_gt (n,1,H,W) int.  Returns (per-image (n,9), stats)."""
    n = pred_depth.shape[0]
    per_image = np.zeros((n, 9))
    conf = np.zeros((2, 2))
    for i in range(n):
        p, valid = clamp_and_mask(pred_depth[i, 0], gt_depth[i, 0], min_d, max_d)
        per_image[i] = depth_errors(gt_depth[i, 0][valid].astype(np.float32), p[valid])
        conf += confusion_matrix(seg_gt[i, 0], seg_logits[i].argmax(0))           # :236-241 (argmax over classes)
    stats = seg_scores(conf)
    mean = per_image.astype(np.float32).sum(0, dtype=np.float32) / np.float32(n)  # :262-263, 316-317 (fp32 running sums)
    stats.update({k: float(v) for k, v in zip(METRIC_NAMES, mean)})
    return per_image, stats
