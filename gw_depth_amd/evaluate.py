"""Evaluation path: mirror of evaluate() (/root/reference/src/engine_glassrgbd.py:174-345) with the dense metrics
accumulated ON THE DEVICE (gwd_eval_accumulate) instead of per-image device->host copies + numpy.

`DenseMetrics` is the device-side accumulator (clamp + validity mask :249-253, compute_depth_errors
src/util/metrics.py:198-218, confusion counts :37-74, running sums :262-263); `evaluate` keeps the reference's
signature and returns the same stats keys (`Background, Glass, Pixel accuracy, Mean accuracy, Mean IU, silog, abs_rel,
log10, rms, sq_rel, log_rms, d1, d2, d3`, plus the line-loss terms with --with_line).  Unlike the reference it accepts
batches of more than one image: the depth measures are per image either way, padded pixels are excluded through the
NestedTensor masks.  The visualisation switches (save_dense / save_line) are outside the accelerated path.
"""
import torch

from . import hip

METRIC_NAMES = ["silog", "abs_rel", "log10", "rms", "sq_rel", "log_rms", "d1", "d2", "d3"]     # engine_glassrgbd.py:204
SEG_LABELS = ["Background", "Glass"]                                                            # util/metrics.py:10-11


def _lib():
    return hip.library()


class DenseMetrics:
    """Running depth / segmentation metrics of an evaluation pass, kept in HBM until compute()."""

    def __init__(self, device, min_depth_eval=1e-3, max_depth_eval=10.0):
        self.device = torch.device(device)
        self.min_d, self.max_d = float(min_depth_eval), float(max_depth_eval)
        self.running = torch.zeros(10, dtype=torch.float64, device=self.device)       # depth_eval_measures (:203)
        self.confusion = torch.zeros(4, dtype=torch.int64, device=self.device)        # confusion_matrix (metrics.py:60)
        self._ws = None

    def reset(self):
        self.running.zero_()
        self.confusion.zero_()

    def update(self, pred_depth=None, gt_depth=None, pred_seg=None, seg_gt=None):
        """pred_depth (B,1,H,W) or (B,H,W) fp32/bf16 metres, gt_depth same shape; pred_seg (B,2,H,W) logits in ANY
        strides whose two pixel dims collapse (the model's pixel-major view qualifies), seg_gt (B,1,H,W)/(B,H,W) int64
        with 255 = ignore.  Either pair may be None.  Returns the (B,9) per-image measures (device, f64) or None."""
        lib = _lib()
        B = (pred_depth if pred_depth is not None else pred_seg).shape[0]
        pred = gt = seg = tgt = measures = None
        strides = (0, 0, 0)
        if pred_depth is not None:
            pred = pred_depth.reshape(B, -1).contiguous()
            gt = gt_depth.reshape(B, -1).to(torch.float32).contiguous()
            if pred.shape != gt.shape:
                raise ValueError("pred_depth %s and gt_depth %s differ" % (tuple(pred_depth.shape), tuple(gt_depth.shape)))
            HW = pred.shape[1]
            measures = torch.empty(B, 9, dtype=torch.float64, device=pred.device)
        if pred_seg is not None:
            if pred_seg.dim() != 4 or pred_seg.shape[1] != 2:
                raise ValueError("pred_seg must be (B, 2, H, W) logits, got %s" % (tuple(pred_seg.shape),))
            Hs, Ws = pred_seg.shape[-2:]
            if Hs > 1 and pred_seg.stride(2) != Ws * pred_seg.stride(3):
                pred_seg = pred_seg.contiguous()
            seg = pred_seg
            strides = (seg.stride(0), seg.stride(3), seg.stride(1))
            tgt = seg_gt.reshape(B, -1).to(torch.int64).contiguous()
            if tgt.shape[1] != Hs * Ws or (pred is not None and HW != Hs * Ws):
                raise ValueError("seg_gt / pred_seg / pred_depth pixel counts differ")
            HW = Hs * Ws
        need = lib.workspace_bytes(hip.WS_EVAL, B, HW)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.running.device)
        lib.eval_accumulate(pred, gt, seg, strides, tgt, self._ws, measures, self.running, self.confusion, B, HW,
                            self.min_d, self.max_d)
        return measures

    def compute(self):
        """One device->host copy; the closing arithmetic of compute_mean_ioU (metrics.py:76-98) and of evaluate()
        (:313-319) on 14 numbers."""
        run = self.running.cpu()
        conf = self.confusion.cpu().to(torch.float64).reshape(2, 2)
        out = {}
        if float(conf.sum()) > 0:
            pos, res, tp = conf.sum(1), conf.sum(0), conf.diagonal()
            iou = tp / torch.clamp(pos + res - tp, min=1.0) * 100
            for lab, v in zip(SEG_LABELS, iou):
                out[lab] = float(v)
            out["Pixel accuracy"] = float(tp.sum() / pos.sum() * 100)
            out["Mean accuracy"] = float((tp / torch.clamp(pos, min=1.0)).mean() * 100)
            out["Mean IU"] = float(iou.mean())
        if float(run[9]) > 0:
            for k, name in enumerate(METRIC_NAMES):
                out[name] = float(run[k] / run[9])
        return out


@torch.no_grad()
def evaluate(model, criterions, postprocessors, data_loader, base_ds, device, output_dir, args, save_dir=None, epoch=0,
             save_dense=False, save_line=False):
    """Same signature and stats as the reference's evaluate() (engine_glassrgbd.py:174-345)."""
    if save_dense or save_line:
        raise NotImplementedError("save_dense / save_line write visualisations; outside the accelerated path (SURVEY.md §2)")
    model.eval()
    criterion = criterions[0]
    if getattr(args, "with_line", False) and criterion is not None:
        criterion.eval()
    dm = DenseMetrics(device, getattr(args, "min_depth_eval", 1e-3), getattr(args, "max_depth_eval", 10.0))
    line_sums, n_batches = {}, 0
    for samples, depth_gt, seg_gt, targets, img_name in data_loader:
        samples = samples.to(device)
        targets = [{k: (v.to(device) if torch.is_tensor(v) else v) for k, v in t.items()} for t in targets]
        outputs = model(samples, reflc_mat=None, img_name=img_name[0].strip() if img_name else None)
        if getattr(args, "with_line", False) and criterion is not None:
            loss = criterion(outputs, targets)                                   # :221-229
            wd = criterion.weight_dict
            for k, v in loss.items():
                line_sums[k + "_unscaled"] = line_sums.get(k + "_unscaled", 0.0) + v.detach().double()
                if k in wd:
                    line_sums[k] = line_sums.get(k, 0.0) + v.detach().double() * wd[k]
            line_sums["loss"] = line_sums.get("loss", 0.0) + sum(v.detach().double() * wd[k] for k, v in loss.items() if k in wd)
            n_batches += 1
        if getattr(args, "with_dense", True):
            pd = outputs["pred_depth"][-1] if isinstance(outputs["pred_depth"], (list, tuple)) else outputs["pred_depth"]
            ps = outputs["pred_seg"][-1] if isinstance(outputs["pred_seg"], (list, tuple)) else outputs["pred_seg"]
            g = depth_gt.tensors.to(device)
            s = seg_gt.tensors.to(device)
            if depth_gt.mask is not None and bool(depth_gt.mask.any()):          # batches > 1: padding is not evaluated
                pad = depth_gt.mask.to(device).unsqueeze(1)
                g = g.masked_fill(pad, 0.0)
                s = s.masked_fill(pad, 255)
            dm.update(pd, g, ps, s)
    stats = {k: float(v / n_batches) for k, v in line_sums.items()}
    stats.update(dm.compute())
    return stats
