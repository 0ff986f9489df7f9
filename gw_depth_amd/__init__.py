"""gw_depth_amd — MI355X-native (gfx950) implementation of the GW-Depth dense-prediction train step.

Public surface = the reference's own model API (/root/reference/src/models/__init__.py:5-6):
    build_model(args) -> (model, [criterion, criterion_depth, criterion_seg, criterion_plane], postprocessors)
"""
from .config import Config


def build_model(args):
    """Drop-in for the reference's models.build_model (glassrgbd.py:509-579)."""
    import torch
    from .criteria import HungarianMatcherLine, PlaneLoss, PostProcessLine, SegLoss, SetCriterion, SilogLoss
    from .model import GlassRGBD
    cfg = Config.from_args(args)
    model = GlassRGBD(cfg)
    matcher = HungarianMatcherLine(cfg.set_cost_class, cfg.set_cost_line)
    weight_dict = {"loss_ce": 1, "loss_line": cfg.line_loss_coef}
    if cfg.aux_loss:
        for i in range(cfg.dec_layers - 1):
            weight_dict.update({f"loss_ce_{i}": 1, f"loss_line_{i}": cfg.line_loss_coef})
    criterion = SetCriterion(1, weight_dict, cfg.eos_coef, ["lines_labels", "lines"], matcher)
    device = torch.device(cfg.device) if (cfg.device != "cuda" or torch.cuda.is_available()) else None
    if device is not None:
        criterion.to(device)
    criterion_depth = SilogLoss(cfg.variance_focus, cfg.log_depth_error)
    criterion_seg = SegLoss()
    criterion_plane = PlaneLoss(28, line_score_thresh=0.6, min_plane_area=100) if cfg.with_plane_norm_loss else None   # glassrgbd.py:573-577
    return model, [criterion, criterion_depth, criterion_seg, criterion_plane], {"line": PostProcessLine()}
