"""Criteria with the reference's call signatures (/root/reference/src/models/glassrgbd.py:133-383,
/root/reference/src/models/matcher.py).  The dense losses run on the fused HIP reductions; the line
losses are a few hundred elements (latency class) and use torch tensor plumbing plus scipy's LSAP on
the host exactly as the reference does."""
import os

import torch
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment
from torch import nn

from . import ops


def _dist_ready():
    return torch.distributed.is_available() and torch.distributed.is_initialized()


# gwd_match_cost + gwd_lsap + gwd_set_losses_* as ONE autograd node (five launches); False keeps forward_packed's torch formulation of the
# same arithmetic - the unit test of the fused node compares the two (tests/test_hip_kernels.py), nothing else clears it
FUSED_SETLOSS = True

class HungarianMatcherLine(nn.Module):
    """cost = cost_line * L1 cdist - cost_class * prob[target class]; scipy LSAP per image (matcher.py:28-82)."""

    def __init__(self, cost_class=1.0, cost_line=5.0):
        super().__init__()
        self.cost_class, self.cost_line = cost_class, cost_line

    @torch.no_grad()
    def cost_matrix(self, outputs, targets):
        B, Q = outputs["pred_logits"].shape[:2]
        prob = outputs["pred_logits"].flatten(0, 1).float().softmax(-1)
        lines = outputs["pred_lines"].flatten(0, 1).float()
        tgt_ids = torch.cat([t["labels"] for t in targets])
        tgt_lines = torch.cat([t["lines"] for t in targets])
        C = self.cost_line * torch.cdist(lines, tgt_lines, p=1) + self.cost_class * (-prob[:, tgt_ids])
        return C.view(B, Q, -1)

    @staticmethod
    def _solve(C, sizes):
        res = [linear_sum_assignment(c[i]) for i, c in enumerate(C.split(sizes, -1))]
        return [(torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)) for i, j in res]

    @torch.no_grad()
    def forward(self, outputs, targets):
        C = self.cost_matrix(outputs, targets).cpu()
        return self._solve(C, [len(t["lines"]) for t in targets])

    @torch.no_grad()
    def prefetch(self, layer_outputs, targets):
        """Cost matrices of ALL decoder layers in one go, copied to pinned host memory asynchronously.  Called by the
        model right after the DETR branch, so the single device->host hand-off overlaps the whole dense branch
        instead of draining the GPU six times per step (the reference syncs in every matcher call, matcher.py:71)."""
        Cs = torch.stack([self.cost_matrix(o, targets) for o in layer_outputs])          # (layers, B, Q, sum T)
        if Cs.is_cuda:
            host = torch.empty(Cs.shape, dtype=Cs.dtype, pin_memory=True)
            host.copy_(Cs, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:
            host, ev = Cs, None
        return {"host": host, "event": ev, "sizes": [len(t["lines"]) for t in targets], "next": 0}

    def from_prefetch(self, handle):
        """Assignment of the next decoder layer (order: final layer first, then aux 0..4) from a prefetch handle."""
        if handle["event"] is not None:
            handle["event"].synchronize()
            handle["event"] = None
        i = handle["next"]
        handle["next"] += 1
        return self._solve(handle["host"][i], handle["sizes"])


class _SetLossFn(torch.autograd.Function):
    """Cost matrices of all decoder layers, device LSAP, weighted cross entropy and matched-pair L1 per layer - and their gradients -
    as five launches (csrc/setloss.hip, csrc/lsap.hip); forward_packed's torch formulation (kept below as the A/B path) took ~65."""

    @staticmethod
    def forward(ctx, logits, lines, tgt_lines, tgt_labels, meta, class_weight, num_items, world, w_line, w_class):
        from . import hip
        lib = hip.library()
        L_, B, Q, K = logits.shape
        cap = tgt_lines.shape[0]
        dev = logits.device
        col_off, bidx, valid = meta[:B + 1], meta[B + 1:B + 1 + cap], meta[B + 1 + cap:]
        cost = torch.empty((L_, B, Q, cap), dtype=torch.float32, device=dev)
        lib.match_cost(logits, lines, tgt_lines, tgt_labels, cost, w_line, w_class)
        qot = torch.empty((L_, cap), dtype=torch.int32, device=dev)
        lib.lsap(cost, col_off.contiguous(), qot, hip.LSAP_MAX_TARGETS)      # padding columns come back as the dummy query Q
        tc = torch.empty((L_, B, Q), dtype=torch.int32, device=dev)
        out = torch.empty((3, L_), dtype=torch.float32, device=dev)           # ce | l1 | sum of class weights
        bidx, valid = bidx.contiguous(), valid.contiguous()
        cw = class_weight.float().contiguous()
        lib.set_losses_forward(logits, lines, tgt_lines, tgt_labels, bidx, valid, qot, cw, num_items, world, tc, out[0], out[1], out[2])
        ctx.save_for_backward(logits, lines, tgt_lines, bidx, valid, qot, cw, num_items, tc, out)
        ctx.world = world
        ctx.mark_non_differentiable(qot)
        return out[0], out[1], qot

    @staticmethod
    def backward(ctx, g_ce, g_l1, _g_qot):
        from . import hip
        logits, lines, tgt_lines, bidx, valid, qot, cw, num_items, tc, out = ctx.saved_tensors
        dlogits = torch.empty_like(logits)
        dlines = torch.zeros_like(lines)
        con = lambda g: None if g is None else g.contiguous().float()
        hip.library().set_losses_backward(logits, lines, tgt_lines, bidx, valid, qot, cw, num_items, ctx.world, tc, out[2], con(g_ce), con(g_l1),
                                          dlogits, dlines)
        return dlogits, dlines, None, None, None, None, None, None, None, None


class SetCriterion(nn.Module):
    """SetCriterion with losses ['lines_labels', 'lines'] (+ aux), glassrgbd.py:133-358."""

    def __init__(self, num_classes, weight_dict, eos_coef, losses, matcher):
        super().__init__()
        self.num_classes, self.weight_dict, self.eos_coef, self.losses, self.matcher = \
            num_classes, weight_dict, eos_coef, losses, matcher
        w = torch.ones(num_classes + 1)
        w[-1] = eos_coef
        self.register_buffer("empty_weight", w)
        self.last_indices = []

    @staticmethod
    def _src_idx(indices):
        b = torch.cat([torch.full_like(s, i) for i, (s, _) in enumerate(indices)])
        return b, torch.cat([s for s, _ in indices])

    def _one(self, out, targets, num_items, suffix, handle=None):
        idx = self.matcher.from_prefetch(handle) if handle is not None else self.matcher(out, targets)
        self.last_indices.append(idx)
        dev = out["pred_logits"].device
        bi, si = self._src_idx(idx)
        bi, si = bi.to(dev), si.to(dev)
        tc = torch.full(out["pred_logits"].shape[:2], self.num_classes, dtype=torch.int64, device=dev)
        tc[bi, si] = torch.cat([t["labels"][j.to(dev)] for t, (_, j) in zip(targets, idx)])
        ce = F.cross_entropy(out["pred_logits"].float().transpose(1, 2), tc, self.empty_weight)          # :168
        tl = torch.cat([t["lines"][j.to(dev)] for t, (_, j) in zip(targets, idx)], dim=0)
        l1 = F.l1_loss(out["pred_lines"].float()[bi, si], tl, reduction="none").sum() / num_items         # :239-242
        return {"loss_ce" + suffix: ce, "loss_line" + suffix: l1}

    def forward_packed(self, outputs, packed, world=1):
        """Same losses as forward(), with no host round trip: cost matrices of all decoder layers, the device LSAP
        (gwd_lsap) and the 2 x layers loss terms are computed from static-shape device tensors, so the whole step can be
        captured in a HIP graph.  `packed` comes from pack_targets(): the targets of the batch concatenated and PADDED to a
        fixed capacity, with the per-image column offsets, the image index of every column and the padding mask as DEVICE
        data - the shapes (and so a captured graph) do not depend on how many lines each image has.  packed["num_items"] must
        already hold the GLOBAL target count when world > 1 (the caller all-reduces it outside the captured region)."""
        from . import hip
        layers = [outputs] + list(outputs.get("aux_outputs", []))
        logits = torch.stack([o["pred_logits"] for o in layers]).float()               # (L,B,Q,2)
        lines = torch.stack([o["pred_lines"] for o in layers]).float()                 # (L,B,Q,6)
        L_, B, Q, _ = logits.shape
        cap = packed["lines"].shape[0]
        meta = packed["meta"]                                                          # int32: col_off (B+1) | image of column (cap) | valid (cap)
        col_off, bidx, valid = meta[:B + 1], meta[B + 1:B + 1 + cap].long(), meta[B + 1 + cap:].float()
        if FUSED_SETLOSS:
            ce, l1, qi = _SetLossFn.apply(logits.contiguous(), lines.contiguous(), packed["lines"], packed["labels"], meta, self.empty_weight,
                                          packed["num_items"], float(world), float(self.matcher.cost_line), float(self.matcher.cost_class))
            self.last_query_of_target = qi.long()
            losses = {"loss_ce": ce[0], "loss_line": l1[0]}
            for i in range(L_ - 1):
                losses[f"loss_ce_{i}"] = ce[i + 1]
                losses[f"loss_line_{i}"] = l1[i + 1]
            # the per-layer vectors themselves, for a caller that weights them as vectors (engine.TrainStep.losses): summed through the
            # twelve select views above, the backward pass is twelve zero-fills, copies and accumulations of 6-element tensors
            self.last_stacks = (ce, l1, ["loss_ce"] + [f"loss_ce_{i}" for i in range(L_ - 1)], ["loss_line"] + [f"loss_line_{i}" for i in range(L_ - 1)])
            return losses
        with torch.no_grad():                                                          # matcher.py:52-70
            prob = logits.softmax(-1)
            # L1 distances as one broadcast |a - b| summed over the 6 coordinates (aten::cdist takes 158 us for this 48x100x56 problem)
            l1 = (lines.reshape(L_ * B, Q, 1, -1) - packed["lines"][None, None]).abs().sum(-1)
            cost = self.matcher.cost_line * l1 + self.matcher.cost_class * (-prob.reshape(L_ * B, Q, -1)[..., packed["labels"]])
            qot = torch.empty((L_, cap), dtype=torch.int32, device=logits.device)
            # per-image target counts are read from col_off ON THE DEVICE; padding columns come back as query Q (a dummy slot)
            hip.library().lsap(cost.reshape(L_, B, Q, cap).contiguous(), col_off, qot, hip.LSAP_MAX_TARGETS)
            qi = qot.long()
        self.last_query_of_target = qi
        li = torch.arange(L_, device=logits.device)[:, None]
        bi = bidx[None].expand(L_, -1)
        tc = torch.full((L_, B, Q + 1), self.num_classes, dtype=torch.int64, device=logits.device)
        tc[li, bi, qi] = packed["labels"][None].expand(L_, -1)                         # padding columns land in the dummy slot Q
        tc = tc[:, :, :Q]
        nll = F.cross_entropy(logits.reshape(L_ * B, Q, -1).transpose(1, 2), tc.reshape(L_ * B, Q), reduction="none")
        w = self.empty_weight[tc.reshape(L_ * B, Q)]
        ce = (nll * w).reshape(L_, -1).sum(1) / w.reshape(L_, -1).sum(1)                  # weighted mean per layer (:168)
        num_items = torch.clamp(packed["num_items"] / world, min=1.0)
        diff = (lines[li, bi, qi.clamp(max=Q - 1)] - packed["lines"][None]).abs().sum(-1)  # (L, cap)
        l1 = (diff * valid[None]).sum(1) / num_items                                       # (:239-242), padding masked out
        losses = {"loss_ce": ce[0], "loss_line": l1[0]}
        for i in range(L_ - 1):
            losses[f"loss_ce_{i}"] = ce[i + 1]
            losses[f"loss_line_{i}"] = l1[i + 1]
        return losses

    def forward(self, outputs, targets, origin_indices=None, depth_gt=None):
        self.last_indices = []
        n = torch.as_tensor([float(sum(len(t["labels"]) for t in targets))], device=outputs["pred_logits"].device)
        world = 1
        if _dist_ready():                                                                               # :323-326
            torch.distributed.all_reduce(n)
            world = torch.distributed.get_world_size()
        num_items = torch.clamp(n / world, min=1).item()
        handle = outputs.get("_match_prefetch") if hasattr(self.matcher, "from_prefetch") else None
        losses = self._one({k: v for k, v in outputs.items() if k != "aux_outputs"}, targets, num_items, "", handle)
        for i, aux in enumerate(outputs.get("aux_outputs", [])):
            losses.update(self._one(aux, targets, num_items, f"_{i}", handle))
        return losses


def target_capacity(total):
    """Padded column count for `total` targets: powers of two from 64 - a handful of distinct static shapes overall."""
    cap = 64
    while cap < total:
        cap *= 2
    return cap


class PackedTargets(dict):
    """Static-shape device form of a batch's line targets for SetCriterion.forward_packed.

    lines (cap,6) fp32, labels (cap,) int64, meta int32 [col_off (B+1) | image index of every column (cap) | valid (cap)],
    num_items (1,) fp32.  `cap` depends only on the size class of the batch's TOTAL target count, so a HIP graph captured over
    these tensors serves every batch of that class whatever the per-image counts are (the reference's dataset has a different
    number of lines in every image, glassrgbd_norhint.py:184-205).  update() refreshes the contents without a host round
    trip: device->device copies of the line data, the host-known bookkeeping through a small ring of pinned buffers."""
    RING = 8

    def __init__(self, batch_size, cap, device):
        super().__init__()
        self.B, self.cap, self.device = int(batch_size), int(cap), torch.device(device)
        self["lines"] = torch.zeros((cap, 6), dtype=torch.float32, device=device)
        self["labels"] = torch.zeros((cap,), dtype=torch.int64, device=device)
        self["meta"] = torch.zeros((self.B + 1 + 2 * cap,), dtype=torch.int32, device=device)
        self["num_items"] = torch.zeros((1,), dtype=torch.float32, device=device)
        self._ring, self._slot = [], 0

    def _staging(self):
        if self.device.type != "cuda":
            return torch.zeros_like(self["meta"], device="cpu"), None
        if len(self._ring) < self.RING:
            self._ring.append([torch.zeros(self["meta"].shape, dtype=torch.int32, pin_memory=True), None])
            ent = self._ring[-1]
        else:
            ent = self._ring[self._slot % self.RING]
            if ent[1] is not None:
                ent[1].synchronize()               # the copy issued RING updates ago has long finished
        self._slot += 1
        return ent[0], ent

    def update(self, targets):
        sizes = [int(len(t["labels"])) for t in targets]
        total = sum(sizes)
        if len(sizes) != self.B or total > self.cap:
            raise ValueError("PackedTargets(B=%d, cap=%d) cannot hold %r" % (self.B, self.cap, sizes))
        host, ent = self._staging()
        B, cap = self.B, self.cap
        host.zero_()
        off = 0
        for i, n in enumerate(sizes):
            host[i] = off
            host[B + 1 + off:B + 1 + off + n] = i
            off += n
        host[B] = off
        host[B + 1 + cap:B + 1 + cap + total] = 1
        self["meta"].copy_(host, non_blocking=True)
        if ent is not None:
            ent[1] = torch.cuda.Event()
            ent[1].record()
        if total:
            self["lines"][:total].copy_(torch.cat([t["lines"] for t in targets]).to(self.device), non_blocking=True)
            self["labels"][:total].copy_(torch.cat([t["labels"] for t in targets]).to(self.device), non_blocking=True)
        self["num_items"].fill_(float(total))
        self.sizes = sizes
        return self


def pack_targets(targets, device, cap=None):
    """PackedTargets of one batch (capacity: target_capacity(total) unless given)."""
    total = sum(int(len(t["labels"])) for t in targets)
    return PackedTargets(len(targets), cap if cap is not None else target_capacity(total), device).update(targets)


class SilogLoss(nn.Module):
    """criterion_depth(pred, gt, mask_bool): glassrgbd.py:360-374 on the fused masked reduction."""

    def __init__(self, variance_focus=0.85, log_depth_error=True):
        super().__init__()
        self.variance_focus, self.log_depth_error = variance_focus, log_depth_error

    def forward(self, depth_est, depth_gt, mask):
        # the kernel derives validity from the GT range [0.2, 10): express an arbitrary mask through it
        gt = torch.where(mask, depth_gt.float().clamp(0.2, 9.999999), torch.zeros_like(depth_gt, dtype=torch.float32))
        return ops.silog_loss(depth_est, gt, 1.0, self.variance_focus, self.log_depth_error)

    def fused(self, depth_est, depth_gt_full, weight):
        """weight * SiLog against the nearest-resized GT / validity mask of engine_glassrgbd.py:65,76-78."""
        return ops.silog_loss(depth_est, depth_gt_full, weight, self.variance_focus, self.log_depth_error)


class PlaneLoss(nn.Module):
    """criterion_plane(depth_pred, depth_gt, line_pred, line_score, valid_mask): glassrgbd.py:385-450, one image per call.
    The reference selects `top_num` lines with a host sync and rasterises every triangle on the CPU (matplotlib); here the
    count stays a device scalar, the num_ref best lines are always gathered and the kernel ignores those beyond it."""

    def __init__(self, num_ref=28, line_score_thresh=0.6, min_plane_area=100):
        super().__init__()
        self.num_ref, self.line_score_thresh, self.min_plane_area = num_ref, line_score_thresh, min_plane_area
        self._consts = {}

    def forward(self, depth_pred, depth_gt, line_pred, line_score, valid_mask):
        if line_score.shape[0] != 1:
            raise AssertionError("one image each iter")                                   # :397
        H, W = depth_pred.shape[-2:]
        score, logit = line_score.detach().float(), line_score.detach().float()[0, :, 0]
        keep = torch.softmax(score, dim=-1)[0, :, 0] > self.line_score_thresh             # :399-400
        n_planes = keep.sum().clamp(max=self.num_ref).to(torch.int32).reshape(1)          # top_num, on the device
        k = min(self.num_ref, logit.shape[0])
        ids = torch.topk(logit, k)[1]                                                     # :402 (raw class-0 logit)
        key = (H, W, str(line_pred.device))
        if key not in self._consts:          # built once (outside any graph capture: a host list -> device copy is not capturable)
            self._consts[key] = (torch.tensor([W, H, W, H, W, H], dtype=torch.float32, device=line_pred.device),
                                 torch.tensor([W - 1, H - 1] * 3, dtype=torch.float32, device=line_pred.device))
        scale, hi = self._consts[key]
        lines = torch.round(line_pred.detach().float()[0][ids] * scale)                   # :412-414
        tri = torch.minimum(lines.clamp(min=0), hi).to(torch.int64).contiguous()          # :415-417, (k, 6)
        valid = valid_mask.reshape(H, W).to(torch.uint8).contiguous()
        return ops.plane_loss(depth_pred.float(), valid, tri, n_planes, self.min_plane_area)


class SegLoss(nn.Module):
    """criterion_seg(logits (B,2,H,W), target (B,H,W) int64): glassrgbd.py:376-383."""

    def forward(self, seg_pred, seg_gt, scale=1.0):
        return ops.seg_cross_entropy(seg_pred.permute(0, 2, 3, 1), seg_gt, scale)


class PostProcessLine(nn.Module):
    """PostProcess_Line 'prediction' branch, glassrgbd.py:452-479."""

    @torch.no_grad()
    def forward(self, outputs, target_sizes, output_type="prediction"):
        prob = F.softmax(outputs["pred_logits"].float(), -1)
        scores, labels = prob[..., :-1].max(-1)
        img_h, img_w = target_sizes.unbind(1)
        scale = torch.stack([img_w, img_h, img_w, img_h], dim=1)
        lines = outputs["pred_lines"][..., :4] * scale[:, None, :]
        return [{"scores": s, "labels": l, "lines": b} for s, l, b in zip(scores, labels, lines)]
