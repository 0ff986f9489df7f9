"""Checkpoint / weight interchange with the reference.

Mirrors the resume / save logic of /root/reference/src/main_glassrgbd.py:104-193,214-226 so that `.pth` files move
both ways unchanged:

* `remap_resume_state_dict`   - `--resume <file>`: DataParallel `module.` stripping and the old `bbox_embed` ->
  `lines_embed` rename (:131-142), quirks included;
* `filter_detr_state_dict`    - `--resume https://...` (DETR-R50): drop class/bbox/query heads and, unless
  `layer1_num == 3`, `input_proj` (:108-115);
* `FlatAdamW`                 - a `torch.optim.Optimizer` whose state IS TrainStep's flat HBM buffers: its
  `state_dict()` / `load_state_dict()` speak `torch.optim.AdamW`'s format with the reference's two parameter groups
  (:59-66), so `torch.optim.lr_scheduler.StepLR(optimizer, args.lr_drop)` (:67) drives it directly and TrainStep
  reads each group's current `lr` from it;
* `save_checkpoint` / `load_checkpoint` - the dict of :216-222 (`model, optimizer, lr_scheduler, epoch, args`).

The model side needs nothing special: `GlassRGBD.state_dict()/load_state_dict()` already use the reference's 970 key
names and (Cout,Cin,KH,KW) conv layout (gw_depth_amd/layers.py).
"""
import re

import torch

_MODULE = re.compile("module.")          # the reference's own pattern (main_glassrgbd.py:131): '.' matches any character


def remap_resume_state_dict(model_sd, log=None):
    """main_glassrgbd.py:129-142.  NB the bbox_embed branch keeps the reference's arithmetic on the ORIGINAL key:
    'lines_embed.' + everything after the first dot (so `bbox_embed.layers.0.weight` -> `lines_embed.layers.0.weight`,
    and `module.bbox_embed.layers.0.weight` -> `lines_embed.bbox_embed.layers.0.weight`, which then simply does not load)."""
    out = {}
    for k, v in model_sd.items():
        k_wo = re.sub(_MODULE, "", k) if re.search("module", k) else k
        if "bbox_embed" in k:
            if log:
                log("bbox_embed from OLD implementation has been replaced with lines_embed")
            out["lines_embed." + ".".join(k.split(".")[1:])] = v
        else:
            out[k_wo] = v
    return out


def filter_detr_state_dict(model_sd, layer1_num=3):
    """main_glassrgbd.py:108-115: the partial load of a DETR-R50 checkpoint."""
    out = {}
    for k, v in model_sd.items():
        if ("class_embed" in k) or ("bbox_embed" in k) or ("query_embed" in k):
            continue
        if ("input_proj" in k) and layer1_num != 3:
            continue
        out[k] = v
    return out


def new_parameters(model, state_dict):
    """The names the reference prints as '... is a new parameter. Not found from load dict.' (:150-152)."""
    return [n for n, _ in model.named_parameters() if n not in state_dict]


class FlatAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW facade over a TrainStep: same parameter groups as main_glassrgbd.py:59-66, state tensors are
    views of the flat moment buffers (so state_dict() copies nothing until it is serialised), step() runs the fused
    device clip + AdamW kernel."""

    def __init__(self, train_step):
        ts = self.ts = train_step
        cfg = ts.cfg
        named = list(ts.model.named_parameters())
        groups = [{"params": [p for n, p in named if "backbone" not in n and p.requires_grad]},
                  {"params": [p for n, p in named if "backbone" in n and p.requires_grad], "lr": cfg.lr_backbone}]
        proto = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=cfg.lr, weight_decay=cfg.weight_decay)
        super().__init__(groups, dict(proto.defaults))
        self._name_of = {id(p): n for n, p in named}
        from .layers import Conv
        # convolution weights (and so their moments) live kernel-native (Cout,KH,KW,Cin); the reference's are (Cout,Cin,KH,KW)
        self._native = {id(m.weight) for m in ts.model.modules() if isinstance(m, Conv)}
        self._step_t = torch.tensor(float(ts.step_count))
        self._link()
        ts.optimizer = self

    def _link(self):
        ts = self.ts
        for g in self.param_groups:
            for p in g["params"]:
                n = self._name_of[id(p)]
                o, k = ts.offsets[n], p.numel()
                self.state[p] = {"step": self._step_t, "exp_avg": ts.flat_m[o:o + k].view(p.shape),
                                 "exp_avg_sq": ts.flat_v[o:o + k].view(p.shape)}

    def lrs(self):
        return self.param_groups[0]["lr"], self.param_groups[1]["lr"]

    @torch.no_grad()
    def step(self, closure=None):
        self.ts.optimizer_step()
        self._step_t.fill_(float(self.ts.step_count))

    def zero_grad(self, set_to_none=False):
        self.ts.zero_grad()

    def _indexed(self):
        i = 0
        for g in self.param_groups:
            for p in g["params"]:
                yield i, p
                i += 1

    def state_dict(self):
        self._step_t.fill_(float(self.ts.step_count))
        sd = super().state_dict()
        state = {k: dict(v) for k, v in sd["state"].items()}
        for i, p in self._indexed():
            st = state[i]
            st["step"] = st["step"].clone()
            if id(p) in self._native:
                st["exp_avg"] = st["exp_avg"].permute(0, 3, 1, 2).clone(memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = st["exp_avg_sq"].permute(0, 3, 1, 2).clone(memory_format=torch.contiguous_format)
        return {"state": state, "param_groups": sd["param_groups"]}

    def load_state_dict(self, state_dict):
        """Accepts a torch.optim.AdamW state dict built over the reference's groups (same parameter order)."""
        ts = self.ts
        groups = state_dict["param_groups"]
        if len(groups) != 2 or [len(g["params"]) for g in groups] != [len(g["params"]) for g in self.param_groups]:
            raise ValueError("optimizer state does not have the reference's two parameter groups of this model")
        steps = set()
        for g_saved, g in zip(groups, self.param_groups):
            for idx, p in zip(g_saved["params"], g["params"]):
                st = state_dict["state"].get(idx)
                n = self._name_of[id(p)]
                o, k = ts.offsets[n], p.numel()
                if st is None:                                   # a parameter the saved run never updated
                    ts.flat_m[o:o + k].zero_()
                    ts.flat_v[o:o + k].zero_()
                    continue
                m, v = st["exp_avg"], st["exp_avg_sq"]
                if id(p) in self._native:
                    m, v = m.permute(0, 2, 3, 1), v.permute(0, 2, 3, 1)
                ts.flat_m[o:o + k].copy_(m.reshape(-1))
                ts.flat_v[o:o + k].copy_(v.reshape(-1))
                steps.add(int(float(st["step"])))
            for key, val in g_saved.items():
                if key != "params":
                    g[key] = val
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ (%s): not a state this fused AdamW can continue" % sorted(steps))
        ts.step_count = steps.pop() if steps else 0
        self._step_t.fill_(float(ts.step_count))
        self._link()


def save_checkpoint(path, model, optimizer, lr_scheduler, epoch, args=None):
    """The dict of main_glassrgbd.py:216-222; tensors are moved to the host."""
    osd = optimizer.state_dict()
    osd["state"] = {k: {kk: (vv.detach().cpu().clone() if torch.is_tensor(vv) else vv) for kk, vv in v.items()}
                    for k, v in osd["state"].items()}
    torch.save({"model": {k: v.detach().cpu() for k, v in model.state_dict().items()}, "optimizer": osd,
                "lr_scheduler": lr_scheduler.state_dict() if lr_scheduler is not None else None, "epoch": epoch,
                "args": args}, path)


def load_checkpoint(checkpoint, model, optimizer=None, lr_scheduler=None, args=None, log=print):
    """`--resume <file>` (main_glassrgbd.py:128-162).  `checkpoint` is a path or an already loaded dict.  Returns the
    epoch to start from (checkpoint epoch + 1 when optimizer + scheduler state were restored, else None)."""
    if not isinstance(checkpoint, dict):
        checkpoint = torch.load(checkpoint, map_location="cpu", weights_only=False)
    sd = remap_resume_state_dict(checkpoint["model"], log)
    for n in new_parameters(model, sd):
        if log:
            log(n, "is a new parameter. Not found from load dict.")
    model.load_state_dict(sd, strict=False)
    ts = getattr(optimizer, "ts", None)
    if ts is not None and ts.flat_p16 is not None:
        ts.flat_p16.copy_(ts.flat_p)                            # the bf16 shadow follows the loaded weights
    no_opt = bool(getattr(args, "no_opt", False)) or bool(getattr(args, "eval", False))
    if optimizer is not None and lr_scheduler is not None and not no_opt and \
            all(checkpoint.get(k) is not None for k in ("optimizer", "lr_scheduler", "epoch")):
        optimizer.load_state_dict(checkpoint["optimizer"])
        lrs = dict(checkpoint["lr_scheduler"])
        if args is not None and hasattr(args, "lr_drop"):
            lrs["step_size"] = args.lr_drop                     # :159 "change the lr_drop epoch"
        lr_scheduler.load_state_dict(lrs)
        return checkpoint["epoch"] + 1
    return None
