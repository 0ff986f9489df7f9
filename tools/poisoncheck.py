#!/usr/bin/env python
"""Eager train step with every free byte of the caching allocator NaN-filled beforehand: any kernel that reads
uninitialised or out-of-bounds memory and lets it reach a result shows up as the first module with a non-finite output."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gw_depth_amd import Config, build_model
from gw_depth_amd.engine import TrainStep
from gw_depth_amd.synth import det_fill_, synth_batch
from gw_depth_amd.criteria import pack_targets

B = int(os.environ.get("B", 8))
cfg = Config(device="cuda", dropout=0.0, log_depth_error=True)
model, crits, _ = build_model(cfg)
model.load_state_dict(det_fill_({k: v.detach().clone() for k, v in model.state_dict().items()}, seed=0))
model.cuda(); crits[0].cuda()
step = TrainStep(model, crits, cfg, compute_dtype=torch.bfloat16, check_finite=False)
b = synth_batch(B, 480, 640, seed=1)
batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
batch["targets"] = [{k: v.cuda() for k, v in t.items()} for t in b["targets"]]
st = {k: batch[k].clone() for k in ("images", "pad_mask", "depth", "seg")}
st["packed"] = pack_targets(batch["targets"], "cuda")
for _ in range(2):
    out, total, terms = step._sync_free_fb(st)
torch.cuda.synchronize()
print("clean total %.6f" % float(total), flush=True)
del out, total, terms


def poison():
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    small = [torch.full((256 * 1024,), float("nan"), device="cuda") for _ in range(int(os.environ.get("SMALL", 3000)))]   # 1 MiB each: small pool
    mid = [torch.full((5 * 1024 * 1024 // 4,), float("nan"), device="cuda") for _ in range(400)]                       # 5 MiB: 20 MiB segments
    free, _ = torch.cuda.mem_get_info()
    big = torch.full((int(free * 0.85) // 4,), float("nan"), device="cuda")
    torch.cuda.synchronize()
    del small, mid, big


def finite(o):
    if torch.is_tensor(o):
        return bool(torch.isfinite(o).all()) if o.is_floating_point() else True
    if isinstance(o, (list, tuple)):
        return all(finite(t) for t in o)
    if isinstance(o, dict):
        return all(finite(t) for t in o.values())
    if hasattr(o, "tensors"):
        return finite(o.tensors)
    return True


first = []
names = {m: n for n, m in model.named_modules()}
def hook(m, inp, o):
    if not first and not finite(o):
        first.append((names[m], type(m).__name__, finite(inp)))
        print("FIRST non-finite output: module %r (%s) inputs finite=%s" % first[0], flush=True)
hs = [m.register_forward_hook(hook) for m in model.modules()]
poison()
out, total, terms = step._sync_free_fb(st)
torch.cuda.synchronize()
print("poisoned total", float(total), "grads finite", bool(torch.isfinite(step.flat_g).all()), flush=True)
if not first and not torch.isfinite(step.flat_g).all():
    bad = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print("forward clean; params with non-finite grads:", len(bad), bad[:10])
    print("terms", {k: float(v) for k, v in terms.items() if not torch.isfinite(v)})
