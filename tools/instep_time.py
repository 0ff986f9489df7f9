#!/usr/bin/env python
"""Times the dominant conv launches INSIDE an eager train step with HIP events around each call, then re-runs the very same calls
(same tensors, same addresses, same data) back to back after the step: separates 'what the launch sees in the step' from 'what the
operands are'."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gw_depth_amd import Config, build_model, hip
from gw_depth_amd.engine import TrainStep
from gw_depth_amd.synth import det_fill_, synth_batch
from gw_depth_amd.criteria import pack_targets

cfg = Config(device="cuda", dropout=0.1, log_depth_error=True)
model, crits, _ = build_model(cfg)
model.load_state_dict(det_fill_({k: v.detach().clone() for k, v in model.state_dict().items()}, seed=0))
model.cuda(); crits[0].cuda()
step = TrainStep(model, crits, cfg, compute_dtype=torch.bfloat16, check_finite=False)
b = synth_batch(8, 480, 640, seed=1)
st = {k: b[k].cuda() for k in ("images", "pad_mask", "depth", "seg")}
st["packed"] = pack_targets([{k: v.cuda() for k, v in t.items()} for t in b["targets"]], "cuda")
for _ in range(2):
    step._sync_free_fb(st)
torch.cuda.synchronize()

lib = hip.library()
orig = lib.conv_forward
rec = []


def spy(x, w, y, dims, **kw):
    hit = tuple(dims) == (8, 120, 160, 160, 120, 160, 160, 3, 3)
    if hit:
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
    orig(x, w, y, dims, **kw)
    if hit:
        e.record()
        rec.append((a, e, x, w, y, dims, dict(kw)))


lib.conv_forward = spy
step._sync_free_fb(st)
torch.cuda.synchronize()
lib.conv_forward = orig
print("in-step launches (us):", " ".join("%.0f" % (a.elapsed_time(e) * 1e3) for a, e, *_ in rec))
again = []
for a, e, x, w, y, dims, kw in rec:
    for _ in range(2):
        orig(x, w, y, dims, **kw)
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    for _ in range(5):
        orig(x, w, y, dims, **kw)
    s1.record()
    torch.cuda.synchronize()
    again.append(s0.elapsed_time(s1) / 5 * 1e3)
print("same calls re-run x5 (us):", " ".join("%.0f" % t for t in again))
for a, e, x, w, y, dims, kw in rec[:3] + rec[-2:]:
    xf = x.float()
    print("  gather %s x: mean %.3f std %.3f zeros %.1f%% | w std %.4f" % (kw.get("gather", 0), float(xf.mean()), float(xf.std()), 100 * float((xf == 0).float().mean()), float(w.float().std())))
