#!/usr/bin/env python
"""Who issues the ATen kernels left in one train step: forward ops by the gw_depth_amd source line, backward ops by the autograd node."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from gw_depth_amd import Config, build_model
from gw_depth_amd.engine import TrainStep
from gw_depth_amd.synth import det_fill_, synth_batch
from gw_depth_amd.criteria import pack_targets

cfg = Config(device="cuda", dropout=0.1, log_depth_error=True)
model, crits, _ = build_model(cfg)
model.load_state_dict(det_fill_({k: v.detach().clone() for k, v in model.state_dict().items()}, seed=0))
model.cuda(); crits[0].cuda()
step = TrainStep(model, crits, cfg, compute_dtype=torch.bfloat16, check_finite=False)
b = synth_batch(8, 480, 640, seed=1)
batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
batch["targets"] = [{k: v.cuda() for k, v in t.items()} for t in b["targets"]]
st = {k: batch[k].clone() for k in ("images", "pad_mask", "depth", "seg")}
st["packed"] = pack_targets(batch["targets"], "cuda")
def _scoped(name, fwd):
    def run(*a, **k):
        with torch.profiler.record_function("M:" + name):
            return fwd(*a, **k)
    return run


for owner, root in (("model", model), ("crit", crits[0])):
    for name, m in root.named_modules():
        m.forward = _scoped(owner + "." + name + "<" + type(m).__name__ + ">", m.forward)
for _ in range(2):
    step._sync_free_fb(st)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step._sync_free_fb(st)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
for e in prof.events():
    if not e.name.startswith("aten::") or e.self_device_time_total <= 0:
        continue
    who = None
    p = e.cpu_parent
    while p is not None:
        if p.name.startswith("autograd::engine::evaluate_function"):
            who = "bwd " + p.name.split(": ")[-1]
            break
        p = p.cpu_parent
    if who is None:
        p = e.cpu_parent
        while p is not None and not p.name.startswith("M:"):
            p = p.cpu_parent
        who = "fwd " + (p.name[2:] if p is not None else "?")
    k = (e.name, who, str(e.input_shapes)[:70])
    agg[k][0] += e.self_device_time_total
    agg[k][1] += 1
rows = sorted(((v[0], v[1], k) for k, v in agg.items()), reverse=True)
print("aten device time %.2f ms, %d ops" % (sum(r[0] for r in rows) / 1e3, sum(r[1] for r in rows)))
for t, n, (name, who, sh) in rows:
    print("%8.1f us %4d x %-22s %-70s %s" % (t, n, name, who[-70:], sh))
