#!/usr/bin/env python
"""ATen ops left in one sync-free train step, grouped by op and input shapes (what is still plumbing on torch)."""
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
for _ in range(2):
    step._sync_free_fb(st)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step._sync_free_fb(st)
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
rows = [(e.self_device_time_total, e.count, e.key, str(e.input_shapes)[:90]) for e in ka if e.self_device_time_total > 0 and e.key.startswith("aten::")]
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("aten device time total %.2f ms in %d launches-ish" % (tot / 1e3, sum(r[1] for r in rows)))
for t, n, k, sh in rows[:70]:
    print("%8.1f us %4d x %-28s %s" % (t, n, k, sh))
