#!/usr/bin/env python
"""Which ATen ops issue hipMemsetAsync in one sync-free train step (memset nodes do not replay correctly in HIP graphs)."""
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
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("hipMemsetAsync", "hipMemcpyAsync", "hipMemcpyWithStream"):
        p = e.cpu_parent
        chain = []
        while p is not None and len(chain) < 3:
            chain.append(p.name + (str(p.input_shapes)[:80] if len(chain) == 0 else ""))
            p = p.cpu_parent
        cnt[(e.name, " <- ".join(chain))] += 1
for k, v in cnt.most_common(60):
    print(v, k[0], "|", k[1])
