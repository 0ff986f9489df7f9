#!/usr/bin/env python
"""torch.profiler view of one train step: which ATen ops (device plumbing still on torch) cost launches / time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from gw_depth_amd import Config, build_model
from gw_depth_amd.engine import TrainStep
from gw_depth_amd.synth import det_fill_, synth_batch

cfg = Config(device="cuda", dropout=0.1, log_depth_error=True)
model, crits, _ = build_model(cfg)
model.load_state_dict(det_fill_({k: v.detach().clone() for k, v in model.state_dict().items()}, seed=0))
model.cuda(); crits[0].cuda()
step = TrainStep(model, crits, cfg, compute_dtype=torch.bfloat16, check_finite=False)
b = synth_batch(8, 480, 640, seed=1)
batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
batch["targets"] = [{k: v.cuda() for k, v in t.items()} for t in b["targets"]]
for _ in range(3):
    step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(batch)
    torch.cuda.synchronize()
ka = prof.key_averages()
print(ka.table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=48))
print(ka.table(sort_by="self_cuda_time_total", row_limit=30, max_name_column_width=48))
