#!/usr/bin/env python
"""Host (enqueue) time vs device time of one train step: is the step launch-bound?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gw_depth_amd import Config, build_model, hip
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
for it in range(4):
    t0 = time.perf_counter()
    step(batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host enqueue %.1f ms, + drain %.1f ms, total %.1f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t2 - t0)))
# split: forward / losses / backward / optimizer host time
from gw_depth_amd.model import NestedTensor
torch.cuda.synchronize()
t0 = time.perf_counter()
match = (step.criterion.matcher, batch["targets"])
out = model(NestedTensor(batch["images"], batch["pad_mask"]), match=match)
t1 = time.perf_counter()
total, terms = step.losses(out, batch["depth"], batch["seg"], batch["targets"])
t2 = time.perf_counter()
step.zero_grad(); total.backward()
t3 = time.perf_counter()
step.optimizer_step()
t4 = time.perf_counter()
torch.cuda.synchronize()
t5 = time.perf_counter()
print("host: forward %.1f  losses %.1f  backward %.1f  optimizer %.1f  drain %.1f ms" % tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)))
