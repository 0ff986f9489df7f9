#!/usr/bin/env python
"""Census of the row-wise backward launches of one eager train step (activation backward with / without column sums, LayerNorm
backward, column-sum batches): one line per (op, rows, C, flags) with the call count and the time measured by HIP events around
each call - which layers still pay an activation-backward pass of their own."""
import os, sys, collections
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
rec = []


def wrap(name, keyfn):
    orig = getattr(lib, name)

    def spy(*a, **kw):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = orig(*a, **kw)
        e.record()
        rec.append((name, keyfn(*a, **kw), s, e))
        return r
    setattr(lib, name, spy)
    return orig


ACT = {0: "none", 1: "relu", 2: "gelu", 3: "elu", 4: "sigmoid"}
wrap("act_backward", lambda gy, ref, gx, scale, rows, C, act, act_scale: (rows, C, ACT.get(act, act)))
wrap("act_backward_colsum", lambda gy, ref, gx, db, rows, C, act, act_scale, mult=None: (rows, C, ACT.get(act, act), "mult" if mult is not None else ""))
wrap("layernorm_backward", lambda gy, x, g, b, m, r, gx, dg, db, rows, C, gelu, ld=0, gskip=None, elu_input=False: (rows, C, "gelu" if gelu else "", "skip" if gskip is not None else ""))
wrap("layernorm_forward", lambda x, g, b, y, m, r, rows, C, gelu, residual=None, ld=0: (rows, C, "gelu" if gelu else "", "res" if residual is not None else ""))
step._sync_free_fb(st)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, key, s, e in rec:
    k = (name,) + tuple(key)
    n, t = agg.get(k, (0, 0.0))
    agg[k] = (n + 1, t + s.elapsed_time(e) * 1e3)
tot = collections.Counter()
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-22s %-40s x%3d  %8.1f us  (%.1f each)" % (k[0], " ".join(str(v) for v in k[1:]), n, t, t / n))
    tot[k[0]] += t
print({k: round(v) for k, v in tot.items()})
