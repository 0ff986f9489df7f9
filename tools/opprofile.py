#!/usr/bin/env python
"""Per-call device time of every C-ABI entry point during one bench-shaped train step, aggregated by
(entry point, shape).  Diagnostic only: each call is bracketed by HIP events and synchronised."""
import collections
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gw_depth_amd import Config, build_model, hip
from gw_depth_amd.engine import TrainStep
from gw_depth_amd.synth import det_fill_, synth_batch


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dtype = torch.bfloat16
    lib = hip.library()
    stats = collections.defaultdict(lambda: [0, 0.0, 0.0])
    enabled = [False]

    def wrap(name):
        fn = getattr(lib, name)

        def inner(*a, **k):
            if not enabled[0]:
                return fn(*a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = fn(*a, **k)
            e.record()
            torch.cuda.synchronize()
            ms = s.elapsed_time(e)
            if name in ("conv_forward", "conv_wgrad"):
                dims = a[3]
                Bq, Hi, Wi, Cin, Ho, Wo, Cout, KH, KW = dims
                flops = 2.0 * Bq * Ho * Wo * Cout * KH * KW * Cin
                key = (name, "g%d" % k.get("gather", 0), dims, k.get("stride", 1))
            else:
                flops = 0.0
                key = (name, tuple(t.shape if torch.is_tensor(t) else t for t in a[:2])[:2])
            st = stats[key]
            st[0] += 1
            st[1] += ms
            st[2] += flops
            return r
        setattr(lib, name, inner)

    for n in ("conv_forward", "conv_wgrad", "layernorm_forward", "layernorm_backward", "colsum", "act_backward",
              "softmax_forward", "softmax_backward", "winattn_forward", "winattn_backward", "resample_forward",
              "resample_backward", "avgpool_forward", "avgpool_backward", "weight_prep"):
        wrap(n)
    cfg = Config(device="cuda", dropout=0.1, log_depth_error=True)
    model, crits, _ = build_model(cfg)
    model.load_state_dict(det_fill_({k: v.detach().clone() for k, v in model.state_dict().items()}, seed=0))
    model.cuda()
    crits[0].cuda()
    step = TrainStep(model, crits, cfg, compute_dtype=dtype)
    b = synth_batch(B, 480, 640, seed=1)
    batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    batch["targets"] = [{k: v.cuda() for k, v in t.items()} for t in b["targets"]]
    step(batch)
    enabled[0] = True
    step(batch)
    enabled[0] = False
    rows = sorted(stats.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for _, v in rows)
    print("total timed ms: %.2f" % tot)
    byname = collections.defaultdict(float)
    for k, v in rows:
        byname[k[0]] += v[1]
    print({k: round(v, 2) for k, v in sorted(byname.items(), key=lambda kv: -kv[1])})
    for k, v in rows[:70]:
        tf = v[2] / (v[1] * 1e-3) / 1e12 if v[1] > 0 and v[2] > 0 else 0
        print("%8.3f ms  n=%3d  %7.1f TF/s  %s" % (v[1], v[0], tf, k))


if __name__ == "__main__":
    main()
