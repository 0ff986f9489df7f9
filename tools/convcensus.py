#!/usr/bin/env python
"""Census of the conv / linear launches of one train step: the library prints one line per call (GWD_TRACE_CONV=1, stderr of a child
process), this groups them by shape and times each distinct shape back to back.  usage: tools/convcensus.py [min_us]"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
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
st = {k: b[k].cuda() for k in ("images", "pad_mask", "depth", "seg")}
st["packed"] = pack_targets([{k: v.cuda() for k, v in t.items()} for t in b["targets"]], "cuda")
step._sync_free_fb(st)
torch.cuda.synchronize()
sys.stderr.write("GWDCONV ---\n")
step._sync_free_fb(st)
torch.cuda.synchronize()
''' % ROOT


def main():
    env = dict(os.environ, GWD_TRACE_CONV="1")
    err = subprocess.run([sys.executable, "-c", CHILD], env=env, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True).stderr
    lines = [l for l in err.splitlines() if l.startswith("GWDCONV")]
    cut = max(i for i, l in enumerate(lines) if l.endswith("---"))
    count = collections.Counter(lines[cut + 1:])
    import torch
    from gw_depth_amd import hip
    lib = hip.library()
    rows = []
    for line, n in count.items():
        what = line.split()[1]
        f = {k: int(v) for k, v in re.findall(r"(\w+)=(-?\d+)", line)}
        if f["dt"] != 1:
            continue
        B, Hi, Wi, Ci, Ho, Wo, Co, k, s, g = (f[x] for x in ("B", "Hi", "Wi", "Cin", "Ho", "Wo", "Cout", "k", "s", "g"))
        x = torch.randn(B, Hi, Wi, Ci, device="cuda").bfloat16()
        y = torch.randn(B, Ho, Wo, Co, device="cuda").bfloat16()
        dims = (B, Hi, Wi, Ci, Ho, Wo, Co, k, k)
        pad = k // 2
        if what == "fwd":
            w = torch.randn(Co, k, k, Ci, device="cuda").bfloat16()
            if g == 2:
                continue
            fn = lambda: lib.conv_forward(x, w, y, dims, stride=s, pad=pad, gather=g)
        else:
            dw = torch.zeros(Co, k, k, Ci, device="cuda")
            if g != 0:
                continue
            fn = lambda: lib.conv_wgrad(x, y, dw, dims, stride=s, pad=pad)
        try:
            for _ in range(3):
                fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                fn()
            b.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(b) * 100
        except Exception as exc:
            us = float("nan")
        fl = 2.0 * B * Ho * Wo * Co * k * k * Ci
        rows.append((us * n, n, us, fl / us / 1e6 if us == us else 0.0, what, (B, Hi, Wi, Ci, Ho, Wo, Co, k, s, g)))
    rows.sort(reverse=True)
    lim = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
    print("%d calls, %d shapes; back-to-back total %.2f ms" % (sum(count.values()), len(count), sum(r[0] for r in rows) / 1e3))
    print("  total us   n    each us   TF/s  kind   (B, Hi, Wi, Cin, Ho, Wo, Cout, k, stride, gather)")
    for tot, n, us, tf, what, shp in rows:
        if tot >= lim:
            print("%9.1f %4d %9.1f %6.1f  %-5s  %s" % (tot, n, us, tf, what, shp))


if __name__ == "__main__":
    main()
