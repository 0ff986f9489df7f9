#!/usr/bin/env python
"""Per-layer-shape time of every convolution / linear forward and data-gradient launch INSIDE the train step, from a rocprofv3 kernel
trace of an eager bench run made with GWD_TRACE_CONV=1 (one stderr line per gwd_conv_forward call, in call order = launch order):

    GWD_TRACE_CONV=1 rocprofv3 --kernel-trace --output-format csv -d D -- python3 bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline 2> conv.log
    python tools/conv_instep.py D conv.log [backbone]

Every gwd_conv_forward call launches exactly one kernel of the families below, so the i-th call is the i-th such kernel of the trace.
With `backbone`: only the ResNet-50 Bottleneck shapes (src/models/backbone.py:90-92), with their sum against the dense bf16 MFMA peak
(north_star: >= 40 % on the backbone conv-GEMMs).  Weight gradients travel in grouped launches and are listed per kernel name instead."""
import collections
import csv
import glob
import os
import re
import sys

FWD_KERNELS = ("igemm_dma_kernel", "igemm_fwd_kernel", "gemm_ksplit_kernel", "tconv_fwd_kernel", "thin_fwd_kernel", "thin_dgrad_kernel")
WGRAD_KERNELS = ("wgrad",)
PEAK = 2500.0


def main():
    d, log = sys.argv[1], sys.argv[2]
    only_backbone = len(sys.argv) > 3 and sys.argv[3] == "backbone"
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    ks = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if any(k in r["Kernel_Name"] for k in FWD_KERNELS)]
    calls = []
    for line in open(log, errors="replace"):
        m = re.match(r"GWDCONV fwd B=(\d+) Hi=(\d+) Wi=(\d+) Cin=(\d+) Ho=(\d+) Wo=(\d+) Cout=(\d+) k=(\d+) s=(\d+) g=(\d+) dt=(\d+)", line)
        if m:
            calls.append(tuple(int(v) for v in m.groups()))
    if len(ks) != len(calls):
        raise SystemExit("trace has %d forward-family kernels, the log %d gwd_conv_forward calls: the 1:1 mapping does not hold" % (len(ks), len(calls)))
    # forward passes in the trace: the fused stem runs once per forward (bench.py adds one forward/backward without an optimizer step)
    passes = sum(1 for r in rows if "stem_kernel" in r["Kernel_Name"]) or (sum(1 for r in rows if "adamw_kernel" in r["Kernel_Name"]) // 2) or 1
    agg = collections.defaultdict(lambda: [0, 0.0])
    for c, us in zip(calls, ks):
        agg[c][0] += 1
        agg[c][1] += us
    size_of = {64: 120, 128: 60, 256: 30, 512: 15}           # planes -> rows of the layer's output map at 480 x 640

    def is_backbone(c):
        """torchvision ResNet-50 v1.5 Bottleneck shapes (stride on the 3x3): conv1 1x1 cin -> p, conv2 3x3 p -> p (stride 1 | 2), conv3 1x1
        p -> 4p, downsample 1x1 cin -> 4p; a data-gradient launch has the same numbers with Cin / Cout and the map sizes swapped.  One
        known collision: dense_input_proj (2048 -> 512 at 15 x 20) has the shape of layer4's conv1 (one of three launches of that row)."""
        B, Hi, Wi, Cin, Ho, Wo, Cout, k, s, g, dt = c
        if Hi * Wi == 1 or B != 8:
            return False
        lo, hi, small = min(Cin, Cout), max(Cin, Cout), min(Hi, Ho)
        if k == 3:
            return Cin == Cout and Cin in size_of and small == size_of[Cin]
        if k != 1 or max(Hi, Ho) not in (120, 60, 30, 15):
            return False
        if (lo, hi) == (64, 64):
            return small == 120
        return lo in (64, 128, 256, 512, 1024) and hi in (256, 512, 1024, 2048) and hi // lo in (2, 4) and hi % lo == 0

    out = []
    for c, (n, us) in agg.items():
        B, Hi, Wi, Cin, Ho, Wo, Cout, k, s, g, dt = c
        if dt != 1 or (only_backbone and not is_backbone(c)):
            continue
        M = B * (Ho * Wo if g != 1 else Hi * Wi)          # a data-gradient launch (g = 1) is described by its INPUT gradient's pixels = Ho x Wo of the desc
        M = B * Ho * Wo
        gflop = 2.0 * M * k * k * Cin * Cout / 1e9
        # algorithmic bytes of the launch: input map + output map + weights (bf16); skip / gate / pre-activation operands of the epilogue
        # are NOT counted (unknown here), so the HBM floor below is a lower bound of what the launch really moves
        mbytes = 2.0 * (B * Hi * Wi * Cin + B * Ho * Wo * Cout + k * k * Cin * Cout) / 1e6
        out.append((us / passes, n / passes, us / n, gflop, c, mbytes))
    out.sort(reverse=True)
    tot_us = sum(o[0] for o in out)
    tot_gf = sum(o[3] * o[1] for o in out)
    print("%s forward + data-gradient launches inside one step (%d passes averaged): %.0f launches, %.2f ms, %.0f GFLOP, %.0f TFLOP/s = %.1f %% of the dense bf16 MFMA peak" %
          ("ResNet-50 Bottleneck" if only_backbone else "all conv / linear", passes, sum(o[1] for o in out), tot_us / 1e3, tot_gf,
           tot_gf / tot_us * 1e3, 100 * tot_gf / tot_us * 1e3 / PEAK))
    floor_us = sum(o[1] * max(o[3] / PEAK * 1e3, o[5] / 6.3) for o in out)      # per launch: max(MFMA time at peak, bytes at 6.3 TB/s)
    print("roofline floor of these launches (per launch max of FLOP / 2500 TFLOP/s and algorithmic bytes / 6.3 TB/s, the copy rate MI355X_MICROARCH.md "
          "measures): %.2f ms = %.0f %% of the measured %.2f ms" % (floor_us / 1e3, 100 * floor_us / tot_us, tot_us / 1e3))
    print("%9s %5s %9s %7s %6s %8s %9s  %s" % ("us/step", "n", "us each", "TF/s", "%MFMA", "MB", "floor us", "(B, Hi, Wi, Cin, Ho, Wo, Cout, k, stride, gather[1 = data gradient])"))
    for us, n, each, gf, c, mb in out[:120]:
        tf = gf / each * 1e3
        print("%9.1f %5.1f %9.1f %7.1f %6.1f %8.1f %9.1f  %s" % (us, n, each, tf, 100 * tf / PEAK, mb, max(gf / PEAK * 1e3, mb / 6.3), c[:10]))
    if not only_backbone:
        wg = collections.defaultdict(lambda: [0, 0.0])
        for r in rows:
            if "wgrad" in r["Kernel_Name"]:
                name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))[:70]
                wg[name][0] += 1
                wg[name][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print("\nweight-gradient kernels per step:")
        for name, (n, us) in sorted(wg.items(), key=lambda kv: -kv[1][1]):
            print("%9.1f us %5.1f x %8.1f us  %s" % (us / passes, n / passes, us / n, name))


if __name__ == "__main__":
    main()
