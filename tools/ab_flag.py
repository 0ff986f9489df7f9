#!/usr/bin/env python
"""Same-box A/B of a module-level switch of the Python layer: tools/ab_flag.py gw_depth_amd.layers.GELU_GATE [bench args] runs
bench.py with the flag False / True, twice, interleaved (each in its own process), and prints ms_per_step."""
import importlib, json, os, runpy, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    mod, name = sys.argv[2].rsplit(".", 1)
    setattr(importlib.import_module(mod), name, sys.argv[3] == "1")
    sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[4:]
    runpy.run_path(sys.argv[0], run_name="__main__")
else:
    extra = sys.argv[2:] or ["--steps", "30", "--warmup", "5", "--no-cpu-baseline"]
    for rep in range(2):
        for v in ("0", "1"):
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", sys.argv[1], v] + extra, stdout=subprocess.PIPE,
                                 stderr=subprocess.DEVNULL, text=True).stdout.strip().splitlines()
            print(sys.argv[1], "=", v, "->", json.loads(out[-1])["ms_per_step"] if out else float("nan"), "ms/step", flush=True)
