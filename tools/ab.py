#!/usr/bin/env python
"""Same-box A/B of environment switches: tools/ab.py VAR=a,b [VAR2=c,d] [-- bench args]; runs bench.py for every combination, twice,
interleaved, and prints ms_per_step (boxes differ by +-0.5 ms, so only same-call comparisons mean anything)."""
import itertools, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--")
    args, extra = args[:i], args[i + 1:]
names = [a.split("=")[0] for a in args]
values = [a.split("=")[1].split(",") for a in args]
for rep in range(2):
    for combo in itertools.product(*values):
        env = dict(os.environ, **dict(zip(names, combo)))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "5", "--no-cpu-baseline"] + extra,
                             env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout.strip().splitlines()
        ms = json.loads(out[-1])["ms_per_step"] if out else float("nan")
        print(" ".join("%s=%s" % kv for kv in zip(names, combo)), "->", ms, "ms/step", flush=True)
