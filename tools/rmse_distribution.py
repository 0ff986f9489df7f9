"""Prints the depth-RMSE deviation table of tests/test_configs.py::rmse_distribution (fp32 mode, bf16 mode, bf16-weights floor)
over eight (weight seed, data seed) pairs; the output is committed as profiles/r03_bf16_rmse_distribution.txt."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from tests.test_configs import rmse_distribution

rows = rmse_distribution()
print("weight_seed data_seed oracle_rms     |fp32-ref|  |bf16-ref|  floor(|oracle, bf16 weights - ref|)")
for r in rows:
    print("%11d %9d %10.6f  %10.2e  %10.2e  %10.2e" % (r["weight_seed"], r["data_seed"], r["oracle_rms"], r["fp32"], r["bf16"], r["floor"]))
bf, fl, fp = (np.array([r[k] for r in rows]) for k in ("bf16", "floor", "fp32"))
print("fp32 : mean %.2e  max %.2e   (bar 1e-3)" % (fp.mean(), fp.max()))
print("bf16 : mean %.2e  max %.2e   (stated bound 3e-2)" % (bf.mean(), bf.max()))
print("floor: mean %.2e  max %.2e   (reference fp32 arithmetic, weight matrices rounded to bf16)" % (fl.mean(), fl.max()))
