"""bench.py's own multi-rank launcher, rehearsed on ONE GPU: `--gpus 2` without a launcher environment starts
two child ranks of bench.py; with GWD_BENCH_BACKEND=gloo both may share the card (RCCL refuses two ranks on one device), so the
spawn-before-GPU structure, the per-rank environment, the barrier / max-over-ranks timing, the `comm` leg, rank 0's post-timed legs
and the barrier in front of destroy_process_group all execute.  A rehearsal of the control path, never a measurement."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_2_over_gloo_prints_one_line_with_a_comm_block():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, GWD_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "1",
                        "--height", "96", "--width", "128", "--no-cpu-baseline"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 2 and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and out["ms_per_step"] > 0
    assert "comm" in out and "error" not in out["comm"], out.get("comm")
    assert out["comm"]["backend"] == "gloo" and out["comm"]["allreduce_mb_per_step"] > 200
    assert out["config"]["launch"] == "hipgraph" and out["config"]["graph_segments"] >= 2
