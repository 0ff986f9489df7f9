#!/usr/bin/env python
"""Two ranks on one GPU (gloo): parameters after four steps in eager and graph mode, with engine.VECTOR_SET_TERMS off / on - which arm moves?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from tests.test_ddp_gpu_graph import _free_port


def worker(rank, world, port, flag, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gw_depth_amd import engine
    from gw_depth_amd.engine import TrainStep
    from gw_depth_amd.synth import synth_batch
    from tests.golden_check import build, to_device
    engine.VECTOR_SET_TERMS = bool(flag)
    counts = [[3 + 2 * rank], [6 - 3 * rank], [2 * (1 - rank)], [5]]
    batches = [to_device(synth_batch(1, 96, 128, seed=30 + rank + 10 * i, n_lines=c), "cuda") for i, c in enumerate(counts)]
    out = {}
    for graph in (False, True):
        cfg, model, crits = build(device="cuda")
        step = TrainStep(model, crits, cfg, compute_dtype=torch.float32, graph=graph, bucket_mb=16.0)
        traj = []
        for b in batches:
            _, total, terms = step(b)
            step.flush()
            torch.cuda.synchronize()
            traj.append((float(total), step.flat_p.double().norm().item(), step.flat_g.double().norm().item()))
        out[graph] = (step.flat_p.clone().cpu(), traj)
    if rank == 0:
        q.put((flag, out))
    dist.destroy_process_group()


if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    res = {}
    for flag in (0, 1):
        q = ctx.Queue()
        port = _free_port()
        ps = [ctx.Process(target=worker, args=(r, 2, port, flag, q)) for r in range(2)]
        for p in ps:
            p.start()
        f, out = q.get(timeout=900)
        for p in ps:
            p.join(timeout=60)
        res[f] = out
    rel = lambda a, b: float((a - b).double().norm() / b.double().norm())
    for f in (0, 1):
        print("flag", f, "graph vs eager:", rel(res[f][True][0], res[f][False][0]))
        for g in (False, True):
            print("   ", "graph" if g else "eager", ["total %.6f |p| %.6f |g| %.6f" % t for t in res[f][g][1]])
    print("eager: flag 1 vs flag 0:", rel(res[1][False][0], res[0][False][0]))
    print("graph: flag 1 vs flag 0:", rel(res[1][True][0], res[0][True][0]))
