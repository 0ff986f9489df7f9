"""world_size-2 data parallelism on CPU (gloo): the bucketed, hook-driven gradient all-reduce of
gw_depth_amd.engine.TrainStep must leave in every rank exactly the SUM of the per-rank gradients (the mean is
folded into the optimizer), on the first (non-overlapped, live-set learning) step and on the hook-driven second
step, and the global num_items normaliser (/root/reference/src/models/glassrgbd.py:321-326) must be shared."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gw_depth_amd import hip
        from gw_depth_amd.engine import TrainStep, never_used
        from gw_depth_amd.synth import synth_batch
        from tests.fake_device import FakeDevice
        from tests.golden_check import build
        hip.set_library(FakeDevice())
        b = synth_batch(1, 64, 96, seed=20 + rank, n_lines=[3 + 2 * rank])     # different data AND target counts per rank

        cfg, model, crits = build()
        local = TrainStep(model, crits, cfg, data_parallel=False)
        assert local.world == 1
        _, loss_local, terms_local = local.forward_backward(b)
        g_local = local.flat_g.clone()
        gathered = [torch.zeros_like(g_local) for _ in range(world)]
        dist.all_gather(gathered, g_local)
        want = sum(gathered)

        cfg2, model2, crits2 = build()
        if rank == 1:                            # the reference seeds rank r with seed + r (main_glassrgbd.py:36): ranks start apart
            with torch.no_grad():
                for p_ in model2.parameters():
                    p_.add_(0.25)
                for b_ in model2.buffers():
                    if b_.is_floating_point():
                        b_.add_(0.5)
        ddp = TrainStep(model2, crits2, cfg2, bucket_mb=16.0)                     # ... and the constructor broadcasts rank 0's
        assert ddp.world == world and len(ddp.buckets) > 3
        assert ddp.names == local.names and ddp.total == local.total
        bcast = bool(torch.equal(ddp.flat_p, local.flat_p)) and all(
            torch.equal(x, y) for x, y in zip(model2.state_dict().values(), model.state_dict().values()))
        counts = []
        launch = ddp._launch
        ddp._launch = lambda bi: (counts.append(bi), launch(bi))[1]
        ddp.forward_backward(b)                 # step 1: counts the hook firings, reduces after backward
        e1 = float((ddp.flat_g - want).abs().max() / want.abs().max())
        seq1, counts[:] = list(counts), []
        n_dead = sum(1 for n in ddp.names if n not in ddp._expect)
        idle_ok = sorted(n for n in ddp.names if n not in ddp._expect) == sorted(n for n in ddp.names if never_used(n))
        early = []
        hook_launch = ddp._launch
        ddp._launch = lambda bi: (early.append((bi, ddp._state is not None)), hook_launch(bi))[1]
        ddp.forward_backward(b)                 # step 2: buckets launched from the hooks, strictly in index order
        e2 = float((ddp.flat_g - want).abs().max() / want.abs().max())
        seq2 = list(counts)
        overlapped = sum(1 for _, during in early if during)
        # step 3: another batch with another target count on this rank only - the collective sequence must not change
        b3 = synth_batch(1, 64, 96, seed=40 + rank, n_lines=[1 + 5 * rank])
        counts[:] = []
        ddp.forward_backward(b3)
        seq3 = list(counts)
        ddp.forward_backward(b)
        # ranks hold bit-identical reduced gradients
        ref = ddp.flat_g.clone()
        dist.broadcast(ref, src=0)
        same = bool(torch.equal(ref, ddp.flat_g))
        # loss_line is normalised by the GLOBAL target count / world (3 + 5) / 2 = 4 on both ranks
        ratio = float(terms_local["loss_line"].detach())
        ddp.optimizer_step()
        p = ddp.flat_p.clone()
        dist.broadcast(p, src=0)
        nb = len(ddp.buckets)
        in_order = seq1 == list(range(nb)) and seq2 == list(range(nb)) and seq3 == list(range(nb))
        q.put((rank, e1, e2, n_dead, overlapped, same, bool(torch.equal(p, ddp.flat_p)), ratio, bcast, idle_ok, in_order))
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_allreduce_matches_sum_of_local_gradients():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, e1, e2, n_dead, overlapped, same, same_p, _, bcast, idle_ok, in_order in res:
        assert bcast, rank                       # every rank starts from rank 0's parameters and buffers
        assert e1 < 1e-5 and e2 < 1e-5, (rank, e1, e2)
        assert n_dead == 54 and idle_ok          # trainable tensors that never get a gradient == engine.never_used
        assert in_order                          # every bucket exactly once per step, in index order, whatever the data
        assert overlapped >= 3 and same and same_p
