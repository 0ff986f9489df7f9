"""The drop-in claim, proven with the REFERENCE'S OWN driver code (build container only: /root/reference does not travel).

`models.build_model` of the reference checkout is pointed at `gw_depth_amd.build_model` (the one-line binding INTEGRATION.md
shows); then the reference's UNMODIFIED `engine_glassrgbd.train_one_epoch` (/root/reference/src/engine_glassrgbd.py:22-171)
runs one epoch of one batch over the product model with a real `torch.optim.AdamW` built exactly as
/root/reference/src/main_glassrgbd.py:59-66 builds it, and the result is compared with the golden vectors the reference's own
model produced through the same engine (tests/golden/tiny_b2_96x128.npz).  Second test: the product model inside
`DistributedDataParallel(model, find_unused_parameters=True)` (main_glassrgbd.py:46) over gloo, world size 2.

CPU, host logic only: the device library is tests/fake_device.py (torch math behind the C-ABI's tensor-level calls).
"""
import os
import random
import socket

import numpy as np
import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="the reference checkout exists only in the build container")


def _bind_product_as_reference_models():
    """What a maintainer does in src/models/__init__.py, done in-process."""
    from oracle import ref_stubs
    ref_stubs.install()
    import models                                    # /root/reference/src/models/__init__.py
    import gw_depth_amd
    models.build_model = gw_depth_amd.build_model
    return models


def _one_batch_loader(case):
    from gw_depth_amd.synth import synth_batch
    from util.misc import NestedTensor              # the REFERENCE's NestedTensor (src/util/misc.py:347)
    b = synth_batch(case["batch"], case["height"], case["width"], seed=case["seed"], n_lines=case["n_lines"], sizes=case["sizes"])
    return [(NestedTensor(b["images"], b["pad_mask"]), NestedTensor(b["depth"], b["pad_mask"]), NestedTensor(b["seg"], b["pad_mask"]),
             b["targets"], ["synthetic\n"])]


def _reference_optimizer(model, args):
    """main_glassrgbd.py:59-66, verbatim semantics."""
    param_dicts = [
        {"params": [p for n, p in model.named_parameters() if "backbone" not in n and p.requires_grad]},
        {"params": [p for n, p in model.named_parameters() if "backbone" in n and p.requires_grad], "lr": args.lr_backbone},
    ]
    return torch.optim.AdamW(param_dicts, lr=args.lr, weight_decay=args.weight_decay)


def test_reference_train_one_epoch_over_the_product_model(golden_dir):
    from gw_depth_amd import hip
    from gw_depth_amd.synth import det_fill_
    from oracle import ref_stubs
    from oracle.make_golden import CASES
    from tests.fake_device import FakeDevice
    from tests.helpers import reference_state_shapes
    models = _bind_product_as_reference_models()
    import engine_glassrgbd as eng                   # /root/reference/src/engine_glassrgbd.py, unmodified
    hip.set_library(FakeDevice())
    try:
        torch.manual_seed(0)
        random.seed(0)
        np.random.seed(0)
        case = CASES["tiny_b2_96x128"]
        g = np.load(os.path.join(golden_dir, "tiny_b2_96x128.npz"))
        args = ref_stubs.reference_args()
        model, criterions, postprocessors = models.build_model(args)          # -> gw_depth_amd.build_model(args)
        assert type(model).__module__.startswith("gw_depth_amd")
        model.load_state_dict(det_fill_(reference_state_shapes(), seed=0), strict=True)
        model.to(torch.device(args.device))
        optimizer = _reference_optimizer(model, args)
        before = {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
        eng.show_labels = lambda *a, **k: None
        stats = eng.train_one_epoch(model, criterions, postprocessors, _one_batch_loader(case), optimizer, torch.device("cpu"), 0,
                                    args.clip_max_norm, args, save_dir=None)
        # the loss terms the reference's engine logged, against what it logged for its own model
        for k, v in stats.items():
            assert abs(v - float(g["stat/" + k])) <= 1e-3 * max(1.0, abs(float(g["stat/" + k]))), (k, v, float(g["stat/" + k]))
        # ... and the AdamW update it applied (per-parameter L2 of the step, reference layout independent)
        names = list(g["grad_names"])
        after = dict(model.named_parameters())
        dl2 = np.array([float((after[n].detach() - before[n]).double().norm()) for n in names])
        big = g["grad_l2"] > 1e-6 * g["grad_l2"].max()
        assert np.max(np.abs(dl2[big] - g["step_delta_l2"][big]) / (g["step_delta_l2"][big] + 1e-12)) < 5e-3
        untouched = [n for n in before if n not in set(names)]
        assert sorted(untouched) == list(g["nograd_names"]) and all(torch.equal(after[n], before[n]) for n in untouched)
    finally:
        hip.set_library(None)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ddp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gw_depth_amd import hip
        from gw_depth_amd.synth import det_fill_, synth_batch
        from oracle import ref_stubs
        from tests.fake_device import FakeDevice
        from tests.helpers import reference_state_shapes
        models = _bind_product_as_reference_models()
        import engine_glassrgbd as eng
        from util.misc import NestedTensor
        hip.set_library(FakeDevice())
        torch.manual_seed(rank)                       # main_glassrgbd.py:36: seed + rank
        args = ref_stubs.reference_args()
        model, criterions, postprocessors = models.build_model(args)
        model.load_state_dict(det_fill_(reference_state_shapes(), seed=0), strict=True)
        if rank == 1:
            with torch.no_grad():
                model.class_embed.bias.add_(1.0)      # DDP's constructor must overwrite this with rank 0's
        ddp = torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=True)     # main_glassrgbd.py:46
        optimizer = _reference_optimizer(ddp.module, args)
        b = synth_batch(1, 96, 128, seed=80 + rank, n_lines=[3 + rank])
        loader = [(NestedTensor(b["images"], b["pad_mask"]), NestedTensor(b["depth"], b["pad_mask"]), NestedTensor(b["seg"], b["pad_mask"]),
                   b["targets"], ["synthetic\n"])]
        eng.show_labels = lambda *a, **k: None
        import util.misc as ref_misc
        # the reference's meter synchronisation hard-codes device='cuda' (src/util/misc.py:50): logging only, no GPU here
        ref_misc.SmoothedValue.synchronize_between_processes = lambda self: None
        stats = eng.train_one_epoch(ddp, criterions, postprocessors, loader, optimizer, torch.device("cpu"), 0, args.clip_max_norm, args)
        flat = torch.cat([p.detach().reshape(-1) for p in ddp.module.parameters()])
        ref = flat.clone()
        dist.broadcast(ref, src=0)
        q.put((rank, bool(torch.equal(ref, flat)), bool(torch.isfinite(flat).all()), float(stats["loss"])))
    except BaseException as exc:                      # fail fast instead of leaving the parent to its queue timeout
        q.put((rank, False, False, repr(exc)))
        raise
    finally:
        dist.destroy_process_group()


def test_product_model_inside_distributed_data_parallel_over_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, same, finite, loss in res:
        assert same and finite and loss == loss, (rank, same, finite, loss)     # ranks hold identical parameters after the step
