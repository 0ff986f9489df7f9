#!/usr/bin/env python
"""bench.py — images/s of the full GW-Depth train step (fwd + 17 losses + bwd + clip + AdamW
[+ RCCL gradient all-reduce when N > 1]) on synthetic 480x640 batches, B=8 per GPU, bf16 storage /
fp32 accumulate (BASELINE.json configs[1]).  One process per GPU; prints ONE JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 8] [--dtype bf16|fp32] [--no-cpu-baseline]

`--gpus N` with N > 1 and no launcher environment (WORLD_SIZE unset): this process starts N child ranks of itself
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set, one per GPU, RCCL) BEFORE touching the GPU,
forwards rank 0's JSON line and exits with the worst child status.  Under `python -m torch.distributed.run` the ranks
already exist and the environment is used as is.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

# "nccl" is RCCL on ROCm.  GWD_BENCH_BACKEND=gloo is a REHEARSAL switch only: it lets several ranks share one GPU (RCCL refuses
# two ranks on one device), which is how the launcher / DDP path is exercised on a one-GPU box; never a measurement.
BACKEND = os.environ.get("GWD_BENCH_BACKEND", "nccl")
PEAK_MFMA_BF16_TFLOPS = 2500.0     # dense bf16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBPS = 8000.0
STEP_GFLOP_PER_IMAGE = 1049.5      # fwd + losses + bwd, conv/mm/addmm/bmm only
BACKBONE_GFLOP_PER_IMAGE = 128.4   # 50.05 fwd + 78.33 bwd
HBM_STEP_FILE = "r03_hbm_step.json"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", dest="graph", action="store_false",
                    help="launch every kernel eagerly instead of replaying zero_grad+forward+losses+backward from one HIP graph")
    ap.add_argument("--cpu-baseline-budget-s", type=float, default=25.0)
    ap.add_argument("--segments", action="store_true",
                    help="cut the captured step at gradient-bucket boundaries even on one GPU (what N > 1 does for the overlap)")
    ap.add_argument("--bucket-mb", type=float, default=32.0)
    return ap.parse_args()


def launch_ranks(a):
    """Parent of a `--gpus N` run without a launcher: N children of this very script, one per GPU.  Nothing here initialises
    the GPU (torch.cuda.device_count() does not, on this image); no exec from a GPU process anywhere."""
    n = torch.cuda.device_count()
    if n < a.gpus and BACKEND == "nccl":
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible" % (a.gpus, n))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    worst, deadline = 0, None
    alive = list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            rc = p.poll()
            if rc is None:
                continue
            alive.remove(p)
            if rc != 0:
                worst = worst or rc
                deadline = deadline or time.time() + 30.0      # a rank died: the others hang in a collective - give them 30 s
        if deadline is not None and time.time() > deadline:
            for p in alive:
                p.kill()                                       # exact PIDs of our own children
    raise SystemExit(worst)


def cpu_baseline(sd_cpu, cfg, budget_s):
    """The oracle (CPU restatement of the reference, checked against the reference's golden vectors) timed on
    the host cores: a bounded sample — ONE fp32 train step at batch 1, 480x640, same synthetic recipe."""
    from gw_depth_amd.synth import synth_batch
    from oracle import gwdepth_ref as R
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))           # the GPU box gives one GPU a 16-core CPU share
    torch.set_num_threads(cores)
    ocfg = R.Cfg(dropout=cfg.dropout, log_depth_error=cfg.log_depth_error)
    b = synth_batch(1, 480, 640, seed=1)
    sd = {k: v.clone() for k, v in sd_cpu.items()}
    opt = {}
    t0 = time.time()
    n = 0
    while True:
        R.train_step(sd, b, ocfg, opt_state=opt, step=n + 1, training=True)
        n += 1
        el = time.time() - t0
        if el + el / n > budget_s or n >= 8:
            break
    return {"value": round(n / el, 4), "unit": "images/s", "cores": cores, "kind": "port", "batch": 1, "cpu_model": cpu_model(),
            "sample": "%d fp32 train step(s), batch 1, 480x640, oracle/gwdepth_ref.py on %d host threads (%.1f s)" % (n, cores, el)}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def fp32_leg(sd_cpu, cfg, a, steps=3):
    """ms per step of the fp32 PARITY mode (the mode that meets north_star's 1e-3 tolerance) on the same workload, beside the timed
    bf16 number: a fresh model + TrainStep, eager launches, 1 untimed + `steps` timed steps bracketed by synchronize."""
    from gw_depth_amd import build_model
    from gw_depth_amd.engine import TrainStep
    from gw_depth_amd.synth import synth_batch
    model, crits, _ = build_model(cfg)
    model.load_state_dict(sd_cpu)
    model.cuda()
    crits[0].cuda()
    step = TrainStep(model, crits, cfg, compute_dtype=torch.float32, graph=False)
    b = synth_batch(a.batch, a.height, a.width, seed=1)
    batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    batch["targets"] = [{k: v.cuda() for k, v in t.items()} for t in b["targets"]]
    step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(batch)
    torch.cuda.synchronize()
    ms = 1000 * (time.perf_counter() - t0) / steps
    return {"ms_per_step": round(ms, 2), "images_per_s": round(a.batch * 1000 / ms, 2), "steps": steps, "launch": "eager",
            "note": "fp32 parity mode (exact v_mfma_f32_32x32x2_f32), same workload; the mode whose depth RMSE is within 1e-3 of the reference"}


def depth_rmse_leg(sd_cpu, cfg, dtype):
    """The other half of BASELINE.json's metric: depth RMSE (`rms` of compute_depth_errors, src/util/metrics.py:203-204,
    after evaluate()'s clamp + validity mask) of the product against the oracle on identical weights and inputs - ONE
    480x640 image, eval mode.  Product: HIP forward in the bench dtype and in fp32, metrics by gwd_eval_accumulate on the
    device.  Oracle (checker): CPU fp32 forward + oracle/eval_ref.py.  Outside the timed region."""
    from gw_depth_amd import build_model
    from gw_depth_amd.evaluate import DenseMetrics
    from gw_depth_amd.model import NestedTensor
    from gw_depth_amd.synth import synth_batch
    from oracle import eval_ref
    from oracle import gwdepth_ref as R
    b = synth_batch(1, 480, 640, seed=1)
    with torch.no_grad():
        ref = R.forward({k: v.clone() for k, v in sd_cpu.items()}, b["images"], b["pad_mask"],
                        R.Cfg(dropout=cfg.dropout, log_depth_error=cfg.log_depth_error), training=False)
    per_image, _ = eval_ref.evaluate_dense(ref["pred_depth"][-1].numpy(), b["depth"].numpy(), ref["pred_seg"].numpy(), b["seg"].numpy())
    out = {"oracle_fp32": float(per_image[0, 3]), "sample": "1 image 480x640, weight seed 0, data seed 1, eval mode"}
    # what bf16 STORAGE OF THE WEIGHTS alone costs with the reference's own fp32 arithmetic (the CPU oracle, weight matrices
    # rounded to bf16, everything else fp32): the floor of any bf16 mode on this sample - DESIGN.md "precision policy"
    with torch.no_grad():
        sd_r = {k: (v.bfloat16().float() if (v.is_floating_point() and v.dim() >= 2) else v.clone()) for k, v in sd_cpu.items()}
        ref_r = R.forward(sd_r, b["images"], b["pad_mask"], R.Cfg(dropout=cfg.dropout, log_depth_error=cfg.log_depth_error), training=False)
    per_r, _ = eval_ref.evaluate_dense(ref_r["pred_depth"][-1].numpy(), b["depth"].numpy(), ref_r["pred_seg"].numpy(), b["seg"].numpy())
    out["oracle_fp32_arithmetic_bf16_weights"] = float(per_r[0, 3])
    out["abs_diff_oracle_bf16_weights"] = abs(float(per_r[0, 3]) - out["oracle_fp32"])
    model, _, _ = build_model(cfg)
    model.load_state_dict(sd_cpu)
    model.cuda().eval()
    for name, dt in (("product_fp32", torch.float32), ("product_" + ("bf16" if dtype == torch.bfloat16 else "fp32"), dtype)):
        if name in out:
            continue
        model.compute_dtype = dt
        with torch.no_grad():
            o = model(NestedTensor(b["images"].cuda(), b["pad_mask"].cuda()))
        dm = DenseMetrics("cuda")
        dm.update(o["pred_depth"][-1], b["depth"].cuda(), o["pred_seg"], b["seg"].cuda())
        out[name] = dm.compute()["rms"]
        out[name.replace("product", "abs_diff")] = abs(out[name] - out["oracle_fp32"])
    return out


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(a)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and rank == 0:
        print("[bench] --gpus %d but the launcher started %d rank(s): reporting n_gpus = %d" % (a.gpus, world, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    local = local % torch.cuda.device_count() if BACKEND != "nccl" else local
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if BACKEND == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(BACKEND, rank=rank, world_size=world)

    from gw_depth_amd import Config, build_model, hip
    from gw_depth_amd.engine import TrainStep
    from gw_depth_amd.synth import det_fill_, synth_batch
    lib = hip.library()
    assert not getattr(lib, "is_fake", False)

    cfg = Config(device="cuda", dropout=0.1, log_depth_error=True)
    model, crits, _ = build_model(cfg)
    sd_cpu = det_fill_({k: v.detach().clone() for k, v in model.state_dict().items()}, seed=0)   # weight seed 0
    model.load_state_dict(sd_cpu)
    model.cuda()
    crits[0].cuda()
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    step = TrainStep(model, crits, cfg, compute_dtype=dtype, graph=a.graph, bucket_mb=a.bucket_mb,
                     segments=True if a.segments else None)
    b = synth_batch(a.batch, a.height, a.width, seed=1 + rank)                                   # data seed 1 + rank
    batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    batch["targets"] = [{k: v.cuda() for k, v in t.items()} for t in b["targets"]]
    torch.manual_seed(100 + rank)

    for _ in range(a.warmup):
        step(batch)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        _, total, _ = step(batch)
    step.flush()                    # graph mode examines each loss one step late: all of them are inside the timed region
    barrier()
    el = time.perf_counter() - t0
    t = torch.tensor([el], device="cuda", dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())
    ips = a.batch * world * a.steps / el

    comm = None
    if world > 1:
        # exposed communication: the same K steps with the bucket all-reduces switched off (every rank does this together);
        # outside the timed region, reported beside the number, not part of it
        try:
            launch, step._launch = step._launch, (lambda bi: None)
            barrier()
            t1 = time.perf_counter()
            for _ in range(a.steps):
                step(batch)
            step.flush()
            barrier()
            t = torch.tensor([time.perf_counter() - t1], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            step._launch = launch
            ms_off = 1000 * float(t.item()) / a.steps
            comm = {"ms_per_step_without_allreduce": round(ms_off, 3), "exposed_allreduce_ms": round(1000 * el / a.steps - ms_off, 3),
                    "allreduce_mb_per_step": round(step.total * 4 / 1e6, 1), "reduce_dtype": "fp32", "backend": BACKEND,
                    "overlap": "graph chain cut at bucket boundaries" if a.graph else "post-accumulate hooks"}
        except Exception as exc:        # never let the side measurement take the bench line down
            comm = {"error": repr(exc)}

    roofline = None
    if rank == 0:
        roofline = dominant_kernel_roofline(lib, dtype)
        if world == 1:
            try:                    # the same kernel INSIDE the step (cold operands, neighbours on the chip): one eager step with events
                roofline.update(in_step_kernel_time(step, batch, roofline["algorithmic_gflop_per_launch"], roofline["peak"]))
            except Exception as exc:
                roofline["in_step_error"] = repr(exc)
        # (N > 1: an extra step on rank 0 alone would issue the step's collectives with no partner - the other ranks are waiting at the
        # final barrier - so the in-step figure is a one-GPU measurement only)
    out = {
        "metric": "images/sec (train fwd+bwd) %dx%d bs=%d/GPU" % (a.height, a.width, a.batch),     # BASELINE.json's metric at the defaults
        "value": round(ips, 3), "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(1000 * el / a.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": "GW-Depth full model (res50 + DETR 6+6 + 4-scale ReferTransformer + decoder) "
                               "fwd+17 losses+bwd+clip+AdamW, %dx%d, bs=%d/GPU, dropout 0.1" % (a.height, a.width, a.batch),
                   "global_batch": a.batch * world, "parallelism": "dp%d" % world,
                   "step_gflop_per_image": STEP_GFLOP_PER_IMAGE,
                   "whole_step_mfma_frac": round(ips / world * STEP_GFLOP_PER_IMAGE / 1000.0 / PEAK_MFMA_BF16_TFLOPS, 5),
                   "launch": "hipgraph" if a.graph and step._graphs and all(e["graph"] is not None for e in step._graphs.values()) else "eager",
                   "graph_segments": max([len(e["graph"]) for e in step._graphs.values() if e["graph"] is not None] or [0]),
                   "grad_buckets": len(step.buckets),
                   "final_loss": round(float(total), 4)},
        "roofline": roofline,
    }
    if comm is not None:
        out["comm"] = comm
    try:                            # step-level HBM-side traffic: PMC bytes of one step (committed profile) over the measured step time
        with open(os.path.join(ROOT, "profiles", HBM_STEP_FILE)) as f:
            hrec = json.load(f)
        hb = hrec["hbm_bytes_per_step"]
        if a.batch == 8 and a.height == 480 and a.width == 640 and a.dtype == "bf16":
            gbps = hb / (el / a.steps) / 1e9
            out["hbm"] = {"bytes_per_step_pmc": hb, "achieved_GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / PEAK_HBM_GBPS, 4),
                          "bytes_measured": "%s, commit %s - a committed PMC measurement divided by THIS run's step time" % (hrec.get("measured", "?"), hrec.get("commit", "?")),
                          "source": "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 correction)" % HBM_STEP_FILE}
    except (OSError, KeyError, ValueError):
        pass
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        print("[bench] GPU leg done: %.2f images/s; timing the CPU baseline (oracle) ..." % ips, file=sys.stderr, flush=True)
        out["cpu_baseline"] = cpu_baseline(sd_cpu, cfg, a.cpu_baseline_budget_s)
        out["depth_rmse"] = depth_rmse_leg(sd_cpu, cfg, dtype)
        if a.dtype == "bf16":
            try:
                del step
                torch.cuda.empty_cache()
                out["fp32_parity_mode"] = fp32_leg(sd_cpu, cfg, a)
            except Exception as exc:
                out["fp32_parity_mode"] = {"error": repr(exc)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        # rank 0's post-timed legs (roofline, in-step trace) run while the other ranks wait HERE: nobody tears the group down under a
        # rank that may still issue a collective, and a rank that died makes the barrier raise on the others (non-zero exit)
        dist.barrier()
        dist.destroy_process_group()


def in_step_kernel_time(step, batch, gflop, peak):
    """Duration of the dominant kernel's launches inside ONE train step: the step runs once eagerly on the graph stream (same kernels,
    same order, same operands as the replayed graph) with a HIP event pair around every conv launch of exactly the dominant shape
    (3x3, 160 -> 160 channels at 8 x 120 x 160: 10 forward + 10 data-gradient launches).  Matching by kernel NAME would average over
    other layers that run the same template (the 800 -> 320 layer takes 7x longer) - tools/instep_time.py is the stand-alone version."""
    from gw_depth_amd import hip
    lib = hip.library()
    orig = lib.conv_forward
    rec = []

    def spy(x, w, y, dims, **kw):
        # the plain launches of the shape: since round 3 the ten FORWARD launches carry the LayerNorm epilogue (another kernel variant,
        # +40 us of epilogue), the ten data-gradient launches are the kernel the roofline leg times
        hit = tuple(dims) == (8, 120, 160, 160, 120, 160, 160, 3, 3) and kw.get("ln") is None
        if hit:
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(torch.cuda.current_stream())
        orig(x, w, y, dims, **kw)
        if hit:
            e.record(torch.cuda.current_stream())
            rec.append((a, e))

    lib.conv_forward = spy
    try:
        if step.use_graph:
            step._on_graph_stream(batch, None)
        else:
            step.forward_backward(batch)
        torch.cuda.synchronize()
    finally:
        del lib.conv_forward                               # back to the class's method
    d = [a.elapsed_time(e) for a, e in rec]
    if not d:
        return {"in_step_error": "no launch of the dominant shape in the step"}
    ms = sum(d) / len(d)
    ach = gflop / ms                                   # GFLOP per ms = TFLOP/s
    return {"in_step_launch_ms": round(ms, 4), "in_step_launches": len(d), "in_step_achieved": round(ach, 2), "in_step_frac": round(ach / peak, 4)}


def hip_gather_transposed():
    from gw_depth_amd import hip
    return hip.GATHER_TRANSPOSED


def dominant_kernel_roofline(lib, dtype):
    """Times the dominant kernel of the step — the 3x3 160->160 ConvLn convolution of
    point_based_pred2.pyramid at 1/4 resolution (B=8: M = 8*120*160 = 153600 output pixels; 10 forward
    instances + the K=7200 lastconv make this pyramid 52 % of all forward FLOPs, SURVEY.md header) — live,
    with HIP events on the launch stream.  Algorithmic FLOPs per launch = 2*M*K*N."""
    B, H, W, C = 8, 120, 160, 160
    x = torch.randn(B, H, W, C, device="cuda").to(dtype)
    w = (torch.randn(C, 3, 3, C, device="cuda") * (9 * C) ** -0.5).to(dtype)
    y = torch.empty(B, H, W, C, device="cuda", dtype=dtype)
    dims = (B, H, W, C, H, W, C, 3, 3)
    # the launch as the step makes it: the DATA GRADIENT of the layer (transposed gather, weights [Cin][3][3][Cout]) - same FLOPs and
    # bytes as the forward product; the step's forward launches of this shape carry the ConvLn epilogue since round 3
    kw = dict(stride=1, pad=1, gather=hip_gather_transposed())
    for _ in range(3):
        lib.conv_forward(x, w, y, dims, **kw)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    torch.cuda.synchronize()
    s.record()
    for _ in range(n):
        lib.conv_forward(x, w, y, dims, **kw)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / n
    flops = 2.0 * B * H * W * 9 * C * C
    ach = flops / (ms * 1e-3) / 1e12
    peak = PEAK_MFMA_BF16_TFLOPS if dtype == torch.bfloat16 else 157.3
    traffic = stamp = None      # HBM bytes per launch from the committed rocprofv3 --pmc passes on this exact launch (profiles/): a
    try:                        # committed measurement, not taken in this run - stamped with when / at which commit it was made
        with open(os.path.join(ROOT, "profiles", "pmc_dominant_kernel.json")) as f:
            rec = json.load(f)
        if dtype == torch.bfloat16:
            traffic = rec["traffic_bytes_per_launch"]
            stamp = "%s, commit %s (profiles/pmc_dominant_kernel.json)" % (rec.get("measured", "?"), rec.get("commit", "?"))
    except (OSError, KeyError, ValueError):
        pass
    return {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "traffic": traffic, "traffic_measured": stamp, "kernel": "igemm_dma_kernel<256,160,8,1,3,GM=1,...,HALO> (%s) conv3x3 160->160 @ 8x120x160, data-gradient launch (halo-patch variant)" % ("bf16" if dtype == torch.bfloat16 else "f32"),
            "avg_launch_ms": round(ms, 4), "algorithmic_gflop_per_launch": round(flops / 1e9, 2)}


if __name__ == "__main__":
    main()
