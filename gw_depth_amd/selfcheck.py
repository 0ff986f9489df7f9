"""smoke(): one tiny fp32 train step on cuda:0 through the HIP kernels, checked against the CPU oracle."""
import torch


def smoke_step():
    from gw_depth_amd import Config, build_model, hip
    from gw_depth_amd.engine import TrainStep
    from gw_depth_amd.synth import det_fill_, synth_batch
    from oracle import gwdepth_ref as R          # checker only (allowed in smoke)

    assert torch.cuda.is_available(), "smoke() needs the MI355X"
    lib = hip.library()
    assert not getattr(lib, "is_fake", False)
    cfg = Config(device="cuda", dropout=0.0, log_depth_error=True)
    model, crits, _ = build_model(cfg)
    sd = det_fill_({k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, seed=0)
    model.load_state_dict(sd)
    model.cuda()
    crits[0].cuda()
    b = synth_batch(1, 96, 128, seed=5, n_lines=[4])
    bg = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    bg["targets"] = [{k: v.cuda() for k, v in t.items()} for t in b["targets"]]
    step = TrainStep(model, crits, cfg, compute_dtype=torch.float32)
    out, total, terms = step(bg)
    torch.cuda.synchronize()

    ocfg = R.Cfg(dropout=0.0, log_depth_error=True)
    ref_out, ref_total, ref_terms, _, _ = R.train_step({k: v.clone() for k, v in sd.items()}, b, ocfg, opt_state=None)
    rel = lambda a, r: float((a.detach().double().cpu() - r.detach().double()).norm() / (r.detach().double().norm() + 1e-12))
    errs = {"pred_depth": rel(out["pred_depth"][-1], ref_out["pred_depth"][-1]),
            "pred_seg": rel(out["pred_seg"], ref_out["pred_seg"]),
            "pred_lines": rel(out["pred_lines"], ref_out["pred_lines"]),
            "loss": abs(float(total) - float(ref_total)) / abs(float(ref_total))}
    print("smoke: loss %.5f (oracle %.5f) rel errors %s" % (float(total), float(ref_total), errs))
    assert all(v < 1e-3 for v in errs.values()), errs
