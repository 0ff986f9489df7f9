"""Where does the bf16 mode's depth-RMSE error come from?  (VERDICT r1, next #2.)

Eval-mode forward of ONE 480x640 image (bench.py's depth_rmse sample: weight seed 0, data seed 1) in fp32 and in bf16, same
teacher-forced sample points: per-stage relative L2 error and mean signed difference of the tapped tensors, then the RMSE
(`rms` of evaluate()) with each stage ALONE kept in fp32 storage, then with growing sets.  Prints a table; run on the GPU box.

    python tools/bf16_taps.py > gpurun_out/bf16_taps.txt
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gw_depth_amd import Config, build_model
from gw_depth_amd.evaluate import DenseMetrics
from gw_depth_amd.model import NestedTensor
from gw_depth_amd.synth import det_fill_, synth_batch

STAGES = ["backbone", "detr", "dense32", "class1", "class2", "pbp1", "class3", "pbp2", "decoder"]
DEC = ["decoder_fuse", "decoder_up1", "decoder_up2", "decoder_head"]
# weight-name prefixes per stage (for the "fp32 arithmetic, bf16-rounded weights" experiment)
WEIGHTS = {"backbone": ("backbone.",), "detr": ("transformer.", "input_proj.", "query_embed.", "class_embed.", "lines_embed."),
           "dense32": ("dense_input_proj.", "dense_encoder.dense_transformer.", "dense_encoder.depth_pred32."),
           "class1": ("dense_encoder.proj_class1.", "dense_encoder.proj_backbn1.", "dense_encoder.class_transformer1.", "dense_encoder.depth_pred16.",
                      "dense_encoder.depth_token", "dense_encoder.seg_token"),
           "class2": ("dense_encoder.proj_class2.", "dense_encoder.proj_backbn2.", "dense_encoder.class_transformer2.", "dense_encoder.old_depth_token_proj8.",
                      "dense_encoder.old_seg_token_proj8."),
           "pbp1": ("dense_encoder.point_based_pred1.",),
           "class3": ("dense_encoder.proj_class3.", "dense_encoder.proj_backbn3.", "dense_encoder.class_transformer3.", "dense_encoder.old_depth_token_proj4.",
                      "dense_encoder.old_seg_token_proj4."),
           "pbp2": ("dense_encoder.point_based_pred2.",), "decoder": ("depth_decoder.",)}


def main():
    cfg = Config(device="cuda", dropout=0.1, log_depth_error=True)
    model, _, _ = build_model(cfg)
    sd = det_fill_({k: v.detach().clone() for k, v in model.state_dict().items()}, seed=0)
    model.load_state_dict(sd)
    model.cuda().eval()
    b = synth_batch(1, 480, 640, seed=1)
    img, msk = b["images"].cuda(), b["pad_mask"].cuda()

    def run(dtype, fp32_stages=(), force=None):
        model.compute_dtype = dtype
        model.fp32_stages = set(fp32_stages)
        taps = {} if force is None else dict(force)
        with torch.no_grad():
            out = model(NestedTensor(img, msk), taps=taps)
        dm = DenseMetrics("cuda")
        dm.update(out["pred_depth"][-1], b["depth"].cuda(), out["pred_seg"], b["seg"].cuda())
        return out, taps, dm.compute()["rms"]

    o32, t32, r32 = run(torch.float32)
    force = {"force_points1": t32["points1"], "force_points2": t32["points2"]}
    o16, t16, r16 = run(torch.bfloat16, (), force)
    print("rms fp32 %.6f   bf16 %.6f   |diff| %.3e   (top-k ids equal: %s)" % (r32, r16, abs(r16 - r32), torch.equal(t32["topk_ids"], t16["topk_ids"])))

    def row(name, a, r):
        a, r = a.double().flatten(), r.double().flatten()
        print("  %-22s rel L2 %.3e   mean signed diff %+.3e   (ref mean %+.3e, ref rms %.3e)" %
              (name, float((a - r).norm() / (r.norm() + 1e-30)), float((a - r).mean()), float(r.mean()), float(r.pow(2).mean().sqrt())))

    print("per-tap error of the all-bf16 forward against fp32 (teacher-forced sample points):")
    for i in range(4):
        row("backbone feat %d" % i, t16["dbg_feats"][i], t32["dbg_feats"][i])
    for k in ("dbg_dense_in", "dbg_x32", "dbg_depth0", "dbg_x1", "dbg_depth1", "dbg_x2", "dbg_depth2", "dbg_x3", "dbg_feat4", "dbg_dtok", "dbg_stok"):
        row(k, t16[k], t32[k])
    row("pred_logits", o16["pred_logits"], o32["pred_logits"])
    row("pred_lines", o16["pred_lines"], o32["pred_lines"])
    for i, (a, r) in enumerate(zip(o16["pred_depth"], o32["pred_depth"])):
        row("pred_depth[%d]" % i, a, r)
    row("pred_seg", o16["pred_seg"], o32["pred_seg"])

    print("one stage alone in fp32 storage (everything else bf16): rms, |diff to fp32|")
    for s in STAGES:
        _, _, r = run(torch.bfloat16, (s,), force)
        print("  %-10s %.6f  %.3e" % (s, r, abs(r - r32)))
    print("growing sets, from the output backwards:")
    acc = []
    for s in reversed(STAGES):
        acc.append(s)
        _, _, r = run(torch.bfloat16, acc, force)
        print("  %-60s %.6f  %.3e" % ("+".join(acc), r, abs(r - r32)))
    print("growing sets, from the input forwards:")
    acc = []
    for s in STAGES:
        acc.append(s)
        _, _, r = run(torch.bfloat16, acc, force)
        print("  %-60s %.6f  %.3e" % ("+".join(acc), r, abs(r - r32)))
    print("decoder sub-stages in fp32 (with dense32 in fp32 as well):")
    for extra in ([], ["decoder"], DEC[:1], DEC[1:2], DEC[2:3], DEC[3:], DEC[2:], DEC[1:], ["decoder", "class2", "class3"], ["decoder", "backbone"],
                  ["decoder", "backbone", "class2", "class3"]):
        _, _, r = run(torch.bfloat16, ["dense32"] + extra, force)
        print("  %-60s %.6f  %.3e" % ("+".join(["dense32"] + extra), r, abs(r - r32)))

    # hypothesis: the error that matters is COHERENT (the same for every pixel) and comes from rounding the WEIGHTS to bf16 -
    # per-pixel activation rounding averages out of an RMSE over 307 200 pixels.  fp32 arithmetic, weights rounded to bf16:
    print("fp32 arithmetic with bf16-rounded weight matrices (dim >= 2 tensors of the named stages):")
    full = {k: v.clone() for k, v in sd.items()}

    def with_rounded(stages):
        pref = tuple(p for s_ in stages for p in WEIGHTS[s_])
        new = {k: (v.bfloat16().float() if (v.dim() >= 2 and v.is_floating_point() and k.startswith(pref)) else v) for k, v in full.items()}
        model.load_state_dict(new)
        _, _, r = run(torch.float32, (), force)
        model.load_state_dict(full)
        return r
    for st in ([], STAGES, ["decoder"], ["dense32"], ["backbone"], ["pbp2"], [x for x in STAGES if x not in ("decoder", "dense32")]):
        r = with_rounded(st)
        print("  %-60s %.6f  %.3e" % ("+".join(st) or "(none)", r, abs(r - r32)))

    # without teacher forcing (what bench.py's depth_rmse leg does)
    _, _, r = run(torch.bfloat16)
    print("bf16 without teacher-forced points: %.6f  %.3e" % (r, abs(r - r32)))


if __name__ == "__main__":
    main()
