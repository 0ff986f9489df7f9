#!/usr/bin/env python
"""HIP-graph train step diagnosis: node census of the captured graph, replay-vs-eager comparison, keep-alive variant."""
import os, sys, re, collections, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gw_depth_amd import Config, build_model
from gw_depth_amd.engine import TrainStep
from gw_depth_amd.synth import det_fill_, synth_batch

mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
B = int(os.environ.get("B", 8))
cfg = Config(device="cuda", dropout=0.0, log_depth_error=True)
model, crits, _ = build_model(cfg)
model.load_state_dict(det_fill_({k: v.detach().clone() for k, v in model.state_dict().items()}, seed=0))
model.cuda(); crits[0].cuda()
step = TrainStep(model, crits, cfg, compute_dtype=torch.bfloat16, check_finite=False, graph=True)
b = synth_batch(B, 480, 640, seed=1)
batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
batch["targets"] = [{k: v.cuda() for k, v in t.items()} for t in b["targets"]]
print("host", socket.gethostname(), torch.cuda.get_device_name(0), torch.version.hip, flush=True)

# eager reference on the same static inputs (no optimizer step: weights stay fixed)
from gw_depth_amd.criteria import pack_targets
st = {k: batch[k].clone() for k in ("images", "pad_mask", "depth", "seg")}
st["packed"] = pack_targets(batch["targets"], "cuda")
es = torch.cuda.current_stream() if mode == "mismatch" else step._graph_stream()
es.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(es):
    for _ in range(2):
        out, total, terms = step._sync_free_fb(st)     # `out` stays referenced: its AccumulateGrad nodes stay alive
torch.cuda.synchronize()
ref_total = float(total); ref_g = step.flat_g.clone(); 
print("eager total %.6f gnorm %.6f finite %s" % (ref_total, float(ref_g.norm()), bool(torch.isfinite(ref_g).all())), flush=True)

keep = []
if mode == "keepalive":
    import torch.autograd.graph as ag
    ctx = ag.saved_tensors_hooks(lambda t: (keep.append(t), t)[1], lambda t: t)
    ctx.__enter__()
    hooks = [m.register_forward_hook(lambda _m, _i, o: keep.append(o)) for m in model.modules()]

orig_graph = torch.cuda.CUDAGraph
dump = None
if dump:
    class G(orig_graph):
        def __new__(cls, *a, **k):
            g = super().__new__(cls, *a, **k); return g
    _g0 = orig_graph.__init__
ent = None
import gw_depth_amd.engine as E
if dump:
    real = torch.cuda.CUDAGraph
    def mk():
        g = real(); g.enable_debug_mode(); return g
    E.torch.cuda.CUDAGraph = mk
clones = []
if mode == "trace":
    names = {m: n for n, m in model.named_modules()}
    def first_tensor(o):
        if torch.is_tensor(o): return o
        if isinstance(o, (list, tuple)):
            for x in o:
                r = first_tensor(x)
                if r is not None: return r
        if isinstance(o, dict):
            for x in o.values():
                r = first_tensor(x)
                if r is not None: return r
        return None
    def hook(m, inp, o):
        if torch.cuda.is_current_stream_capturing():
            ti, to = first_tensor(inp), first_tensor(o)
            clones.append((names[m], type(m).__name__, None if ti is None else ti.detach().clone(), None if to is None else to.detach().clone()))
    hs = [m.register_forward_hook(hook) for m in model.modules()]
import gw_depth_amd.model as M
if mode == "fine":
    M.TRACE = []
ent = step._graph_entry(batch)
E.torch.cuda.CUDAGraph = orig_graph
torch.cuda.synchronize()
if dump:
    ent["graph"].debug_dump(dump)
    txt = open(dump).read()
    print("dot bytes", len(txt))
    kinds = collections.Counter(re.findall(r'label="?\{?\s*\n?([A-Za-z_]+)', txt))
    print("node label heads", kinds.most_common(12))
    print("H2D mentions", len(re.findall(r"HostToDevice|H2D|hipMemcpyHostToDevice", txt)), "D2H", len(re.findall(r"DeviceToHost|D2H", txt)))
    for m in re.findall(r"[^\n]*(?:HostToDevice|H2D)[^\n]*", txt)[:10]:
        print("  ", m[:300])
for r in range(3):
    step.flat_g.fill_(float("nan"))                       # anything the graph does not rewrite shows up
    ent["graph"].replay()
    torch.cuda.synchronize()
    tot = float(ent["result"][1]); g = step.flat_g
    nf = int((~torch.isfinite(g)).sum())
    d = float((g - ref_g).norm() / ref_g.norm()) if nf == 0 else float("nan")
    print("replay %d total %.6f (eager %.6f) nonfinite grads %d rel grad diff %.3e" % (r, tot, ref_total, nf, d), flush=True)
if M.TRACE:
    for n, tt in M.TRACE[:8]:
        f = bool(torch.isfinite(tt.float()).all())
        if n in ("mu", "var", "upd", "ra0"):
            print("     min %.4e max %.4e" % (float(tt.float().min()), float(tt.float().max())), tt.float().flatten()[:6].tolist())
        print("  trace %-8s finite=%s shape %s %s" % (n, f, tuple(tt.shape), "" if f else "nonfinite count %d" % int((~torch.isfinite(tt.float())).sum())))
if clones:
    shown = 0
    for n, ty, ti, to in clones:
        fi = True if ti is None or not ti.is_floating_point() else bool(torch.isfinite(ti).all())
        fo = True if to is None or not to.is_floating_point() else bool(torch.isfinite(to).all())
        if not fo or not fi:
            print("  module %-60s %-22s in finite=%s out finite=%s out shape %s" % (n, ty, fi, fo, None if to is None else tuple(to.shape)))
            shown += 1
            if shown > 12: break
if nf:
    bad = []
    for n, p in model.named_parameters():
        if p.grad is not None and not torch.isfinite(p.grad).all():
            bad.append(n)
    print("params with non-finite grads: %d of %d; first %s" % (len(bad), len(list(model.parameters())), bad[:6]))
    outs = ent["result"][0]
    for k, v in outs.items():
        if torch.is_tensor(v):
            print("  out", k, bool(torch.isfinite(v).all()))
        elif isinstance(v, (list, tuple)):
            print("  out", k, [bool(torch.isfinite(t).all()) if torch.is_tensor(t) else None for t in v])
