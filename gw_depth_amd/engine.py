"""The train step: forward, 17 loss terms, backward, gradient all-reduce, clip, AdamW.

Mirror of the body of train_one_epoch (/root/reference/src/engine_glassrgbd.py:45-166) and of the
optimizer / DDP setup of /root/reference/src/main_glassrgbd.py:46-67, re-laid for MI355X:

* all trainable parameters, their gradients and both Adam moments live in four flat fp32 HBM
  buffers (plus a bf16 shadow of the parameters when activations are bf16), so zero_grad is one
  memset, the global-norm is one reduction and clip+AdamW is one streaming kernel per LR group;
* data parallelism is one process per GPU: the flat gradient buffer is cut into contiguous buckets
  in (approximate) backward order and each bucket is all-reduced over RCCL as soon as the last of its
  live parameters has accumulated its gradient, overlapping with the rest of backward; the 54
  parameters that never receive a gradient are learned on the first step and simply
  stay zero — no find_unused_parameters graph walk, no buffer broadcasts.
"""
import math
import warnings

import torch
import torch.distributed as dist

from . import hip, ops
from .criteria import pack_targets
from .model import NestedTensor

FORWARD_ORDER = ["backbone", "input_proj", "query_embed", "transformer", "class_embed", "lines_embed",
                 "dense_input_proj", "dense_encoder", "depth_decoder"]
ALIGN = 8   # elements: keeps every bf16 shadow slice 16-byte aligned


def _align(n):
    return (n + ALIGN - 1) // ALIGN * ALIGN


class TrainStep:
    def __init__(self, model, criterions, cfg, compute_dtype=torch.float32, bucket_mb=32.0, process_group=None,
                 check_finite=True, data_parallel=True, graph=False):
        self.model, self.cfg = model, cfg
        self.criterion, self.criterion_depth, self.criterion_seg, self.criterion_plane = criterions
        self.compute_dtype = compute_dtype
        model.compute_dtype = compute_dtype
        self.check_finite = check_finite
        self.device_matcher = True        # gwd_lsap + sync-free criterion (taps / teacher-forced tests use the host matcher)
        self._pack_cache = {}
        self.use_graph = bool(graph)      # capture zero_grad+forward+losses+backward of a batch signature in one HIP graph
        self._graphs = {}
        self._gstream = None
        self._pending_checks = []
        self.weights = None           # built after the parameters moved into the flat buffer
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (data_parallel and dist.is_available() and dist.is_initialized()) else 1
        # the line-loss normaliser is the GLOBAL target count / world whenever a process group exists (glassrgbd.py:323-326)
        self.norm_world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.step_count = 0

        named = dict(model.named_parameters())
        order = []
        for top in FORWARD_ORDER:
            order += [n for n in named if n.split(".")[0] == top and named[n].requires_grad]
        assert len(order) == sum(p.requires_grad for p in named.values()), "FORWARD_ORDER misses a top-level module"
        order.reverse()                                   # ~ the order in which backward produces gradients
        head = [n for n in order if "backbone" not in n]  # LR group 0 (main_glassrgbd.py:59-64)
        tail = [n for n in order if "backbone" in n]      # LR group 1: lr_backbone
        self.names = head + tail
        offs, off = {}, 0
        for n in self.names:
            offs[n] = off
            off += _align(named[n].numel())
        self.offsets, self.total = offs, off
        self.split = offs[tail[0]] if tail else off
        dev = next(model.parameters()).device
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        self.flat_p16 = torch.zeros(off, dtype=torch.bfloat16, device=dev) if compute_dtype == torch.bfloat16 else None
        self.sq = torch.zeros(1, dtype=torch.float64, device=dev)
        self.params = {}
        for n in self.names:
            p, o, k = named[n], offs[n], named[n].numel()
            self.flat_p[o:o + k].copy_(p.data.reshape(-1))
            p.data = self.flat_p[o:o + k].view(p.shape)
            p.grad = self.flat_g[o:o + k].view(p.shape)
            if self.flat_p16 is not None:
                p._gwd_bf16 = self.flat_p16[o:o + k].view(p.shape)
            p._gwd_grad = p.grad          # kernels accumulate weight/bias/LN gradients straight into the flat buffer
            p._gwd_hook = None
            self.params[n] = p
        if self.flat_p16 is not None:
            self.flat_p16.copy_(self.flat_p)
        spans = [(self.flat_p.data_ptr(), self.flat_p.data_ptr() + self.flat_p.numel() * 4)]
        spans += [(p.data_ptr(), p.data_ptr() + p.numel() * p.element_size()) for p in model.parameters() if not p.requires_grad]
        self.weights = ops.WeightCache(spans)

        # ---- bucket plan: contiguous flat ranges of ~bucket_mb
        per = max(int(bucket_mb * (1 << 20) / 4), 1)
        self.buckets, start, members = [], 0, []
        for n in self.names:
            members.append(n)
            end = offs[n] + _align(named[n].numel())
            if end - start >= per:
                self.buckets.append((start, end, members))
                start, members = end, []
        if members:
            self.buckets.append((start, off, members))
        self.live = None            # learned on the first step
        self._pending, self._works = None, []
        self._expect = {}           # gradient contributions per parameter and step (a shared weight fires once per use)
        if self.world > 1:
            for bi, (_, _, mem) in enumerate(self.buckets):
                for n in mem:
                    hook = self._make_hook(bi, n)
                    self.params[n].register_post_accumulate_grad_hook(hook)     # gradients that arrive through autograd
                    self.params[n]._gwd_hook = (lambda h=hook: h(None))          # gradients accumulated by the kernels

    # ------------------------------------------------------------------ DDP
    def _make_hook(self, bi, name):
        def hook(_p):
            if self._pending is None:
                return
            self._got[name] = self._got.get(name, 0) + 1
            if self.live is None:
                self._seen.add(name)
            elif name not in self.live:            # a parameter that was dead on step 1 woke up
                self.live.add(name)
                self._late.append(name)
            elif self._got[name] == self._expect.get(name, 1):
                self._pending[bi] -= 1
                if self._pending[bi] == 0:
                    self._launch(self.buckets[bi][0], self.buckets[bi][1])
                    self._launched.add(bi)
        return hook

    def _launch(self, s, e):
        self._works.append(dist.all_reduce(self.flat_g[s:e], group=self.pg, async_op=True))

    def _begin_backward(self):
        if self.world == 1:
            return
        self._works, self._late, self._launched, self._got = [], [], set(), {}
        if self.live is None:
            self._seen, self._pending = set(), [0] * len(self.buckets)
        else:
            self._pending = [sum(n in self.live for n in mem) for _, _, mem in self.buckets]

    def _finish_backward(self):
        if self.world == 1:
            return
        if self.live is None:                      # first step: no overlap, learn the live set
            self.live = set(self._seen)
            self._expect = dict(self._got)
            for s, e, _ in self.buckets:
                self._launch(s, e)
        else:
            for bi, left in enumerate(self._pending):
                if left > 0 and bi not in self._launched:     # a live parameter got no gradient this step
                    self._launch(self.buckets[bi][0], self.buckets[bi][1])
            for n in self._late:
                bi = next(i for i, b in enumerate(self.buckets) if n in b[2])
                if bi in self._launched:           # its bucket already went out without it: reduce the slice alone
                    o = self.offsets[n]
                    self._launch(o, o + self.params[n].numel())
        for w in self._works:
            w.wait()
        self._pending = None

    def _packed(self, targets):
        """Static-shape device form of the targets for the sync-free criterion; the index scaffolding is cached per
        tuple of target counts, only the line coordinates / labels are gathered each step (two small device ops)."""
        sizes = tuple(int(len(t["labels"])) for t in targets)
        if sum(sizes) == 0 or max(sizes) > 64:
            return None                                # device LSAP limits; fall back to the host matcher
        ent = self._pack_cache.get(sizes)
        if ent is None:
            ent = self._pack_cache[sizes] = pack_targets(targets, self.flat_p.device)
        p = dict(ent)
        p["lines"] = torch.cat([t["lines"] for t in targets]).float()
        p["labels"] = torch.cat([t["labels"] for t in targets])
        if self.norm_world > 1:                         # global target count (glassrgbd.py:323-326), stays on the device
            n = ent["num_items"].clone()
            dist.all_reduce(n, group=self.pg)
            p["num_items"] = n
        return p

    # ------------------------------------------------------------------ losses (engine_glassrgbd.py:62-115)
    def losses(self, out, depth_gt, seg_gt, targets, packed=None):
        cfg = self.cfg
        terms = self.criterion.forward_packed(out, packed, self.norm_world) if packed is not None else self.criterion(out, targets)
        wd = self.criterion.weight_dict
        keys = [k for k in terms if k in wd]
        parts, coef = [terms[k] for k in keys], [float(wd[k]) for k in keys]
        names = ["1/16", "1/8", "1/4", "1"]
        for i, pd in enumerate(out["pred_depth"]):
            ld = self.criterion_depth.fused(pd, depth_gt, cfg.depth_loss_weights[i])
            terms["loss_depth_" + names[i]] = ld
            parts.append(ld)
            coef.append(1.0)
        if self.criterion_plane is not None:
            # --with_plane_norm_loss, one image per step (engine_glassrgbd.py:85-86).  The reference LOGS 50 x this loss
            # (:133-135) but never adds it to `losses` (:109-115), so it is a reported term only here as well - no gradient.
            with torch.no_grad():
                mask = (depth_gt >= 0.2) & (depth_gt < 10.0)
                lp = self.criterion_plane(out["pred_depth"][-1], depth_gt, out["pred_lines"], out["pred_logits"], mask)
            terms["loss_plane"] = lp * float(cfg.plane_norm_loss_coef)
        ls = self.criterion_seg(out["pred_seg"], seg_gt.reshape(seg_gt.shape[0], *seg_gt.shape[-2:]), cfg.seg_loss_weight)
        terms["loss_seg"] = ls
        parts.append(ls)
        coef.append(1.0)
        # engine_glassrgbd.py:120-134: the weighted sum of the 17 terms, as one stack/multiply/sum instead of 34 scalar kernels
        ck = (tuple(coef), parts[0].device)
        if getattr(self, "_coef_key", None) != ck:
            self._coef_key, self._coef = ck, torch.tensor(coef, dtype=torch.float32, device=parts[0].device)
        total = (torch.stack([p.float().reshape(()) for p in parts]) * self._coef).sum()
        return total, terms

    # ------------------------------------------------------------------ the step
    def zero_grad(self):
        self.flat_g.zero_()

    def optimizer_step(self):
        cfg, lib = self.cfg, hip.library()
        self.step_count += 1
        t = self.step_count
        bc1, bc2 = 1 - 0.9 ** t, 1 - 0.999 ** t
        self.sq.zero_()
        lib.sqnorm(self.flat_g, self.sq, self.total)
        gs = 1.0 / self.world
        opt = getattr(self, "optimizer", None)      # checkpoint.FlatAdamW: a StepLR may have moved the groups' rates
        lr0, lr1 = opt.lrs() if opt is not None else (cfg.lr, cfg.lr_backbone)
        for lo, hi, lr in ((0, self.split, lr0), (self.split, self.total, lr1)):
            if hi > lo:
                lib.adamw_step(self.flat_p[lo:hi], self.flat_g[lo:hi], self.flat_m[lo:hi], self.flat_v[lo:hi],
                               None if self.flat_p16 is None else self.flat_p16[lo:hi], self.sq, hi - lo, lr, 0.9, 0.999,
                               1e-8, cfg.weight_decay, bc1, bc2, cfg.clip_max_norm, gs)

    def grad_norm(self):
        """Un-clipped global gradient norm of the last step (host sync)."""
        return math.sqrt(float(self.sq.item())) / self.world

    def forward_backward(self, batch, taps=None):
        """Forward, losses, backward and (N > 1) the bucketed gradient all-reduce; leaves SUMMED grads in flat_g."""
        self.model.train()
        packed = self._packed(batch["targets"]) if self.device_matcher else None
        match = (self.criterion.matcher, batch["targets"]) if (packed is None and hasattr(self.criterion.matcher, "prefetch")) else None
        weights = self.weights if (self.compute_dtype == torch.bfloat16 and batch["images"].is_cuda) else None
        if weights is not None:
            weights.begin_pass()
        try:
            out = self.model(NestedTensor(batch["images"], batch["pad_mask"]), taps=taps, match=match)
            total, terms = self.losses(out, batch["depth"], batch["seg"], batch["targets"], packed=packed)
            self.zero_grad()
            self._begin_backward()
            with ops.COLSUMS, ops.WGRADS:               # bias / weight gradients of small layers run as grouped launches
                total.backward()
            self._finish_backward()
        finally:
            if weights is not None:
                weights.end_pass()
        return out, total, terms

    # ------------------------------------------------------------------ HIP-graph path (no host sync inside)
    def _sync_free_fb(self, st):
        """zero_grad + forward + 17 losses + backward on the static tensors `st`; contains no host round trip
        (device LSAP, device CertainSample, fused losses), hence capturable."""
        self.model.train()
        weights = self.weights if (self.compute_dtype == torch.bfloat16 and st["images"].is_cuda) else None
        if weights is not None:
            weights.begin_pass()                       # every transposed / BN-folded bf16 weight copy, one launch
        try:
            out = self.model(NestedTensor(st["images"], st["pad_mask"]), taps=st.get("taps"))
            total, terms = self.losses(out, st["depth"], st["seg"], None, packed=st["packed"])
            self.flat_g.zero_()
            with ops.COLSUMS, ops.WGRADS:
                total.backward()
        finally:
            if weights is not None:
                weights.end_pass()
        return out, total.detach(), {k: v.detach() for k, v in terms.items()}

    def _graph_stream(self):
        """The one non-default stream every graph-mode forward/backward of this TrainStep runs on.  autograd binds
        each parameter's AccumulateGrad node to the stream it was created under and keeps it as long as any autograd
        graph referencing it is alive; if such a node predates the capture on another stream, its in-place gradient
        accumulation is captured on a forked branch whose input buffers the allocator recycles without ordering."""
        if self._gstream is None:
            self._gstream = torch.cuda.Stream()
        return self._gstream

    def _count_memsets(self, st):
        """Run one sync-free pass under the profiler and count the hipMemsetAsync runtime calls it makes."""
        from torch.profiler import ProfilerActivity, profile
        try:
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
                self._sync_free_fb(st)
                torch.cuda.synchronize()
            return sum(1 for e in prof.events() if e.name == "hipMemsetAsync")
        except Exception as exc:                     # no tracer on this host: without the audit there is no capture
            warnings.warn("gw_depth_amd: capture audit unavailable (%s)" % exc)
            self._sync_free_fb(st)
            return -1

    def _graph_entry(self, batch):
        sizes = tuple(int(len(t["labels"])) for t in batch["targets"])
        key = (tuple(batch["images"].shape), sizes)
        ent = self._graphs.get(key)
        if ent is not None:
            return ent
        dev = self.flat_p.device
        st = {k: batch[k].clone() for k in ("images", "pad_mask", "depth", "seg")}
        st["packed"] = pack_targets(batch["targets"], dev)
        side = self._graph_stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            self._sync_free_fb(st)                     # allocator / lazily built caches / AccumulateGrad nodes
            memsets = self._count_memsets(st)          # second warm-up pass, audited
        torch.cuda.current_stream().wait_stream(side)
        reason = None
        if any("AccumulateGrad node's stream does not match" in str(w.message) for w in caught):
            reason = ("an autograd graph built on another stream is still alive (e.g. the outputs of an eager step): its "
                      "AccumulateGrad nodes would be captured on a forked stream")
        elif memsets < 0:
            reason = "the capture audit (torch.profiler runtime-call trace) is not available on this host"
        elif memsets:
            reason = ("%d hipMemsetAsync call(s) in the step (ATen multi-block reductions zero their semaphores that way); "
                      "memset nodes do not replay correctly in HIP graphs on this ROCm" % memsets)
        if reason is not None:
            warnings.warn("gw_depth_amd: HIP-graph capture refused for batch signature %r, running eager: %s" % (key, reason))
            ent = self._graphs[key] = {"graph": None, "reason": reason}
            return ent
        if self.norm_world > 1:                     # communicator set-up (allocations, helper threads) finishes before capture
            dist.all_reduce(torch.zeros(1, device=dev), group=self.pg)
            torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        # thread_local: RCCL's watchdog / proxy threads may touch the runtime while this thread captures
        try:
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                res = self._sync_free_fb(st)
        except RuntimeError as e:                   # capture invalidated: keep training, eagerly, and say so
            torch.cuda.synchronize()
            warnings.warn("gw_depth_amd: HIP-graph capture failed for batch signature %r, running eager: %s" % (key, e))
            ent = self._graphs[key] = {"graph": None, "reason": str(e)}
            return ent
        ent = self._graphs[key] = {"graph": g, "static": st, "result": res}
        return ent

    def _on_graph_stream(self, batch, taps):
        """Eager forward/backward of a graph-mode TrainStep: same stream as the captures (see _graph_stream)."""
        side = self._graph_stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            res = self.forward_backward(batch, taps)
        torch.cuda.current_stream().wait_stream(side)
        return res

    def _graph_step(self, batch):
        ent = self._graph_entry(batch)
        if ent["graph"] is None:
            return self._on_graph_stream(batch, None)
        st = ent["static"]
        for k in ("images", "pad_mask", "depth", "seg"):
            st[k].copy_(batch[k], non_blocking=True)
        lines = torch.cat([t["lines"] for t in batch["targets"]])
        st["packed"]["lines"].copy_(lines, non_blocking=True)
        st["packed"]["labels"].copy_(torch.cat([t["labels"] for t in batch["targets"]]), non_blocking=True)
        n = st["packed"]["num_items"]
        n.fill_(float(lines.shape[0]))
        if self.norm_world > 1:                     # global target count, outside the captured region
            dist.all_reduce(n, group=self.pg)
        ent["graph"].replay()
        if self.world > 1:                          # gradients: bucketed all-reduce after the replay (no overlap in graph mode)
            works = [dist.all_reduce(self.flat_g[s:e], group=self.pg, async_op=True) for s, e, _ in self.buckets]
            for w in works:
                w.wait()
        return ent["result"]

    def __call__(self, batch, taps=None):
        """batch: dict(images (B,3,H,W), pad_mask (B,H,W) bool, depth (B,1,H,W), seg (B,1,H,W) i64, targets)."""
        if self.use_graph and batch["images"].is_cuda:
            out, total, terms = self._graph_step(batch) if taps is None else self._on_graph_stream(batch, taps)
        else:
            out, total, terms = self.forward_backward(batch, taps)
        self.optimizer_step()
        if self.check_finite:                       # engine_glassrgbd.py:143-153
            if total.is_cuda and self.use_graph:
                # graph mode keeps the host a step ahead of the device: the loss goes to pinned memory asynchronously and
                # is examined when the NEXT step has been enqueued (or in flush()), so the check costs no device idle
                # time (a blocking read here leaves the GPU idle for the ~1 ms the host needs to launch the next graph)
                self._queue_finite_check(total)
                self._poll_finite_checks(keep=1)
            else:
                v = float(total.detach())
                if not math.isfinite(v):
                    bad = {k: float(t) for k, t in terms.items() if not math.isfinite(float(t))}
                    raise FloatingPointError("Loss is %r at step %d, stopping training (non-finite terms: %r)" % (v, self.step_count, bad))
        return out, total.detach(), terms

    def _queue_finite_check(self, total):
        slot = torch.empty(1, dtype=torch.float32, pin_memory=True)
        slot.copy_(total.detach().reshape(1).float(), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._pending_checks.append((self.step_count, slot, ev))

    def _poll_finite_checks(self, keep=0):
        while len(self._pending_checks) > keep:
            step_no, slot, ev = self._pending_checks.pop(0)
            ev.synchronize()
            v = float(slot[0])
            if not math.isfinite(v):
                raise FloatingPointError("Loss is %r at step %d, stopping training" % (v, step_no))

    def flush(self):
        """Examine every outstanding loss (graph mode checks one step late); call at the end of an epoch."""
        self._poll_finite_checks(keep=0)
