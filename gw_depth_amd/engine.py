"""The train step: forward, 17 loss terms, backward, gradient all-reduce, clip, AdamW.

Mirror of the body of train_one_epoch (/root/reference/src/engine_glassrgbd.py:45-166) and of the
optimizer / DDP setup of /root/reference/src/main_glassrgbd.py:46-67, re-laid for MI355X:

* all trainable parameters, their gradients and both Adam moments live in four flat fp32 HBM
  buffers (plus a bf16 shadow of the parameters when activations are bf16), so zero_grad is one
  memset, the global-norm is one reduction and clip+AdamW is one streaming kernel per LR group;
* data parallelism is one process per GPU: rank 0's parameters and buffers are broadcast at construction
  (as DistributedDataParallel does, main_glassrgbd.py:46); the flat gradient buffer is cut into
  contiguous buckets in (approximate) backward order and EVERY rank all-reduces EVERY bucket exactly
  once per step in bucket-index order - a bucket leaves as soon as it and all buckets before it are
  complete, overlapping with the rest of backward.  The collective sequence of a step is therefore the
  same on every rank whatever its data, launch mode (eager / HIP graph / refused capture) or cache
  state.  The 54 parameters that never receive a gradient sit at the end of their LR
  group, where nothing waits for them - no find_unused_parameters graph walk.
* HIP-graph mode captures zero_grad + forward + 17 losses + backward as a CHAIN of graphs cut at bucket
  boundaries: after graph k has been enqueued its buckets go to the communication stream while graph
  k+1 runs, so the gradient all-reduce overlaps with backward in the replayed step as well.
"""
import gc
import math
import os
import warnings
from collections import OrderedDict

import torch
import torch.distributed as dist

from . import hip, ops
from .criteria import PackedTargets, target_capacity
from .model import NestedTensor

FORWARD_ORDER = ["backbone", "input_proj", "query_embed", "transformer", "class_embed", "lines_embed",
                 "dense_input_proj", "dense_encoder", "depth_decoder"]
ALIGN = 8   # elements: keeps every bf16 shadow slice 16-byte aligned
# trainable tensors no forward ever touches (SURVEY.md §3.5, 54 tensors; tests/test_product_wiring.py checks the list against
# the reference's own backward): they are laid out behind the live parameters of their LR group so that no bucket waits for them
NEVER_USED = ("border_mu", "border_logsigma", ".proj_seg.", ".pyramid.layer4.", ".pre_depth_pred.", ".depth_pred4.", ".depth_pred32.")


def never_used(name):
    """True for the trainable tensors that receive no gradient in any step (dead code of the reference kept for state-dict
    parity: glassrgbd.py / multiscale_transformerr.py build them, no forward reads them; the 1/32 depth map feeds only the
    gradient-free CertainSample).  A wrong answer here costs overlap, never correctness: see TrainStep._bucket_ready."""
    return any(k in name for k in NEVER_USED) or ("class_transformer" in name and (".diff_mu" in name or ".diff_logsigma" in name))
MAX_GRAPHS = 6          # captured batch signatures kept (least recently used goes first); all share one memory pool


def _align(n):
    return (n + ALIGN - 1) // ALIGN * ALIGN


class TrainStep:
    def __init__(self, model, criterions, cfg, compute_dtype=torch.float32, bucket_mb=32.0, process_group=None,
                 check_finite=True, data_parallel=True, graph=False, segments=None):
        self.model, self.cfg = model, cfg
        self.criterion, self.criterion_depth, self.criterion_seg, self.criterion_plane = criterions
        self.compute_dtype = compute_dtype
        model.compute_dtype = compute_dtype
        self.check_finite = check_finite
        self.device_matcher = True        # gwd_lsap + sync-free criterion (taps / teacher-forced tests use the host matcher)
        self._packs = {}                  # (batch size, capacity) -> PackedTargets of the eager path
        self.use_graph = bool(graph)      # capture zero_grad+forward+losses+backward of a batch signature as a chain of HIP graphs
        self._graphs = OrderedDict()
        self._pool = None
        self._gstream = None
        self._pending_checks = []
        self.weights = None           # built after the parameters moved into the flat buffer
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (data_parallel and dist.is_available() and dist.is_initialized()) else 1
        # the line-loss normaliser is the GLOBAL target count / world whenever a process group exists (glassrgbd.py:323-326)
        self.norm_world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        # cut the captured step at bucket boundaries (needed for the overlap when world > 1; may be forced for measurements)
        self.segments = (self.world > 1) if segments is None else bool(segments)
        self.step_count = 0

        named = dict(model.named_parameters())
        order = []
        for top in FORWARD_ORDER:
            order += [n for n in named if n.split(".")[0] == top and named[n].requires_grad]
        assert len(order) == sum(p.requires_grad for p in named.values()), "FORWARD_ORDER misses a top-level module"
        order.reverse()                                   # ~ the order in which backward produces gradients
        live = [n for n in order if not never_used(n)]
        idle = [n for n in order if never_used(n)]
        head = [n for n in live if "backbone" not in n] + [n for n in idle if "backbone" not in n]   # LR group 0 (main_glassrgbd.py:59-64)
        tail = [n for n in live if "backbone" in n] + [n for n in idle if "backbone" in n]           # LR group 1: lr_backbone
        self.names = head + tail
        self.idle = set(idle)
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
        if self.world > 1:
            self._broadcast_from_rank0()
        if self.flat_p16 is not None:
            self.flat_p16.copy_(self.flat_p)
        spans = [(self.flat_p.data_ptr(), self.flat_p.data_ptr() + self.flat_p.numel() * 4)]
        spans += [(p.data_ptr(), p.data_ptr() + p.numel() * p.element_size()) for p in model.parameters() if not p.requires_grad]
        self.weights = ops.WeightCache(spans)

        # ---- bucket plan: contiguous flat ranges of ~bucket_mb, in backward order
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
        self._expect = None         # hook firings per parameter and step (a shared weight fires once per use): counted on step 1
        self._state = None          # per-backward bookkeeping of the bucket hooks
        self._works = []
        self._cut = None            # graph capture: called with the bucket ids that just became ready
        self._const_cache = {}      # small constant device tensors of losses(), never freed (captured graphs read them by address)
        if self.world > 1 or self.segments:
            for bi, (_, _, mem) in enumerate(self.buckets):
                for n in mem:
                    hook = self._make_hook(bi, n)
                    self.params[n].register_post_accumulate_grad_hook(hook)     # gradients that arrive through autograd
                    self.params[n]._gwd_hook = (lambda h=hook: h(None))          # gradients accumulated by the kernels

    # ------------------------------------------------------------------ DDP
    def _broadcast_from_rank0(self):
        """What DistributedDataParallel's constructor does for the reference (main_glassrgbd.py:46 - its ranks are seeded
        seed + rank, :36, and rely on it): every rank starts from rank 0's parameters and buffers.  Also the communicator's
        first collectives, so its set-up (allocations, helper threads) is over before any graph capture."""
        dist.broadcast(self.flat_p, 0, group=self.pg)
        rest = [p.data for p in self.model.parameters() if not p.requires_grad]
        rest += [b for b in self.model.buffers() if b.is_floating_point()]
        if rest:
            flat = torch.cat([t.reshape(-1).float() for t in rest])
            dist.broadcast(flat, 0, group=self.pg)
            o = 0
            for t in rest:
                t.copy_(flat[o:o + t.numel()].view(t.shape))
                o += t.numel()
        chk = torch.stack([self.flat_p.double().sum(), self.flat_p.double().abs().sum()])
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.pg)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.pg)
        if not torch.equal(lo, hi):
            raise RuntimeError("gw_depth_amd: ranks hold different parameters after the rank-0 broadcast")

    def _make_hook(self, bi, name):
        def hook(_p):
            st = self._state
            if st is None:
                return
            st["got"][name] = st["got"].get(name, 0) + 1
            if st["expect"] is None or name in self.idle:
                return
            if st["got"][name] == st["expect"].get(name, 1):
                st["pending"][bi] -= 1
                if st["pending"][bi] == 0:
                    self._bucket_ready()
        return hook

    def _bucket_ready(self):
        """Buckets leave strictly in index order, each exactly once per step, on every rank: bucket i goes as soon as all of
        its expected parameters and buckets 0..i-1 are done; whatever is left (a parameter that got no gradient this step,
        the never-used tensors at the end of each group) goes at the end of backward.  No data-dependent collective exists."""
        st = self._state
        ready = []
        while st["next"] < len(self.buckets) and st["pending"][st["next"]] == 0:
            ready.append(st["next"])
            st["next"] += 1
        if ready and st["next"] == len(self.buckets):
            # the last bucket closes the step: leave it (and what became ready with it) to _finish_backward, so that a captured
            # chain does not end in an empty graph
            st["next"] -= len(ready)
            st["pending"][st["next"]] = -1              # never "ready" again from a hook
            return
        if ready:
            self._dispatch(ready)

    def _dispatch(self, bucket_ids, final=False):
        if self._cut is not None:                     # capturing: end this graph segment here, its buckets go after its replay
            self._cut(bucket_ids, final)
        else:
            for bi in bucket_ids:
                self._launch(bi)

    def _launch(self, bi):
        if self.world > 1:
            s, e, _ = self.buckets[bi]
            self._works.append(dist.all_reduce(self.flat_g[s:e], group=self.pg, async_op=True))

    def _begin_backward(self):
        if self.world == 1 and not self.segments:
            return
        self._works = []
        exp = self._expect
        pending = None if exp is None else [sum(1 for n in mem if n not in self.idle) for _, _, mem in self.buckets]
        self._state = {"got": {}, "expect": exp, "pending": pending, "next": 0}
        if pending is not None and pending[0] == 0:
            self._bucket_ready()

    def _finish_backward(self):
        st = self._state
        if st is None:
            return
        self._state = None
        if st["expect"] is None:                   # first pass: count the firings per parameter (structural, the same on every rank)
            self._expect = {n: max(1, c) for n, c in st["got"].items()}
        rest = list(range(st["next"], len(self.buckets)))
        if rest:
            self._dispatch(rest, final=True)
        self._wait_works()

    def _wait_works(self):
        for w in self._works:
            w.wait()
        self._works = []

    def _packed(self, targets, store=None):
        """Static-shape device form of the targets for the sync-free criterion (criteria.PackedTargets, one per capacity
        class); None when the device LSAP cannot take the batch (no targets at all, or more than 64 in one image): the
        caller then uses the host matcher."""
        sizes = [int(len(t["labels"])) for t in targets]
        if sum(sizes) == 0 or max(sizes) > hip.LSAP_MAX_TARGETS:
            return None
        store = self._packs if store is None else store
        key = (len(sizes), target_capacity(sum(sizes)))
        p = store.get(key)
        if p is None:
            p = store[key] = PackedTargets(key[0], key[1], self.flat_p.device)
        p.update(targets)
        if self.norm_world > 1:                         # global target count (glassrgbd.py:323-326), stays on the device
            dist.all_reduce(p["num_items"], group=self.pg)
        return p

    # ------------------------------------------------------------------ losses (engine_glassrgbd.py:62-115)
    def losses(self, out, depth_gt, seg_gt, targets, packed=None):
        cfg = self.cfg
        self.criterion.last_stacks = None
        terms = self.criterion.forward_packed(out, packed, self.norm_world) if packed is not None else self.criterion(out, targets)
        wd = self.criterion.weight_dict
        keys = [k for k in terms if k in wd]
        stacks = getattr(self.criterion, "last_stacks", None)
        if stacks is not None and all(k in wd for k in stacks[2] + stacks[3]):
            # the set criterion's per-layer terms as two weighted vector sums (same weights, same terms; the dict entries stay for logging)
            ce, l1, kce, kl1 = stacks
            skip = set(kce) | set(kl1)
            keys = [k for k in keys if k not in skip]
            wk = (tuple(float(wd[k]) for k in kce), tuple(float(wd[k]) for k in kl1), ce.device)
            sw = self._const_cache.get(wk)
            if sw is None:
                sw = self._const_cache[wk] = (torch.tensor(wk[0], dtype=torch.float32, device=ce.device),
                                              torch.tensor(wk[1], dtype=torch.float32, device=ce.device))
            parts, coef = [(ce.float() * sw[0]).sum(), (l1.float() * sw[1]).sum()], [1.0, 1.0]
        else:
            parts, coef = [], []
        parts += [terms[k] for k in keys]
        coef += [float(wd[k]) for k in keys]
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
        # constant tensors read by the step are kept for the life of the TrainStep, one per distinct value: a captured HIP graph reads them
        # by address, and a step that takes another path (the host-matcher fallback has 17 terms, the packed path 7) must not free the
        # tensor a captured chain still uses - with ONE cached slot the replay after such a step read recycled memory (found by the
        # two-rank graph test: |g| 187 instead of 250 on the step behind the fallback)
        ck = (tuple(coef), parts[0].device)
        cf = self._const_cache.get(ck)
        if cf is None:
            cf = self._const_cache[ck] = torch.tensor(coef, dtype=torch.float32, device=parts[0].device)
        total = (torch.stack([p.float().reshape(()) for p in parts]) * cf).sum()
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
            try:
                with ops.COLSUMS, ops.WGRADS:               # bias / weight gradients of small layers run as grouped launches
                    total.backward()
            except BaseException:
                self._state = None
                raise
            self._finish_backward()
        finally:
            if weights is not None:
                weights.end_pass()
        return out, total, terms

    # ------------------------------------------------------------------ HIP-graph path (no host sync inside)
    def _sync_free_fb(self, st):
        """zero_grad + forward + 17 losses + backward on the static tensors `st`; contains no host round trip
        (device LSAP, device CertainSample, fused losses) and no collective, hence capturable."""
        self.model.train()
        weights = self.weights if (self.compute_dtype == torch.bfloat16 and st["images"].is_cuda) else None
        if weights is not None:
            weights.begin_pass()                       # every transposed / BN-folded bf16 weight copy, one launch
        try:
            out = self.model(NestedTensor(st["images"], st["pad_mask"]), taps=st.get("taps"))
            total, terms = self.losses(out, st["depth"], st["seg"], None, packed=st["packed"])
            self.flat_g.zero_()
            self._begin_backward()
            try:
                with ops.COLSUMS, ops.WGRADS:
                    total.backward()
            except BaseException:
                self._state = None
                raise
            self._finish_backward()
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
            self._state = None
            self._sync_free_fb(st)
            return -1

    def _capture(self, st):
        """Capture one sync-free pass as a chain of HIP graphs on the graph stream -> [(graph, bucket ids to all-reduce once
        it has been enqueued)].  With bucket hooks installed (world > 1) the chain is cut whenever buckets become ready:
        the hook ends the running capture and begins the next one on the same stream and memory pool, in the middle of
        backward (autograd runs single-threaded here, so begin and end happen on one thread, which thread-local capture
        mode requires).  Without hooks it is one graph."""
        side = self._graph_stream()
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()
        chain, cur = [], [None]

        def begin():
            g = torch.cuda.CUDAGraph()
            # thread_local: RCCL's watchdog / proxy threads may touch the runtime while this thread captures
            g.capture_begin(pool=self._pool, capture_error_mode="thread_local")
            cur[0] = g

        def cut(bucket_ids, final):
            cur[0].capture_end()
            chain.append((cur[0], list(bucket_ids)))
            cur[0] = None
            if not final:                              # the last buckets close the step: nothing is enqueued after them
                begin()

        torch.cuda.synchronize()
        gc.collect()
        res = None
        try:
            with torch.cuda.stream(side), torch.autograd.set_multithreading_enabled(False):
                begin()
                self._cut = cut
                try:
                    res = self._sync_free_fb(st)
                finally:
                    self._cut = None
                    if cur[0] is not None:
                        cur[0].capture_end()
                        chain.append((cur[0], []))
                        cur[0] = None
        finally:
            self._works = []
        return chain, res

    def _graph_entry(self, batch):
        sizes = [int(len(t["labels"])) for t in batch["targets"]]
        if sum(sizes) == 0 or max(sizes) > hip.LSAP_MAX_TARGETS:
            return None                                # device LSAP limits: this batch runs eagerly with the host matcher
        key = (tuple(batch["images"].shape), target_capacity(sum(sizes)))
        ent = self._graphs.get(key)
        if ent is not None:
            self._graphs.move_to_end(key)
            return ent
        while len(self._graphs) >= MAX_GRAPHS:          # bounded cache; the executables go, the shared pool keeps the memory
            self._graphs.popitem(last=False)
        dev = self.flat_p.device
        st = {k: batch[k].clone() for k in ("images", "pad_mask", "depth", "seg")}
        st["packed"] = PackedTargets(len(sizes), key[1], dev).update(batch["targets"])
        side = self._graph_stream()
        side.wait_stream(torch.cuda.current_stream())
        launch, self._launch = self._launch, (lambda bi: None)     # warm-up passes and capture issue NO collective
        try:
            with torch.cuda.stream(side), warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                self._sync_free_fb(st)                     # allocator / lazily built caches / AccumulateGrad nodes / hook counts
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
            try:
                chain, res = self._capture(st)
            except RuntimeError as e:                   # capture invalidated: keep training, eagerly, and say so
                torch.cuda.synchronize()
                self._state = None
                warnings.warn("gw_depth_amd: HIP-graph capture failed for batch signature %r, running eager: %s" % (key, e))
                ent = self._graphs[key] = {"graph": None, "reason": str(e)}
                return ent
        finally:
            self._launch = launch
        ent = self._graphs[key] = {"graph": chain, "static": st, "result": res}
        return ent

    def _on_graph_stream(self, batch, taps):
        """Eager forward/backward of a graph-mode TrainStep: same stream as the captures (see _graph_stream), and the same
        collective sequence as a replayed step (one num_items all-reduce, then every bucket once, in index order)."""
        side = self._graph_stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            res = self.forward_backward(batch, taps)
        torch.cuda.current_stream().wait_stream(side)
        return res

    def _graph_step(self, batch):
        ent = self._graph_entry(batch)
        if ent is None or ent["graph"] is None:
            return self._on_graph_stream(batch, None)
        st = ent["static"]
        for k in ("images", "pad_mask", "depth", "seg"):
            st[k].copy_(batch[k], non_blocking=True)
        st["packed"].update(batch["targets"])
        if self.norm_world > 1:                     # global target count, outside the captured region
            dist.all_reduce(st["packed"]["num_items"], group=self.pg)
        self._works = []
        for g, bucket_ids in ent["graph"]:
            g.replay()
            for bi in bucket_ids:                   # the communication stream picks the bucket up behind this segment and
                self._launch(bi)                    # reduces it while the next segment runs
        self._wait_works()
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
