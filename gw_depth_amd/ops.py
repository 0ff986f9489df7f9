"""torch.autograd.Function wrappers over the C ABI (include/gwdepth.h).

Layout conventions: feature maps are (B, H, W, C) contiguous ("pixel-major"), token tensors
(..., C) contiguous; conv weights are (Cout, KH, KW, Cin); Linear weights (out, in) as in torch.
Master parameters are fp32; with bf16 activations the kernels read a bf16 copy of the weights
(the flat shadow maintained by the fused optimizer when present, otherwise made on the fly).
"""
import os

import torch
import torch.nn.functional as F

from . import hip
from .hip import (ACT_ELU, ACT_GELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, GATHER_CONV, GATHER_TRANSPOSED,
                  GATHER_UPSAMPLED)

__all__ = ["conv2d", "linear", "layer_norm", "softmax_lastdim", "silog_loss", "seg_cross_entropy",
           "ACT_NONE", "ACT_RELU", "ACT_GELU", "ACT_ELU", "ACT_SIGMOID"]


def _lib():
    return hip.library()


class WeightCache:
    """bf16 kernel-side copies of the conv / linear weights (row-scaled forward copy for folded FrozenBN, transposed
    copy for the data gradient), refreshed by ONE gwd_weight_prep_batch launch per train step instead of one small
    launch per layer.  engine.TrainStep owns one and brackets each forward/backward with begin_pass() / end_pass();
    outside such a pass (plain ops calls, fp32 runs) every copy is made on the fly as before.  A weight first seen inside a pass is
    prepared on the fly and joins the table for the next pass."""

    def __init__(self, persistent=()):
        # [start, end) address ranges of PARAMETER storage: only weights living there are cached - a temporary (e.g. a
        # zero-padded copy of a parameter) has a new address every step and must be prepared on the fly
        self.ranges = sorted((int(a), int(b)) for a, b in persistent)
        self.jobs = {}           # key -> dict(w, rs, fwd, t)
        self.table = None
        self.n_jobs = self.blocks = 0
        self.dirty = False
        self.active = False
        self._retired = []

    @staticmethod
    def _key(w, rs, geom=None):
        return (w.data_ptr(), tuple(w.shape), 0 if rs is None else rs.data_ptr(), geom)

    def get(self, w, rs, kind, geom=None):
        """kind 'fwd' | 't' -> cached tensor, or None when the cache is not in a pass / not applicable.
        geom = (Np, Cg, Cgp): the zero-padded copies of padded_weight()."""
        if not self.active or not w.is_cuda or w.dtype != torch.float32:
            return None
        p = w.data_ptr()
        if not any(a <= p < b for a, b in self.ranges):
            return None
        job = self.jobs.get(self._key(w, rs, geom))
        if job is not None and job[kind] is not None:
            return job[kind]        # refreshed by this pass's batch launch, or made earlier in this very pass
        return self._add(w, rs, kind, geom)

    def _add(self, w, rs, kind, geom=None):
        key = self._key(w, rs, geom)
        job = self.jobs.get(key)
        if job is None:
            job = self.jobs[key] = {"w": w.detach(), "rs": rs, "fwd": None, "t": None, "geom": geom}
        N, C = w.shape[0], w.shape[-1]
        taps = w.numel() // (N * C)
        if geom is not None:
            # the padding is written here, once; the per-step batch launch refreshes the real entries only
            job[kind] = padded_weight(w.detach(), geom, kind, torch.bfloat16)
            self.dirty = True
            return job[kind]
        if kind == "fwd":
            out = torch.empty(w.shape, dtype=torch.bfloat16, device=w.device)
            _lib().weight_prep(w.detach(), rs, out, None, N, taps, C, hip.BF16)
        else:
            out = torch.empty((C,) + tuple(w.shape[1:-1]) + (N,), dtype=torch.bfloat16, device=w.device)
            _lib().weight_prep(w.detach(), rs, None, out, N, taps, C, hip.BF16)
        job[kind] = out
        self.dirty = True
        return out

    def begin_pass(self):
        if self.dirty and self.jobs:
            recs = (hip.PrepJob * len(self.jobs))()
            b0 = 0
            owners = []
            for i, job in enumerate(self.jobs.values()):
                w = job["w"]
                N, C = w.shape[0], w.shape[-1]
                r = recs[i]
                r.w, r.row_scale = w.data_ptr(), (0 if job["rs"] is None else job["rs"].data_ptr())
                r.w_fwd = 0 if job["fwd"] is None else job["fwd"].data_ptr()
                r.w_dgrad = 0 if job["t"] is None else job["t"].data_ptr()
                r.N, r.taps, r.C, r.block0 = N, w.numel() // (N * C), C, b0
                r.Np, r.Cg, r.Cgp = job["geom"] if job.get("geom") is not None else (0, 0, 0)
                nblk = (w.numel() // (N * C)) * ((N + 31) // 32) * ((C + 31) // 32)
                owners.append(torch.full((nblk,), i, dtype=torch.int32))
                b0 += nblk
            raw = torch.frombuffer(bytearray(bytes(recs)), dtype=torch.uint8)
            dev = next(iter(self.jobs.values()))["w"].device
            self._retired.append((self.table, getattr(self, "block_job", None)))      # a captured HIP graph may still launch with the old tables
            self.table = raw.to(dev)
            self.block_job = torch.cat(owners).to(dev)
            self.n_jobs, self.blocks = len(self.jobs), b0
            self.dirty = False
        if self.table is not None:
            _lib().weight_prep_batch(self.table, self.n_jobs, self.blocks, getattr(self, "block_job", None))
        self.active = True
        global _ACTIVE_WEIGHTS
        _ACTIVE_WEIGHTS = self

    def end_pass(self):
        global _ACTIVE_WEIGHTS
        self.active = False
        _ACTIVE_WEIGHTS = None


class ColsumQueue:
    """Bias gradients of a backward pass, batched: the column sums of up to hip.COLSUM_BATCH layers run as ONE launch
    (gwd_colsum_batch) instead of one 6-25 us launch each.  engine.TrainStep opens the queue around backward and
    flushes it at the end; outside (plain ops calls) every column sum runs at once, as before.  A queued job keeps its
    gradient tensor alive until the flush; the hooks of the DDP bucket plan fire after the sums have been enqueued."""

    def __init__(self):
        self.jobs, self.active = [], False

    def add(self, g, out, rows, C, hook=None):
        lib = _lib()
        if not self.active or rows <= 0 or not lib.colsum_batchable(g, C) or (self.jobs and self.jobs[0][0].dtype != g.dtype):
            if self.jobs and self.active:
                self.flush()                        # keep program order between jobs that may share `out`
            lib.colsum(g, out, rows, C)
            if hook is not None:
                hook()
            return
        self.jobs.append((g, out, rows, C, hook))
        if len(self.jobs) == hip.COLSUM_BATCH:
            self.flush()

    def flush(self):
        if not self.jobs:
            return
        jobs, self.jobs = self.jobs, []
        _lib().colsum_batch([j[:4] for j in jobs])
        for j in jobs:
            if j[4] is not None:
                j[4]()

    def __enter__(self):
        self.active = True
        return self

    def __exit__(self, *exc):
        try:
            if exc[0] is None:
                self.flush()
        finally:
            self.jobs, self.active = [], False
        return False


COLSUMS = ColsumQueue()


class WgradQueue:
    """Weight gradients that accumulate into the flat gradient buffer, handed to the library WGRAD_BATCH at a time
    (gwd_conv_wgrad_batch runs the small plain-GEMM ones of a batch as one grouped launch).  Same life cycle as
    ColsumQueue: open around backward, flushed at its end; a queued job keeps its activation and gradient alive."""
    BATCH = 32

    def __init__(self):
        self.jobs, self.active = [], False
        self._pool, self._used, self._want, self._dev = None, 0, 0, None
        self._retired = []

    def scratch(self, shape, device):
        """Zeroed fp32 scratch for a weight gradient formed in a padded shape.  Inside a backward pass the pieces come from ONE
        buffer zeroed by one fill when the pass opens (sized by the previous pass; a piece that does not fit gets its own zeros)."""
        n = 1
        for v in shape:
            n *= int(v)
        n16 = (n + 15) // 16 * 16
        self._want += n16
        self._dev = device
        if self.active and self._pool is not None and self._pool.device == device and self._used + n16 <= self._pool.numel():
            out = self._pool[self._used:self._used + n].view(shape)
            self._used += n16
            return out
        return torch.zeros(shape, dtype=torch.float32, device=device)

    def add(self, x, dv, dw, dims, kw, hook=None, unpad=None):
        """unpad = (dst, N, taps, G, Cg, Cgp): dw is a zeroed scratch tensor in the PADDED weight shape; after the launch its real
        entries are added to dst (the parameter's slot of the flat gradient buffer) by gwd_unpad_add_batch, then the hook fires."""
        if not self.active:
            _lib().conv_wgrad(x, dv, dw, dims, **kw)
            if unpad is not None:
                _lib().unpad_add_batch([(dw,) + tuple(unpad)])
            if hook is not None:
                hook()
            return
        self.jobs.append((x, dv, dw, dims, kw, hook, unpad))
        if len(self.jobs) == self.BATCH:
            self.flush()

    def flush(self):
        if not self.jobs:
            return
        jobs, self.jobs = self.jobs, []
        _lib().conv_wgrad_batch([j[:5] for j in jobs])
        folds = [(j[2],) + tuple(j[6]) for j in jobs if j[6] is not None]
        if folds:
            _lib().unpad_add_batch(folds)
        for j in jobs:
            if j[5] is not None:
                j[5]()

    def __enter__(self):
        self.active = True
        if self._pool is not None:
            self._pool.zero_()
        self._used = self._want = 0
        return self

    def __exit__(self, *exc):
        try:
            if exc[0] is None:
                self.flush()
        finally:
            self.jobs, self.active = [], False
            if self._want and self._dev is not None and (self._pool is None or self._pool.numel() < self._want) \
                    and not torch.cuda.is_current_stream_capturing():
                if self._pool is not None:
                    self._retired.append(self._pool)          # a captured HIP graph may still zero / write the old buffer: never freed
                self._pool = torch.empty(self._want, dtype=torch.float32, device=self._dev)     # for the next pass
        return False


WGRADS = WgradQueue()


_ACTIVE_WEIGHTS = None


def _weight_for(w, row_scale, dtype, shadow=None):
    """Kernel-ready forward weights in the activation dtype (fp32 master -> as is)."""
    if row_scale is None:
        if dtype == torch.float32:
            return w.detach()
        if shadow is not None and shadow.dtype == dtype:
            return shadow
    if dtype == torch.bfloat16 and _ACTIVE_WEIGHTS is not None:
        cached = _ACTIVE_WEIGHTS.get(w, row_scale, "fwd")
        if cached is not None:
            return cached
    out = torch.empty(w.shape, dtype=dtype, device=w.device)
    N, C = w.shape[0], w.shape[-1]
    _lib().weight_prep(w.detach(), row_scale, out, None, N, w.numel() // (N * C), C, hip.F32 if dtype == torch.float32 else hip.BF16)
    return out


def _weight_transposed(w, row_scale, dtype):
    """[Cin][taps][Cout] copy for the data gradient."""
    if dtype == torch.bfloat16 and _ACTIVE_WEIGHTS is not None:
        cached = _ACTIVE_WEIGHTS.get(w, row_scale, "t")
        if cached is not None:
            return cached
    N, C = w.shape[0], w.shape[-1]
    taps = w.numel() // (N * C)
    out = torch.empty((C,) + tuple(w.shape[1:-1]) + (N,), dtype=dtype, device=w.device)
    _lib().weight_prep(w.detach(), row_scale, None, out, N, taps, C, hip.F32 if dtype == torch.float32 else hip.BF16)
    return out


_DIM_T = {}


def _upsampled_dgrad_weight(w, dtype):
    """Data gradient of `3x3 conv over a 2x nearest-upsampled map` onto the LOW-resolution input, as ONE 4x4 / stride 2 / pad 1
    convolution of the output gradient: summing the 2 x 2 children of a low-res pixel commutes into the taps,
        gx[y', x'] = sum_{t, s = 0..3} gy[2 y' - 1 + t, 2 x' - 1 + s] . Wk[t][s],   Wk[t][s] = sum_{kh in G(t), kw in G(s)} W[kh][kw],
        G(0) = {2}, G(1) = {1, 2}, G(2) = {0, 1}, G(3) = {0}
    - 16 taps per low-res pixel = 4 per high-res one instead of 9, and no high-resolution gradient map in memory (the transposed
    gather wrote it, 314 MB for the last decoder stage, and the footprint sum read it back).  w (Cout, 3, 3, Cin) fp32 master ->
    (Cin, 4, 4, Cout) in `dtype`: the weight operand of gwd_conv_forward with x = gy."""
    Cout, _, _, Cin = w.shape
    wk = torch.empty((Cin, 4, 4, Cout), dtype=dtype, device=w.device)
    _lib().upsample_taps_collapse(w.detach().float().contiguous(), wk)       # one launch (was ~8 element-wise ones; an einsum would be a library GEMM)
    return wk




def mask_levels(pad_mask, sizes):
    """Padding masks of the feature levels (F.interpolate(mask[None].float(), size).to(bool), backbone.py:81-88) with the cumulative
    counts of their un-masked pixels attached (`_gwd_counts`, what pos_sine needs): one launch per level."""
    lib = _lib()
    B = pad_mask.shape[0]
    full = pad_mask.contiguous().view(torch.uint8) if pad_mask.dtype == torch.bool else pad_mask.contiguous()
    out = []
    for h, w in sizes:
        lvl = torch.empty((B, h, w), dtype=torch.uint8, device=pad_mask.device)
        counts = torch.empty((B, h, w, 2), dtype=torch.int16, device=pad_mask.device)
        lib.pos_counts(full, lvl, counts)
        m = lvl.view(torch.bool)
        m._gwd_counts = counts
        out.append(m)
    return out


def pos_sine(mask, num_pos_feats, normalize, temperature=10000.0):
    """PositionEmbeddingSine (position_encoding.py:28-48) of a level mask from mask_levels(): (B,h,w,2F) fp32, one launch."""
    counts = mask._gwd_counts
    key = (num_pos_feats, float(temperature), str(mask.device))
    dim_t = _DIM_T.get(key)
    if dim_t is None:
        t = torch.arange(num_pos_feats, dtype=torch.float32, device=mask.device)
        dim_t = _DIM_T[key] = temperature ** (2 * torch.div(t, 2, rounding_mode="floor") / num_pos_feats)
    B, h, w, _ = counts.shape
    out = torch.empty((B, h, w, 2 * num_pos_feats), dtype=torch.float32, device=mask.device)
    _lib().pos_emit(counts, dim_t, out, normalize)
    return out


_STEM_PACKED = {}
_STEM_RETIRED = []


def stem(x, w, scale, shift):
    """ResNet stem (conv 7x7 s2 p3, 3 -> 64, folded FrozenBN, ReLU, max-pool 3x3 s2 p1) as ONE forward-only kernel
    (gwd_stem_forward); x bf16 (B,H,W,3) -> (B,Hp,Wp,64).  The stem is frozen on this path (backbone.py:62-64): no gradient.
    The operand-order copy of the weights is rebuilt when the weight or the scale changes (load_state_dict)."""
    lib = _lib()
    key = (w.data_ptr(), w._version, scale.data_ptr(), scale._version, str(w.device))
    ent = _STEM_PACKED.get(w.data_ptr())
    if ent is None or ent[0] != key:
        packed = torch.empty(hip.STEM_PACKED_ELEMS, dtype=torch.bfloat16, device=w.device)
        lib.stem_pack(w.detach(), scale.detach().float().contiguous(), packed)
        if ent is not None:
            _STEM_RETIRED.append(ent[1])                      # a captured graph may still read the previous copy
        ent = _STEM_PACKED[w.data_ptr()] = (key, packed)
    B, H, W, _ = x.shape
    y = torch.empty((B, hip.stem_out(H), hip.stem_out(W), 64), dtype=x.dtype, device=x.device)
    lib.stem_forward(x.detach().contiguous(), ent[1], shift.detach().float().contiguous(), y)
    return y


def padded_weight(w, geom, kind, dtype):
    """Zero-padded kernel-side copy of w (N, KH, KW, C), geom = (Np, Cg, Cgp): rows N -> Np and each of the G = C / Cg groups of
    input channels Cg -> Cgp.  kind 'fwd': (Np, KH, KW, G*Cgp); 't' (data gradient): (G*Cgp, KH, KW, Np)."""
    Np, Cg, Cgp = geom
    N, KH, KW, C = w.shape
    G = C // Cg
    if G * Cg != C or Np < N or Cgp < Cg:
        raise ValueError("padded_weight: geometry %r does not fit a weight of shape %r" % (geom, tuple(w.shape)))
    src = w.view(N, KH * KW, G, Cg)
    if kind == "fwd":
        out = torch.zeros((Np, KH, KW, G * Cgp), dtype=dtype, device=w.device)
        out.view(Np, KH * KW, G, Cgp)[:N, :, :, :Cg] = src
    else:
        out = torch.zeros((G * Cgp, KH, KW, Np), dtype=dtype, device=w.device)
        out.view(G, Cgp, KH * KW, Np)[:, :Cg, :, :N] = src.permute(2, 3, 1, 0)
    return out


def _padded_weight_for(w, geom, kind, dtype):
    if dtype == torch.bfloat16 and _ACTIVE_WEIGHTS is not None:
        cached = _ACTIVE_WEIGHTS.get(w, None, kind, geom)
        if cached is not None:
            return cached
    return padded_weight(w.detach(), geom, kind, dtype)


class _PadConvFn(torch.autograd.Function):
    """Stride-1 convolution (no bias, no activation) of a layer whose channel counts are not multiples of the MFMA / LDS-DMA
    granule, run on ZERO-PADDED channel counts: x (B, H, W, G*Cgp) holds G groups of Cg real channels each padded to Cgp with zeros,
    the result (B, Ho, Wo, Np) has its Cout real channels followed by zeros.  The kernels see an ordinary (G*Cgp -> Np) layer (the
    padding lives in the kernel-side weight copies: WeightCache / gwd_weight_prep_batch write the real entries into zeroed
    copies), the parameter and its gradient keep their own shape: the weight gradient is formed in the padded shape and folded
    back by gwd_unpad_add_batch.  User: the 30 / 60 / 300-channel pyramid of points_sample.py:45-125 (32 / 64 / 320 here)."""

    @staticmethod
    def forward(ctx, x, w, pad, geom, sink, fanout=False):
        lib = _lib()
        ctx.fan = bool(fanout)
        Np, Cg, Cgp = geom
        B, Hi, Wi, Cin = x.shape
        N, KH, KW, C = w.shape
        G = C // Cg
        if Cin != G * Cgp:
            raise ValueError("conv2d_padded: input has %d channels, the padded weight expects %d" % (Cin, G * Cgp))
        Ho, Wo = Hi + 2 * pad - KH + 1, Wi + 2 * pad - KW + 1
        x = x.contiguous()
        y = torch.empty((B, Ho, Wo, Np), dtype=x.dtype, device=x.device)
        dims = (B, Hi, Wi, Cin, Ho, Wo, Np, KH, KW)
        lib.conv_forward(x, _padded_weight_for(w, geom, "fwd", x.dtype), y, dims, stride=1, pad=pad)
        ctx.save_for_backward(x, w)
        ctx.cfg = (dims, pad, geom, sink)
        if fanout:
            return y, x.view_as(x)                   # as _ConvFn: the input again, for its second consumer
        return y

    @staticmethod
    def backward(ctx, gy, *g_fan):
        lib = _lib()
        g_in = g_fan[0].contiguous() if (ctx.fan and g_fan and g_fan[0] is not None) else None
        x, w = ctx.saved_tensors
        dims, pad, geom, sink = ctx.cfg
        B, Hi, Wi, Cin, Ho, Wo, Np, KH, KW = dims
        _, Cg, Cgp = geom
        N, C = w.shape[0], w.shape[-1]
        gy = gy.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            lib.conv_forward(gy, _padded_weight_for(w, geom, "t", x.dtype), gx, (B, Ho, Wo, Np, Hi, Wi, Cin, KH, KW), stride=1, pad=pad,
                             gather=GATHER_TRANSPOSED, residual=g_in)
        elif g_in is not None:
            gx = g_in
        if ctx.needs_input_grad[1]:
            tmp = WGRADS.scratch((Np, KH, KW, Cin), x.device)
            fold = (N, KH * KW, C // Cg, Cg, Cgp)
            if sink is not None:
                WGRADS.add(x, gy, tmp, dims, dict(stride=1, pad=pad), sink[1], unpad=(sink[0].view(-1),) + fold)
            else:
                lib.conv_wgrad(x, gy, tmp, dims, stride=1, pad=pad)
                gw = torch.zeros(w.shape, dtype=torch.float32, device=w.device)
                lib.unpad_add_batch([(tmp, gw.view(-1)) + fold])
        return gx, gw, None, None, None, None


def conv2d_padded(x, w, pad, geom, fanout=False):
    """See _PadConvFn.  x (B, H, W, G*Cgp) zero-padded, w (Cout, KH, KW, G*Cg) fp32 master, geom = (Np, Cg, Cgp)."""
    return _PadConvFn.apply(x, w, int(pad), tuple(int(v) for v in geom), _sink(w), bool(fanout))


class _ConvFn(torch.autograd.Function):
    """y = act_scale * act(conv(x, w * row_scale) + shift + residual), all in one kernel launch; with `mult` (an element-wise
    dropout multiplier) y = act_scale * act(conv + shift) * mult + residual: the skip is added after the dropout."""

    @staticmethod
    def forward(ctx, x, w, bias, residual, row_scale, shift_const, stride, pad, act, act_scale, virt, shadow, sinks, mult=None, fanout=False,
                in_gate=ACT_NONE, defer=False, gate_src=None):
        lib = _lib()
        ctx.fan = bool(fanout)
        # in_gate / defer: the activation backward of a ReLU / ELU layer moves into the data-gradient epilogue of its ONLY consumer
        # (gwd_conv_desc.gate = the consumer's saved input): the producer (defer) passes the incoming gradient on unchanged, the
        # consumer (in_gate = the producer's activation) returns the gradient w.r.t. the producer's PRE-activation value
        # GELU (the MLPs): its derivative needs the producer's PRE-activation value, so the producer (defer) returns it as a second
        # output and the consumer gets it as gate_src (in_gate = ACT_GELU); ReLU / ELU gates are rebuilt from the consumer's own input
        ctx.in_gate, ctx.defer = int(in_gate), bool(defer)
        if defer and (act not in (ACT_RELU, ACT_ELU, ACT_GELU) or act_scale != 1.0 or mult is not None or (act == ACT_GELU and fanout)):
            raise ValueError("conv2d: defer needs ReLU / ELU / GELU with act_scale 1 and no dropout multiplier")
        if in_gate not in (ACT_NONE, ACT_RELU, ACT_ELU, ACT_GELU) or (in_gate == ACT_GELU) != (gate_src is not None):
            raise ValueError("conv2d: in_gate is ReLU or ELU, or GELU with the producer's pre-activation tensor as gate_src")
        B, Hi, Wi, Cin = x.shape
        Cout, KH, KW, Cw = w.shape
        if Cw != Cin:
            raise ValueError("conv2d: weight expects %d input channels, input has %d" % (Cw, Cin))
        if virt is None:
            Ho = (Hi + 2 * pad - KH) // stride + 1
            Wo = (Wi + 2 * pad - KW) // stride + 1
            gather, vv = GATHER_CONV, (0, 0)
        else:
            Ho, Wo = virt[0] + 2 * pad - KH + 1, virt[1] + 2 * pad - KW + 1
            gather, vv = GATHER_UPSAMPLED, tuple(virt)
        x = x.contiguous()
        wk = _weight_for(w, row_scale, x.dtype, shadow)
        shift = shift_const if bias is None else (bias.detach().float() if shift_const is None else shift_const + bias.detach().float())
        y = torch.empty((B, Ho, Wo, Cout), dtype=x.dtype, device=x.device)
        need_grad = any(ctx.needs_input_grad[:4])
        z = torch.empty_like(y) if (act == ACT_GELU and need_grad) else None
        if residual is not None:
            residual = residual.contiguous()
        dims = (B, Hi, Wi, Cin, Ho, Wo, Cout, KH, KW)
        if mult is not None:
            mult = mult.contiguous()
        lib.conv_forward(x, wk, y, dims, z=z, shift=shift, residual=residual, stride=stride, pad=pad, gather=gather,
                         virt=vv, act=act, act_scale=act_scale, mult=mult)
        ctx.cfg = (dims, stride, pad, act, act_scale, gather, vv, bias is not None, residual is not None)
        ctx.sinks = sinks
        if mult is not None and (act not in (ACT_NONE, ACT_RELU) or (act == ACT_RELU and residual is not None)):
            raise ValueError("conv2d: a dropout multiplier is supported behind no activation (+ skip) or behind ReLU (no skip) only")
        # ReLU's backward needs only the sign of the output: y * mult has the sign of relu(v) wherever mult > 0, and where
        # mult == 0 the incoming gradient is multiplied by zero anyway
        ctx.save_for_backward(x, w, row_scale, z if act == ACT_GELU else (y if act != ACT_NONE else None), mult, gate_src)
        if defer and act == ACT_GELU:
            if z is None:                               # no gradient wanted anywhere: nobody will read it
                z = y
            ctx.mark_non_differentiable(z)
            ctx.set_materialize_grads(False)            # or every backward starts with a zero fill the size of z for "its" gradient
            return y, z
        if fanout:
            # second output: x again, for a second consumer of the layer's input (a skip connection).  With it this node is x's only
            # consumer, both gradients arrive in ONE backward call and the skip's joins the data gradient in the kernel epilogue -
            # no accumulation pass over the map (PyrBlock, Bottleneck)
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, gy, *g_fan):
        lib = _lib()
        if gy is None:                                 # (only where materialize_grads is off: a GELU producer nobody consumed)
            return (None,) * 18
        g_in = g_fan[0].contiguous() if (ctx.fan and g_fan and g_fan[0] is not None) else None
        x, w, row_scale, ref, mult, gate_src = ctx.saved_tensors
        dims, stride, pad, act, act_scale, gather, vv, has_bias, has_res = ctx.cfg
        B, Hi, Wi, Cin, Ho, Wo, Cout, KH, KW = dims
        gy = gy.contiguous()
        g_skip = gy                                    # the post-dropout skip connection gets the incoming gradient as is
        rows = B * Ho * Wo
        w_sink, b_sink = ctx.sinks if ctx.sinks is not None else (None, None)
        bias_done = False
        dv = None
        if mult is not None and has_bias and ctx.needs_input_grad[2] and b_sink is not None:
            # dropout multiplier, activation backward and the bias gradient in ONE pass (was: an ATen multiply, then the rest)
            dv = torch.empty_like(gy)
            bias_done = lib.act_backward_colsum(gy, ref, dv, b_sink[0], rows, Cout, act, act_scale, mult=mult)
            if bias_done:
                if b_sink[1] is not None:
                    b_sink[1]()
            else:
                dv = None
        if ctx.defer:
            dv = gy                                    # the consumer's data-gradient epilogue has applied act'(.) already
        if dv is None:
            if mult is not None:
                gy = gy * mult
            if act != ACT_NONE or act_scale != 1.0:
                dv = torch.empty_like(gy)
                if has_bias and ctx.needs_input_grad[2] and b_sink is not None:
                    # activation backward and the bias gradient (column sums of dv) in one pass, straight into the flat gradient
                    bias_done = lib.act_backward_colsum(gy, ref, dv, b_sink[0], rows, Cout, act, act_scale)
                    if bias_done and b_sink[1] is not None:
                        b_sink[1]()
                if not bias_done:
                    lib.act_backward(gy, ref, dv, None, rows, Cout, act, act_scale)
            else:
                dv = gy
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            wt = _weight_transposed(w, row_scale, x.dtype)
            gate = dict(gate=(x if gate_src is None else gate_src), gate_act=ctx.in_gate) if ctx.in_gate != ACT_NONE else {}
            if (gather == GATHER_UPSAMPLED and x.is_cuda and x.dtype == torch.bfloat16 and KH == 3 and KW == 3
                    and pad == 1 and vv == (2 * Hi, 2 * Wi) and row_scale is None and (g_in is None or not gate)):
                # one strided convolution of the gradient with the 4x4 collapse of the weights (see _upsampled_dgrad_weight)
                gx = torch.empty_like(x)
                lib.conv_forward(dv, _upsampled_dgrad_weight(w, x.dtype), gx, (B, Ho, Wo, Cout, Hi, Wi, Cin, 4, 4), stride=2, pad=1,
                                 residual=g_in, **gate)
            elif gather == GATHER_UPSAMPLED:
                gxv = torch.empty((B, vv[0], vv[1], Cin), dtype=x.dtype, device=x.device)
                lib.conv_forward(dv, wt, gxv, (B, Ho, Wo, Cout, vv[0], vv[1], Cin, KH, KW), stride=1, pad=pad,
                                 gather=GATHER_TRANSPOSED)
                if gate and g_in is None:              # the gate rides on the footprint sum
                    gx = _nearest_upsample_backward(gxv, Hi, Wi, gate=x, gate_act=ctx.in_gate)
                else:
                    gx = _nearest_upsample_backward(gxv, Hi, Wi)
                    if g_in is not None:
                        gx = gx + g_in
                    if gate:                           # behind the skip gradient: a pass of its own
                        gated = torch.empty_like(gx)
                        lib.act_backward(gx, x, gated, None, B * Hi * Wi, Cin, ctx.in_gate, 1.0)
                        gx = gated
            else:
                gx = torch.empty_like(x)
                done = False
                if KH == 1 and KW == 1 and stride > 1 and pad == 0 and not gate:
                    # 1x1 / stride s: only the pixels (s i, s j) get a gradient - a plain GEMM over the OUTPUT pixels, then placement
                    # (+ the skip gradient) in one pass; the transposed gather spent 3/4 of its work on zero-page products
                    q = torch.empty((B, Ho, Wo, Cin), dtype=x.dtype, device=x.device)
                    lib.conv_forward(dv, wt, q, (B, Ho, Wo, Cout, Ho, Wo, Cin, 1, 1), stride=1, pad=0, gather=GATHER_TRANSPOSED)
                    done = lib.stride_place(q, g_in, gx, stride)
                if not done:
                    if lib.conv_forward(dv, wt, gx, (B, Ho, Wo, Cout, Hi, Wi, Cin, KH, KW), stride=stride, pad=pad,
                                        gather=GATHER_TRANSPOSED, residual=g_in, **gate) is False:
                        # no kernel with the GELU gate for this shape: the data gradient, then the gate as a pass of its own
                        lib.conv_forward(dv, wt, gx, (B, Ho, Wo, Cout, Hi, Wi, Cin, KH, KW), stride=stride, pad=pad,
                                         gather=GATHER_TRANSPOSED, residual=g_in)
                        gated = torch.empty_like(gx)
                        lib.act_backward(gx, gate["gate"], gated, None, B * Hi * Wi, Cin, ctx.in_gate, 1.0)
                        gx = gated
        elif g_in is not None:
            gx = g_in
        if ctx.needs_input_grad[1]:
            # row_scale (folded FrozenBN: the layer ran with w * scale) multiplies the gradient inside the kernel's epilogue
            if (w_sink is not None and gather == GATHER_UPSAMPLED and x.is_cuda and x.dtype == torch.bfloat16
                    and KH == 3 and KW == 3 and pad == 1 and vv == (2 * Hi, 2 * Wi) and row_scale is None):
                # the weight gradient through the same 4x4 / stride 2 form: D[ci][t][s][co] = sum_{y', x'} x[y', x', ci] gy[2 y' - 1 + t, 2 x' - 1 + s, co]
                # is the ordinary weight gradient of that strided convolution (operands swapped: its input is gy, its output side x) - 16 taps per
                # low-res pixel instead of 36, on the grouped GEMM kernels - and dW[co][kh][kw][ci] = sum_{t in T(kh), s in T(kw)} D[ci][t][s][co],
                # T(0) = {2, 3}, T(1) = {1, 2}, T(2) = {0, 1}, added to the flat gradient when the batch has been issued
                D = WGRADS.scratch((Cin, 4, 4, Cout), x.device)

                def fold(D=D, sink=w_sink, shape=(Cout, KH, KW, Cin)):
                    lib.upsample_taps_fold(D, sink[0].view(shape))
                    if sink[1] is not None:
                        sink[1]()

                WGRADS.add(dv, x, D, (B, Ho, Wo, Cout, Hi, Wi, Cin, 4, 4), dict(stride=2, pad=1), fold)
            elif w_sink is not None:
                # the kernel ACCUMULATES (fp32 atomics): add straight into the flat gradient buffer, no temporary
                WGRADS.add(x, dv, w_sink[0], dims, dict(stride=stride, pad=pad, gather=gather, virt=vv, scale=row_scale), w_sink[1])
            else:
                gw = torch.zeros(w.shape, dtype=torch.float32, device=w.device)
                lib.conv_wgrad(x, dv, gw, dims, stride=stride, pad=pad, gather=gather, virt=vv, scale=row_scale)
        if has_bias and ctx.needs_input_grad[2] and not bias_done:
            if b_sink is not None:
                COLSUMS.add(dv, b_sink[0], rows, Cout, b_sink[1])
            else:
                gb = torch.zeros(Cout, dtype=torch.float32, device=gy.device)
                lib.colsum(dv, gb, rows, Cout)
        gres = (g_skip if mult is not None else dv) if (has_res and ctx.needs_input_grad[3]) else None
        return gx, gw, gb, gres, None, None, None, None, None, None, None, None, None, None, None, None, None, None


def _nearest_upsample_backward(gv, Hi, Wi, gate=None, gate_act=ACT_NONE):
    """Sum the gradient of a nearest-upsampled view back onto its source pixels: one gather pass of gwd_resample_backward
    (each source pixel sums its own footprint; was a strided aten::sum at 1.5 TB/s).  gate: the source map when it is the output of
    an activation whose backward is applied to the sums in the same pass (conv2d defer / in_gate)."""
    B, Hv, Wv, C = gv.shape
    gx = torch.empty((B, Hi, Wi, C), dtype=gv.dtype, device=gv.device)
    lib = _lib()
    done = lib.resample_backward(gv.contiguous(), gx, B, Hi, Wi, Hv, Wv, C, hip.RESAMPLE_NEAREST, gate=gate, gate_act=gate_act)
    if gate is not None and done is False:
        gated = torch.empty_like(gx)
        lib.act_backward(gx, gate, gated, None, B * Hi * Wi, C, gate_act, 1.0)
        gx = gated
    return gx


def _sink(p, shape=None):
    """(flat-gradient view, hook) of a parameter managed by engine.TrainStep, else None."""
    g = getattr(p, "_gwd_grad", None)
    if g is None:
        return None
    return (g if shape is None else g.view(shape), getattr(p, "_gwd_hook", None))


def conv2d(x, w, bias=None, *, stride=1, pad=0, act=ACT_NONE, act_scale=1.0, residual=None, row_scale=None,
           shift=None, upsample_to=None, mult=None, fanout=False, in_gate=ACT_NONE, defer=False):
    """x (B,H,W,Cin); w (Cout,KH,KW,Cin) fp32 master; bias fp32 parameter or None.
    row_scale / shift: constant per-Cout tensors of a folded FrozenBatchNorm.
    upsample_to=(Hv,Wv): convolve a nearest-upsampled view of x without materialising it.
    fanout: return (y, x'), x' = x for the OTHER consumer of the input (use x' instead of x there): see _ConvFn.forward.
    defer / in_gate: a ReLU / ELU layer whose output has exactly ONE consumer (this function again, or the fan-out chain that starts
    with it) is called with defer=True and that consumer with in_gate=<the producer's activation>: the activation's backward then
    runs in the consumer's data-gradient epilogue instead of as a pass of its own (act_gate_enabled(): both or neither)."""
    sinks = (_sink(w), _sink(bias) if bias is not None else None)
    return _ConvFn.apply(x, w, bias, residual, row_scale, shift, stride, pad, act, float(act_scale), upsample_to,
                         getattr(w, "_gwd_bf16", None), sinks if (sinks[0] or sinks[1]) else None, mult, bool(fanout), in_gate, defer)


def act_gate_enabled():
    """ReLU / ELU layers with a single consumer run their backward in that consumer's data-gradient epilogue (conv2d defer / in_gate)."""
    return True


def linear(x, w, bias=None, act=ACT_NONE, rows=None, residual=None, mult=None, fanout=False, defer=False, in_gate=ACT_NONE, gate_src=None):
    """x (..., K) @ w(N, K)^T + bias, optional fused activation; same kernel as conv2d (1x1, one pixel per row).
    rows=(r0, r1): use only that row range of a packed parameter (the q/k/v blocks of an attention in-projection);
    the gradient then goes straight into that slice of the parameter's flat gradient instead of through a
    zero-filled full-size SliceBackward temporary."""
    K = x.shape[-1]
    lead = x.shape[:-1]
    x2 = x.reshape(-1, 1, 1, K)
    shadow = getattr(w, "_gwd_bf16", None)
    ws, bs = _sink(w), (_sink(bias) if bias is not None else None)
    if rows is not None:
        r0, r1 = rows
        if ws is not None:
            ws = (ws[0][r0:r1], ws[1])
        if bs is not None:
            bs = (bs[0][r0:r1], bs[1])
        shadow = None if shadow is None else shadow[r0:r1]
        w = w[r0:r1]
        bias = None if bias is None else bias[r0:r1]
    n = w.shape[0]
    sinks = (None if ws is None else (ws[0].view(n, 1, 1, K), ws[1]), bs)
    res = None if residual is None else residual.reshape(-1, 1, 1, n)
    mul = None if mult is None else mult.reshape(-1, 1, 1, n)
    fan = bool(fanout)
    # defer (act = GELU): the activation's backward runs in the data-gradient epilogue of the layer's ONLY consumer; returns (y, z), z the
    # pre-activation tensor that consumer takes as gate_src (with in_gate = ACT_GELU) - the fc1 / fc2 pair of an MLP (layers.Mlp)
    gs = None if gate_src is None else gate_src.reshape(-1, 1, 1, K)
    y = _ConvFn.apply(x2, w.view(n, 1, 1, K), bias, res, None, None, 1, 0, act, 1.0, None,
                      None if shadow is None else shadow.view(n, 1, 1, K),
                      sinks if (sinks[0] or sinks[1]) else None, mul, fan, int(in_gate), bool(defer), gs)
    if defer and act == ACT_GELU:
        return y[0].view(*lead, n), y[1].view(*lead, n)
    if fanout:                              # (y, x again for the input's second consumer): see _ConvFn.forward
        return (y[0].view(*lead, n), y[1].view(x.shape)) if fan else (y.view(*lead, n), x)
    return y.view(*lead, n)


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, gelu, sinks, residual, fanout=False, in_gate=ACT_NONE):
        lib = _lib()
        ctx.sinks = sinks
        ctx.fan = bool(fanout)
        if in_gate not in (ACT_NONE, ACT_ELU):
            raise ValueError("layer_norm: in_gate is ELU (the gate is rebuilt from the normalised value: continuous derivatives only)")
        ctx.in_gate = int(in_gate)                  # x is the output of a conv2d(..., act=ELU, defer=True): its backward runs in ours
        ctx.has_res = residual is not None
        x = x.contiguous()
        ld = x.shape[-1]
        C = gamma.shape[0] if gamma is not None else ld       # fewer affine entries than channels: the rest is zero padding (_PadConvFn)
        rows = x.numel() // ld
        y = torch.empty_like(x)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        g = gamma.detach() if gamma is not None else None
        b = beta.detach() if beta is not None else None
        lib.layernorm_forward(x, g, b, y, mean, rstd, rows, C, gelu, residual=None if residual is None else residual.contiguous(),
                              ld=0 if ld == C else ld)
        ctx.save_for_backward(x, g, b, mean, rstd)
        ctx.gelu = gelu
        if fanout:
            return y, x.view_as(x)        # x again for its second consumer (the block's skip): see _ConvFn.forward
        return y

    @staticmethod
    def backward(ctx, gy, *g_fan):
        lib = _lib()
        g_in = g_fan[0].contiguous() if (ctx.fan and g_fan and g_fan[0] is not None) else None
        x, g, b, mean, rstd = ctx.saved_tensors
        ld = x.shape[-1]
        C = g.shape[0] if g is not None else ld
        rows = x.numel() // ld
        gy = gy.contiguous()
        gx = torch.empty_like(x)
        dg = db = None
        direct = ctx.sinks is not None
        if direct:
            (dg, h1), (db, h2) = ctx.sinks
        elif g is not None:
            dg = torch.zeros(C, dtype=torch.float32, device=x.device)
            db = torch.zeros(C, dtype=torch.float32, device=x.device)
        fused = lib.layernorm_backward(gy, x, g, b, mean, rstd, gx, dg, db, rows, C, ctx.gelu, ld=0 if ld == C else ld, gskip=g_in,
                                       elu_input=ctx.in_gate == ACT_ELU)
        if fused is False:                                 # no vector kernel for this width: the skip gradient / the gate as passes
            if g_in is not None:
                gx = gx + g_in
            if ctx.in_gate != ACT_NONE:
                gated = torch.empty_like(gx)
                lib.act_backward(gx, x, gated, None, rows, ld, ctx.in_gate, 1.0)
                gx = gated
        gres = gy if ctx.has_res else None                # y = LN(x) + residual: the skip gets the incoming gradient as is
        if direct:
            for h in (h1, h2):
                if h is not None:
                    h()
            return gx, None, None, None, None, gres, None, None
        return gx, dg, db, None, None, gres, None, None


def layer_norm(x, gamma, beta, gelu=False, residual=None, fanout=False, in_gate=ACT_NONE):
    """LayerNorm over the last dim (eps 1e-5) with optional fused exact GELU; residual (same shape) is added afterwards.
    fanout: return (y, x') with x' = x for the OTHER consumer of the input (the skip of a pre-norm block): x's two gradients then
    meet inside the LayerNorm backward kernel instead of an accumulation pass."""
    sinks = None
    if gamma is not None:
        sg, sb = _sink(gamma), _sink(beta)
        sinks = (sg, sb) if (sg is not None and sb is not None) else None
    return _LayerNormFn.apply(x, gamma, beta, bool(gelu), sinks, residual, bool(fanout), in_gate)


class _ConvLnFn(torch.autograd.Function):
    """ConvLn (points_sample.py:12-25): stride-1 convolution without bias -> LayerNorm over channels [-> GELU] [+ residual] as ONE
    forward launch - the LayerNorm runs in the implicit GEMM's epilogue on the fp32 accumulators (gwd_conv_desc.ln_*), so the
    normalisation pass and its read of the convolution's output disappear.  The kernel leaves what the unfused pair left: the
    convolution's output z and the row statistics, so the backward is the unfused one (gwd_layernorm_backward, then data / weight
    gradient of the convolution).  geom = None, or (Np, Cg, Cgp) for a layer on zero-padded channel counts (_PadConvFn).  Where the
    library has no fused kernel for the shape the two forward kernels run here instead - same results, same saved tensors."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, residual, pad, gelu, geom, w_sink, ln_sinks, fanout):
        lib = _lib()
        ctx.fan = bool(fanout)
        B, Hi, Wi, Cin = x.shape
        N, KH, KW, Cw = w.shape
        if geom is None:
            Np = N
            if Cw != Cin:
                raise ValueError("conv_ln: weight expects %d input channels, input has %d" % (Cw, Cin))
            wk = _weight_for(w, None, x.dtype, getattr(w, "_gwd_bf16", None))
        else:
            Np, Cg, Cgp = geom
            if Cin != (Cw // Cg) * Cgp:
                raise ValueError("conv_ln: input has %d channels, the padded weight expects %d" % (Cin, (Cw // Cg) * Cgp))
            wk = _padded_weight_for(w, geom, "fwd", x.dtype)
        Ho, Wo = Hi + 2 * pad - KH + 1, Wi + 2 * pad - KW + 1
        x = x.contiguous()
        rows = B * Ho * Wo
        y = torch.empty((B, Ho, Wo, Np), dtype=x.dtype, device=x.device)
        need_grad = any(ctx.needs_input_grad)
        z = torch.empty_like(y) if need_grad else None      # inference: the convolution's own output is never written
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        g, b = gamma.detach(), beta.detach()
        res = None if residual is None else residual.contiguous()
        dims = (B, Hi, Wi, Cin, Ho, Wo, Np, KH, KW)
        fused = lib.conv_forward(x, wk, y, dims, z=z, scale=g, shift=b, residual=res, stride=1, pad=pad,
                                 act=ACT_GELU if gelu else ACT_NONE, ln=(mean, rstd, N))
        if fused is False:
            z = torch.empty_like(y) if z is None else z
            lib.conv_forward(x, wk, z, dims, stride=1, pad=pad)
            lib.layernorm_forward(z, g, b, y, mean, rstd, rows, N, gelu, residual=res, ld=0 if Np == N else Np)
        if need_grad:
            ctx.save_for_backward(x, w, z, g, b, mean, rstd)
        ctx.cfg = (dims, pad, bool(gelu), geom, w_sink, ln_sinks, residual is not None)
        if fanout:
            return y, x.view_as(x)                   # the input again, for its second consumer (the block's skip): see _ConvFn
        return y

    @staticmethod
    def backward(ctx, gy, *g_fan):
        lib = _lib()
        g_in = g_fan[0].contiguous() if (ctx.fan and g_fan and g_fan[0] is not None) else None
        x, w, z, g, b, mean, rstd = ctx.saved_tensors
        dims, pad, gelu, geom, w_sink, ln_sinks, has_res = ctx.cfg
        B, Hi, Wi, Cin, Ho, Wo, Np, KH, KW = dims
        N = g.shape[0]
        rows = B * Ho * Wo
        gy = gy.contiguous()
        # ---- LayerNorm backward (as _LayerNormFn.backward): gradient w.r.t. the convolution's output, d gamma / d beta into their sinks
        gz = torch.empty_like(z)
        dg = db = None
        if ln_sinks is not None:
            (dg, h1), (db, h2) = ln_sinks
        else:
            dg = torch.zeros(N, dtype=torch.float32, device=x.device)
            db = torch.zeros(N, dtype=torch.float32, device=x.device)
        lib.layernorm_backward(gy, z, g, b, mean, rstd, gz, dg, db, rows, N, gelu, ld=0 if Np == N else Np)
        if ln_sinks is not None:
            for h in (h1, h2):
                if h is not None:
                    h()
        # ---- convolution backward (as _ConvFn / _PadConvFn.backward)
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            wt = _weight_transposed(w, None, x.dtype) if geom is None else _padded_weight_for(w, geom, "t", x.dtype)
            lib.conv_forward(gz, wt, gx, (B, Ho, Wo, Np, Hi, Wi, Cin, KH, KW), stride=1, pad=pad, gather=GATHER_TRANSPOSED, residual=g_in)
        elif g_in is not None:
            gx = g_in
        if ctx.needs_input_grad[1]:
            if geom is None:
                if w_sink is not None:
                    WGRADS.add(x, gz, w_sink[0], dims, dict(stride=1, pad=pad), w_sink[1])
                else:
                    gw = torch.zeros(w.shape, dtype=torch.float32, device=w.device)
                    lib.conv_wgrad(x, gz, gw, dims, stride=1, pad=pad)
            else:
                _, Cg, Cgp = geom
                C = w.shape[-1]
                tmp = WGRADS.scratch((Np, KH, KW, Cin), x.device)
                fold = (N, KH * KW, C // Cg, Cg, Cgp)
                if w_sink is not None:
                    WGRADS.add(x, gz, tmp, dims, dict(stride=1, pad=pad), w_sink[1], unpad=(w_sink[0].view(-1),) + fold)
                else:
                    lib.conv_wgrad(x, gz, tmp, dims, stride=1, pad=pad)
                    gw = torch.zeros(w.shape, dtype=torch.float32, device=w.device)
                    lib.unpad_add_batch([(tmp, gw.view(-1)) + fold])
        gres = gy if has_res else None                      # y = ... + residual: the skip gets the incoming gradient as is
        if ln_sinks is not None:
            return gx, gw, None, None, gres, None, None, None, None, None, None
        return gx, gw, dg, db, gres, None, None, None, None, None, None


def conv_ln(x, w, gamma, beta, pad, gelu=False, residual=None, geom=None, fanout=False):
    """See _ConvLnFn.  x (B,H,W,Cin) bf16 on the device; w (Cout,KH,KW,Cin') fp32 master; gamma / beta (Cout,) fp32."""
    sg, sb = _sink(gamma), _sink(beta)
    ln_sinks = (sg, sb) if (sg is not None and sb is not None) else None
    return _ConvLnFn.apply(x, w, gamma, beta, residual, int(pad), bool(gelu), None if geom is None else tuple(int(v) for v in geom),
                           _sink(w), ln_sinks, bool(fanout))


def _as4(t):
    """(..., rows, cols) tensor -> a 4-D (b0, b1, rows, cols) view with unit inner stride (a copy only when the inner stride is not 1)."""
    if t.stride(-1) != 1:
        t = t.contiguous()
    while t.dim() < 4:
        t = t.unsqueeze(0)
    if t.dim() > 4:
        t = t.reshape(-1, *t.shape[-3:])
    return t


def _bmm_raw(a, a_km, b, b_km, M, N, K, alpha=1.0, out_dtype=None, splits=1):
    """alpha * A @ B^T with A (M x K), B (N x K) given as 4-D views, either possibly k-major; batch dims broadcast from size 1."""
    nb0, nb1 = max(a.shape[0], b.shape[0]), max(a.shape[1], b.shape[1])
    acc = splits > 1
    c = (torch.zeros if acc else torch.empty)((nb0, nb1, M, N), dtype=torch.float32 if acc else a.dtype, device=a.device)
    _lib().bmm(a, b, c, M, N, K, a_kmajor=a_km, b_kmajor=b_km, alpha=alpha, accumulate=acc, splits=splits)
    return c if (out_dtype is None or c.dtype == out_dtype) else c.to(out_dtype)


def _bmm_splits(M, N, K, batches):
    """Workgroups along the reduction when the tiles alone leave the chip idle (d refer = d rg^T @ xg: 19 200 pixels onto 80 x 64)."""
    tiles = ((M + 63) // 64) * ((N + 63) // 64) * batches
    if tiles >= 256 or K < 2048:
        return 1
    return max(1, min(K // 512, 512 // tiles))


class _MatmulFn(torch.autograd.Function):
    """alpha * a @ b^T (trans_b) or alpha * a @ b over leading batch dims, forward and both gradients on gwd_bmm - strided operands,
    no transposed copies.  a (..., M, K); b (..., N, K) when trans_b else (..., K, N); batch dims of a and b equal (or 1: broadcast)."""

    @staticmethod
    def forward(ctx, a, b, trans_b, alpha):
        a4, b4 = _as4(a), _as4(b)
        M, K = a4.shape[-2:]
        N = b4.shape[-2] if trans_b else b4.shape[-1]
        c = _bmm_raw(a4, False, b4, not trans_b, M, N, K, alpha)
        ctx.save_for_backward(a4, b4)
        ctx.cfg = (trans_b, alpha, a.shape, b.shape, M, N, K)
        lead = torch.broadcast_shapes(a.shape[:-2], b.shape[:-2])
        return c.reshape(*lead, M, N)

    @staticmethod
    def backward(ctx, gc):
        a4, b4 = ctx.saved_tensors
        trans_b, alpha, ashape, bshape, M, N, K = ctx.cfg
        g4 = _as4(gc)
        ga = gb = None
        batches = g4.shape[0] * g4.shape[1]
        if ctx.needs_input_grad[0]:
            # d a (M x K) = alpha * gc (M x N) @ B'^T, B' (K x N): b (N x K) is k-major for it, b (K x N) is not
            ga = _bmm_raw(g4, False, b4, trans_b, M, K, N, alpha)
            if a4.shape[0] != g4.shape[0] or a4.shape[1] != g4.shape[1]:
                ga = ga.sum(dim=[i for i in (0, 1) if a4.shape[i] != g4.shape[i]], keepdim=True)
            ga = ga.reshape(ashape)
        if ctx.needs_input_grad[1]:
            sp = _bmm_splits(N if trans_b else K, K if trans_b else N, M, batches)
            if trans_b:       # d b (N x K) = alpha * gc^T (N x M) @ a'^T with a' (K x M): both k-major (the reduction runs over M)
                gb = _bmm_raw(g4, True, a4, True, N, K, M, alpha, out_dtype=b4.dtype, splits=sp)
            else:             # d b (K x N) = alpha * a^T (K x M) @ gc'^T with gc' (N x M): both k-major
                gb = _bmm_raw(a4, True, g4, True, K, N, M, alpha, out_dtype=b4.dtype, splits=sp)
            if b4.shape[0] != g4.shape[0] or b4.shape[1] != g4.shape[1]:
                gb = gb.sum(dim=[i for i in (0, 1) if b4.shape[i] != g4.shape[i]], keepdim=True)
            gb = gb.reshape(bshape)
        return ga, gb, None, None


def matmul_nt(a, b, alpha=1.0):
    """alpha * a @ b.transpose(-1, -2) on the library's own batched GEMM (gwd_bmm)."""
    return _MatmulFn.apply(a, b, True, float(alpha))


def matmul_nn(a, b, alpha=1.0):
    """alpha * a @ b on the library's own batched GEMM (gwd_bmm)."""
    return _MatmulFn.apply(a, b, False, float(alpha))


class _SoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        L = x.shape[-1]
        y = torch.empty_like(x)
        _lib().softmax_forward(x, y, x.numel() // L, L)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        L = y.shape[-1]
        gx = torch.empty_like(y)
        _lib().softmax_backward(gy.contiguous(), y, gx, y.numel() // L, L)
        return gx


def softmax_lastdim(x):
    return _SoftmaxFn.apply(x)


class _AttnSoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, key_mask, scale):
        x = x.contiguous()
        L = x.shape[-1]
        rows = x.numel() // L
        y = torch.empty_like(x)
        if key_mask is not None:
            key_mask = key_mask.contiguous().view(torch.uint8) if key_mask.dtype == torch.bool else key_mask.contiguous()
            rpm = rows // (key_mask.numel() // L)
        else:
            rpm = 1
        _lib().softmax_masked_forward(x, key_mask, y, rows, L, rpm, scale)
        ctx.save_for_backward(y)
        ctx.scale = scale
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        L = y.shape[-1]
        gx = torch.empty_like(y)
        _lib().softmax_scaled_backward(gy.contiguous(), y, gx, y.numel() // L, L, ctx.scale)
        return gx, None, None


def attention_softmax(scores, key_padding_mask=None, scale=1.0):
    """softmax(scale * scores + key-padding mask) over the last dim; scores (B, H, L, S), mask (B, S) bool (True = pad)."""
    return _AttnSoftmaxFn.apply(scores, key_padding_mask, float(scale))


class _MhaFlashFn(torch.autograd.Function):
    """dropout(softmax(scale q k^T + key mask)) v with the heads merged, on the matrix cores (gwd_mha_flash_forward /
    _backward, csrc/mfattn.hip): only the merged output and one log-sum-exp per (image, head, query) are saved.
    `qk` is either one packed (B, L, 2E) projection (self-attention: q = [..., :E], k = [..., E:], ONE gradient tensor, no
    slice-backward zero-fill + add) or a (q, k) pair."""

    @staticmethod
    def forward(ctx, qk, k_sep, v, H, key_padding_mask, mult, scale):
        lib = _lib()
        packed = k_sep is None
        E = v.shape[-1]
        q, k = (qk[..., :E], qk[..., E:]) if packed else (qk, k_sep)
        B, L, S = q.shape[0], q.shape[1], k.shape[1]
        out = torch.empty((B, L, E), dtype=q.dtype, device=q.device)
        lse = torch.empty((B, H, L), dtype=torch.float32, device=q.device)
        if key_padding_mask is not None:
            key_padding_mask = key_padding_mask.contiguous().view(torch.uint8) if key_padding_mask.dtype == torch.bool else key_padding_mask.contiguous()
        lib.mha_flash_forward(q, k, v, key_padding_mask, mult, out, lse, H, scale)
        ctx.save_for_backward(qk, k_sep, v, key_padding_mask, mult, out, lse)
        ctx.cfg = (H, scale, packed)
        return out

    @staticmethod
    def backward(ctx, go):
        qk, k_sep, v, kpm, mult, out, lse = ctx.saved_tensors
        H, scale, packed = ctx.cfg
        E = v.shape[-1]
        q, k = (qk[..., :E], qk[..., E:]) if packed else (qk, k_sep)
        go = go.contiguous()
        gqk = torch.empty_like(qk)
        gk_sep = None if packed else torch.empty_like(k_sep)
        gq, gk = (gqk[..., :E], gqk[..., E:]) if packed else (gqk, gk_sep)
        gv = torch.empty_like(v)
        delta = torch.empty_like(lse)
        _lib().mha_flash_backward(q, k, v, go, out, kpm, mult, lse, delta, gq, gk, gv, H, scale)
        return gqk, gk_sep, gv, None, None, None, None


def _dense_rows(t):
    """(B, tokens, C) view acceptable to the attention kernels: unit channel stride, dense token rows, 16-byte aligned rows."""
    ok = (t.stride(2) == 1 and t.stride(0) == t.shape[1] * t.stride(1) and (t.stride(1) * t.element_size()) % 16 == 0
          and t.data_ptr() % 16 == 0)
    return t if ok else t.contiguous()


def mha_core(qk, k, v, heads, key_padding_mask, dropout_p, training, scale, mult=None):
    """Attention core of one MultiheadAttention call (multi_head_attention.py:329-375) as ONE kernel each way.  qk: the packed
    (B, L, 2E) q|k projection with k=None, or q with a separate k (B, S, E); v (B, S, E).  Returns the merged (B, L, E) output,
    or None when the kernels do not cover the call (not bf16 on a HIP device, head_dim != 32): the caller keeps the unfused
    path (the fp32 parity mode)."""
    E = v.shape[-1]
    if not v.is_cuda or v.dtype != torch.bfloat16 or E // heads != 32:
        return None
    B, L = qk.shape[0], qk.shape[1]
    S = v.shape[1]
    if mult is None and training and dropout_p > 0:      # no pool: ATen's graph-safe Philox stream decides which probabilities are dropped
        mult = F.dropout(torch.ones((B, heads, L, S), dtype=v.dtype, device=v.device), dropout_p, True)
    return _MhaFlashFn.apply(_dense_rows(qk), None if k is None else _dense_rows(k), _dense_rows(v), int(heads), key_padding_mask,
                             mult, float(scale))


class _SilogFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, weight, lam, log_err):
        lib = _lib()
        pred = pred.contiguous()
        B = pred.shape[0]
        # accepted layouts: (B,h,w,1) pixel-major or (B,1,h,w); both are the same bytes
        if pred.dim() == 4 and pred.shape[-1] == 1:
            h, w = pred.shape[1], pred.shape[2]
        elif pred.dim() == 4 and pred.shape[1] == 1:
            h, w = pred.shape[2], pred.shape[3]
        else:
            h, w = pred.shape[-2], pred.shape[-1]
        H, W = gt.shape[-2], gt.shape[-1]
        sums = torch.zeros(3, dtype=torch.float64, device=pred.device)
        lib.silog_sums(pred, gt, sums, B, h, w, H, W, log_err)
        loss = torch.empty((), dtype=torch.float32, device=pred.device)
        lib.silog_finalize(sums, lam, 10.0 * weight, loss)       # 10 w sqrt(E[d^2] - lam E[d]^2), one launch
        ctx.save_for_backward(pred, gt, sums)
        ctx.cfg = (B, h, w, H, W, weight, lam, log_err)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        pred, gt, sums = ctx.saved_tensors
        B, h, w, H, W, weight, lam, log_err = ctx.cfg
        gp = torch.empty_like(pred)
        _lib().silog_backward(pred, gt, sums, gloss.contiguous().float(), weight, lam, gp, B, h, w, H, W, log_err)
        return gp, None, None, None, None


def silog_loss(pred, gt_full, weight=1.0, variance_focus=0.85, log_depth_error=True):
    """weight * SiLog(pred, nearest_resize(gt), nearest_resize(0.2 <= gt < 10)); gt_full (B,1,H,W)/(B,H,W) fp32."""
    return _SilogFn.apply(pred, gt_full.contiguous(), float(weight), float(variance_focus), bool(log_depth_error))


class _SegCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, scale):
        logits = logits.contiguous()
        P = logits.numel() // 2
        s = torch.zeros(1, dtype=torch.float64, device=logits.device)
        _lib().seg_ce_sum(logits, target, s, P)
        ctx.save_for_backward(logits, target)
        ctx.cfg = (P, scale)
        return (s[0] * (scale / P)).float()

    @staticmethod
    def backward(ctx, gloss):
        logits, target = ctx.saved_tensors
        P, scale = ctx.cfg
        gl = torch.empty_like(logits)
        _lib().seg_ce_backward(logits, target, gloss.contiguous().float(), scale, gl, P)
        return gl, None, None


def seg_cross_entropy(logits_pixel_major, target, scale=1.0):
    """scale * mean CE of (B,H,W,2) logits against (B,H,W) int64 targets."""
    return _SegCEFn.apply(logits_pixel_major, target.contiguous(), float(scale))


class _ResampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size, mode):
        x = x.contiguous()
        B, Hs, Ws, C = x.shape
        y = torch.empty((B, size[0], size[1], C), dtype=x.dtype, device=x.device)
        _lib().resample_forward(x, y, B, Hs, Ws, size[0], size[1], C, mode)
        ctx.cfg = (B, Hs, Ws, size[0], size[1], C, mode)
        return y

    @staticmethod
    def backward(ctx, gy):
        B, Hs, Ws, Ho, Wo, C, mode = ctx.cfg
        gy = gy.contiguous()
        gx = torch.empty((B, Hs, Ws, C), dtype=gy.dtype, device=gy.device)
        lib = _lib()
        done = False
        if C % 4 == 0 and (Ho > Hs or Wo > Ws):
            tmp = torch.empty(lib.workspace_bytes(hip.WS_RESAMPLE_BWD, B, Ho, Ws, C) // 4, dtype=torch.float32, device=gy.device)   # x pass -> y pass scratch
            done = lib.resample_backward_sep(gy, tmp, gx, B, Hs, Ws, Ho, Wo, C, mode)
        if not done:
            lib.resample_backward(gy, gx, B, Hs, Ws, Ho, Wo, C, mode)
        return gx, None, None


class _PspPoolFn(torch.autograd.Function):
    """x -> (x itself, avg_pool 16, 8, 4, 2 of x): the pooling side of the PSP module (points_sample.py:107-113) as one pass forward
    and one pass backward.  The first output is x again, for the consumer that takes the map itself (the concat): with it this node
    is x's ONLY consumer, so the five gradients arrive together and leave as one tensor (was four pool backward kernels and four
    accumulation passes over the full map)."""

    @staticmethod
    def forward(ctx, x):
        B, H, W, C = x.shape
        outs = [torch.empty((B, H // k, W // k, C), dtype=x.dtype, device=x.device) for k in (16, 8, 4, 2)]
        if not _lib().psp_pool_forward(x, *outs):
            raise ValueError("psp_pools: unsupported shape %r" % (tuple(x.shape),))
        ctx.shape = tuple(x.shape)
        return (x.view_as(x),) + tuple(outs)

    @staticmethod
    def backward(ctx, g_pass, g16, g8, g4, g2):
        B, H, W, C = ctx.shape
        like = next(g for g in (g_pass, g16, g8, g4, g2) if g is not None)
        gx = torch.empty((B, H, W, C), dtype=like.dtype, device=like.device)
        con = lambda g: None if g is None else g.contiguous()
        _lib().psp_pool_backward(g_pass, con(g16), con(g8), con(g4), con(g2), gx)
        return gx


def psp_pools(x, pools):
    """(x, [avg_pool(x, k) for k in pools]); fused into one pass each way for the reference's pools (16, 8, 4, 2)."""
    B, H, W, C = x.shape
    vec = 8 if x.dtype == torch.bfloat16 else 4
    if tuple(pools) == (16, 8, 4, 2) and H >= 16 and W >= 16 and C % vec == 0 and x.is_contiguous():
        outs = _PspPoolFn.apply(x)
        return outs[0], list(outs[1:])
    return x, [avg_pool(x, k) for k in pools]


class _PyramidCatFn(torch.autograd.Function):
    """cat([x, up(y_1), ..., up(y_n)], channels) with up = bilinear(align_corners) to x's size (the PSP tail of
    points_sample.py:114-122) WITHOUT the concat pass: the up-sampling kernels write their channel slice of the result directly
    (pixel pitch = (n + 1) C), only x itself is copied; backward reads the slices in place (no .contiguous() copies) and hands
    x its slice of the gradient as a view."""

    @staticmethod
    def forward(ctx, x, *ys):
        lib = _lib()
        x = x.contiguous()
        B, H, W, C = x.shape
        n = len(ys)
        out = torch.empty((B, H, W, (n + 1) * C), dtype=x.dtype, device=x.device)
        out[..., :C].copy_(x)
        shapes = []
        for k, y in enumerate(ys):
            y = y.contiguous()
            if y.shape[0] != B or y.shape[3] != C:
                raise ValueError("pyramid_concat: branch %d has shape %r" % (k, tuple(y.shape)))
            lib.resample_forward(y, out[..., (k + 1) * C:(k + 2) * C], B, y.shape[1], y.shape[2], H, W, C, hip.RESAMPLE_BILINEAR_AC)
            shapes.append((y.shape[1], y.shape[2]))
        ctx.cfg = (B, H, W, C, shapes)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib()
        B, H, W, C, shapes = ctx.cfg
        g = g.contiguous()
        grads = [g[..., :C]]
        for k, (h, w) in enumerate(shapes):
            gk = g[..., (k + 1) * C:(k + 2) * C]
            gy = torch.empty((B, h, w, C), dtype=g.dtype, device=g.device)
            tmp = torch.empty(lib.workspace_bytes(hip.WS_RESAMPLE_BWD, B, H, w, C) // 4, dtype=torch.float32, device=g.device)
            if not lib.resample_backward_sep(gk, tmp, gy, B, h, w, H, W, C, hip.RESAMPLE_BILINEAR_AC):
                lib.resample_backward(gk.contiguous(), gy, B, h, w, H, W, C, hip.RESAMPLE_BILINEAR_AC)
            grads.append(gy)
        return tuple(grads)


def pyramid_concat(x, ys):
    """x (B,H,W,C) and low-resolution maps ys[k] (B,h_k,w_k,C) -> (B,H,W,(1+len(ys)) C) = [x | bilinear_ac(ys[k] -> H,W) ...]."""
    C = x.shape[-1]
    if C % 8 or not ys:
        return torch.cat([x] + [upsample_bilinear_ac(y, x.shape[1:3]) for y in ys], dim=-1)
    return _PyramidCatFn.apply(x, *ys)


def upsample_bilinear_ac(x, size):
    """(B,Hs,Ws,C) -> (B,H,W,C), bilinear with align_corners=True."""
    return _ResampleFn.apply(x, tuple(size), hip.RESAMPLE_BILINEAR_AC)


def upsample_nearest(x, size):
    """(B,Hs,Ws,C) -> (B,H,W,C), legacy nearest (floor(dst * in / out))."""
    return _ResampleFn.apply(x, tuple(size), hip.RESAMPLE_NEAREST)


class _AvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k):
        x = x.contiguous()
        B, H, W, C = x.shape
        y = torch.empty((B, H // k, W // k, C), dtype=x.dtype, device=x.device)
        _lib().avgpool_forward(x, y, B, H, W, C, k)
        ctx.cfg = (B, H, W, C, k)
        return y

    @staticmethod
    def backward(ctx, gy):
        B, H, W, C, k = ctx.cfg
        gx = torch.empty((B, H, W, C), dtype=gy.dtype, device=gy.device)
        _lib().avgpool_backward(gy.contiguous(), gx, B, H, W, C, k)
        return gx, None


def avg_pool(x, k):
    """k x k average pooling with stride k on a pixel-major map."""
    return _AvgPoolFn.apply(x, int(k))


class _WinAttnPackedFn(torch.autograd.Function):
    """qkv (W, 49, 3, heads, hd) packed as the qkv Linear writes it -> (W, 49, heads*hd).  `table` is the relative-position
    bias PARAMETER (n_rel, heads), gathered through rel (49*49 int32) inside the kernels; its gradient is accumulated by the
    backward kernel straight into the flat gradient buffer when the parameter is managed by engine.TrainStep (sink)."""

    @staticmethod
    def forward(ctx, qkv, table, rel, region, wpi, scale, sink):
        qkv = qkv.contiguous()
        W, N, _, H, D = qkv.shape
        out = torch.empty((W, N, H, D), dtype=qkv.dtype, device=qkv.device)
        tb = table.detach()
        _lib().winattn_forward(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], out, tb, region, wpi, scale, rel_index=rel)
        ctx.save_for_backward(qkv, tb, rel, region)
        ctx.cfg = (wpi, scale)
        ctx.sink = sink
        return out.view(W, N, H * D)

    @staticmethod
    def backward(ctx, go):
        qkv, tb, rel, region = ctx.saved_tensors
        wpi, scale = ctx.cfg
        W, N, _, H, D = qkv.shape
        go = go.contiguous().view(W, N, H, D)
        g = torch.empty_like(qkv)
        gtab = _winattn_table_grad(tb, ctx.sink, lambda dtab, hm: _lib().winattn_backward(
            qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], go, g[:, :, 0], g[:, :, 1], g[:, :, 2], tb, dtab, region, wpi, scale, rel_index=rel, head_major=hm), rel)
        return g, gtab, None, None, None, None, None


def _winattn_table_grad(tb, sink, run, rel):
    """Backward of the relative-position table of a window attention: run(dbias, head_major) launches the kernel.  On the device the
    kernel accumulates into a zeroed (heads, n_rel) scratch - a wave's flush is then one contiguous run of atomics instead of n_rel
    4-byte adds in n_rel different 64-byte segments, 41-59 us of every launch - and the transposed scratch is added to the table's
    gradient (the flat-buffer slice when there is a sink).  Returns what autograd gets for the table (None with a sink)."""
    direct = sink is not None
    if tb.is_cuda and rel is not None:
        tmp = WGRADS.scratch((tb.shape[1], tb.shape[0]), tb.device)
        run(tmp, True)
        if not direct:
            return tmp.t().contiguous()
        sink[0].view(tb.shape).add_(tmp.t())
    else:
        dtab = sink[0] if direct else torch.zeros_like(tb)
        run(dtab, False)
        if not direct:
            return dtab
    if sink[1] is not None:
        sink[1]()
    return None


class _WinAttnFn(torch.autograd.Function):
    """Separate q (W,49,H,D), k, v operands (any window/token/head strides, unit channel stride); bias table as above."""

    @staticmethod
    def forward(ctx, q, k, v, table, rel, region, wpi, scale, sink):
        W, N, H, D = q.shape
        out = torch.empty((W, N, H, D), dtype=q.dtype, device=q.device)
        tb = table.detach()
        fix = lambda t: t if t.stride(3) == 1 else t.contiguous()
        q, k, v = fix(q), fix(k), fix(v)
        _lib().winattn_forward(q, k, v, out, tb, region, wpi, scale, rel_index=rel)
        ctx.save_for_backward(q, k, v, tb, rel, region)
        ctx.cfg = (wpi, scale)
        ctx.sink = sink
        return out.view(W, N, H * D)

    @staticmethod
    def backward(ctx, go):
        q, k, v, tb, rel, region = ctx.saved_tensors
        wpi, scale = ctx.cfg
        W, N, H, D = q.shape
        go = go.contiguous().view(W, N, H, D)
        gq, gk, gv = (torch.empty((W, N, H, D), dtype=q.dtype, device=q.device) for _ in range(3))
        gtab = _winattn_table_grad(tb, ctx.sink, lambda dtab, hm: _lib().winattn_backward(
            q, k, v, go, gq, gk, gv, tb, dtab, region, wpi, scale, rel_index=rel, head_major=hm), rel)
        return gq, gk, gv, gtab, None, None, None, None, None


class GradLink:
    """Hands the packed qkv gradient of window_attention_qkv (k / v slots written) to ref_scores' backward, which writes the q slot and
    returns the ONE tensor: q feeds ref_scores -> ... -> q_new -> window_attention_qkv, so that backward always runs first.  Without the
    link each node returns a packed tensor with the other's slots zero-filled and autograd adds the two (2 fills + 1 add per block)."""
    __slots__ = ("g",)

    def __init__(self):
        self.g = None


class _WinAttnQFn(torch.autograd.Function):
    """Rewritten query q_new (W,49,H,D) against the k / v slots of the packed projection qkv (W,49,3,H,D): the gradient of qkv
    comes back as ONE packed tensor (q slot zero) instead of two zero-filled select-backward temporaries and their sum."""

    @staticmethod
    def forward(ctx, q_new, qkv, table, rel, region, wpi, scale, sink, link=None):
        ctx.link = link
        q_new, qkv = q_new.contiguous(), qkv.contiguous()
        W, N, H, D = q_new.shape
        out = torch.empty((W, N, H, D), dtype=q_new.dtype, device=q_new.device)
        tb = table.detach()
        _lib().winattn_forward(q_new, qkv[:, :, 1], qkv[:, :, 2], out, tb, region, wpi, scale, rel_index=rel)
        ctx.save_for_backward(q_new, qkv, tb, rel, region)
        ctx.cfg = (wpi, scale)
        ctx.sink = sink
        return out.view(W, N, H * D)

    @staticmethod
    def backward(ctx, go):
        q_new, qkv, tb, rel, region = ctx.saved_tensors
        wpi, scale = ctx.cfg
        W, N, H, D = q_new.shape
        go = go.contiguous().view(W, N, H, D)
        gq = torch.empty_like(q_new)
        g = torch.empty_like(qkv)
        if ctx.link is None:
            g[:, :, 0].zero_()
        gtab = _winattn_table_grad(tb, ctx.sink, lambda dtab, hm: _lib().winattn_backward(
            q_new, qkv[:, :, 1], qkv[:, :, 2], go, gq, g[:, :, 1], g[:, :, 2], tb, dtab, region, wpi, scale, rel_index=rel, head_major=hm), rel)
        if ctx.link is not None:                       # ref_scores' backward completes and returns it
            ctx.link.g, g = g, None
        return gq, g, gtab, None, None, None, None, None, None


def window_attention_qkv(q_new, qkv, table, rel, region, windows_per_image, scale, link=None):
    """softmax(scale*q_new k^T + bias (+shift mask)) v with k, v = qkv[:, :, 1], qkv[:, :, 2] (the 1/32 stage).  link: the GradLink also given
    to the ref_scores call q_new descends from."""
    return _WinAttnQFn.apply(q_new, qkv, table, rel, region, int(windows_per_image), float(scale), _sink(table), link)


def window_attention_packed(qkv, table, rel, region, windows_per_image, scale):
    """softmax(scale*q k^T + bias (+shift mask)) v over 49-token windows; qkv (W,49,3,H,D); bias(h,i,j) = table[rel[i*49+j], h]."""
    return _WinAttnPackedFn.apply(qkv, table, rel, region, int(windows_per_image), float(scale), _sink(table))


def window_attention(q, k, v, table, rel, region, windows_per_image, scale):
    """Separate q / k / v operands (W, 49, H, D): bf16 on the matrix cores (csrc/mfattn.hip, head_dim 4..32), fp32 on the
    lane-per-row kernels (csrc/winattn.hip)."""
    return _WinAttnFn.apply(q, k, v, table, rel, region, int(windows_per_image), float(scale), _sink(table))


class _RowAffineFn(torch.autograd.Function):
    """mu + exp(logsigma) * x with per-channel mu / logsigma (the reference tokens' re-parameterisation,
    multiscale_transformerr.py:289-292).  The parameter gradients are column sums over the rows: own colsum kernel straight into
    the flat gradient buffer - ATen's multi-block reduction zeroes its semaphore with hipMemsetAsync once the row count grows
    (batch 16), which a HIP graph cannot replay (engine.TrainStep refuses the capture)."""

    @staticmethod
    def forward(ctx, x, mu, logsigma, sinks):
        e = logsigma.detach().exp().to(x.dtype)
        ctx.save_for_backward(x, e)
        ctx.sinks = sinks
        ctx.shapes = (mu.shape, logsigma.shape)
        return mu.detach().to(x.dtype) + e * x

    @staticmethod
    def backward(ctx, g):
        x, e = ctx.saved_tensors
        g = g.contiguous()
        C = g.shape[-1]
        rows = g.numel() // C
        gl = (g * x * e).contiguous()
        outs = []
        for t, sink, shape in ((g, ctx.sinks[0], ctx.shapes[0]), (gl, ctx.sinks[1], ctx.shapes[1])):
            if sink is not None:
                COLSUMS.add(t, sink[0].view(-1), rows, C, sink[1])
                outs.append(None)
            else:
                o = torch.zeros(C, dtype=torch.float32, device=g.device)
                _lib().colsum(t, o, rows, C)
                outs.append(o.view(shape))
        return g * e, outs[0], outs[1], None


def row_affine(x, mu, logsigma):
    """mu + exp(logsigma) * x, x (..., C), mu / logsigma (1, 1, C) parameters."""
    return _RowAffineFn.apply(x, mu, logsigma, (_sink(mu), _sink(logsigma)))


class _RefScoresFn(torch.autograd.Function):
    """ra (B, nwin*49, R, H) = scale * q . ref_k per head (multiscale_transformerr.py:296-298); q is read in place from the packed
    qkv projection (W, 49, 3, H, hd) and its gradient comes back as ONE packed tensor (k and v slots zero)."""

    @staticmethod
    def forward(ctx, qkv, ref_k, B, scale, link=None):
        ctx.link = link
        qkv, ref_k = qkv.contiguous(), ref_k.contiguous()
        W, N, _, H, hd = qkv.shape
        R = ref_k.shape[1]
        ra = torch.empty((B, (W // B) * N, R, H), dtype=qkv.dtype, device=qkv.device)
        _lib().ref_scores_forward(qkv[:, :, 0], ref_k, ra, B, W // B, scale)
        ctx.save_for_backward(qkv, ref_k)
        ctx.cfg = (B, scale)
        return ra

    @staticmethod
    def backward(ctx, g):
        qkv, ref_k = ctx.saved_tensors
        B, scale = ctx.cfg
        gqkv = None
        if ctx.link is not None:
            gqkv, ctx.link.g = ctx.link.g, None         # k / v slots already hold window_attention_qkv's gradient
        if gqkv is None:
            gqkv = torch.zeros_like(qkv)
        dk = torch.empty(ref_k.shape, dtype=torch.float32, device=g.device)
        _lib().ref_scores_backward(qkv[:, :, 0], ref_k, g.contiguous(), gqkv[:, :, 0], dk, B, qkv.shape[0] // B, scale)
        return gqkv, dk.to(ref_k.dtype), None, None, None


class _RefMixFn(torch.autograd.Function):
    """q_new (B, T, C) = softmax_r(ra) . ref_v per head (multiscale_transformerr.py:304-309)."""

    @staticmethod
    def forward(ctx, ra, ref_v, H):
        ra, ref_v = ra.contiguous(), ref_v.contiguous()
        B, T = ra.shape[0], ra.shape[1]
        q_new = torch.empty((B, T, ref_v.shape[2]), dtype=ra.dtype, device=ra.device)
        att = torch.empty_like(ra) if (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) else None
        _lib().ref_mix_forward(ra, ref_v, q_new, att, H)
        ctx.save_for_backward(att, ref_v)
        ctx.H = H
        return q_new

    @staticmethod
    def backward(ctx, g):
        att, ref_v = ctx.saved_tensors
        d_ra = torch.empty_like(att)
        dv = torch.empty(ref_v.shape, dtype=torch.float32, device=g.device)
        _lib().ref_mix_backward(att, ref_v, g.contiguous(), d_ra, dv, ctx.H)
        return d_ra, dv.to(ref_v.dtype), None


def ref_scores(qkv, ref_k, images, scale, link=None):
    """qkv (images*nwin, 49, 3, H, hd) packed projection, ref_k (images, R, H*hd) -> (images, nwin*49, R, H)."""
    return _RefScoresFn.apply(qkv, ref_k, int(images), float(scale), link)


def ref_mix(ra, ref_v, heads):
    """ra (images, T, R, H), ref_v (images, R, H*hd) -> (images, T, H*hd)."""
    return _RefMixFn.apply(ra, ref_v, int(heads))


class _TokAttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, scale):
        fix = lambda t: t if t.stride(3) == 1 else t.contiguous()
        q, k, v = fix(q), fix(k), fix(v)
        W, N, H, R = q.shape
        out = torch.empty((W, N, H, R), dtype=q.dtype, device=q.device)
        _lib().tokattn_forward(q, k, v, out, scale)
        ctx.save_for_backward(q, k, v)
        ctx.scale = scale
        return out.view(W, N, H * R)

    @staticmethod
    def backward(ctx, go):
        q, k, v = ctx.saved_tensors
        W, N, H, R = q.shape
        go = go.contiguous().view(W, N, H, R)
        gq = torch.empty((W, N, H, R), dtype=q.dtype, device=q.device)
        gk = torch.empty(k.shape, dtype=k.dtype, device=k.device)
        gv = torch.empty(v.shape, dtype=v.dtype, device=v.device)
        _lib().tokattn_backward(q, k, v, go, gq, gk, gv, ctx.scale)
        return gq, gk, gv, None


def token_attention(q, k, v, scale):
    """Class-token attention: q (W,49,H,4), k/v (W,49,H,e) -> (W,49,4H); softmax over the e feature channels."""
    return _TokAttnFn.apply(q, k, v, float(scale))


class _TokAttnPairFn(torch.autograd.Function):
    """token_attention for both class tokens at once (bf16): the pair shares k / v, one launch each way, and the backward's gk / gv are
    already the sum autograd would have formed from two nodes."""

    @staticmethod
    def forward(ctx, q, q2, k, v, scale):
        fix = lambda t: t if t.stride(3) == 1 else t.contiguous()
        q, q2, k, v = fix(q), fix(q2), fix(k), fix(v)
        W, N, H, R = q.shape
        o = torch.empty((W, N, H, R), dtype=q.dtype, device=q.device)
        o2 = torch.empty_like(o)
        _lib().tokattn_pair_forward(q, q2, k, v, o, o2, scale)
        ctx.save_for_backward(q, q2, k, v)
        ctx.scale = scale
        return o.view(W, N, H * R), o2.view(W, N, H * R)

    @staticmethod
    def backward(ctx, go, go2):
        q, q2, k, v = ctx.saved_tensors
        W, N, H, R = q.shape
        go, go2 = go.contiguous().view(W, N, H, R), go2.contiguous().view(W, N, H, R)
        gq, gq2 = torch.empty_like(go), torch.empty_like(go)
        gk = torch.empty(k.shape, dtype=k.dtype, device=k.device)
        gv = torch.empty(v.shape, dtype=v.dtype, device=v.device)
        _lib().tokattn_pair_backward(q, q2, k, v, go, go2, gq, gq2, gk, gv, ctx.scale)
        return gq, gq2, gk, gv, None


def token_attention_pair(q, q2, k, v, scale):
    """token_attention(q, k, v), token_attention(q2, k, v); bf16 with e in {12, 16, 24}: one launch for the pair."""
    if q.dtype == torch.bfloat16 and k.shape[3] in (12, 16, 24):
        return _TokAttnPairFn.apply(q, q2, k, v, float(scale))
    return token_attention(q, k, v, scale), token_attention(q2, k, v, scale)


class _WindowGatherFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, shift):
        x = x.contiguous()
        B, H, W, C = x.shape
        Hp, Wp = (H + 6) // 7 * 7, (W + 6) // 7 * 7
        out = torch.empty((B * (Hp // 7) * (Wp // 7), 49, C), dtype=x.dtype, device=x.device)
        _lib().window_map(x, out, B, H, W, C, shift, True)
        ctx.cfg = (B, H, W, C, shift)
        return out

    @staticmethod
    def backward(ctx, g):
        B, H, W, C, shift = ctx.cfg
        gx = torch.empty((B, H, W, C), dtype=g.dtype, device=g.device)
        _lib().window_map(g.contiguous(), gx, B, H, W, C, shift, False)
        return gx, None


class _WindowScatterFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, win, B, H, W, shift, residual):
        win = win.contiguous()
        C = win.shape[-1]
        out = torch.empty((B, H, W, C), dtype=win.dtype, device=win.device)
        res = None if residual is None else residual.contiguous()
        _lib().window_map(win, out, B, H, W, C, shift, False, residual=res)
        ctx.cfg = (B, H, W, C, shift, tuple(win.shape), residual is not None and tuple(residual.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        B, H, W, C, shift, shape, rshape = ctx.cfg
        g = g.contiguous()
        gw = torch.empty(shape, dtype=g.dtype, device=g.device)
        _lib().window_map(g, gw, B, H, W, C, shift, True)
        return gw, None, None, None, None, (g.view(rshape) if rshape else None)


class _WindowGatherMultiFn(torch.autograd.Function):
    """window_gather of several maps of one geometry (features + class tokens of a Swin block) as ONE launch each way."""

    @staticmethod
    def forward(ctx, shift, *xs):
        xs = [x.contiguous() for x in xs]
        B, H, W = xs[0].shape[:3]
        Hp, Wp = (H + 6) // 7 * 7, (W + 6) // 7 * 7
        outs = [torch.empty((B * (Hp // 7) * (Wp // 7), 49, x.shape[-1]), dtype=x.dtype, device=x.device) for x in xs]
        _lib().window_map_multi(xs, outs, B, H, W, [x.shape[-1] for x in xs], shift, True)
        ctx.cfg = (B, H, W, shift, [x.shape[-1] for x in xs])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        B, H, W, shift, Cs = ctx.cfg
        gs = [g.contiguous() for g in gs]
        gxs = [torch.empty((B, H, W, c), dtype=g.dtype, device=g.device) for g, c in zip(gs, Cs)]
        _lib().window_map_multi(gs, gxs, B, H, W, Cs, shift, False)
        return (None,) + tuple(gxs)


class _WindowScatterMultiFn(torch.autograd.Function):
    """window_scatter (+ residual streams) of several maps of one geometry as ONE launch each way; args: n windows, then n residuals."""

    @staticmethod
    def forward(ctx, B, H, W, shift, n, *ts):
        wins = [t.contiguous() for t in ts[:n]]
        ress = [None if r is None else r.contiguous() for r in ts[n:]]
        Cs = [w.shape[-1] for w in wins]
        outs = [torch.empty((B, H, W, c), dtype=w.dtype, device=w.device) for w, c in zip(wins, Cs)]
        _lib().window_map_multi(wins, outs, B, H, W, Cs, shift, False, residuals=ress)
        ctx.cfg = (B, H, W, shift, Cs, [tuple(w.shape) for w in wins], [r is not None and tuple(r.shape) for r in ts[n:]])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        B, H, W, shift, Cs, shapes, rshapes = ctx.cfg
        gs = [g.contiguous() for g in gs]
        gws = [torch.empty(sh, dtype=g.dtype, device=g.device) for g, sh in zip(gs, shapes)]
        _lib().window_map_multi(gs, gws, B, H, W, Cs, shift, True)
        return (None,) * 5 + tuple(gws) + tuple((g.view(rs) if rs else None) for g, rs in zip(gs, rshapes))


def window_gather_multi(xs, shift):
    """[window_gather(x, shift) for x in xs] (maps of one (B, H, W), any channel counts, at most 4) in one launch."""
    return _WindowGatherMultiFn.apply(int(shift), *xs)


def window_scatter_multi(wins, B, H, W, shift, residuals):
    """[window_scatter(w, B, H, W, shift, r) for w, r in zip(wins, residuals)] in one launch."""
    return _WindowScatterMultiFn.apply(int(B), int(H), int(W), int(shift), len(wins), *wins, *residuals)


def window_gather(x, shift):
    """(B,H,W,C) -> (B*nWin, 49, C): zero-pad to multiples of 7, cyclic shift by -shift, 7x7 window partition."""
    return _WindowGatherFn.apply(x, int(shift))


def window_scatter(win, B, H, W, shift, residual=None):
    """Inverse of window_gather (window reverse, un-shift, crop) -> (B,H,W,C) [+ residual, any shape with B*H*W*C elements]."""
    return _WindowScatterFn.apply(win, int(B), int(H), int(W), int(shift), residual)


class _InormGeluFn(torch.autograd.Function):
    SLICES = 32

    @staticmethod
    def forward(ctx, a, u, eps):
        a, u = a.contiguous(), u.contiguous()
        B, C = u.shape[0], u.shape[-1]
        L = u.numel() // (B * C)
        S = _InormGeluFn.SLICES
        y = torch.empty_like(u)
        part = torch.empty(_lib().workspace_bytes(hip.WS_INORM_GELU, B, S, C) // 4, dtype=torch.float32, device=u.device)
        stat = torch.empty((B, C, 2), dtype=torch.float32, device=u.device)
        _lib().inorm_gelu_forward(a, u, y, part, stat, B, L, C, S, float(eps))
        ctx.save_for_backward(u, stat)
        ctx.cfg = (B, L, C, S)
        return y

    @staticmethod
    def backward(ctx, gy):
        u, stat = ctx.saved_tensors
        B, L, C, S = ctx.cfg
        gy = gy.contiguous()
        du = torch.empty_like(u)
        part = torch.empty(_lib().workspace_bytes(hip.WS_INORM_GELU, B, S, C) // 4, dtype=torch.float32, device=u.device)
        _lib().inorm_gelu_backward(gy, u, stat, part, du, B, L, C, S)
        return gy, du, None


class _AnchorDepthFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, att, anchor):
        B, P, R = att.shape
        att = att.contiguous()
        anchor = anchor.float().contiguous()
        pred = torch.empty((B, P), dtype=torch.float32, device=att.device)
        _lib().anchor_depth_forward(att, anchor, pred, B, P, R)
        ctx.save_for_backward(att, anchor)
        return pred

    @staticmethod
    def backward(ctx, g):
        att, anchor = ctx.saved_tensors
        B, P, R = att.shape
        datt = torch.empty_like(att) if ctx.needs_input_grad[0] else None
        danchor = torch.zeros((B, R), dtype=torch.float32, device=att.device)
        _lib().anchor_depth_backward(att, anchor, g.float().contiguous(), datt, danchor, B, P, R)
        return datt, danchor


def anchor_depth(att, anchor):
    """sum_r att[b,p,r] * anchor[b,r] (points_sample.py:277-279): att (B,P,R) softmax over the point channels, anchor (B,R) fp32
    -> (B,P) fp32; one streaming kernel each way instead of matrix-vector products in the batched-GEMM library."""
    return _AnchorDepthFn.apply(att, anchor)


class _PlaneLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, valid, tri, n_planes, min_area):
        lib = _lib()
        H, W = depth.shape[-2:]
        P = tri.shape[0]
        d = depth.reshape(H, W).contiguous()
        stats = torch.empty(4 * P + 1, dtype=torch.float64, device=d.device)
        loss = torch.empty(1, dtype=torch.float32, device=d.device)
        ws = torch.empty(lib.workspace_bytes(hip.WS_PLANE, P, H * W), dtype=torch.uint8, device=d.device)
        lib.plane_loss_forward(d, valid, tri, n_planes, P, H, W, int(min_area), ws, stats, loss)
        ctx.save_for_backward(d, valid, tri, n_planes, stats)
        ctx.shape = depth.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        d, valid, tri, n_planes, stats = ctx.saved_tensors
        H, W = d.shape
        gd = torch.empty_like(d)
        _lib().plane_loss_backward(d, valid, tri, n_planes, tri.shape[0], H, W, stats, g.reshape(1).float().contiguous(), gd)
        return gd.view(ctx.shape), None, None, None, None


def plane_loss(depth, valid, tri, n_planes, min_area):
    """PlaneLoss of one image (glassrgbd.py:385-450): depth (1,1,H,W), valid (H,W) uint8, tri (P,6) int64 rounded / clamped
    vertices, n_planes device int32 (how many of the P triangles count).  Returns the scalar loss; gradient to depth."""
    return _PlaneLossFn.apply(depth, valid, tri, n_planes, int(min_area))


def inorm_gelu_residual(a, u, eps=1e-5):
    """a + gelu(instance_norm(u)): statistics over every dim but the first (image) and last (channel)."""
    return _InormGeluFn.apply(a, u, eps)


class _BroadcastRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, row, B, L, dtype):
        ctx.src = (row.shape, row.dtype)
        return row.reshape(1, 1, -1).to(dtype).expand(B, L, -1).contiguous()

    @staticmethod
    def backward(ctx, g):
        shape, dtype = ctx.src
        g = g.contiguous()
        C = g.shape[-1]
        out = torch.zeros(C, dtype=torch.float32, device=g.device)
        _lib().colsum(g, out, g.numel() // C, C)
        return out.to(dtype).reshape(shape), None, None, None


def broadcast_rows(row, B, L, dtype):
    """(.., C) parameter row -> materialised (B, L, C); the gradient is one column-sum kernel."""
    return _BroadcastRowsFn.apply(row, int(B), int(L), dtype)


class _PointSampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fmap, coords, mode):
        fmap = fmap.contiguous()
        B, H, W, C = fmap.shape
        coords = coords.detach().reshape(B, -1, 2).float().contiguous()
        S = coords.shape[1]
        out = torch.empty((B, S, C), dtype=torch.float32, device=fmap.device)
        _lib().point_sample_forward(fmap, coords, out, B, H, W, C, S, mode)
        ctx.save_for_backward(coords)
        ctx.cfg = (B, H, W, C, S, mode, fmap.dtype)
        return out

    @staticmethod
    def backward(ctx, gout):
        (coords,) = ctx.saved_tensors
        B, H, W, C, S, mode, dtype = ctx.cfg
        gmap = torch.empty((B, H, W, C), dtype=dtype, device=gout.device)
        g = gout.float().contiguous()
        if not _lib().point_sample_backward_gather(g, coords, gmap, B, H, W, C, S, mode):      # gather: every element written once
            gmap.zero_()
            _lib().point_sample_backward(g, coords, gmap, B, H, W, C, S, mode)
        return gmap, None, None


class _PointSampleFramedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fmap, coords, frame):
        fmap = fmap.contiguous()
        B, H, W, C = fmap.shape
        coords = coords.detach().reshape(B, -1, 2).float().contiguous()
        S = coords.shape[1]
        out = torch.empty((B, S, C), dtype=torch.float32, device=fmap.device)
        _lib().point_sample_framed_forward(fmap, coords, out, B, H, W, C, S, frame)
        ctx.save_for_backward(coords)
        ctx.cfg = (B, H, W, C, S, frame, fmap.dtype)
        return out

    @staticmethod
    def backward(ctx, gout):
        (coords,) = ctx.saved_tensors
        B, H, W, C, S, frame, dtype = ctx.cfg
        gmap = torch.empty((B, H, W, C), dtype=dtype, device=gout.device)
        _lib().point_sample_framed_backward(gout.float().contiguous(), coords, gmap, B, H, W, C, S, frame)
        return gmap, None, None


def point_sample(fmap, coords, nearest=False, frame=None):
    """F.grid_sample(map, coords (B,S,1,2) or (B,S,2), align_corners=False, zeros padding) on a pixel-major map
    (B,H,W,C) -> fp32 (B,S,C); gradient to the map only.  frame = (Hf, Wf, shift), nearest only: the map as
    torch.roll(F.pad(map, to (Hf, Wf)), (-shift, -shift), (1, 2)) presents it (the shifted-window frame of the 1/32 stage), sampled in place."""
    if frame is not None:
        Hf, Wf, shift = (int(v) for v in frame)
        B, H, W, _ = fmap.shape
        if (Hf, Wf, shift) == (H, W, 0):
            frame = None
        elif not nearest:
            raise ValueError("point_sample: a frame needs nearest sampling")
        elif coords.numel() // (2 * B) <= 256:
            return _PointSampleFramedFn.apply(fmap, coords, (Hf, Wf, shift))
        else:                                               # more points than the gather backward stages: build the frame
            fmap = torch.nn.functional.pad(fmap, (0, 0, 0, Wf - W, 0, Hf - H))
            if shift:
                fmap = torch.roll(fmap, shifts=(-shift, -shift), dims=(1, 2))
    return _PointSampleFn.apply(fmap, coords, 1 if nearest else 0)
