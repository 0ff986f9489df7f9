"""A stand-in for libgwdepth_hip.so used ONLY by the CPU test-suite (tests/, never the product).

It implements the tensor-level methods of gw_depth_amd.hip.HipLibrary with plain torch-CPU math so
that the host logic of the product (module wiring, layouts, autograd plumbing, state-dict mapping,
flat optimizer buffers, DDP bucket plan) can be checked against the oracle without a GPU.  The
numerical behaviour of the real kernels is checked on the GPU box by the `-m gpu` tests.
"""
import math

import torch
import torch.nn.functional as F

from gw_depth_amd import hip


def _act(v, act):
    if act == hip.ACT_RELU:
        return F.relu(v)
    if act == hip.ACT_GELU:
        return F.gelu(v)
    if act == hip.ACT_ELU:
        return F.elu(v)
    if act == hip.ACT_SIGMOID:
        return torch.sigmoid(v)
    return v


@torch.no_grad()
def _certain_sample_reference(pred_small, pred_large, interval, sample_num, min_depth):
    """CertainSample.forward, points_sample.py:291-364: interval histogram decides HOW MANY points,
    every top-k is over the whole variance map; integer coordinates, no gradient."""
    B, _, H, W = pred_large.shape
    small = F.interpolate(pred_small, size=(H, W), mode="bilinear", align_corners=True)
    var = ((small - pred_large) ** 2).flatten(1)
    edges = torch.tensor([min_depth] + list(interval) + [1.0], device=pred_large.device, dtype=pred_large.dtype)
    flat = pred_large.flatten(1)
    n_i = ((flat[:, None, :] >= edges[:-1, None]) & (flat[:, None, :] < edges[1:, None])).sum(-1)       # (B, I)
    import numpy as np
    n_host = np.asarray(n_i.tolist(), dtype=np.int64)                                                    # one host sync
    k_host = np.floor(n_host.astype(np.float32) / np.float32(H * W) * np.float32(sample_num))
    k_i = np.minimum(k_host, n_host.astype(np.float32)).astype(np.int64).tolist()
    order = torch.argsort(var, dim=1, descending=True, stable=True)                                       # lowest index wins ties
    outs = []
    for b in range(B):
        groups = [order[b, :k].sort()[0] for k in k_i[b] if k > 0]
        counts = [int(g.numel()) for g in groups]
        already = sum(counts)
        if groups:
            cat = torch.cat(groups)
            remain = sample_num - already
        else:
            cat = order[b, :sample_num].sort()[0]
            remain = 0
        if remain > 0 and remain >= already:
            times = remain // already + 1
            cat = cat.repeat(times)
            remain = sample_num - already * times
        if remain > 0:
            cat = torch.cat([cat, cat[-remain:]])
        if remain < 0:
            mid = max(range(len(counts)), key=lambda i: (counts[i], -i))
            groups[mid] = groups[mid][:remain]
            cat = torch.cat(groups)
        outs.append(cat)
    idx = torch.stack(outs)
    col, row = (idx % W).float(), torch.div(idx, W, rounding_mode="floor").float()
    # tensor / tensor is an IEEE division on the device; tensor / python_scalar multiplies by 1/W (1 ulp off the CPU path)
    wt, ht = torch.full_like(col, float(W)), torch.full_like(row, float(H))
    return torch.stack([(col / wt) * 2 - 1, (row / ht) * 2 - 1], dim=-1)[:, :, None]



class FakeDevice:
    is_fake = True

    def version(self):
        return 2

    @staticmethod
    def _conv_core(x, w, dims, stride, pad, gather, virt):
        B, Hi, Wi, Cin, Ho, Wo, Cout, KH, KW = dims
        xn = x.reshape(B, Hi, Wi, Cin).permute(0, 3, 1, 2).float()
        wn = w.reshape(Cout, KH, KW, Cin).permute(0, 3, 1, 2).float()
        if gather == hip.GATHER_CONV:
            out = F.conv2d(xn, wn, stride=stride, padding=pad)
        elif gather == hip.GATHER_UPSAMPLED:
            out = F.conv2d(F.interpolate(xn, size=tuple(virt), mode="nearest"), wn, stride=1, padding=pad)
        else:
            opad = (Ho - ((Hi - 1) * stride - 2 * pad + KH), Wo - ((Wi - 1) * stride - 2 * pad + KW))
            out = F.conv_transpose2d(xn, wn.permute(1, 0, 2, 3), stride=stride, padding=pad, output_padding=opad)
        assert out.shape == (B, Cout, Ho, Wo), (out.shape, dims)
        return out.permute(0, 2, 3, 1)

    def conv_forward(self, x, w, y, dims, z=None, scale=None, shift=None, residual=None, stride=1, pad=0,
                     gather=hip.GATHER_CONV, virt=(0, 0), act=hip.ACT_NONE, act_scale=1.0, mult=None, gate=None, gate_act=hip.ACT_NONE):
        v = self._conv_core(x, w, dims, stride, pad, gather, virt)
        if scale is not None:
            v = v * scale
        if shift is not None:
            v = v + shift
        if residual is not None and mult is None:
            v = v + residual.reshape(v.shape).float()
        if z is not None:
            z.copy_(v.reshape(z.shape))
        out = _act(v, act) * act_scale
        if mult is not None:                          # y = act_scale * act(v) * mult + residual (dropout, then the skip)
            out = out * mult.reshape(v.shape).float()
            if residual is not None:
                out = out + residual.reshape(v.shape).float()
        if gate is not None:                          # gwd_conv_desc.gate: act'(.) of the activation whose output is `gate`, last
            r = gate.reshape(v.shape).float()
            if gate_act == hip.ACT_GELU:              # from the producer's PRE-activation value
                out = out * (0.5 * (1 + torch.erf(r * 0.7071067811865476)) + r * torch.exp(-0.5 * r * r) * 0.3989422804014327)
            else:
                out = out * ((r > 0).float() if gate_act == hip.ACT_RELU else torch.where(r > 0, torch.ones_like(r), r + 1))
        y.copy_(out.reshape(y.shape))

    def conv_wgrad(self, x, gy, dw, dims, stride=1, pad=0, gather=hip.GATHER_CONV, virt=(0, 0), scale=None, **_):
        B, Hi, Wi, Cin, Ho, Wo, Cout, KH, KW = dims
        w0 = torch.zeros(Cout, KH, KW, Cin, requires_grad=True)
        with torch.enable_grad():
            out = self._conv_core(x, w0, dims, stride, pad, gather, virt)
            (g,) = torch.autograd.grad(out, w0, gy.reshape(out.shape).float())
        g = g.reshape(dw.shape)
        dw.add_(g if scale is None else g * scale.view(-1, 1, 1, 1))

    def weight_prep(self, w, row_scale, w_fwd, w_dgrad, N, taps, C, dtype):
        v = w.reshape(N, taps, C).float()
        if row_scale is not None:
            v = v * row_scale.view(-1, 1, 1)
        if w_fwd is not None:
            w_fwd.copy_(v.reshape(w_fwd.shape))
        if w_dgrad is not None:
            w_dgrad.copy_(v.permute(2, 1, 0).reshape(w_dgrad.shape))

    def act_backward(self, gy, ref, gx, scale, rows, C, act, act_scale):
        g = gy.float()
        if act == hip.ACT_RELU:
            g = g * (ref.float() > 0)
        elif act == hip.ACT_GELU:
            r = ref.float()
            g = g * (0.5 * (1 + torch.erf(r * 0.7071067811865476)) + r * torch.exp(-0.5 * r * r) * 0.3989422804014327)
        elif act == hip.ACT_ELU:
            r = ref.float()
            g = g * torch.where(r > 0, torch.ones_like(r), r / act_scale + 1)
        elif act == hip.ACT_SIGMOID:
            sg = ref.float() / act_scale
            g = g * sg * (1 - sg)
        g = g * act_scale
        if scale is not None:
            g = g * scale
        gx.copy_(g.reshape(gx.shape))

    @staticmethod
    def _plane_loss_torch(depth, valid, tri, n_planes, P, H, W, min_area):
        """glassrgbd.py:385-450 in torch (differentiable in depth); returns (loss, stats)."""
        k = torch.tensor([[[1., 0., -1.], [2., 0., -2.], [1., 0., -1.]], [[1., 2., 1.], [0., 0., 0.], [-1., -2., -1.]]]).view(2, 1, 3, 3)
        g = F.conv2d(depth.reshape(1, 1, H, W).float(), k, padding=1)
        nx, ny = -g[0, 0].flatten(), -g[0, 1].flatten()
        ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        px, py, v = xs.flatten(), ys.flatten(), valid.flatten().bool()
        stats = torch.zeros(4 * P + 1, dtype=torch.float64)
        total, cnt = depth.new_zeros((), dtype=torch.float32), 0
        for j in range(min(int(n_planes[0]), P)):
            t = tri[j].reshape(3, 2)
            inside = torch.zeros_like(v)
            for a, b in ((0, 1), (1, 2), (2, 0)):
                ax, ay, bx, by = t[a, 0], t[a, 1], t[b, 0], t[b, 1]
                f0, f1 = ay >= py, by >= py
                inside ^= (f0 != f1) & (((by - py) * (ax - bx) >= (bx - px) * (ay - by)) == f1)
            m = inside & v
            n = int(m.sum())
            stats[4 * j] = n
            if n < min_area:
                continue
            stats[4 * j + 1], stats[4 * j + 2], stats[4 * j + 3] = float(nx[m].detach().mean()), float(ny[m].detach().mean()), 1.0
            total = total + torch.var(nx[m], unbiased=False) + torch.var(ny[m], unbiased=False)
            cnt += 1
        stats[4 * P] = cnt
        return total / max(1, cnt), stats

    def plane_loss_forward(self, depth, valid, tri, n_planes, P, H, W, min_area, workspace, stats, loss):
        with torch.no_grad():
            l, st = self._plane_loss_torch(depth, valid, tri, n_planes, P, H, W, min_area)
        loss.copy_(l.reshape(1))
        stats.copy_(st)
        self._plane_min_area = min_area

    def plane_loss_backward(self, depth, valid, tri, n_planes, P, H, W, stats, gloss, gdepth):
        d = depth.detach().float().clone().requires_grad_(True)
        with torch.enable_grad():
            l, _ = self._plane_loss_torch(d, valid, tri, n_planes, P, H, W, self._plane_min_area)
        if l.requires_grad:
            (g,) = torch.autograd.grad(l, d)
            gdepth.copy_((g * gloss[0]).reshape(gdepth.shape))
        else:
            gdepth.zero_()

    @staticmethod
    def _mha_ref(q, k, v, key_padding_mask, mult, H, scale):
        """multi_head_attention.py:329-375 in fp32 torch: merged output and log-sum-exp of the scaled, masked scores."""
        B, L, S, hd = q.shape[0], q.shape[1], k.shape[1], q.shape[-1] // H
        qh = q.float().reshape(B, L, H, hd).transpose(1, 2) * scale
        kh = k.float().reshape(B, S, H, hd).transpose(1, 2)
        vh = v.float().reshape(B, S, H, hd).transpose(1, 2)
        s_ = qh @ kh.transpose(-2, -1)
        if key_padding_mask is not None:
            s_ = s_.masked_fill(key_padding_mask.bool().view(B, 1, 1, S), float("-inf"))
        p = torch.softmax(s_, dim=-1)
        pd = p if mult is None else p * mult.float()
        return (pd @ vh).transpose(1, 2).reshape(B, L, H * hd), torch.logsumexp(s_, -1)

    def mha_flash_forward(self, q, k, v, key_padding_mask, mult, out, lse, H, scale):
        o, l = self._mha_ref(q, k, v, key_padding_mask, mult, H, scale)
        out.copy_(o)
        lse.copy_(l)

    def mha_flash_backward(self, q, k, v, go, out, key_padding_mask, mult, lse, delta, gq, gk, gv, H, scale):
        qf, kf, vf = (t.detach().float().clone().requires_grad_(True) for t in (q, k, v))
        with torch.enable_grad():
            o, _ = self._mha_ref(qf, kf, vf, key_padding_mask, mult, H, scale)
            g = torch.autograd.grad(o, [qf, kf, vf], go.float())
        gq.copy_(g[0])
        gk.copy_(g[1])
        gv.copy_(g[2])

    def anchor_depth_forward(self, att, anchor, pred, B, P, R):
        pred.copy_((att.float().reshape(B, P, R) * anchor.reshape(B, 1, R)).sum(-1))

    def anchor_depth_backward(self, att, anchor, gpred, datt, danchor, B, P, R):
        g = gpred.reshape(B, P, 1).float()
        if datt is not None:
            datt.copy_((g * anchor.reshape(B, 1, R)).to(datt.dtype).reshape(datt.shape))
        danchor.add_((att.float().reshape(B, P, R) * g).sum(1).reshape(danchor.shape))

    def collate(self, samples, H, W, mean, std, images, mask, depth, seg):
        """gwd_collate in torch (transforms_depth.py:618-660, glassrgbd_norhint.py:277-281, util/misc.py:273-313)."""
        m, s_ = torch.tensor(mean, dtype=torch.float32), torch.tensor(std, dtype=torch.float32)
        images.zero_(); mask.fill_(1); depth.zero_(); seg.zero_()
        for b, (rgb, dmm, lab) in enumerate(samples):
            h, w = rgb.shape[:2]
            images[b, :h, :w] = ((rgb.to(torch.float32).div(255) - m) / s_).to(images.dtype)
            mask[b, :h, :w] = 0
            if dmm is not None:
                depth[b, :h, :w] = dmm / 1000.0
            if lab is not None:
                seg[b, :h, :w] = (lab > 0).long()

    def eval_accumulate(self, pred, gt, seg, seg_strides, seg_gt, workspace, measures, running, confusion, B, HW, dmin, dmax):
        """gwd_eval_accumulate in torch: fp32 per-pixel terms, f64 sums (src/engine_glassrgbd.py:249-263, util/metrics.py:37-99,198-218)."""
        if pred is not None:
            f32 = torch.float32
            for b in range(B):
                p = pred[b].float().clone()
                p[p < dmin] = dmin
                p[p > dmax] = dmax
                p[torch.isnan(p)] = dmin
                v = (gt[b] > dmin) & (gt[b] < dmax)
                g, p = gt[b][v].to(f32), p[v]
                n = float(g.numel())
                thr = torch.maximum(g / p, p / g)
                err = (p.double().log().to(f32) - g.double().log().to(f32))
                l10 = (p.double().log10().to(f32) - g.double().log10().to(f32)).abs()
                sq = (g - p) ** 2
                me, me2 = err.double().sum() / n if n else float("nan"), (err * err).double().sum() / n if n else float("nan")
                m = [torch.sqrt(torch.as_tensor(me2 - me * me)) * 100, ((g - p).abs() / g).double().sum() / n if n else float("nan"),
                     l10.double().sum() / n if n else float("nan"), torch.sqrt(sq.double().sum() / n) if n else float("nan"),
                     (sq / g).double().sum() / n if n else float("nan"), torch.sqrt(torch.as_tensor(me2)),
                     (thr < 1.25).double().sum() / n if n else float("nan"), (thr < 1.5625).double().sum() / n if n else float("nan"),
                     (thr < 1.953125).double().sum() / n if n else float("nan")]
                measures[b] = torch.as_tensor([float(x) for x in m], dtype=torch.float64)
                running[:9] += measures[b]
                running[9] += 1
        if seg is not None:
            lg = seg.as_strided((B, HW, 2), tuple(seg_strides)).float()
            c = (lg[..., 1] > lg[..., 0]).long()
            for t in (0, 1):
                for k in (0, 1):
                    confusion[t * 2 + k] += int(((seg_gt == t) & (c == k)).sum())

    def conv_wgrad_batch(self, jobs):
        for x, gy, dw, dims, kw in jobs:
            self.conv_wgrad(x, gy, dw, dims, **kw)

    def colsum_batch(self, jobs):
        for g, out, rows, C in jobs:
            self.colsum(g, out, rows, C)

    @staticmethod
    def colsum_batchable(g, C):
        return C % 4 == 0

    def colsum(self, g, out, rows, C):
        out.add_(g.reshape(rows, C).float().sum(0))

    def layernorm_forward(self, x, gamma, beta, y, mean, rstd, rows, C, gelu, residual=None, ld=0):
        if ld and ld != C:                      # padded rows: the real channels are normalised, the padding is written as zeros
            yv = y.view(rows, ld)
            yv.zero_()
            yc = torch.empty(rows, C, dtype=y.dtype)
            self.layernorm_forward(x.reshape(rows, ld)[:, :C].contiguous(), gamma, beta, yc, mean, rstd, rows, C, gelu,
                                   None if residual is None else residual.reshape(rows, ld)[:, :C].contiguous())
            yv[:, :C] = yc
            return
        xf = x.reshape(rows, C).float()
        mu = xf.mean(1)
        rs = (xf.var(1, unbiased=False) + 1e-5).rsqrt()
        o = (xf - mu[:, None]) * rs[:, None]
        if gamma is not None:
            o = o * gamma + beta
        if gelu:
            o = F.gelu(o)
        if residual is not None:
            o = o + residual.reshape(rows, C).float()
        y.copy_(o.reshape(y.shape))
        mean.copy_(mu)
        rstd.copy_(rs)

    def resample_u8_pass(self, src, dst, bounds, kk, axis, row_stride, base0, step0, base1, step1):
        n_out, ksize = kk.shape
        C = dst.shape[2]
        other = dst.shape[0] if axis == 1 else dst.shape[1]
        flat = src.as_strided((src.untyped_storage().nbytes() - src.storage_offset(),), (1,)).to(torch.int64)
        o = torch.arange(n_out)[:, None, None, None]
        j = torch.arange(other)[None, :, None, None]
        c = torch.arange(C)[None, None, :, None]
        t = torch.arange(ksize)[None, None, None, :]
        a = base0 + step0 * (bounds[:, 0].long()[:, None, None, None] + t)
        jo = base1 + step1 * j
        off = (jo * row_stride + a * C + c) if axis == 1 else (a * row_stride + jo * C + c)
        valid = t < bounds[:, 1].long()[:, None, None, None]
        vals = flat[torch.where(valid, off, torch.zeros_like(off))] * valid
        acc = (vals * kk.long()[:, None, None, :]).sum(-1) + (1 << 21)
        res = (acc >> 22).clamp(0, 255).to(torch.uint8)                 # (n_out, other, C)
        dst.copy_(res.permute(1, 0, 2) if axis == 1 else res)

    def gather2d(self, src, dst, ytab, xtab, row_stride_bytes, elem_bytes):
        eb = src.element_size()
        rs = row_stride_bytes // eb
        per = elem_bytes // eb                                          # elements per gathered item (3 for RGB bytes)
        flat = src.as_strided((src.untyped_storage().nbytes() // eb - src.storage_offset(),), (1,))
        off = ytab.long()[:, None, None] * rs + xtab.long()[None, :, None] * per + torch.arange(per)[None, None, :]
        dst.view(-1).copy_(flat[off].reshape(-1))

    def color_adjust(self, rgb, out, mode, factor, scratch=None):
        from oracle import pil_color_ref as C                       # the CPU stand-in is test infrastructure: it may use the oracle
        a = rgb.cpu().numpy()
        if mode == "hue":
            hsv = C.rgb_to_hsv(a)
            hsv[..., 0] = ((hsv[..., 0].astype("int32") + int(factor)) & 255).astype("uint8")
            r = C.hsv_to_rgb(hsv)
        else:
            r = {"brightness": C.adjust_brightness, "contrast": C.adjust_contrast, "saturation": C.adjust_saturation}[mode](a, factor)
        out.copy_(torch.from_numpy(r))

    def match_cost(self, logits, lines, tgt_lines, tgt_labels, cost, w_line, w_class):
        prob = logits.softmax(-1)
        l1 = (lines[..., None, :] - tgt_lines).abs().sum(-1)
        cost.copy_(w_line * l1 + w_class * (-prob[..., tgt_labels]))

    def set_losses_forward(self, logits, lines, tgt_lines, tgt_labels, bidx, valid, qot, class_weight, num_items, world, target_class, ce, l1, wsum):
        L_, B, Q, K = logits.shape
        li, bi, qi = torch.arange(L_)[:, None], bidx.long()[None].expand(L_, -1), qot.long()
        tc = torch.full((L_, B, Q + 1), K - 1, dtype=torch.int64)
        tc[li, bi, qi] = tgt_labels[None].expand(L_, -1)
        tc = tc[:, :, :Q]
        nll = F.cross_entropy(logits.reshape(L_ * B, Q, K).transpose(1, 2), tc.reshape(L_ * B, Q), reduction="none")
        w = class_weight[tc.reshape(L_ * B, Q)]
        ce.copy_((nll * w).reshape(L_, -1).sum(1) / w.reshape(L_, -1).sum(1))
        wsum.copy_(w.reshape(L_, -1).sum(1))
        n = torch.clamp(num_items / world, min=1.0)
        diff = (lines[li, bi, qi.clamp(max=Q - 1)] - tgt_lines[None]).abs().sum(-1)
        l1.copy_((diff * valid.float()[None]).sum(1) / n)
        target_class.copy_(tc.to(torch.int32))

    def set_losses_backward(self, logits, lines, tgt_lines, bidx, valid, qot, class_weight, num_items, world, target_class, wsum, g_ce, g_l1,
                            dlogits, dlines):
        L_, B, Q, K = logits.shape
        lg, ln = logits.detach().clone().requires_grad_(True), lines.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            tc = target_class.long()
            nll = F.cross_entropy(lg.reshape(L_ * B, Q, K).transpose(1, 2), tc.reshape(L_ * B, Q), reduction="none")
            w = class_weight[tc.reshape(L_ * B, Q)]
            ce = (nll * w).reshape(L_, -1).sum(1) / w.reshape(L_, -1).sum(1)
            n = torch.clamp(num_items / world, min=1.0)
            li, bi, qi = torch.arange(L_)[:, None], bidx.long()[None].expand(L_, -1), qot.long()
            diff = (ln[li, bi, qi.clamp(max=Q - 1)] - tgt_lines[None]).abs().sum(-1)
            l1 = (diff * valid.float()[None]).sum(1) / n
            tot = (ce * (g_ce if g_ce is not None else torch.zeros(L_))).sum() + (l1 * (g_l1 if g_l1 is not None else torch.zeros(L_))).sum()
            gl, gn = torch.autograd.grad(tot, [lg, ln], allow_unused=True)
        dlogits.copy_(gl if gl is not None else torch.zeros_like(logits))
        dlines.add_(gn if gn is not None else torch.zeros_like(lines))

    def pos_counts(self, mask_full, mask_level, counts):
        h, w = mask_level.shape[1:]
        m = F.interpolate(mask_full.bool()[None].float(), size=(h, w)).to(torch.bool)[0]
        mask_level.copy_(m.to(mask_level.dtype))
        counts[..., 0] = (~m).cumsum(1).to(torch.int16)
        counts[..., 1] = (~m).cumsum(2).to(torch.int16)

    def pos_emit(self, counts, dim_t, out, normalize):
        y, x = counts[..., 0].float(), counts[..., 1].float()
        if normalize:
            y = y / (y[:, -1:, :] + 1e-6) * (2 * math.pi)
            x = x / (x[:, :, -1:] + 1e-6) * (2 * math.pi)
        px, py = x[..., None] / dim_t, y[..., None] / dim_t
        px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).flatten(3)
        py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).flatten(3)
        out.copy_(torch.cat((py, px), dim=3))

    def stem_pack(self, w, scale, packed):
        ws = w * scale.view(64, 1, 1, 1) if scale is not None else w
        i = torch.arange(14 * 2 * 64 * 8)
        j, lane, mt, s = i & 7, (i >> 3) & 63, (i >> 9) & 1, i >> 10
        co, kh, kw, c = 32 * mt + (lane & 31), s >> 1, 4 * (s & 1) + 2 * (lane >> 5) + (j >> 2), j & 3
        ok = (kw < 7) & (c < 3)
        v = ws[co, kh, kw.clamp(max=6), c.clamp(max=2)]
        packed.copy_(torch.where(ok, v, torch.zeros_like(v)).to(packed.dtype))
        self._stem_weights = getattr(self, "_stem_weights", {})
        self._stem_weights[packed.data_ptr()] = ws.to(torch.bfloat16).float()

    def stem_forward(self, x, packed, shift, y):
        w = self._stem_weights[packed.data_ptr()]                       # (64,7,7,3), bf16-rounded like the kernel's operands
        v = F.conv2d(x.float().permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), None, stride=2, padding=3)
        if shift is not None:
            v = v + shift.view(1, -1, 1, 1)
        v = F.max_pool2d(F.relu(v), 3, 2, 1)
        y.copy_(v.permute(0, 2, 3, 1))

    def unpad_add_batch(self, jobs):
        for src, dst, N, taps, G, Cg, Cgp in jobs:
            dst.view(N, taps, G, Cg).add_(src.view(-1, taps, G, Cgp)[:N, :, :, :Cg])

    def layernorm_backward(self, gy, x, gamma, beta, mean, rstd, gx, dgamma, dbeta, rows, C, gelu, ld=0, gskip=None, elu_input=False):
        if gskip is not None or elu_input:
            self.layernorm_backward(gy, x, gamma, beta, mean, rstd, gx, dgamma, dbeta, rows, C, gelu, ld)
            if gskip is not None:
                gx.add_(gskip.reshape(gx.shape).to(gx.dtype))
            if elu_input:
                r = x.reshape(gx.shape).float()
                gx.mul_(torch.where(r > 0, torch.ones_like(r), r + 1).to(gx.dtype))
            return True
        if ld and ld != C:
            gxv = gx.view(rows, ld)
            gxv.zero_()
            gc = torch.empty(rows, C, dtype=gx.dtype)
            self.layernorm_backward(gy.reshape(rows, ld)[:, :C].contiguous(), x.reshape(rows, ld)[:, :C].contiguous(), gamma, beta, mean, rstd,
                                    gc, dgamma, dbeta, rows, C, gelu)
            gxv[:, :C] = gc
            return
        xf = x.reshape(rows, C).float().requires_grad_(True)
        ga = gamma.clone().requires_grad_(True) if gamma is not None else None
        be = beta.clone().requires_grad_(True) if beta is not None else None
        with torch.enable_grad():
            o = F.layer_norm(xf, (C,), ga, be, 1e-5)
            if gelu:
                o = F.gelu(o)
            ins = [xf] + ([ga, be] if ga is not None else [])
            gs = torch.autograd.grad(o, ins, gy.reshape(rows, C).float())
        gx.copy_(gs[0].reshape(gx.shape))
        if ga is not None:
            dgamma.add_(gs[1])
            dbeta.add_(gs[2])

    def softmax_forward(self, x, y, rows, L):
        y.copy_(F.softmax(x.float(), dim=-1))

    def softmax_backward(self, gy, y, gx, rows, L):
        yf, g = y.float(), gy.float()
        gx.copy_(yf * (g - (yf * g).sum(-1, keepdim=True)))

    @staticmethod
    def _silog_terms(pred, gt, B, h, w, H, W, log_err):
        p = pred.reshape(B, 1, h, w).float()
        g = F.interpolate(gt.reshape(B, 1, H, W), size=(h, w), mode="nearest")
        m = (g >= 0.2) & (g < 10.0)
        d = (torch.log(p) - torch.log(g)) if log_err else ((p + torch.log(p)) - (g + torch.log(g)))
        return p, m, torch.where(m, d, torch.zeros_like(d))

    def silog_sums(self, pred, gt, sums, B, h, w, H, W, log_err):
        _, m, d = self._silog_terms(pred, gt, B, h, w, H, W, log_err)
        sums.add_(torch.stack([d.double().sum(), (d.double() ** 2).sum(), m.double().sum()]))

    def silog_finalize(self, sums, lam, scale, loss):
        n = sums[2]
        mean = sums[0] / n
        loss.copy_((torch.sqrt(sums[1] / n - lam * mean * mean) * scale).float().reshape(loss.shape))

    def silog_backward(self, pred, gt, sums, gloss, weight, lam, gpred, B, h, w, H, W, log_err):
        p, m, d = self._silog_terms(pred, gt, B, h, w, H, W, log_err)
        n = sums[2]
        mean = sums[0] / n
        var = sums[1] / n - lam * mean * mean
        c = (10.0 / torch.sqrt(var) / n).float() * weight * gloss
        dd = (1.0 / p) if log_err else (1.0 + 1.0 / p)
        gpred.copy_(torch.where(m, c * (d - (lam * mean).float()) * dd, torch.zeros_like(d)).reshape(gpred.shape))

    def seg_ce_sum(self, logits, target, out, P):
        out.add_(F.cross_entropy(logits.reshape(P, 2).float(), target.reshape(P), reduction="sum").double())

    def seg_ce_backward(self, logits, target, gloss, scale, glogits, P):
        sm = F.softmax(logits.reshape(P, 2).float(), dim=-1)
        sm = sm - F.one_hot(target.reshape(P), 2).float()
        glogits.copy_((sm * (gloss * scale / P)).reshape(glogits.shape))

    def sqnorm(self, g, sq, n):
        sq.add_((g[:n].double() ** 2).sum())

    def adamw_step(self, p, g, m, v, p16, sq, n, lr, b1, b2, eps, wd, bc1, bc2, max_norm, grad_scale):
        coef = grad_scale
        if sq is not None and max_norm > 0:
            total = float(sq[0]) ** 0.5 * grad_scale
            coef *= min(max_norm / (total + 1e-6), 1.0)
        gi = g[:n] * coef
        p[:n].mul_(1 - lr * wd)
        m[:n].lerp_(gi, 1 - b1)
        v[:n].mul_(b2).addcmul_(gi, gi, value=1 - b2)
        p[:n].addcdiv_(m[:n], (v[:n].sqrt() / (bc2 ** 0.5)).add_(eps), value=-lr / bc1)
        if p16 is not None:
            p16[:n].copy_(p[:n])

    def resample_forward(self, x, y, B, Hs, Ws, Ho, Wo, C, mode):
        xn = x.reshape(B, Hs, Ws, C).permute(0, 3, 1, 2).float()
        o = F.interpolate(xn, size=(Ho, Wo), mode="nearest") if mode == hip.RESAMPLE_NEAREST else \
            F.interpolate(xn, size=(Ho, Wo), mode="bilinear", align_corners=True)
        y.copy_(o.permute(0, 2, 3, 1).reshape(y.shape))

    def resample_backward(self, gy, gx, B, Hs, Ws, Ho, Wo, C, mode, gate=None, gate_act=hip.ACT_NONE):
        x0 = torch.zeros(B, C, Hs, Ws, requires_grad=True)
        with torch.enable_grad():
            o = F.interpolate(x0, size=(Ho, Wo), mode="nearest") if mode == hip.RESAMPLE_NEAREST else \
                F.interpolate(x0, size=(Ho, Wo), mode="bilinear", align_corners=True)
            (g,) = torch.autograd.grad(o, x0, gy.reshape(B, Ho, Wo, C).permute(0, 3, 1, 2).float())
        g = g.permute(0, 2, 3, 1).reshape(gx.shape)
        if gate is not None:
            r = gate.reshape(gx.shape).float()
            g = g * ((r > 0).float() if gate_act == hip.ACT_RELU else torch.where(r > 0, torch.ones_like(r), r + 1))
        gx.copy_(g)
        return True

    def stride_place(self, src, residual, dst, stride):
        out = torch.zeros(dst.shape) if residual is None else residual.float().clone()
        Ho, Wo = src.shape[1], src.shape[2]
        out[:, 0:Ho * stride:stride, 0:Wo * stride:stride] += src.float()
        dst.copy_(out)
        return True

    def psp_pool_forward(self, x, p16, p8, p4, p2):
        B, H, W, C = x.shape
        for k, out in ((16, p16), (8, p8), (4, p4), (2, p2)):
            self.avgpool_forward(x, out, B, H, W, C, k)
        return True

    def psp_pool_backward(self, g_pass, g16, g8, g4, g2, gx):
        B, H, W, C = gx.shape
        acc = torch.zeros(B, H, W, C) if g_pass is None else g_pass.float().clone()
        for k, g in ((16, g16), (8, g8), (4, g4), (2, g2)):
            if g is not None:
                tmp = torch.empty(B, H, W, C)
                self.avgpool_backward(g.float(), tmp, B, H, W, C, k)
                acc += tmp
        gx.copy_(acc)

    def avgpool_forward(self, x, y, B, H, W, C, k):
        y.copy_(F.avg_pool2d(x.reshape(B, H, W, C).permute(0, 3, 1, 2).float(), k, k).permute(0, 2, 3, 1).reshape(y.shape))

    def avgpool_backward(self, gy, gx, B, H, W, C, k):
        x0 = torch.zeros(B, C, H, W, requires_grad=True)
        with torch.enable_grad():
            (g,) = torch.autograd.grad(F.avg_pool2d(x0, k, k), x0, gy.reshape(B, H // k, W // k, C).permute(0, 3, 1, 2).float())
        gx.copy_(g.permute(0, 2, 3, 1).reshape(gx.shape))

    @staticmethod
    def _winattn_scores(q, k, bias, region, wpi, scale):
        # q,k: (W, N, H, D) -> scores (W, H, N, N)
        s = torch.einsum("wihd,wjhd->whij", q.float() * scale, k.float()) + bias[None]
        if region is not None:
            r = region.long()                                             # (wpi, N)
            m = torch.where(r[:, :, None] != r[:, None, :], -100.0, 0.0)  # (wpi, N, N)
            W = q.shape[0]
            s = s + m.repeat(W // wpi, 1, 1)[:, None]
        return s

    @staticmethod
    def _dense_bias(bias, rel_index):
        """(n_rel, heads) table + rel_index (49*49) -> (heads, 49, 49) (multiscale_transformerr.py:313-315)."""
        if rel_index is None:
            return bias
        return bias[rel_index.long()].view(49, 49, -1).permute(2, 0, 1)

    def winattn_forward(self, q, k, v, o, bias, region, wpi, scale, rel_index=None):
        p = F.softmax(self._winattn_scores(q, k, self._dense_bias(bias, rel_index), region, wpi, scale), dim=-1)
        o.copy_(torch.einsum("whij,wjhd->wihd", p, v.float()))

    def winattn_backward(self, q, k, v, go, gq, gk, gv, bias, dbias, region, wpi, scale, rel_index=None, head_major=False):
        qf, kf, vf = (t.detach().float().clone().requires_grad_(True) for t in (q, k, v))
        bf = bias.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            p = F.softmax(self._winattn_scores(qf, kf, self._dense_bias(bf, rel_index), region, wpi, scale), dim=-1)
            out = torch.einsum("whij,wjhd->wihd", p, vf)
            g = torch.autograd.grad(out, [qf, kf, vf, bf], go.float())
        gq.copy_(g[0])
        gk.copy_(g[1])
        gv.copy_(g[2])
        if dbias is not None:
            dbias.add_(g[3].t() if head_major else g[3])

    def ref_scores_forward(self, q, ref_k, ra, B, nwin, scale):
        """multiscale_transformerr.py:296-298: (q * scale) @ ref_k^T per head, written pixel-major (B, nwin*49, R, H)."""
        H, hd, R = q.shape[2], q.shape[3], ref_k.shape[1]
        qs = q.float().reshape(B, nwin * 49, H, hd) * scale
        ra.copy_(torch.einsum("bthd,brhd->btrh", qs, ref_k.float().reshape(B, R, H, hd)))

    def ref_scores_backward(self, q, ref_k, g, dq, d_ref_k, B, nwin, scale):
        H, hd, R = q.shape[2], q.shape[3], ref_k.shape[1]
        gf, kf = g.float(), ref_k.float().reshape(B, R, H, hd)
        dq.copy_((torch.einsum("btrh,brhd->bthd", gf, kf) * scale).reshape(dq.shape))
        d_ref_k.copy_((torch.einsum("btrh,bthd->brhd", gf, q.float().reshape(B, nwin * 49, H, hd)) * scale).reshape(d_ref_k.shape))

    def ref_mix_forward(self, ra, ref_v, q_new, att, H):
        B, T, R = ra.shape[0], ra.shape[1], ra.shape[2]
        a = F.softmax(ra.float(), dim=2)
        if att is not None:
            att.copy_(a)
        q_new.copy_(torch.einsum("btrh,brhd->bthd", a, ref_v.float().reshape(B, R, H, -1)).reshape(q_new.shape))

    def ref_mix_backward(self, att, ref_v, g, d_ra, d_ref_v, H):
        B, T, R = att.shape[0], att.shape[1], att.shape[2]
        a, gf = att.float(), g.float().reshape(B, T, H, -1)
        da = torch.einsum("bthd,brhd->btrh", gf, ref_v.float().reshape(B, R, H, -1))
        d_ra.copy_(a * (da - (a * da).sum(2, keepdim=True)))
        d_ref_v.copy_(torch.einsum("btrh,bthd->brhd", a, gf).reshape(d_ref_v.shape))

    def tokattn_forward(self, q, k, v, o, scale):
        a = F.softmax(torch.einsum("wnhr,wnhc->whrc", q.float(), k.float()) * scale, dim=-1)
        o.copy_(torch.einsum("whrc,wnhc->wnhr", a, v.float()))

    def tokattn_backward(self, q, k, v, go, gq, gk, gv, scale):
        qf, kf, vf = (t.detach().float().clone().requires_grad_(True) for t in (q, k, v))
        with torch.enable_grad():
            a = F.softmax(torch.einsum("wnhr,wnhc->whrc", qf, kf) * scale, dim=-1)
            g = torch.autograd.grad(torch.einsum("whrc,wnhc->wnhr", a, vf), [qf, kf, vf], go.float())
        gq.copy_(g[0])
        gk.copy_(g[1])
        gv.copy_(g[2])

    def upsample_taps_collapse(self, w, wk):
        wp = w.detach().float().permute(3, 1, 2, 0)                                             # (Cin, kh, kw, Cout)
        r = torch.stack([wp[:, 2], wp[:, 1] + wp[:, 2], wp[:, 0] + wp[:, 1], wp[:, 0]], dim=1)
        wk.copy_(torch.stack([r[:, :, 2], r[:, :, 1] + r[:, :, 2], r[:, :, 0] + r[:, :, 1], r[:, :, 0]], dim=2))

    def upsample_taps_fold(self, D, dw):
        r = torch.stack([D[:, 2] + D[:, 3], D[:, 1] + D[:, 2], D[:, 0] + D[:, 1]], dim=1)
        q = torch.stack([r[:, :, 2] + r[:, :, 3], r[:, :, 1] + r[:, :, 2], r[:, :, 0] + r[:, :, 1]], dim=2)
        dw.add_(q.permute(3, 1, 2, 0))

    def tokattn_pair_forward(self, q, q2, k, v, o, o2, scale):
        self.tokattn_forward(q, k, v, o, scale)
        self.tokattn_forward(q2, k, v, o2, scale)

    def tokattn_pair_backward(self, q, q2, k, v, go, go2, gq, gq2, gk, gv, scale):
        gk2, gv2 = torch.empty_like(gk, dtype=torch.float32), torch.empty_like(gv, dtype=torch.float32)
        gk1, gv1 = torch.empty_like(gk2), torch.empty_like(gv2)
        self.tokattn_backward(q, k, v, go, gq, gk1, gv1, scale)
        self.tokattn_backward(q2, k, v, go2, gq2, gk2, gv2, scale)
        gk.copy_(gk1 + gk2)
        gv.copy_(gv1 + gv2)

    def certain_sample(self, small, large, coords, edges, sample_num):
        e = [float(v) for v in edges.tolist()]
        coords.copy_(_certain_sample_reference(small, large, e[1:-1], sample_num, e[0]))

    def lsap(self, cost, col_offsets, out, max_targets):
        from scipy.optimize import linear_sum_assignment
        L_, B, Q, sumT = cost.shape
        off = col_offsets.tolist()
        for l in range(L_):
            for b in range(B):
                c = cost[l, b, :, off[b]:off[b + 1]].double().cpu().numpy()
                qi, ti = linear_sum_assignment(c)
                for q, t in zip(qi, ti):
                    out[l, off[b] + t] = int(q)
            out[l, off[B]:] = Q                      # padding columns -> the dummy query slot

    def window_map(self, src, dst, B, H, W, C, shift, gather, residual=None):
        Hp, Wp = (H + 6) // 7 * 7, (W + 6) // 7 * 7
        if gather:
            x = F.pad(src.reshape(B, H, W, C), (0, 0, 0, Wp - W, 0, Hp - H))
            if shift:
                x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
            x = x.view(B, Hp // 7, 7, Wp // 7, 7, C).permute(0, 1, 3, 2, 4, 5)
            dst.copy_(x.reshape(dst.shape))
        else:
            x = src.reshape(B, Hp // 7, Wp // 7, 7, 7, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
            if shift:
                x = torch.roll(x, shifts=(shift, shift), dims=(1, 2))
            x = x[:, :H, :W]
            if residual is not None:
                x = (x.float() + residual.reshape(x.shape).float()).to(dst.dtype)
            dst.copy_(x.reshape(dst.shape))

    def window_map_multi(self, srcs, dsts, B, H, W, Cs, shift, gather, residuals=None):
        for i, (s_, d_, c_) in enumerate(zip(srcs, dsts, Cs)):
            self.window_map(s_, d_, B, H, W, c_, shift, gather, residual=None if residuals is None else residuals[i])

    def inorm_gelu_forward(self, a, u, y, part, stat, B, L, C, S, eps):
        uf = u.reshape(B, L, C).float()
        mu = uf.mean(dim=1, keepdim=True)
        var = uf.var(dim=1, keepdim=True, unbiased=False)
        rstd = torch.rsqrt(var + eps)
        y.copy_((a.reshape(B, L, C).float() + F.gelu((uf - mu) * rstd)).reshape(y.shape))
        stat.copy_(torch.stack([mu.reshape(B, C), rstd.reshape(B, C)], dim=-1).reshape(stat.shape))

    def inorm_gelu_backward(self, gy, u, stat, part, du, B, L, C, S):
        st = stat.reshape(B, 1, C, 2)
        n = (u.reshape(B, L, C).float() - st[..., 0]) * st[..., 1]
        cdf = 0.5 * (1 + torch.erf(n * 0.7071067811865476))
        pdf = 0.3989422804014327 * torch.exp(-0.5 * n * n)
        gn = gy.reshape(B, L, C).float() * (cdf + n * pdf)
        du.copy_((st[..., 1] * (gn - gn.mean(dim=1, keepdim=True) - n * (gn * n).mean(dim=1, keepdim=True))).reshape(du.shape))

    def weight_prep_batch(self, table, n_jobs, total_blocks, block_job=None):
        raise NotImplementedError("the CPU stand-in never activates the batched weight cache")

    def point_sample_forward(self, fmap, coords, out, B, H, W, C, S, mode):
        r = F.grid_sample(fmap.reshape(B, H, W, C).permute(0, 3, 1, 2).float(), coords.reshape(B, S, 1, 2),
                          mode="nearest" if mode == 1 else "bilinear", align_corners=False)
        out.copy_(r.reshape(B, C, S).permute(0, 2, 1).reshape(out.shape))

    @staticmethod
    def _framed(fmap, frame):
        """The padded / rolled frame itself, as the reference builds it (multiscale_transformerr.py:662-676)."""
        Hf, Wf, shift = frame
        t = F.pad(fmap, (0, 0, 0, Wf - fmap.shape[2], 0, Hf - fmap.shape[1]))
        return torch.roll(t, shifts=(-shift, -shift), dims=(1, 2)) if shift else t

    def point_sample_framed_forward(self, fmap, coords, out, B, H, W, C, S, frame):
        self.point_sample_forward(self._framed(fmap.reshape(B, H, W, C), frame), coords, out, B, frame[0], frame[1], C, S, 1)

    def point_sample_framed_backward(self, gout, coords, gmap, B, H, W, C, S, frame):
        Hf, Wf, shift = frame
        gf = torch.zeros(B, Hf, Wf, C)
        self.point_sample_backward(gout, coords, gf, B, Hf, Wf, C, S, 1)
        if shift:
            gf = torch.roll(gf, shifts=(shift, shift), dims=(1, 2))
        gmap.copy_(gf[:, :H, :W].to(gmap.dtype))

    def point_sample_backward_gather(self, gout, coords, gmap, B, H, W, C, S, mode):
        gmap.zero_()
        self.point_sample_backward(gout, coords, gmap, B, H, W, C, S, mode)
        return True

    def point_sample_backward(self, gout, coords, gmap, B, H, W, C, S, mode):
        with torch.enable_grad():
            x = torch.zeros(B, C, H, W, requires_grad=True)
            r = F.grid_sample(x, coords.reshape(B, S, 1, 2), mode="nearest" if mode == 1 else "bilinear", align_corners=False)
            r.backward(gout.reshape(B, S, C).permute(0, 2, 1).reshape(B, C, S, 1))
        gmap.add_(x.grad.permute(0, 2, 3, 1).reshape(gmap.shape).to(gmap.dtype))

    def act_backward_colsum(self, gy, ref, gx, dbias, rows, C, act, act_scale, mult=None):
        if mult is not None:
            gy = (gy.float() * mult.float()).to(gy.dtype)
        self.act_backward(gy, ref, gx, None, rows, C, act, act_scale)
        dbias.add_(gx.reshape(rows, C).float().sum(0))
        return True

    def resample_backward_sep(self, gy, tmp, gx, B, Hs, Ws, Ho, Wo, C, mode):
        self.resample_backward(gy, gx, B, Hs, Ws, Ho, Wo, C, mode)
        return True

    def softmax_masked_forward(self, x, key_mask, y, rows, L, rows_per_mask, scale):
        s = x.float().reshape(rows, L) * scale
        if key_mask is not None:
            m = key_mask.reshape(-1, L).bool().repeat_interleave(rows_per_mask, dim=0)
            s = s.masked_fill(m, float("-inf"))
        y.copy_(F.softmax(s, dim=-1).reshape(y.shape))

    def softmax_scaled_backward(self, gy, y, gx, rows, L, scale):
        yf, g = y.float(), gy.float()
        gx.copy_(scale * yf * (g - (yf * g).sum(-1, keepdim=True)))

    def workspace_bytes(self, op, *dims):
        n = 4
        for d in dims:
            n *= int(d)
        return n * (2 if op == hip.WS_INORM_GELU else 1)
