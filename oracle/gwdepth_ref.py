"""CPU restatement of the GW-Depth train-step hot path (TEST INFRASTRUCTURE ONLY).

Plain functional PyTorch-CPU over a flat state dict with the reference's key names.  Every
function cites the reference lines it restates (paths relative to /root/reference).  Pinned by
tests/test_oracle_golden.py against tests/golden/*.npz, which oracle/make_golden.py produced by
running the reference's own code in the build container.  The product (gw_depth_amd) never
imports this file.
"""
import math

import torch
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment

WS = 7          # window size, multiscale_transformerr.py:1039,1057
HEADS = 16      # args.dense_trans_heads, src/args.py:136


class Cfg:
    """The defaults of src/args.py that define the published model."""
    num_queries = 100
    hidden_dim = 256
    nheads = 8
    enc_layers = 6
    dec_layers = 6
    dropout = 0.0
    num_ref = 20
    dense_trans_dim = 512
    dense_trans_layers = (4,)
    class_trans_layers = (2, 2, 1)
    class_token_dim = 64
    depth_interval = (0.1, 0.3, 0.5, 0.7, 0.9)
    interval_sample_num = (30, 80)
    min_depth_eval = 1e-3
    max_depth_eval = 10.0
    max_depth = 10
    depth_loss_weights = (0.25, 0.25, 0.25, 1.0)
    seg_loss_weight = 2.0
    variance_focus = 0.85
    log_depth_error = True
    set_cost_class = 1.0
    set_cost_line = 5.0
    line_loss_coef = 5.0
    eos_coef = 0.1
    aux_loss = True
    lr = 1e-4
    lr_backbone = 1e-5
    weight_decay = 1e-4
    clip_max_norm = 0.1

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)


class View:
    """Prefix view on the flat state dict."""

    def __init__(self, sd, prefix=""):
        self.sd, self.prefix = sd, prefix

    def __getitem__(self, name):
        return self.sd[self.prefix + name]

    def sub(self, name):
        return View(self.sd, self.prefix + name + ".")

    def has(self, name):
        return (self.prefix + name) in self.sd


def linear(x, p, name):
    return F.linear(x, p[name + ".weight"], p[name + ".bias"] if p.has(name + ".bias") else None)


def layer_norm(x, p, name):
    w = p[name + ".weight"]
    return F.layer_norm(x, (w.shape[0],), w, p[name + ".bias"], 1e-5)


# ------------------------------------------------------------------------------ backbone
def frozen_bn(x, p):
    """src/models/backbone.py:45-55 (eps inside rsqrt)."""
    scale = p["weight"] * (p["running_var"] + 1e-5).rsqrt()
    bias = p["bias"] - p["running_mean"] * scale
    return x * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)


def bottleneck(x, p, stride):
    """torchvision ResNet v1.5 Bottleneck (stride on the 3x3); see oracle/resnet50_v15.py."""
    out = F.relu(frozen_bn(F.conv2d(x, p["conv1.weight"]), p.sub("bn1")))
    out = F.relu(frozen_bn(F.conv2d(out, p["conv2.weight"], stride=stride, padding=1), p.sub("bn2")))
    out = frozen_bn(F.conv2d(out, p["conv3.weight"]), p.sub("bn3"))
    if p.has("downsample.0.weight"):
        x = frozen_bn(F.conv2d(x, p["downsample.0.weight"], stride=stride), p.sub("downsample.1"))
    return F.relu(out + x)


def resnet50_features(x, p):
    """conv1 7x7/2 + FrozenBN + ReLU + maxpool 3x3/2, layers 1-4 (3,4,6,3 blocks);
    src/models/backbone.py:65-69,90-92 -> 4 maps at strides 4, 8, 16, 32."""
    x = F.relu(frozen_bn(F.conv2d(x, p["conv1.weight"], stride=2, padding=3), p.sub("bn1")))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for li, (blocks, stride) in enumerate(((3, 1), (4, 2), (6, 2), (3, 2)), start=1):
        for bi in range(blocks):
            x = bottleneck(x, p.sub(f"layer{li}.{bi}"), stride if bi == 0 else 1)
        feats.append(x)
    return feats


def resize_mask(mask, size):
    """src/models/backbone.py:79 — nearest resize of the bool pad mask."""
    return F.interpolate(mask[None].float(), size=size).to(torch.bool)[0]


def pos_sine(mask, num_pos_feats, normalize):
    """src/models/position_encoding.py:28-48."""
    not_mask = ~mask
    y = not_mask.cumsum(1, dtype=torch.float32)
    x = not_mask.cumsum(2, dtype=torch.float32)
    if normalize:
        y = y / (y[:, -1:, :] + 1e-6) * (2 * math.pi)
        x = x / (x[:, :, -1:] + 1e-6) * (2 * math.pi)
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = 10000 ** (2 * (dim_t // 2) / num_pos_feats)
    px = x[:, :, :, None] / dim_t
    py = y[:, :, :, None] / dim_t
    px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).flatten(3)
    py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).flatten(3)
    return torch.cat((py, px), dim=3).permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------ DETR branch
def mha(query, key, value, p, nheads, key_padding_mask, drop, training):
    """src/models/multi_head_attention.py:188-380 (packed in-proj sliced per operand,
    q*scale, bmm, -inf key padding, softmax, dropout, bmm, out-proj). (L,B,E) layout."""
    L, B, E = query.shape
    S = key.shape[0]
    W, b = p["in_proj_weight"], p["in_proj_bias"]
    hd = E // nheads
    q = F.linear(query, W[:E], b[:E]) * (float(hd) ** -0.5)
    k = F.linear(key, W[E:2 * E], b[E:2 * E])
    v = F.linear(value, W[2 * E:], b[2 * E:])
    q = q.contiguous().view(L, B * nheads, hd).transpose(0, 1)
    k = k.contiguous().view(S, B * nheads, hd).transpose(0, 1)
    v = v.contiguous().view(S, B * nheads, hd).transpose(0, 1)
    att = torch.bmm(q, k.transpose(1, 2))
    if key_padding_mask is not None:
        att = att.view(B, nheads, L, S).masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
        att = att.view(B * nheads, L, S)
    att = F.dropout(F.softmax(att, dim=-1), drop, training)
    out = torch.bmm(att, v).transpose(0, 1).contiguous().view(L, B, E)
    return linear(out, p, "out_proj")


def detr_transformer(src, mask, query_embed, pos, p, cfg, training):
    """src/models/transformer.py:47-61, 149-162 (encoder post-norm), 212-233 (decoder post-norm),
    96-125 (shared final LayerNorm on every decoder layer's output)."""
    B = src.shape[0]
    d = cfg.dropout

    def drop(x):
        return F.dropout(x, d, training)

    src = src.flatten(2).permute(2, 0, 1)
    pos = pos.flatten(2).permute(2, 0, 1)
    qpos = query_embed.unsqueeze(1).repeat(1, B, 1)
    kpm = mask.flatten(1)
    x = src
    for i in range(cfg.enc_layers):
        lp = p.sub(f"encoder.layers.{i}")
        qk = x + pos
        x = layer_norm(x + drop(mha(qk, qk, x, lp.sub("self_attn"), cfg.nheads, kpm, d, training)), lp, "norm1")
        ff = linear(drop(F.relu(linear(x, lp, "linear1"))), lp, "linear2")
        x = layer_norm(x + drop(ff), lp, "norm2")
    memory = x
    tgt = torch.zeros_like(qpos)
    inter = []
    for i in range(cfg.dec_layers):
        lp = p.sub(f"decoder.layers.{i}")
        qk = tgt + qpos
        tgt = layer_norm(tgt + drop(mha(qk, qk, tgt, lp.sub("self_attn"), cfg.nheads, None, d, training)), lp, "norm1")
        t2 = mha(tgt + qpos, memory + pos, memory, lp.sub("multihead_attn"), cfg.nheads, kpm, d, training)
        tgt = layer_norm(tgt + drop(t2), lp, "norm2")
        ff = linear(drop(F.relu(linear(tgt, lp, "linear1"))), lp, "linear2")
        tgt = layer_norm(tgt + drop(ff), lp, "norm3")
        inter.append(layer_norm(tgt, p, "decoder.norm"))
    return torch.stack(inter).transpose(1, 2)        # (layers, B, Q, E)


# ------------------------------------------------------------------------------ window stages
def window_partition(x, ws=WS):
    """multiscale_transformerr.py:120-132."""
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws * ws, C)


def window_reverse(win, H, W, ws=WS):
    """multiscale_transformerr.py:135-149."""
    B = win.shape[0] // ((H // ws) * (W // ws))
    x = win.view(B, H // ws, W // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


def shift_mask(Hp, Wp, ws=WS):
    """multiscale_transformerr.py:937-955: 9-region SW-MSA mask with fill -100."""
    shift = ws // 2
    img = torch.zeros(1, Hp, Wp, 1)
    cnt = 0
    for h in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for w in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, h, w, :] = cnt
            cnt += 1
    mw = window_partition(img).view(-1, ws * ws)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return am.masked_fill(am != 0, -100.0).masked_fill(am == 0, 0.0)


def rel_bias(p):
    """multiscale_transformerr.py:313-315."""
    n = WS * WS
    idx = p["relative_position_index"].view(-1)
    return p["relative_position_bias_table"][idx].view(n, n, -1).permute(2, 0, 1).contiguous()


def softmax_attn(scores, v, p, mask):
    """shared tail of both attentions (multiscale_transformerr.py:313-329 / 541-558)."""
    B_, nH, N, _ = scores.shape
    att = scores + rel_bias(p).unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        att = att.view(B_ // nW, nW, nH, N, N) + mask.unsqueeze(1).unsqueeze(0)
        att = att.view(-1, nH, N, N)
    att = F.softmax(att, dim=-1)
    x = (att @ v).transpose(1, 2).reshape(B_, N, -1)
    return linear(x, p, "proj")


def ref_window_attention(xw, x_ref, p, mask):
    """WindowAttention.forward, multiscale_transformerr.py:267-332 (1/32 stage). q is scaled
    twice (:295 and :310); the ref "diffusion" is 3x x += gelu(LN_[nWin*49, nrf](conv3x3(x)))
    with one shared 16->16 conv over (B,16,nWin*49,nrf) (:299-302)."""
    B_, N, C = xw.shape
    hd = C // HEADS
    scale = hd ** -0.5
    qkv = linear(xw, p, "qkv").reshape(B_, N, 3, HEADS, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    rqk = linear(x_ref, p, "ref_qk").reshape(x_ref.shape[0], x_ref.shape[1], 2, C).permute(2, 0, 1, 3)
    ref_q, ref_v = rqk[0], rqk[1]
    rB, nrf, _ = ref_q.shape
    nwin = B_ // rB
    ref_q = p["diff_mu"] + p["diff_logsigma"].exp() * ref_q
    ref_k = ref_q.reshape(rB, nrf, HEADS, hd).permute(0, 2, 1, 3).repeat_interleave(nwin, dim=0)
    ref_v = ref_v.reshape(rB, nrf, HEADS, hd).permute(0, 2, 1, 3).repeat_interleave(nwin, dim=0)
    q = q * scale
    ra = q @ ref_k.transpose(-2, -1)                                  # (B_, nH, N, nrf)
    ra = ra.view(rB, nwin, HEADS, N, nrf).permute(0, 2, 1, 3, 4).reshape(rB, HEADS, nwin * N, nrf).contiguous()
    for _ in range(3):
        upd = F.conv2d(ra, p["ref_attn_diffusion.weight"], p["ref_attn_diffusion.bias"], padding=1)
        ra = ra + F.gelu(F.layer_norm(upd, [nwin * N, nrf]))
    ra = ra.reshape(rB, HEADS, nwin, N, nrf).permute(0, 2, 1, 3, 4).reshape(rB * nwin, HEADS, N, nrf)
    q_new = (F.softmax(ra, dim=-1) @ ref_v) * scale
    return softmax_attn(q_new @ k.transpose(-2, -1), v, p, mask)


def class_window_attention(xw, dtok, stok, p, mask):
    """WindowClassAttention.forward with group_attention=False, multiscale_transformerr.py:455-580.
    Token attention is "transposed" (:568-571) and BOTH tokens go through proj_dth (:572,578)."""
    B_, N, C = xw.shape
    hd = C // HEADS
    scale = hd ** -0.5
    qkv = linear(xw, p, "qkv").reshape(B_, N, 3, HEADS, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    x = softmax_attn((q * scale) @ k.transpose(-2, -1), v, p, mask)
    tdim = dtok.shape[-1]
    dq = linear(dtok, p, "cls_dth_q").reshape(B_, N, HEADS, tdim // HEADS).permute(0, 2, 1, 3)
    sq = linear(stok, p, "cls_seg_q").reshape(B_, N, HEADS, tdim // HEADS).permute(0, 2, 1, 3)
    tx = torch.cat([x, dtok, stok], dim=-1)
    tC = tx.shape[-1]
    tk = linear(tx, p, "global_k").reshape(B_, N, HEADS, tC // HEADS).permute(0, 2, 1, 3)
    tv = linear(tx, p, "global_v").reshape(B_, N, HEADS, tC // HEADS).permute(0, 2, 1, 3)

    def tok(qh):
        a = F.softmax((qh * scale).transpose(-2, -1) @ tk, dim=-1)            # (B_, nH, 4, tC/16)
        t = (a @ tv.transpose(-2, -1)).reshape(B_, -1, N).permute(0, 2, 1)   # (B_, N, 64)
        return linear(t, p, "proj_dth")

    return x, tok(dq), tok(sq)


def mlp2(x, p):
    """Mlp, multiscale_transformerr.py:55-73 (GELU exact, drop=0)."""
    return linear(F.gelu(linear(x, p, "fc1")), p, "fc2")


def pad_roll_partition(t, H, W, shift):
    """SwinTransformerBlock.forward :667-676, 705-707 for one (B,H,W,C) map."""
    pr, pb = (WS - W % WS) % WS, (WS - H % WS) % WS
    t = F.pad(t, (0, 0, 0, pr, 0, pb))
    if shift:
        t = torch.roll(t, shifts=(-shift, -shift), dims=(1, 2))
    return t, H + pb, W + pr


def unpartition_unroll_crop(win, H, W, Hp, Wp, shift):
    """SwinTransformerBlock.forward :730-747."""
    t = window_reverse(win, Hp, Wp)
    if shift:
        t = torch.roll(t, shifts=(shift, shift), dims=(1, 2))
    return t[:, :H, :W, :].contiguous()


def swin_block(x, H, W, shift, mask, p, ref_coors=None, ref_pos=None, dtok=None, stok=None):
    """SwinTransformerBlock.forward, multiscale_transformerr.py:646-788."""
    B, L, C = x.shape
    shortcut = x
    xn = layer_norm(x, p, "norm1").view(B, H, W, C)
    sx, Hp, Wp = pad_roll_partition(xn, H, W, shift)
    amask = mask if shift else None
    if dtok is None:
        # ---- 1/32 stage: line end-point tokens by nearest grid_sample (:678-701)
        if shift:
            rc = torch.zeros_like(ref_coors)
            rc[..., 0] = ref_coors[..., 0] - (shift / (Wp - 1)) * 2
            rc[..., 1] = ref_coors[..., 1] - (shift / (Hp - 1)) * 2
            rc = torch.where(rc < -1, -1 - (1 + rc), rc)
            rpos = torch.roll(ref_pos, shifts=(-shift, -shift), dims=(2, 3))
        else:
            rc, rpos = ref_coors, ref_pos
        x_ref = F.grid_sample(sx.permute(0, 3, 1, 2), rc, mode="nearest", align_corners=False)
        x_ref = x_ref + F.grid_sample(rpos, rc, mode="nearest", align_corners=False)
        x_ref = x_ref.reshape(B, C, -1).permute(0, 2, 1)
        aw = ref_window_attention(window_partition(sx), x_ref, p.sub("attn"), amask)
    else:
        dsc, ssc = dtok, stok
        tC = dtok.shape[2]
        dn, _, _ = pad_roll_partition(layer_norm(dtok, p, "norm_depth1").view(B, H, W, tC), H, W, shift)
        sn, _, _ = pad_roll_partition(layer_norm(stok, p, "norm_seg1").view(B, H, W, tC), H, W, shift)
        aw, dw, sw = class_window_attention(window_partition(sx), window_partition(dn), window_partition(sn),
                                            p.sub("attn"), amask)
    xo = unpartition_unroll_crop(aw, H, W, Hp, Wp, shift).view(B, H * W, C)
    x = shortcut + xo
    x = x + mlp2(layer_norm(x, p, "norm2"), p.sub("mlp"))
    if dtok is None:
        return x, None, None
    d = dsc.reshape(B, H, W, -1) + unpartition_unroll_crop(dw, H, W, Hp, Wp, shift)
    d = d + mlp2(layer_norm(d, p, "norm_depth2"), p.sub("mlp_depth"))
    s = ssc.reshape(B, H, W, -1) + unpartition_unroll_crop(sw, H, W, Hp, Wp, shift)
    s = s + mlp2(layer_norm(s, p, "norm_seg2"), p.sub("mlp_seg"))
    return x, d.view(B, H * W, tC), s.view(B, H * W, tC)


def basic_layer(x, H, W, depth, p, **kw):
    """BasicLayer.forward, multiscale_transformerr.py:926-979 (blocks alternate shift 0 / 3)."""
    Hp, Wp = math.ceil(H / WS) * WS, math.ceil(W / WS) * WS
    mask = shift_mask(Hp, Wp)
    d, s = kw.pop("dtok", None), kw.pop("stok", None)
    for i in range(depth):
        x, d, s = swin_block(x, H, W, 0 if i % 2 == 0 else WS // 2, mask, p.sub(f"blocks.{i}"),
                             dtok=d, stok=s, **kw)
    return x, d, s


# ------------------------------------------------------------------------------ point sampling
def certain_sample(pred_small, pred_large, interval, sample_num, min_depth):
    """CertainSample.forward, src/models/points/points_sample.py:291-364.  The interval masks only
    set HOW MANY points are taken; each top-k is over the whole variance map (:319)."""
    B, _, H, W = pred_large.shape
    small = F.interpolate(pred_small, size=(H, W), mode="bilinear", align_corners=True)
    var = (small - pred_large) ** 2
    edges = [min_depth] + list(interval) + [1.0]
    outs = []
    total = H * W
    for b in range(B):
        groups, counts, already = [], [], 0
        for i in range(len(edges) - 1):
            m = (pred_large[b] >= edges[i]) & (pred_large[b] < edges[i + 1])
            n_i = torch.sum(m)
            k = int(torch.min(torch.floor((n_i / total) * sample_num), n_i))
            if k > 0:
                idx = torch.topk(var[b].flatten(0), k)[1].sort()[0]
                groups.append(torch.stack([idx % W, torch.div(idx, W, rounding_mode="floor")]))
                counts.append(k)
                already += k
        if groups:
            cat = torch.cat(groups, dim=1)
            remain = sample_num - already
        else:
            idx = torch.topk(var[b].flatten(1), sample_num, dim=-1)[1].sort()[0]
            cat = torch.cat([idx % W, torch.div(idx, W, rounding_mode="floor")], dim=0)
            remain = 0
        if remain > 0 and remain >= already:
            times = remain // already + 1
            cat = cat.repeat(1, times)
            remain = sample_num - already * times
        if remain > 0:
            cat = torch.cat([cat, cat[:, -remain:]], dim=1)
        if remain < 0:
            mid = int(torch.argmax(torch.tensor(counts)))
            groups[mid] = groups[mid][:, :remain]
            cat = torch.cat(groups, dim=1)
        outs.append(cat)
    c = torch.stack(outs).float().permute(0, 2, 1)
    c = torch.stack([(c[:, :, 0] / W) * 2 - 1, (c[:, :, 1] / H) * 2 - 1], dim=-1)
    return c[:, :, None]


def conv_ln(x, p, gelu=False, padding=1):
    """ConvLn, points_sample.py:12-25: conv (no bias) -> LayerNorm over channels."""
    x = F.conv2d(x, p["conv.weight"], padding=padding)
    x = layer_norm(x.permute(0, 2, 3, 1), p, "layer_norm").permute(0, 3, 1, 2)
    return F.gelu(x) if gelu else x


def basic_block(x, p):
    """BasicBlock, points_sample.py:27-43."""
    out = conv_ln(x, p.sub("conv1.0"), gelu=True)
    out = conv_ln(out, p.sub("conv2"))
    if p.has("downsample.conv.weight"):
        x = conv_ln(x, p.sub("downsample"), padding=0)
    return out + x


def pyramid(x, p, pools=(16, 8, 4, 2)):
    """PyramidLayer.forward, points_sample.py:106-125 (layer4 is never called)."""
    x = conv_ln(x, p.sub("firstconv.0"), gelu=True)
    x = conv_ln(x, p.sub("firstconv.2"), gelu=True)
    for name, n in (("layer1", 1), ("layer2", 2), ("layer3", 2)):
        for i in range(n):
            x = basic_block(x, p.sub(f"{name}.{i}"))
    H, W = x.shape[-2:]
    if H < pools[0] or W < pools[0]:                                 # pad_before_pool :94-104
        x = F.pad(x, (0, max(pools[0] - W, 0), 0, max(pools[0] - H, 0)))
    size = x.shape[-2:]
    outs = [x]
    for bi, k in enumerate(pools, start=1):
        y = conv_ln(F.avg_pool2d(x, k, k), p.sub(f"branch{bi}.1"), gelu=True)
        outs.append(F.interpolate(y, size=size, mode="bilinear", align_corners=True))
    x = conv_ln(torch.cat(outs, dim=1), p.sub("lastconv.0"), gelu=True)
    return F.conv2d(x, p["lastconv.2.weight"])


def point_based_pred(x, dtok, pre_depth, coords, H, W, pos, p):
    """PointBasedPred.forward, points_sample.py:257-280."""
    dim = p["pre_proj.weight"].shape[0]
    xg_xr = linear(linear(torch.cat([x, dtok], dim=-1), p, "pre_proj"), p, "refer_proj")
    xg, xr = xg_xr[:, :, :dim], xg_xr[:, :, dim:]
    B = xr.shape[0]
    xr = xr.permute(0, 2, 1).reshape(B, -1, H, W)
    refer = F.grid_sample(xr, coords, align_corners=False) + F.grid_sample(pos, coords, align_corners=False)
    anchor = F.grid_sample(pre_depth, coords, align_corners=False).permute(0, 2, 1, 3)
    rg = (xg @ refer.flatten(2)) * (dim ** -2)
    rg = rg.permute(0, 2, 1).reshape(B, -1, H, W)
    att = F.softmax(pyramid(rg, p.sub("pyramid")), dim=1)
    return torch.sum(att * anchor, dim=1, keepdim=True)


# ------------------------------------------------------------------------------ dense encoder
def conv_a(x, p):
    """ConvA, multiscale_transformerr.py:104-118."""
    return F.gelu(F.conv2d(x, p["conv.weight"], p["conv.bias"], padding=1))


def mlp_norm(x, p):
    """MlpNorm with act_layer=None, norm_layer=LayerNorm; multiscale_transformerr.py:75-102."""
    return layer_norm(linear(linear(x, p, "fc1"), p, "fc2"), p, "norm")


def sig_head(x, p, name):
    """nn.Sequential(Linear, Linear, Sigmoid); multiscale_transformerr.py:1044,1064."""
    return torch.sigmoid(linear(linear(x, p, name + ".0"), p, name + ".1"))


def tokens_up(tok, Hs, Ws, size):
    B = tok.shape[0]
    t = tok.reshape(B, Hs, Ws, -1).permute(0, 3, 1, 2)
    return F.interpolate(t, size=size, mode="nearest").flatten(2).permute(0, 2, 1)


def refer_transformer(top, feats, masks, pred_lines, pred_logits, sizes, p, cfg, taps=None):
    """ReferTransformer.forward, multiscale_transformerr.py:1151-1319."""
    B, C, H, W = top.shape
    ids = torch.topk(pred_logits[:, :, 0], cfg.num_ref, dim=-1)[1]            # :1166 raw logit
    pts = torch.stack([pred_lines[i][ids[i]] for i in range(B)]).reshape(B, cfg.num_ref, -1, 2)
    pts = (pts * 2 - 1.0)[:, :, :2]                                            # :1175-1179
    pos = pos_sine(masks[3], cfg.dense_trans_dim // 2, False)                  # :1035,1181
    x, _, _ = basic_layer(top.flatten(2).permute(0, 2, 1), H, W, cfg.dense_trans_layers[0],
                          p.sub("dense_transformer"), ref_coors=pts, ref_pos=pos)
    depth0 = sig_head(x, p, "depth_pred32").permute(0, 2, 1).reshape(B, -1, H, W)
    dense_out = x.permute(0, 2, 1).reshape(-1, C, H, W)

    # ---- 1/16 (:1191-1218)
    H1, W1 = sizes[0]
    up = F.interpolate(dense_out, size=(H1, W1), mode="nearest")
    x1 = linear(up.flatten(2).permute(0, 2, 1), p, "proj_class1") + conv_a(feats[2], p.sub("proj_backbn1")).flatten(2).permute(0, 2, 1)
    dtok = p["depth_token"].expand(B, H1 * W1, -1)
    stok = p["seg_token"].expand(B, H1 * W1, -1)
    x1, dtok, stok = basic_layer(x1, H1, W1, cfg.class_trans_layers[0], p.sub("class_transformer1"), dtok=dtok, stok=stok)
    depth1 = sig_head(torch.cat([x1, dtok], dim=-1), p, "depth_pred16").permute(0, 2, 1).reshape(B, -1, H1, W1)
    pts1 = certain_sample(depth0, depth1, cfg.depth_interval, cfg.interval_sample_num[0],
                          cfg.min_depth_eval / cfg.max_depth_eval)

    # ---- 1/8 (:1226-1260)
    H2, W2 = sizes[1]
    C1 = x1.shape[-1]
    up = F.interpolate(x1.permute(0, 2, 1).reshape(-1, C1, H1, W1), size=(H2, W2), mode="nearest")
    x2 = linear(up.flatten(2).permute(0, 2, 1), p, "proj_class2") + conv_a(feats[1], p.sub("proj_backbn2")).flatten(2).permute(0, 2, 1)
    pos2 = pos_sine(masks[1], cfg.dense_trans_dim // 8, False)
    dtok = mlp_norm(tokens_up(dtok, H1, W1, (H2, W2)), p.sub("old_depth_token_proj8"))
    stok = mlp_norm(tokens_up(stok, H1, W1, (H2, W2)), p.sub("old_seg_token_proj8"))
    x2, dtok, stok = basic_layer(x2, H2, W2, cfg.class_trans_layers[1], p.sub("class_transformer2"), dtok=dtok, stok=stok)
    depth2 = point_based_pred(x2, dtok, depth1, pts1, H2, W2, pos2, p.sub("point_based_pred1"))
    pts2 = certain_sample(depth1, depth2, cfg.depth_interval, cfg.interval_sample_num[1],
                          cfg.min_depth_eval / cfg.max_depth_eval)

    # ---- 1/4 (:1263-1292)
    H3, W3 = sizes[2]
    C2 = x2.shape[-1]
    up = F.interpolate(x2.permute(0, 2, 1).reshape(-1, C2, H2, W2), size=(H3, W3), mode="nearest")
    x3 = linear(up.flatten(2).permute(0, 2, 1), p, "proj_class3") + conv_a(feats[0], p.sub("proj_backbn3")).flatten(2).permute(0, 2, 1)
    pos3 = pos_sine(masks[0], cfg.dense_trans_dim // 16, False)
    dtok = mlp_norm(tokens_up(dtok, H2, W2, (H3, W3)), p.sub("old_depth_token_proj4"))
    stok = mlp_norm(tokens_up(stok, H2, W2, (H3, W3)), p.sub("old_seg_token_proj4"))
    x3, dtok, stok = basic_layer(x3, H3, W3, cfg.class_trans_layers[2], p.sub("class_transformer3"), dtok=dtok, stok=stok)
    depth3 = point_based_pred(x3, dtok, depth2, pts2, H3, W3, pos3, p.sub("point_based_pred2"))

    C3 = x3.shape[-1]
    if taps is not None:
        taps.update(topk_ids=ids, points1=pts1, points2=pts2, depth0=depth0)
    to_map = lambda t: t.permute(0, 2, 1).reshape(-1, C3, H3, W3)
    return to_map(x3), to_map(dtok), to_map(stok), [depth1, depth2, depth3]


# ------------------------------------------------------------------------------ full-res decoder
def upconv(x, w, size=None):
    """upconv.forward, src/models/dense_upsample.py:82-90."""
    x = F.interpolate(x, scale_factor=2, mode="nearest") if size is None else F.interpolate(x, size=size, mode="nearest")
    return F.elu(F.conv2d(x, w, padding=1))


def dense_prediction(feat, depth3, dtok, stok, size, p, cfg):
    """DensePrediction.forward, src/models/dense_upsample.py:160-182."""
    def branch(fuse_in, tag, fuse):
        B, _, H, W = fuse_in.shape
        f = mlp2(fuse_in.flatten(2).permute(0, 2, 1), p.sub(fuse)).permute(0, 2, 1).reshape(B, -1, H, W)
        u1 = layer_norm(upconv(f, p[f"upconv1_{tag}.conv.weight"]).permute(0, 2, 3, 1), p, f"norm_{tag}")
        c1 = F.elu(F.conv2d(u1.permute(0, 3, 1, 2), p[f"conv1_{tag}.0.weight"], padding=1))
        u2 = upconv(c1, p[f"upconv2_{tag}.conv.weight"], size)
        return F.elu(F.conv2d(u2, p[f"conv2_{tag}.0.weight"], padding=1))

    d = branch(torch.cat([feat, depth3, dtok], dim=1), "depth", "depth_token_fuse")
    depth = cfg.max_depth * torch.sigmoid(F.conv2d(d, p["get_depth.0.weight"], padding=1))
    s = branch(torch.cat([feat, stok], dim=1), "seg", "seg_token_fuse")
    return depth, F.conv2d(s, p["get_seg.weight"], padding=1)


# ------------------------------------------------------------------------------ model forward
def forward(sd, images, pad_mask, cfg, training=False, taps=None):
    """GlassRGBD.forward, src/models/glassrgbd.py:74-123 (flags --with_line --with_center --with_dense)."""
    p = View(sd)
    feats = resnet50_features(images, p.sub("backbone.0.body"))
    masks = [resize_mask(pad_mask, f.shape[-2:]) for f in feats]
    pos3 = pos_sine(masks[3], cfg.hidden_dim // 2, True)
    src = feats[3]
    hs = detr_transformer(F.conv2d(src, p["input_proj.weight"], p["input_proj.bias"]), masks[3],
                          p["query_embed.weight"], pos3, p.sub("transformer"), cfg, training)
    logits = linear(hs, p, "class_embed")
    lp = p.sub("lines_embed")
    lines = torch.sigmoid(linear(F.relu(linear(F.relu(linear(hs, lp, "layers.0")), lp, "layers.1")), lp, "layers.2"))
    out = {"pred_logits": logits[-1], "pred_lines": lines[-1]}
    if cfg.aux_loss:
        out["aux_outputs"] = [{"pred_logits": a, "pred_lines": b} for a, b in zip(logits[:-1], lines[:-1])]
    dense_in = F.conv2d(src, p["dense_input_proj.weight"], p["dense_input_proj.bias"])
    sizes = [tuple(f.shape[-2:]) for f in feats[:-1]][::-1]
    feat4, dtok, stok, depths = refer_transformer(dense_in, feats, masks, out["pred_lines"], out["pred_logits"],
                                                  sizes, p.sub("dense_encoder"), cfg, taps)
    depth, seg = dense_prediction(feat4, depths[-1], dtok, stok, tuple(images.shape[-2:]), p.sub("depth_decoder"), cfg)
    out["pred_depth"] = depths + [depth]
    out["pred_seg"] = seg
    return out


# ------------------------------------------------------------------------------ criteria
@torch.no_grad()
def hungarian(logits, lines, targets, cfg):
    """HungarianMatcher_Line.forward, src/models/matcher.py:28-82."""
    B, Q = logits.shape[:2]
    prob = logits.flatten(0, 1).softmax(-1)
    tgt_ids = torch.cat([t["labels"] for t in targets])
    tgt_lines = torch.cat([t["lines"] for t in targets])
    C = cfg.set_cost_line * torch.cdist(lines.flatten(0, 1), tgt_lines, p=1) + cfg.set_cost_class * (-prob[:, tgt_ids])
    C = C.view(B, Q, -1).cpu()
    sizes = [len(t["lines"]) for t in targets]
    res = [linear_sum_assignment(c[i]) for i, c in enumerate(C.split(sizes, -1))]
    return [(torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)) for i, j in res]


def set_criterion(out, targets, cfg, world_size=1, num_items_global=None, taps=None):
    """SetCriterion.forward, src/models/glassrgbd.py:308-358 with losses ['lines_labels','lines']."""
    n = float(sum(len(t["labels"]) for t in targets)) if num_items_global is None else float(num_items_global)
    num_items = max(n / world_size, 1.0)                          # :321-326
    empty_weight = torch.tensor([1.0, cfg.eos_coef])

    def one(o, suffix):
        idx = hungarian(o["pred_logits"], o["pred_lines"], targets, cfg)
        if taps is not None:
            taps.setdefault("matches", []).append(idx)
        bi = torch.cat([torch.full_like(s, i) for i, (s, _) in enumerate(idx)])
        si = torch.cat([s for s, _ in idx])
        tc = torch.full(o["pred_logits"].shape[:2], 1, dtype=torch.int64)
        tc[bi, si] = torch.cat([t["labels"][j] for t, (_, j) in zip(targets, idx)])
        ce = F.cross_entropy(o["pred_logits"].transpose(1, 2), tc, empty_weight)       # :168
        tl = torch.cat([t["lines"][j] for t, (_, j) in zip(targets, idx)], dim=0)
        l1 = F.l1_loss(o["pred_lines"][bi, si], tl, reduction="none").sum() / num_items  # :239-242
        return {"loss_ce" + suffix: ce, "loss_line" + suffix: l1}

    losses = one(out, "")
    for i, a in enumerate(out.get("aux_outputs", [])):
        losses.update(one(a, f"_{i}"))
    return losses


def silog(est, gt, mask, cfg):
    """SilogLoss.forward, src/models/glassrgbd.py:366-374."""
    if cfg.log_depth_error:
        d = torch.log(est[mask]) - torch.log(gt[mask])
    else:
        e, g = est[mask], gt[mask]
        d = (e + torch.log(e)) - (g + torch.log(g))
    return torch.sqrt((d ** 2).mean() - cfg.variance_focus * (d.mean() ** 2)) * 10.0


def step_losses(out, depth_gt, seg_gt, targets, cfg, world_size=1, num_items_global=None, taps=None):
    """Loss assembly of train_one_epoch, src/engine_glassrgbd.py:62-115."""
    terms = set_criterion(out, targets, cfg, world_size, num_items_global, taps)
    total = sum(v * (cfg.line_loss_coef if "loss_line" in k else 1.0) for k, v in terms.items())
    mask = (depth_gt >= 0.2) & (depth_gt < 10.0)
    names = ["1/16", "1/8", "1/4", "1"]
    for i, pd in enumerate(out["pred_depth"]):
        size = pd.shape[-2:]
        g = F.interpolate(depth_gt, size=size, mode="nearest")
        m = F.interpolate(mask.to(torch.uint8), size=size, mode="nearest").to(torch.bool)
        ld = silog(pd, g, m, cfg) * cfg.depth_loss_weights[i]
        terms["loss_depth_" + names[i]] = ld
        total = total + ld
    ls = F.cross_entropy(out["pred_seg"], seg_gt.squeeze(1)) * cfg.seg_loss_weight
    terms["loss_seg"] = ls
    return total + ls, terms


# ------------------------------------------------------------------------------ parameters / optimizer
def is_buffer(name):
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "relative_position_index":
        return True
    return ".bn" in name or "downsample.1." in name          # FrozenBatchNorm2d buffers (backbone.py:30-33)


def is_trainable(name):
    """src/models/backbone.py:62-64: backbone conv1 + layer1 frozen; everything else trains."""
    if is_buffer(name):
        return False
    if name.startswith("backbone."):
        return any(k in name for k in ("layer2", "layer3", "layer4"))
    return True


def clip_and_adamw(params, grads, state, cfg, step):
    """clip_grad_norm_(max_norm) then torch.optim.AdamW single step, src/engine_glassrgbd.py:157-159,
    src/main_glassrgbd.py:59-66 (two LR groups, betas (0.9,0.999), eps 1e-8, decoupled wd)."""
    names = [n for n in params if grads.get(n) is not None]
    total = torch.norm(torch.stack([grads[n].norm(2) for n in names]), 2)
    coef = torch.clamp(cfg.clip_max_norm / (total + 1e-6), max=1.0)
    b1, b2, eps = 0.9, 0.999, 1e-8
    with torch.no_grad():
        for n in names:
            g = grads[n] * coef
            lr = cfg.lr_backbone if "backbone" in n else cfg.lr
            m, v = state.setdefault(n, (torch.zeros_like(g), torch.zeros_like(g)))
            params[n].mul_(1 - lr * cfg.weight_decay)
            m.lerp_(g, 1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
            denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
            params[n].addcdiv_(m, denom, value=-lr / bc1)
    return float(total)


def train_step(sd, batch, cfg, opt_state=None, step=1, training=True, taps=None, world_size=1, num_items_global=None):
    """One full step on CPU: forward, 17 loss terms, backward, clip, AdamW (in place on `sd`)."""
    leaves = {n: t.requires_grad_(True) for n, t in sd.items() if is_trainable(n)}
    out = forward(sd, batch["images"], batch["pad_mask"], cfg, training=training, taps=taps)
    total, terms = step_losses(out, batch["depth"], batch["seg"], batch["targets"], cfg, world_size, num_items_global, taps)
    grads_list = torch.autograd.grad(total, list(leaves.values()), allow_unused=True)
    grads = dict(zip(leaves, grads_list))
    for t in leaves.values():
        t.requires_grad_(False)
    gnorm = None
    if opt_state is not None:
        gnorm = clip_and_adamw(leaves, grads, opt_state, cfg, step)
    return out, total.detach(), {k: v.detach() for k, v in terms.items()}, grads, gnorm
