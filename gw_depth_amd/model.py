"""GW-Depth model on the MI355X kernels: host-side mirror of the reference's model interface.

Same module tree / state-dict key names as the reference (970 entries), same forward contract
(GlassRGBD.forward, /root/reference/src/models/glassrgbd.py:74-123) — but every feature map is kept
pixel-major (B, H, W, C), so that map <-> token reshapes are free, channel LayerNorms and softmaxes
run over the contiguous dim, and the convolutions are implicit GEMMs over NHWC.  Dense compute goes
through gw_depth_amd.ops (hand-written gfx950 kernels behind the C ABI); torch is used for device
memory, autograd bookkeeping and small index plumbing (pad / roll / gather).
"""
import math
import os

import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .layers import Conv, FrozenBN, LayerNorm, Linear, Mlp, MlpNorm, Seq
from .ops import ACT_ELU, ACT_GELU, ACT_NONE, ACT_RELU, ACT_SIGMOID

WS = 7
HEADS = 16
DEFAULT_FP32_STAGES = ()


class NestedTensor:
    """(tensors, mask) pair, /root/reference/src/util/misc.py:347-367."""

    def __init__(self, tensors, mask):
        self.tensors, self.mask = tensors, mask

    def to(self, device):
        return NestedTensor(self.tensors.to(device), None if self.mask is None else self.mask.to(device))

    def decompose(self):
        return self.tensors, self.mask


def nested_tensor_from_tensor_list(tensor_list):
    """Zero-pad to the largest (H, W); mask True = padding (/root/reference/src/util/misc.py:291-313)."""
    if isinstance(tensor_list, torch.Tensor) and tensor_list.dim() == 4:
        tensor_list = list(tensor_list)
    if tensor_list[0].dim() != 3:
        raise ValueError("not supported")
    c = tensor_list[0].shape[0]
    h = max(t.shape[1] for t in tensor_list)
    w = max(t.shape[2] for t in tensor_list)
    out = torch.zeros((len(tensor_list), c, h, w), dtype=tensor_list[0].dtype, device=tensor_list[0].device)
    mask = torch.ones((len(tensor_list), h, w), dtype=torch.bool, device=out.device)
    for img, pad, m in zip(tensor_list, out, mask):
        pad[:, : img.shape[1], : img.shape[2]].copy_(img)
        m[: img.shape[1], : img.shape[2]] = False
    return NestedTensor(out, mask)


def to_nchw(x):
    """(B,H,W,C) contiguous -> (B,C,H,W) channels-last view (no copy)."""
    return x.permute(0, 3, 1, 2)


def to_pixel_major(x):
    return x.permute(0, 2, 3, 1).contiguous()


def pos_sine(mask, num_pos_feats, normalize):
    """PositionEmbeddingSine (/root/reference/src/models/position_encoding.py:28-48) -> (B,h,w,2F) fp32."""
    if hasattr(mask, "_gwd_counts"):                      # level mask from ops.mask_levels: one kernel (csrc/posenc.hip)
        return ops.pos_sine(mask, num_pos_feats, normalize)
    not_mask = ~mask
    y = not_mask.cumsum(1, dtype=torch.float32)
    x = not_mask.cumsum(2, dtype=torch.float32)
    if normalize:
        y = y / (y[:, -1:, :] + 1e-6) * (2 * math.pi)
        x = x / (x[:, :, -1:] + 1e-6) * (2 * math.pi)
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32, device=mask.device)
    dim_t = 10000 ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_pos_feats)
    px = x[:, :, :, None] / dim_t
    py = y[:, :, :, None] / dim_t
    px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).flatten(3)
    py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).flatten(3)
    return torch.cat((py, px), dim=3)


# ------------------------------------------------------------------------------------ backbone
class Bottleneck(nn.Module):
    """torchvision ResNet v1.5 Bottleneck with FrozenBN folded into each conv and the residual +
    ReLU fused into the third conv's epilogue."""

    def __init__(self, inplanes, planes, stride, down):
        super().__init__()
        self.conv1, self.bn1 = Conv(inplanes, planes, 1), FrozenBN(planes)
        self.conv2, self.bn2 = Conv(planes, planes, 3), FrozenBN(planes)
        self.conv3, self.bn3 = Conv(planes, planes * 4, 1), FrozenBN(planes * 4)
        self.stride = stride
        self.downsample = Seq(_0=Conv(inplanes, planes * 4, 1), _1=FrozenBN(planes * 4)) if down else None

    def forward(self, x, gated_in=False, defer_out=False):
        """gated_in: x is the previous block's output and that block left its ReLU backward to us (defer_out there); inside the block
        conv1 -> conv2 -> conv3 hand theirs on the same way (ops.conv2d: defer / in_gate), so a block's three activation-backward
        passes run in the data-gradient epilogues of the layers behind them."""
        g = ACT_RELU if ops.act_gate_enabled() else ACT_NONE
        d = g != ACT_NONE
        s, b = self.bn1.folded()
        out, x = ops.conv2d(x, self.conv1.weight, row_scale=s, shift=b, act=ACT_RELU, fanout=True,     # x again, for the skip path
                            in_gate=ACT_RELU if gated_in else ACT_NONE, defer=d)
        s, b = self.bn2.folded()
        out = ops.conv2d(out, self.conv2.weight, stride=self.stride, pad=1, row_scale=s, shift=b, act=ACT_RELU, in_gate=g, defer=d)
        idt = x
        if self.downsample is not None:
            s, b = self.downsample[1].folded()
            idt = ops.conv2d(x, self.downsample[0].weight, stride=self.stride, row_scale=s, shift=b)
        s, b = self.bn3.folded()
        return ops.conv2d(out, self.conv3.weight, row_scale=s, shift=b, residual=idt, act=ACT_RELU, in_gate=g, defer=defer_out)


class ResNetBody(nn.Module):
    """conv1 .. layer4 (keys as torchvision's IntermediateLayerGetter keeps them, backbone.py:65-69)."""

    def __init__(self):
        super().__init__()
        self.conv1, self.bn1 = Conv(3, 64, 7), FrozenBN(64)
        inpl = 64
        for li, (planes, blocks, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)), start=1):
            layer = [Bottleneck(inpl, planes, stride, True)]
            inpl = planes * 4
            layer += [Bottleneck(inpl, planes, 1, False) for _ in range(1, blocks)]
            setattr(self, f"layer{li}", nn.Sequential(*layer))
        for m in self.modules():
            if isinstance(m, Conv):
                nn.init.normal_(m.weight, std=math.sqrt(2.0 / (m.weight.shape[0] * m.weight.shape[1] * m.weight.shape[2])))

    def forward(self, x):
        s, b = self.bn1.folded()
        if x.is_cuda and x.dtype == torch.bfloat16 and not (self.conv1.weight.requires_grad and torch.is_grad_enabled()) \
                and not x.requires_grad:
            x = ops.stem(x, self.conv1.weight, s, b)             # conv1 + bn1 + relu + maxpool, one forward-only kernel
        else:
            x = ops.conv2d(x, self.conv1.weight, stride=2, pad=3, row_scale=s, shift=b, act=ACT_RELU)
            x = to_pixel_major(F.max_pool2d(to_nchw(x), 3, 2, 1))
        feats = []
        # a block whose output feeds only the next block of its layer defers its last ReLU backward to that block's first conv
        # (which, with the fan-out, is the single consumer of the map); the last block's output also leaves as a feature level
        chain = ops.act_gate_enabled()
        for li in range(1, 5):
            blocks = getattr(self, f"layer{li}")
            for bi, blk in enumerate(blocks):
                x = blk(x, gated_in=chain and bi > 0, defer_out=chain and bi + 1 < len(blocks))
            feats.append(x)
        return feats


class BackboneBase(nn.Module):
    def __init__(self, train_backbone=True):
        super().__init__()
        self.body = ResNetBody()
        for name, p in self.body.named_parameters():            # backbone.py:62-64
            if not train_backbone or ("layer2" not in name and "layer3" not in name and "layer4" not in name):
                p.requires_grad_(False)
        self.num_channels = 2048


class PosHolder(nn.Module):
    """Parameter-free slot '1' of the reference's Joiner (keeps module indices aligned)."""

    def __init__(self, num_pos_feats):
        super().__init__()
        self.num_pos_feats = num_pos_feats


class Joiner(nn.Module):
    """backbone.0 = body holder, backbone.1 = sine position embedding (backbone.py:101-110)."""

    def __init__(self, hidden_dim, train_backbone=True):
        super().__init__()
        self.add_module("0", BackboneBase(train_backbone))
        self.add_module("1", PosHolder(hidden_dim // 2))
        self.num_channels = 2048

    def forward(self, images_pm, pad_mask):
        feats = self._modules["0"].body(images_pm)
        if pad_mask.is_cuda:
            masks = ops.mask_levels(pad_mask, [tuple(f.shape[1:3]) for f in feats])      # + the counts pos_sine() builds on
        else:
            masks = [F.interpolate(pad_mask[None].float(), size=f.shape[1:3]).to(torch.bool)[0] for f in feats]
        return feats, masks


# ------------------------------------------------------------------------------------ DETR branch
class DropPool:
    """Every dropout multiplier (0 or 1/(1-p)) of one DETR forward pass from ONE ATen launch: F.dropout over a persistent
    tensor of ones (ATen's graph-safe Philox stream decides), handed out as slices.  The consumers are kernel epilogues
    (ops.linear(mult=...): x + dropout(sublayer) in the producing GEMM) and the attention kernels - no dropout, masked-scale
    or residual-add launch is left in the DETR branch.  take() returns None when dropout is off."""

    def __init__(self):
        self.ones, self.buf, self.off = None, None, 0

    def begin(self, total, p, training, dtype, device):
        self.buf, self.off = None, 0
        if not training or p <= 0 or total <= 0 or device.type != "cuda":
            return self
        total = (total + 7) // 8 * 8
        if self.ones is None or self.ones.numel() < total or self.ones.dtype != dtype or self.ones.device != device:
            self.ones = torch.ones(total, dtype=dtype, device=device)
        self.buf = F.dropout(self.ones[:total], p, True)
        return self

    def take(self, *shape):
        if self.buf is None:
            return None
        n = 1
        for d in shape:
            n *= int(d)
        out = self.buf[self.off:self.off + n].view(*shape)
        self.off += (n + 7) // 8 * 8                     # 16-byte aligned slices
        return out


def _drop(t, p, training, pool):
    """F.dropout for the paths without a multiplier pool (CPU host-logic tests, fp32 parity mode with dropout on)."""
    return F.dropout(t, p, training)


class MultiheadAttention(nn.Module):
    """Packed in-proj attention, /root/reference/src/models/multi_head_attention.py:117-380, batch-major."""

    def __init__(self, dim, heads, dropout):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * dim, dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * dim))
        self.out_proj = Linear(dim, dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        self.dim, self.heads, self.dropout = dim, heads, dropout

    def forward(self, query, key, value, key_padding_mask=None, pool=None, residual=None, out_mult=None):
        """out_proj(attention) [* out_mult + residual]: the sub-layer's dropout and skip ride in the out-projection's epilogue
        when the caller passes them (transformer.py:149-162)."""
        B, L, E = query.shape
        S = key.shape[1]
        H, hd = self.heads, E // self.heads
        W, b = self.in_proj_weight, self.in_proj_bias
        scale = float(hd) ** -0.5
        amult = pool.take(B, H, L, S) if pool is not None else None
        if query is key:
            qk = ops.linear(query, W, b, rows=(0, 2 * E))
            if residual is value:            # value also feeds the skip: both gradients of it meet in the v projection's data gradient
                v, residual = ops.linear(value, W, b, rows=(2 * E, 3 * E), fanout=True)
            else:
                v = ops.linear(value, W, b, rows=(2 * E, 3 * E))
            fused = ops.mha_core(qk, None, v, H, key_padding_mask, self.dropout, self.training, scale, amult)
            q, k = qk[..., :E], qk[..., E:]
        else:
            q = ops.linear(query, W, b, rows=(0, E))
            k = ops.linear(key, W, b, rows=(E, 2 * E))
            v = ops.linear(value, W, b, rows=(2 * E, 3 * E))
            fused = ops.mha_core(q, k, v, H, key_padding_mask, self.dropout, self.training, scale, amult)
        if fused is None:                   # fp32 parity mode: the unfused arithmetic of the reference
            q = q.reshape(B, L, H, hd).transpose(1, 2)
            k = k.reshape(B, S, H, hd).transpose(1, 2)
            v = v.reshape(B, S, H, hd).transpose(1, 2)
            # q * scaling and masked_fill(-inf) (multi_head_attention.py:329-352) are folded into the softmax kernel
            own = q.is_cuda                   # gwd_bmm on the strided head views (exact fp32 MFMA); the CPU stand-in of the tests uses torch
            att = ops.attention_softmax(ops.matmul_nt(q, k) if own else q @ k.transpose(-2, -1), key_padding_mask, scale)
            att = att * amult if amult is not None else F.dropout(att, self.dropout, self.training)
            fused = (ops.matmul_nn(att, v) if own else att @ v).transpose(1, 2).reshape(B, L, E)
        # bf16: one matrix-core kernel each way, no L x S tensor, no head split / merge copies
        return self.out_proj(fused, residual=residual, mult=out_mult)


class EncoderLayer(nn.Module):
    def __init__(self, d, heads, ff, dropout):
        super().__init__()
        self.self_attn = MultiheadAttention(d, heads, dropout)
        self.linear1, self.linear2 = Linear(d, ff), Linear(ff, d)
        self.norm1, self.norm2 = LayerNorm(d), LayerNorm(d)
        self.p = dropout

    def forward(self, x, pos, kpm, pool=None):
        """TransformerEncoderLayer.forward_post, /root/reference/src/models/transformer.py:149-162."""
        qk = x + pos
        if pool is not None and pool.buf is not None:      # dropout multipliers + skips inside the producing GEMMs' epilogues
            x = self.norm1(self.self_attn(qk, qk, x, kpm, pool, residual=x, out_mult=pool.take(*x.shape)))
            h, x = self.linear1(x, ACT_RELU, mult=pool.take(*x.shape[:-1], self.linear1.weight.shape[0]), fan=True)    # x again, for the skip
            return self.norm2(self.linear2(h, residual=x, mult=pool.take(*x.shape)))
        x = self.norm1(x + F.dropout(self.self_attn(qk, qk, x, kpm), self.p, self.training))
        ff = self.linear2(F.dropout(self.linear1(x, ACT_RELU), self.p, self.training))
        return self.norm2(x + F.dropout(ff, self.p, self.training))


class DecoderLayer(nn.Module):
    def __init__(self, d, heads, ff, dropout):
        super().__init__()
        self.self_attn = MultiheadAttention(d, heads, dropout)
        self.multihead_attn = MultiheadAttention(d, heads, dropout)
        self.linear1, self.linear2 = Linear(d, ff), Linear(ff, d)
        self.norm1, self.norm2, self.norm3 = LayerNorm(d), LayerNorm(d), LayerNorm(d)
        self.p = dropout

    def forward(self, tgt, memory, mem_pos, pos, qpos, kpm, pool=None):
        """TransformerDecoderLayer.forward_post, /root/reference/src/models/transformer.py:212-233."""
        qk = tgt + qpos
        if pool is not None and pool.buf is not None:
            tgt = self.norm1(self.self_attn(qk, qk, tgt, None, pool, residual=tgt, out_mult=pool.take(*tgt.shape)))
            tgt = self.norm2(self.multihead_attn(tgt + qpos, mem_pos, memory, kpm, pool, residual=tgt, out_mult=pool.take(*tgt.shape)))
            h, tgt = self.linear1(tgt, ACT_RELU, mult=pool.take(*tgt.shape[:-1], self.linear1.weight.shape[0]), fan=True)
            return self.norm3(self.linear2(h, residual=tgt, mult=pool.take(*tgt.shape)))
        tgt = self.norm1(tgt + F.dropout(self.self_attn(qk, qk, tgt), self.p, self.training))
        t2 = self.multihead_attn(tgt + qpos, mem_pos, memory, kpm)
        tgt = self.norm2(tgt + F.dropout(t2, self.p, self.training))
        ff = self.linear2(F.dropout(self.linear1(tgt, ACT_RELU), self.p, self.training))
        return self.norm3(tgt + F.dropout(ff, self.p, self.training))


class Stack(nn.Module):
    def __init__(self, layers, norm=None):
        super().__init__()
        self.layers = nn.ModuleList(layers)
        if norm is not None:
            self.norm = norm


class Transformer(nn.Module):
    """DETR encoder/decoder, /root/reference/src/models/transformer.py:18-125 (post-norm)."""

    def __init__(self, d=256, heads=8, enc=6, dec=6, ff=2048, dropout=0.1):
        super().__init__()
        self.encoder = Stack([EncoderLayer(d, heads, ff, dropout) for _ in range(enc)])
        self.decoder = Stack([DecoderLayer(d, heads, ff, dropout) for _ in range(dec)], LayerNorm(d))
        self.d_model, self.nhead = d, heads
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, src, mask, query_embed, pos):
        """src, pos (B,h,w,d); mask (B,h,w) -> hs (layers, B, Q, d)."""
        B = src.shape[0]
        x = src.flatten(1, 2)
        pos = pos.flatten(1, 2).to(x.dtype)
        kpm = mask.flatten(1)
        L, d, Q, H = x.shape[1], x.shape[2], query_embed.shape[0], self.nhead
        p = self.encoder.layers[0].p if len(self.encoder.layers) else 0.0
        ff = self.encoder.layers[0].linear1.weight.shape[0] if len(self.encoder.layers) else 0
        pool = None
        if x.is_cuda and x.dtype == torch.bfloat16 and self.training and p > 0:
            n_enc = len(self.encoder.layers) * (B * H * L * L + 2 * B * L * d + B * L * ff + 32)
            n_dec = len(self.decoder.layers) * (B * H * Q * Q + B * H * Q * L + 3 * B * Q * d + B * Q * ff + 48)
            if not hasattr(self, "_pool"):
                self._pool = DropPool()
            pool = self._pool.begin(n_enc + n_dec, p, True, x.dtype, x.device)
        for layer in self.encoder.layers:
            x = layer(x, pos, kpm, pool)
        memory, mem_pos = x, x + pos
        qpos = query_embed.to(x.dtype).unsqueeze(0).expand(B, -1, -1)
        tgt = torch.zeros_like(qpos)
        inter = []
        for layer in self.decoder.layers:
            tgt = layer(tgt, memory, mem_pos, pos, qpos, kpm, pool)
            inter.append(self.decoder.norm(tgt))
        return torch.stack(inter)


class MLP(nn.Module):
    """glassrgbd.py:30-42."""

    def __init__(self, cin, hidden, cout, n):
        super().__init__()
        dims = [cin] + [hidden] * (n - 1) + [cout]
        self.layers = nn.ModuleList(Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))
        self.n = n

    def forward(self, x):
        for i, l in enumerate(self.layers):
            x = l(x, ACT_RELU if i < self.n - 1 else ACT_NONE)
        return x


# ------------------------------------------------------------------------------------ window stages
def window_partition(x):
    B, H, W, C = x.shape
    x = x.view(B, H // WS, WS, W // WS, WS, C)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, WS * WS, C)


def window_reverse(win, H, W):
    B = win.shape[0] // ((H // WS) * (W // WS))
    x = win.view(B, H // WS, W // WS, WS, WS, -1)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, -1)


_REGION_CACHE = {}


def shift_regions(Hp, Wp, device):
    """SW-MSA mask (multiscale_transformerr.py:937-955) in compact form: the 9-region label of every token of
    every window, int32 (nW, 49).  The kernels add -100 where two tokens of a window carry different labels."""
    key = (Hp, Wp, str(device))
    if key not in _REGION_CACHE:
        shift = WS // 2
        img = torch.zeros(1, Hp, Wp, 1)
        cnt = 0
        for h in (slice(0, -WS), slice(-WS, -shift), slice(-shift, None)):
            for w in (slice(0, -WS), slice(-WS, -shift), slice(-shift, None)):
                img[:, h, w, :] = cnt
                cnt += 1
        _REGION_CACHE[key] = window_partition(img).view(-1, WS * WS).to(torch.int32).to(device).contiguous()
    return _REGION_CACHE[key]


def relative_position_index():
    coords = torch.stack(torch.meshgrid(torch.arange(WS), torch.arange(WS), indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += WS - 1
    rel[:, :, 1] += WS - 1
    rel[:, :, 0] *= 2 * WS - 1
    return rel.sum(-1)


class WindowAttnBase(nn.Module):
    def __init__(self, dim, border=False):
        super().__init__()
        self.dim = dim
        self.scale = (dim // HEADS) ** -0.5
        # registration order = the reference's (multiscale_transformerr.py:222-247, 401-410): torch.optim state dicts are
        # positional, so optimizer checkpoints only interchange if named_parameters() enumerates in the same order
        self.diff_mu = nn.Parameter(torch.randn(1, 1, dim))
        self.diff_logsigma = nn.Parameter(torch.zeros(1, 1, dim))
        nn.init.xavier_uniform_(self.diff_logsigma)
        if border:
            self.border_mu = nn.Parameter(torch.randn(1, 1, dim))
            self.border_logsigma = nn.Parameter(torch.zeros(1, 1, dim))
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * WS - 1) ** 2, HEADS))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        self.register_buffer("relative_position_index", relative_position_index())
        self.qkv = Linear(dim, dim * 3)
        self.proj = Linear(dim, dim)

    def rel_index(self):
        """relative_position_index (49, 49) int64 buffer as the flat int32 vector the kernels gather the bias table through
        (multiscale_transformerr.py:313-315: bias(h, i, j) = table[index[i, j], h]); built once per device."""
        r = getattr(self, "_rel32", None)
        if r is None or r.device != self.relative_position_index.device:
            r = self._rel32 = self.relative_position_index.reshape(-1).to(torch.int32).contiguous()
        return r


class WindowAttention(WindowAttnBase):
    """Line-point-guided window attention of the 1/32 stage (multiscale_transformerr.py:202-332)."""

    def __init__(self, dim):
        super().__init__(dim)
        self.ref_qk = Linear(dim, dim * 2)
        self.ref_attn_diffusion = Conv(HEADS, HEADS, 3, bias=True)

    def forward(self, xw, x_ref, regions):
        B_, N, C = xw.shape
        hd = C // HEADS
        qkv = self.qkv(xw).view(B_, N, 3, HEADS, hd)
        rq, rv = self.ref_qk(x_ref).split(C, dim=-1)       # split, not two slices: its backward is ONE cat (two slices = 2 fills + 2 copies + 1 add)
        rB = rq.shape[0]
        ref_k = ops.row_affine(rq, self.diff_mu, self.diff_logsigma)              # mu + exp(logsigma) * x, (rB, nrf, C), :289-292
        link = ops.GradLink()                                                     # one packed qkv gradient from the two consumers below
        ra = ops.ref_scores(qkv, ref_k, rB, self.scale, link)                    # (rB, nWin*N, nrf, heads): pixel-major map, :295-298
        for _ in range(3):                                                        # :299-302
            upd, ra = ops.conv2d(ra, self.ref_attn_diffusion.weight, self.ref_attn_diffusion.bias, pad=1, fanout=True)     # ra again, for the skip
            ra = ops.inorm_gelu_residual(ra, upd, 1e-5)
        q_new = ops.ref_mix(ra, rv, HEADS).view(B_, N, HEADS, hd)      # softmax over the ref tokens, . ref_v; second *scale: in-kernel
        wpi = regions.shape[0] if regions is not None else 1
        x = ops.window_attention_qkv(q_new, qkv, self.relative_position_bias_table, self.rel_index(), regions, wpi, self.scale, link)
        return self.proj(x)


class WindowClassAttention(WindowAttnBase):
    """Window attention + depth/seg class-token cross attention (multiscale_transformerr.py:375-580,
    group_attention=False).  border_* and proj_seg exist for state-dict parity and never get a gradient."""

    def __init__(self, dim, tdim):
        super().__init__(dim, border=True)
        self.cls_dth_q, self.cls_seg_q = Linear(tdim, tdim), Linear(tdim, tdim)
        self.global_k, self.global_v = Linear(dim + 2 * tdim, dim + 2 * tdim), Linear(dim + 2 * tdim, dim + 2 * tdim)
        self.proj_dth, self.proj_seg = Linear(tdim, tdim), Linear(tdim, tdim)

    def forward(self, xw, dtok, stok, regions):
        B_, N, C = xw.shape
        qkv = self.qkv(xw).view(B_, N, 3, HEADS, C // HEADS)
        wpi = regions.shape[0] if regions is not None else 1
        x = self.proj(ops.window_attention_packed(qkv, self.relative_position_bias_table, self.rel_index(), regions, wpi, self.scale))
        tdim = dtok.shape[-1]
        tx = torch.cat([x, dtok, stok], dim=-1)
        tC = tx.shape[-1]
        tk = self.global_k(tx).view(B_, N, HEADS, tC // HEADS)
        tv = self.global_v(tx).view(B_, N, HEADS, tC // HEADS)

        # :561-578; the two tokens query the same tk / tv (one launch for the pair), and both go through proj_dth
        qd = self.cls_dth_q(dtok).view(B_, N, HEADS, tdim // HEADS)
        qs = self.cls_seg_q(stok).view(B_, N, HEADS, tdim // HEADS)
        od, os_ = ops.token_attention_pair(qd, qs, tk, tv, self.scale)
        return x, self.proj_dth(od), self.proj_dth(os_)


def pad_roll(t, H, W, shift):
    pr, pb = (WS - W % WS) % WS, (WS - H % WS) % WS
    if pr or pb:
        t = F.pad(t, (0, 0, 0, pr, 0, pb))
    if shift:
        t = torch.roll(t, shifts=(-shift, -shift), dims=(1, 2))
    return t, H + pb, W + pr


def unroll_crop(win, H, W, Hp, Wp, shift):
    t = window_reverse(win, Hp, Wp)
    if shift:
        t = torch.roll(t, shifts=(shift, shift), dims=(1, 2))
    return t[:, :H, :W, :]


class SwinBlock(nn.Module):
    """SwinTransformerBlock.forward, multiscale_transformerr.py:646-788."""

    def __init__(self, dim, shift, tdim=None):
        super().__init__()
        self.shift = shift
        self.norm1 = LayerNorm(dim)
        self.attn = WindowAttention(dim) if tdim is None else WindowClassAttention(dim, tdim)
        self.norm2 = LayerNorm(dim)
        self.mlp = Mlp(dim, dim * 2)
        if tdim is not None:
            self.norm_seg1, self.norm_depth1 = LayerNorm(tdim), LayerNorm(tdim)
            self.mlp_seg, self.norm_seg2 = Mlp(tdim, tdim * 2), LayerNorm(tdim)
            self.mlp_depth, self.norm_depth2 = Mlp(tdim, tdim * 2), LayerNorm(tdim)

    def forward(self, x, H, W, ref_coors=None, ref_pos=None, dtok=None, stok=None):
        B, L, C = x.shape
        shift = self.shift
        xn, x = self.norm1(x, fan=True)                      # x again, for the skip below (pre-norm block: x feeds the norm AND the skip)
        xn = xn.view(B, H, W, C)
        Hp, Wp = (H + WS - 1) // WS * WS, (W + WS - 1) // WS * WS
        mask = shift_regions(Hp, Wp, x.device) if shift else None
        if dtok is None:
            if shift:                                                             # :678-686
                # the shifted blocks of a layer get the same points: shifted coordinates / rolled position map made once per forward
                memo = getattr(ref_coors, "_gwd_shifted", None)
                if memo is None or memo[0] != (shift, Hp, Wp) or memo[1] is not ref_pos:
                    rc = torch.stack([ref_coors[..., 0] - (shift / (Wp - 1)) * 2,
                                      ref_coors[..., 1] - (shift / (Hp - 1)) * 2], dim=-1)
                    rc = torch.where(rc < -1, -2 - rc, rc)
                    rpos = torch.roll(ref_pos, shifts=(-shift, -shift), dims=(1, 2))
                    ref_coors._gwd_shifted = ((shift, Hp, Wp), ref_pos, rc, rpos)
                else:
                    rc, rpos = memo[2], memo[3]
            else:
                rc, rpos = ref_coors, ref_pos
            # the padded / rolled map (sx of :662-676) is never built: the points are sampled in its frame
            x_ref = (ops.point_sample(xn, rc, nearest=True, frame=(Hp, Wp, shift)) + ops.point_sample(rpos, rc, nearest=True)).to(x.dtype)   # (B, S, C)
            aw = self.attn(ops.window_gather(xn, shift), x_ref, mask)          # == window_partition(sx), one index-remapping copy
        else:
            tC = dtok.shape[-1]
            dn, dtok = self.norm_depth1(dtok, fan=True)
            sn, stok = self.norm_seg1(stok, fan=True)
            # the three maps of the block are partitioned / reversed together: one launch each way instead of three
            xw, dn, sn = ops.window_gather_multi([xn, dn.view(B, H, W, tC), sn.view(B, H, W, tC)], shift)
            aw, dw, sw = self.attn(xw, dn, sn, mask)
            x, d, s = ops.window_scatter_multi([aw, dw, sw], B, H, W, shift, [x, dtok, stok])
            x = self.mlp(*self.norm2(x.view(B, H * W, C), fan=True))
            d = self.mlp_depth(*self.norm_depth2(d.view(B, H * W, tC), fan=True))
            s = self.mlp_seg(*self.norm_seg2(s.view(B, H * W, tC), fan=True))
            return x, d, s
        x = ops.window_scatter(aw, B, H, W, shift, residual=x).view(B, H * W, C)     # window reverse + un-shift + crop + skip
        x = self.mlp(*self.norm2(x, fan=True))              # mlp(norm(x), residual = x)
        return x, None, None


class BasicLayer(nn.Module):
    def __init__(self, dim, depth, tdim=None, pre_class_pred=False):
        super().__init__()
        self.blocks = nn.ModuleList(SwinBlock(dim, 0 if i % 2 == 0 else WS // 2, tdim) for i in range(depth))
        if pre_class_pred:      # built by the reference (multiscale_transformerr.py:911-915), never executed
            self.pre_depth_pred = Seq(_0=Linear(dim + tdim, tdim), _1=Linear(tdim, 1))

    def forward(self, x, H, W, **kw):
        d, s = kw.pop("dtok", None), kw.pop("stok", None)
        for blk in self.blocks:
            x, d, s = blk(x, H, W, dtok=d, stok=s, **kw)
        return x, d, s


# ------------------------------------------------------------------------------------ pyramid heads
class ConvLn(nn.Module):
    """conv (no bias) -> LayerNorm over channels [-> GELU], points_sample.py:12-25."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.conv = Conv(cin, cout, k)
        self.layer_norm = LayerNorm(cout)
        self.pad = k // 2

    def forward(self, x, gelu=False, residual=None, geom=None, fan=False):
        """fan: also return the input again for its second consumer (the block's skip): ops._ConvFn fan-out."""
        if x.is_cuda and x.dtype == torch.bfloat16:
            # one launch: the LayerNorm (+ GELU / + skip) runs in the convolution's epilogue (ops._ConvLnFn)
            return ops.conv_ln(x, self.conv.weight, self.layer_norm.weight, self.layer_norm.bias, self.pad, gelu=gelu, residual=residual,
                               geom=geom, fanout=fan)
        if geom is not None:                # zero-padded channel counts (PyramidLayer.padded_width); the LayerNorm sees the pitch
            y = ops.conv2d_padded(x, self.conv.weight, self.pad, geom, fanout=fan)
        else:
            y = ops.conv2d(x, self.conv.weight, pad=self.pad, fanout=fan)
        if fan:
            return self.layer_norm(y[0], gelu, residual=residual), y[1]
        return self.layer_norm(y, gelu, residual=residual)


class PyrBlock(nn.Module):
    """BasicBlock, points_sample.py:27-43 (no downsample at these widths)."""

    def __init__(self, c):
        super().__init__()
        self.conv1 = Seq(_0=ConvLn(c, c, 3))
        self.conv2 = ConvLn(c, c, 3)

    def forward(self, x, geom=None):
        h, xs = self.conv1[0](x, True, geom=geom, fan=True)      # xs = x: both gradients of x then meet in conv1's data-gradient epilogue
        return self.conv2(h, residual=xs, geom=geom)              # the skip is added in the LayerNorm kernel


class PyramidLayer(nn.Module):
    """PyramidLayer, points_sample.py:45-125; layer4 is constructed (state dict) but never called."""

    def __init__(self, c, pools=(16, 8, 4, 2)):
        super().__init__()
        self.pools = pools
        self.firstconv = Seq(_0=ConvLn(c, c, 3), _2=ConvLn(c, 2 * c, 3))
        c2 = 2 * c
        self.layer1 = nn.Sequential(PyrBlock(c2))
        self.layer2 = nn.Sequential(PyrBlock(c2), PyrBlock(c2))
        self.layer3 = nn.Sequential(PyrBlock(c2), PyrBlock(c2))
        self.layer4 = nn.Sequential(PyrBlock(c2))
        for i in range(1, 5):
            setattr(self, f"branch{i}", Seq(_1=ConvLn(c2, c2, 3)))
        self.lastconv = Seq(_0=ConvLn(5 * c2, 2 * c2, 3), _2=Conv(2 * c2, c, 1))

    FORCE_PAD = False           # tests: the padded route on the CPU stand-in too

    def padded_width(self, ref):
        """Width of the map this pyramid wants: c, or c rounded up to 32 when c is not a multiple of 8 (the 30-point pyramid: 30 / 60 /
        300 / 120 channels are rows of 60 - 600 bytes, which keep every conv on the register-staged odd-width kernels at 80 - 190
        TFLOP/s and the weight gradients out of the grouped launches; as 32 / 64 / 320 / 128 they take the LDS-DMA route).  The
        caller hands over a map of that width whose extra channels are ZERO and gets the result in the same width."""
        c = self.firstconv[0].conv.weight.shape[0]
        if c % 8 and ((ref.dtype == torch.bfloat16 and ref.is_cuda) or self.FORCE_PAD):
            return (c + 31) // 32 * 32
        return c

    def forward(self, x):
        c = self.firstconv[0].conv.weight.shape[0]
        if x.shape[-1] != c:
            return self.forward_padded(x, c)
        x = self.firstconv[2](self.firstconv[0](x, True), True)
        x = self.layer3(self.layer2(self.layer1(x)))
        B, H, W, C = x.shape
        if H < self.pools[0] or W < self.pools[0]:
            x = F.pad(x, (0, 0, 0, max(self.pools[0] - W, 0), 0, max(self.pools[0] - H, 0)))
        x, pooled = ops.psp_pools(x, self.pools)                  # the four average pools in one pass
        ys = [getattr(self, f"branch{i}")[1](p, True) for i, p in enumerate(pooled, start=1)]
        x = self.lastconv[0](ops.pyramid_concat(x, ys), True)      # the up-sampling kernels write the concat's channel slices
        return ops.conv2d(x, self.lastconv[2].weight)

    def forward_padded(self, x, c):
        """The same layers on zero-padded channel counts (ops._PadConvFn); pooling, resampling and the concat are channel-agnostic
        and keep zeros zero, the LayerNorms normalise the real channels and write zeros into the padding."""
        r32 = lambda n: (n + 31) // 32 * 32
        cp, c2, c2p, c4, c4p = r32(c), 2 * c, r32(2 * c), 4 * c, r32(4 * c)
        if x.shape[-1] != cp:
            raise ValueError("PyramidLayer: a map of %d or %d channels expected, got %d" % (c, cp, x.shape[-1]))
        x = self.firstconv[2](self.firstconv[0](x, True, geom=(cp, c, cp)), True, geom=(c2p, c, cp))
        g2 = (c2p, c2, c2p)
        for layer in (self.layer1, self.layer2, self.layer3):
            for blk in layer:
                x = blk(x, g2)
        B, H, W, C = x.shape
        if H < self.pools[0] or W < self.pools[0]:
            x = F.pad(x, (0, 0, 0, max(self.pools[0] - W, 0), 0, max(self.pools[0] - H, 0)))
        x, pooled = ops.psp_pools(x, self.pools)
        ys = [getattr(self, f"branch{i}")[1](p, True, geom=g2) for i, p in enumerate(pooled, start=1)]
        x = self.lastconv[0](ops.pyramid_concat(x, ys), True, geom=(c4p, c2, c2p))      # five groups of c2 channels, each padded
        return ops.conv2d_padded(x, self.lastconv[2].weight, 0, (cp, c4, c4p))


class PointBasedPred(nn.Module):
    """PointBasedPred.forward, points_sample.py:257-280."""

    def __init__(self, dim, tdim, point_num):
        super().__init__()
        self.dim = dim
        self.pre_proj = Linear(dim + tdim, dim)
        self.refer_proj = Linear(dim, dim * 2)
        self.pyramid = PyramidLayer(point_num)

    def forward(self, x, dtok, pre_depth, coords, H, W, pos):
        B = x.shape[0]
        xg_xr = self.refer_proj(self.pre_proj(torch.cat([x, dtok], dim=-1)))
        xg, xr = xg_xr.split(self.dim, dim=-1)
        refer = ops.point_sample(xr.reshape(B, H, W, self.dim), coords) + ops.point_sample(pos, coords)   # (B, S, dim) fp32
        hp, wp = pre_depth.shape[-2:]
        anchor = ops.point_sample(pre_depth.float().reshape(B, hp, wp, 1), coords)       # (B, S, 1)
        S = refer.shape[1]
        Sp = self.pyramid.padded_width(xg)                   # S, or S rounded up to 32: zero rows in `refer` = zero channels of the map
        if Sp != S:
            refer = F.pad(refer, (0, 0, 0, Sp - S))
        if xg.is_cuda:                                       # own batched GEMM (gwd_bmm): no vendor library on the path
            rg = ops.matmul_nt(xg, refer.to(xg.dtype), alpha=self.dim ** -2)            # (B, HW, S) = pixel-major map
        else:
            rg = torch.bmm(xg, refer.transpose(1, 2).to(xg.dtype)) * (self.dim ** -2)
        # NB: when H or W < 16 the pyramid zero-pads its map and the reference keeps the padded size (:94-125)
        logits = self.pyramid(rg.view(B, H, W, -1))
        att = ops.softmax_lastdim(logits if Sp == S else logits[..., :S])
        Ho, Wo, R = att.shape[1], att.shape[2], att.shape[3]
        if R <= 256:
            pred = ops.anchor_depth(att.view(B, Ho * Wo, R), anchor.view(B, R))          # sum_r att * anchor depth, one pass
        else:
            pred = (ops.matmul_nn if att.is_cuda else torch.bmm)(att.float().view(B, Ho * Wo, R), anchor.view(B, R, 1))
        return pred.view(B, 1, Ho, Wo)                                                  # (B,1,H',W') fp32


@torch.no_grad()
def certain_sample(pred_small, pred_large, interval, sample_num, min_depth):
    """CertainSample.forward, points_sample.py:291-364, as ONE device kernel (no host round trip): the interval
    histogram decides HOW MANY points, every top-k is over the whole variance map; integer-valued, no gradient.
    pred_small (B,1,hs,ws), pred_large (B,1,H,W) fp32 -> (B, S, 1, 2) normalised coordinates."""
    B = pred_large.shape[0]
    edges = _edges(tuple([min_depth] + list(interval) + [1.0]), pred_large.device)
    coords = torch.empty((B, sample_num, 1, 2), dtype=torch.float32, device=pred_large.device)
    ops.hip.library().certain_sample(pred_small.float().contiguous(), pred_large.float().contiguous(), coords, edges, sample_num)
    return coords


_EDGE_CACHE = {}


def _edges(values, device):
    key = (values, str(device))
    if key not in _EDGE_CACHE:
        _EDGE_CACHE[key] = torch.tensor(values, dtype=torch.float32, device=device)
    return _EDGE_CACHE[key]


# ------------------------------------------------------------------------------------ dense encoder
class ConvA(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = Conv(cin, cout, 3, bias=True)

    def forward(self, x):
        return ops.conv2d(x, self.conv.weight, self.conv.bias, pad=1, act=ACT_GELU)


def sig_head(x, seq):
    return ops.linear(seq[0](x), seq[1].weight, seq[1].bias, ACT_SIGMOID)


def nearest_up_tokens(tok, Hs, Ws, size):
    """(B, Hs*Ws, C) -> nearest-upsampled (B, H*W, C)."""
    B, _, C = tok.shape
    return ops.upsample_nearest(tok.reshape(B, Hs, Ws, C), size).view(B, size[0] * size[1], C)


class ReferTransformer(nn.Module):
    """ReferTransformer, multiscale_transformerr.py:1025-1319 (flags with_line, not with_line_depth)."""

    def __init__(self, cfg):
        super().__init__()
        D, T = cfg.dense_trans_dim, cfg.class_token_dim
        self.cfg = cfg
        self.depth_token = nn.Parameter(torch.zeros(1, 1, T))
        self.seg_token = nn.Parameter(torch.zeros(1, 1, T))
        nn.init.trunc_normal_(self.depth_token, std=0.02)
        nn.init.trunc_normal_(self.seg_token, std=0.02)
        self.dense_transformer = BasicLayer(D, cfg.dense_trans_layers[0])
        self.depth_pred32 = Seq(_0=Linear(D, T), _1=Linear(T, 1))
        self.proj_class1, self.proj_backbn1 = Linear(D, D // 2), ConvA(1024, D // 2)
        self.class_transformer1 = BasicLayer(D // 2, cfg.class_trans_layers[0], T, pre_class_pred=True)
        self.depth_pred16 = Seq(_0=Linear(D // 2 + T, T), _1=Linear(T, 1))
        self.point_based_pred1 = PointBasedPred(D // 4, T, cfg.interval_sample_num[0])
        self.old_depth_token_proj8, self.old_seg_token_proj8 = MlpNorm(T, 2 * T), MlpNorm(T, 2 * T)
        self.proj_class2, self.proj_backbn2 = Linear(D // 2, D // 4), ConvA(512, D // 4)
        self.class_transformer2 = BasicLayer(D // 4, cfg.class_trans_layers[1], T)
        self.point_based_pred2 = PointBasedPred(D // 8, T, cfg.interval_sample_num[1])
        self.old_depth_token_proj4, self.old_seg_token_proj4 = MlpNorm(T, 2 * T), MlpNorm(T, 2 * T)
        self.proj_class3, self.proj_backbn3 = Linear(D // 4, D // 8), ConvA(256, D // 8)
        self.class_transformer3 = BasicLayer(D // 8, cfg.class_trans_layers[2], T)
        self.depth_pred4 = Seq(_0=Linear(D // 8 + T, T), _1=Linear(T, 1))      # never executed (:1288-1292)

    def forward(self, top, feats, masks, pred_lines, pred_logits, taps=None, cast=None):
        cfg = self.cfg
        B, H, W, C = top.shape
        dt = top.dtype
        cast = cast or (lambda stage, *ts: ts if len(ts) > 1 else ts[0])       # GlassRGBD.stage_cast: per-stage storage precision
        top = cast("dense32", top)
        ids = torch.topk(pred_logits[:, :, 0].float(), cfg.num_ref, dim=-1)[1]                  # :1166 (raw logit)
        if taps is not None and "force_topk_ids" in taps:       # parity tests: identical index operands downstream (the ORDER of
            taps["own_topk_ids"] = ids                          # the reference points matters: a 3x3 conv runs over that axis)
            ids = taps["force_topk_ids"]
        pts = torch.gather(pred_lines.float(), 1, ids[..., None].expand(-1, -1, pred_lines.shape[-1]))
        pts = (pts.reshape(B, cfg.num_ref, -1, 2) * 2 - 1.0)[:, :, :2]                          # :1175-1179
        pos = pos_sine(masks[3], cfg.dense_trans_dim // 2, False)
        x, _, _ = self.dense_transformer(top.flatten(1, 2), H, W, ref_coors=pts, ref_pos=pos)
        depth0 = sig_head(x, self.depth_pred32).float().view(B, 1, H, W)

        def stage(x_prev, Hs, Ws, feat, proj, proj_bb):
            Hn, Wn = feat.shape[1:3]
            up = nearest_up_tokens(x_prev, Hs, Ws, (Hn, Wn))
            return proj(up) + proj_bb(feat).flatten(1, 2), Hn, Wn

        x, f2 = cast("class1", x, feats[2])
        x1, H1, W1 = stage(x, H, W, f2, self.proj_class1, self.proj_backbn1)
        dtok = ops.broadcast_rows(self.depth_token, B, H1 * W1, x1.dtype)
        stok = ops.broadcast_rows(self.seg_token, B, H1 * W1, x1.dtype)
        x1, dtok, stok = self.class_transformer1(x1, H1, W1, dtok=dtok, stok=stok)
        depth1 = sig_head(torch.cat([x1, dtok], dim=-1), self.depth_pred16).float().view(B, 1, H1, W1)
        md = cfg.min_depth_eval / cfg.max_depth_eval
        pts1 = certain_sample(depth0, depth1, cfg.depth_interval, cfg.interval_sample_num[0], md)
        if taps is not None:
            taps["points1"] = pts1
            pts1 = taps.get("force_points1", pts1)     # teacher forcing for parity tests (identical index operands)

        x1, dtok, stok, f1 = cast("class2", x1, dtok, stok, feats[1])
        x2, H2, W2 = stage(x1, H1, W1, f1, self.proj_class2, self.proj_backbn2)
        pos2 = pos_sine(masks[1], cfg.dense_trans_dim // 8, False)
        dtok = self.old_depth_token_proj8(nearest_up_tokens(dtok, H1, W1, (H2, W2)))
        stok = self.old_seg_token_proj8(nearest_up_tokens(stok, H1, W1, (H2, W2)))
        x2, dtok, stok = self.class_transformer2(x2, H2, W2, dtok=dtok, stok=stok)
        px, pd = cast("pbp1", x2, dtok)
        depth2 = self.point_based_pred1(px, pd, depth1, pts1, H2, W2, pos2)
        pts2 = certain_sample(depth1, depth2, cfg.depth_interval, cfg.interval_sample_num[1], md)
        if taps is not None:
            taps["points2"] = pts2
            pts2 = taps.get("force_points2", pts2)

        x2, dtok, stok, f0 = cast("class3", x2, dtok, stok, feats[0])
        x3, H3, W3 = stage(x2, H2, W2, f0, self.proj_class3, self.proj_backbn3)
        pos3 = pos_sine(masks[0], cfg.dense_trans_dim // 16, False)
        dtok = self.old_depth_token_proj4(nearest_up_tokens(dtok, H2, W2, (H3, W3)))
        stok = self.old_seg_token_proj4(nearest_up_tokens(stok, H2, W2, (H3, W3)))
        x3, dtok, stok = self.class_transformer3(x3, H3, W3, dtok=dtok, stok=stok)
        px, pd = cast("pbp2", x3, dtok)
        depth3 = self.point_based_pred2(px, pd, depth2, pts2, H3, W3, pos3)
        if taps is not None:
            taps["topk_ids"] = ids
            taps.update(dbg_x32=x, dbg_depth0=depth0, dbg_x1=x1, dbg_depth1=depth1, dbg_x2=x2, dbg_depth2=depth2, dbg_x3=x3,
                        dbg_pts=pts, dbg_pos=pos)
        as_map = lambda t: t.view(B, H3, W3, -1)
        return as_map(x3), as_map(dtok), as_map(stok), [depth1, depth2, depth3]


# ------------------------------------------------------------------------------------ full-res decoder
class UpConv(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = Conv(cin, cout, 3)

    def forward(self, x, size, defer=False, in_gate=ACT_NONE):
        """nearest upsample fused into the conv's gather, ELU in its epilogue (dense_upsample.py:82-90)."""
        return ops.conv2d(x, self.conv.weight, pad=1, act=ACT_ELU, upsample_to=size, defer=defer, in_gate=in_gate)


class DensePrediction(nn.Module):
    """DensePrediction.forward, /root/reference/src/models/dense_upsample.py:160-182."""

    def __init__(self, max_depth, tdim, feat=64):
        super().__init__()
        self.max_depth = max_depth
        self.depth_token_fuse = Mlp(feat + 1 + tdim, None, tdim)
        self.seg_token_fuse = Mlp(feat + tdim, None, tdim)
        for tag in ("depth", "seg"):                      # registration order of dense_upsample.py:131-158
            setattr(self, f"upconv1_{tag}", UpConv(tdim, tdim))
            setattr(self, f"norm_{tag}", LayerNorm(tdim))
            setattr(self, f"conv1_{tag}", Seq(_0=Conv(tdim, tdim, 3)))
            setattr(self, f"upconv2_{tag}", UpConv(tdim, tdim // 2))
            setattr(self, f"conv2_{tag}", Seq(_0=Conv(tdim // 2, tdim // 2, 3)))
            if tag == "depth":
                self.get_depth = Seq(_0=Conv(tdim // 2, 1, 3))
        self.get_seg = Conv(tdim // 2, 2, 3)

    def fuse_padded(self, x):
        """depth_token_fuse (129 -> 129 -> 64) with the odd 129-channel width zero-padded to 160: identical values (the
        extra input channels are zero, the extra hidden units get zero weights and bias, GELU(0) = 0), but the width is a
        multiple of 32 channels, so the three GEMMs and their gradients run on the LDS-DMA kernels (the 136-wide version
        sat on the register-staged odd-width path: 284 us for one data gradient)."""
        fc1, fc2 = self.depth_token_fuse.fc1, self.depth_token_fuse.fc2
        pad = x.shape[-1] - fc1.weight.shape[-1]            # forward() has already appended the zero channels (in its concat)
        if pad == 0:
            return self.depth_token_fuse(x)
        h = ops.linear(x, F.pad(fc1.weight, (0, pad, 0, pad)), F.pad(fc1.bias, (0, pad)), ACT_GELU)
        return ops.linear(h, F.pad(fc2.weight, (0, pad)), fc2.bias)

    _ZERO_BLOCKS = {}

    @classmethod
    def zero_channels(cls, like, n):
        """A constant (B,H,W,n) block of zeros (cached per shape): concatenated behind the fuse input it makes the width a multiple
        of 32 in the concat pass itself (was: concat, then F.pad = one more pass over the map, and its backward)."""
        key = (tuple(like.shape[:3]), n, like.dtype, str(like.device))
        z = cls._ZERO_BLOCKS.get(key)
        if z is None:
            z = cls._ZERO_BLOCKS[key] = torch.zeros(tuple(like.shape[:3]) + (n,), dtype=like.dtype, device=like.device)
        return z

    def branch(self, fuse_in, tag, fuse, size, cast):
        B, H, W, _ = fuse_in.shape
        f = fuse(cast("decoder_fuse", fuse_in))
        f = cast("decoder_up1", f)
        # the ELU backward of three of the four convolutions runs inside the backward of the layer behind each: the LayerNorm (upconv1),
        # the footprint sum of the up-sampling conv (conv1), the data-gradient epilogue of conv2 (upconv2)
        g = ACT_ELU if ops.act_gate_enabled() else ACT_NONE
        u1 = getattr(self, f"norm_{tag}")(getattr(self, f"upconv1_{tag}")(f, (2 * H, 2 * W), defer=g != ACT_NONE), in_gate=g)
        c1 = ops.conv2d(u1, getattr(self, f"conv1_{tag}")[0].weight, pad=1, act=ACT_ELU, defer=g != ACT_NONE)
        c1 = cast("decoder_up2", c1)
        u2 = getattr(self, f"upconv2_{tag}")(c1, size, defer=g != ACT_NONE, in_gate=g)
        # conv2 keeps its own pass: in the 1- / 2-channel heads' data gradient (one thread per pixel, 64 bytes of gate each) the gate
        # costs 90-140 us inside the step against the 87 us of the pass it would replace (tools/gatebench.py, per-kernel trace)
        return cast("decoder_head", ops.conv2d(u2, getattr(self, f"conv2_{tag}")[0].weight, pad=1, act=ACT_ELU, in_gate=g))

    def forward(self, feat, depth3, dtok, stok, size, cast=None):
        cast = cast or (lambda stage, t: t)
        B, H, W, _ = feat.shape
        d3 = depth3.view(B, H, W, 1).to(feat.dtype)
        parts = [feat, d3, dtok]
        pad = (-sum(p.shape[-1] for p in parts)) % 32
        if pad and feat.is_cuda:
            parts.append(self.zero_channels(feat, pad))
        d = self.branch(torch.cat(parts, dim=-1), "depth", self.fuse_padded, size, cast)
        depth = ops.conv2d(d, self.get_depth[0].weight, pad=1, act=ACT_SIGMOID, act_scale=float(self.max_depth))
        s = self.branch(torch.cat([feat, stok], dim=-1), "seg", self.seg_token_fuse, size, cast)
        seg = ops.conv2d(s, self.get_seg.weight, pad=1)
        return depth.float().view(B, 1, size[0], size[1]), seg.permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------------ assembly
class GlassRGBD(nn.Module):
    """GlassRGBD, /root/reference/src/models/glassrgbd.py:44-131 (with_line, with_center, with_dense)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.num_queries = cfg.num_queries
        self.transformer = Transformer(cfg.hidden_dim, cfg.nheads, cfg.enc_layers, cfg.dec_layers,
                                       cfg.dim_feedforward, cfg.dropout)
        self.class_embed = Linear(cfg.hidden_dim, 2)
        self.query_embed = nn.Embedding(cfg.num_queries, cfg.hidden_dim)
        self.input_proj = Conv(2048, cfg.hidden_dim, 1, bias=True)
        self.lines_embed = MLP(cfg.hidden_dim, cfg.hidden_dim, 6, 3)
        self.backbone = Joiner(cfg.hidden_dim, cfg.lr_backbone > 0)
        self.aux_loss = cfg.aux_loss
        self.dense_input_proj = Conv(2048, 512, 1, bias=True)
        self.dense_encoder = ReferTransformer(cfg)
        self.depth_decoder = DensePrediction(cfg.max_depth, cfg.class_token_dim)
        self.compute_dtype = torch.float32
        # stages kept in fp32 storage when compute_dtype is bf16 (names: backbone detr dense32 class1 pbp1 class2 pbp2 class3
        # decoder); the shipped set is chosen from measurements (tools/bf16_taps.py, DESIGN.md "precision policy")
        self.fp32_stages = set(DEFAULT_FP32_STAGES)

    def stage_cast(self, stage, *ts):
        """Storage precision of a stage's inputs: fp32 for the stages listed in fp32_stages, compute_dtype otherwise."""
        dt = torch.float32 if stage in self.fp32_stages else self.compute_dtype
        out = tuple(t if t.dtype == dt else t.to(dt) for t in ts)
        return out if len(out) > 1 else out[0]

    def decoder_cast(self, stage, t):
        """The decoder's sub-stages (decoder_fuse / _up1 / _up2 / _head); naming "decoder" keeps all four in fp32."""
        dt = torch.float32 if (stage in self.fp32_stages or "decoder" in self.fp32_stages) else self.compute_dtype
        return t if t.dtype == dt else t.to(dt)

    def forward(self, samples, reflc_points=None, reflc_mat=None, img_name=None, taps=None, match=None):
        if isinstance(samples, (list, torch.Tensor)):
            samples = nested_tensor_from_tensor_list(samples)
        images, pad_mask = samples.decompose()
        H, W = images.shape[-2:]
        cast = self.stage_cast
        x = cast("backbone", to_pixel_major(images))
        feats, masks = self.backbone(x, pad_mask)
        src, mask = feats[3], masks[3]
        pos = pos_sine(mask, self.cfg.hidden_dim // 2, True)
        src_d = cast("detr", src)
        hs = self.transformer(ops.conv2d(src_d, self.input_proj.weight, self.input_proj.bias), mask,
                              self.query_embed.weight, pos)
        logits = self.class_embed(hs).float()
        lines = torch.sigmoid(self.lines_embed(hs).float())
        out = {"pred_logits": logits[-1], "pred_lines": lines[-1]}
        if self.aux_loss:
            out["aux_outputs"] = [{"pred_logits": a, "pred_lines": b} for a, b in zip(logits[:-1], lines[:-1])]
        if match is not None:      # (matcher, targets): start the Hungarian hand-off now, consume it in the criterion
            matcher, targets = match
            out["_match_prefetch"] = matcher.prefetch([out] + out.get("aux_outputs", []), targets)
        dense_in = ops.conv2d(cast("dense32", src), self.dense_input_proj.weight, self.dense_input_proj.bias)
        if taps is not None:
            taps.update(dbg_feats=feats, dbg_dense_in=dense_in, dbg_src=src)
        feat4, dtok, stok, depths = self.dense_encoder(dense_in, feats, masks, out["pred_lines"], out["pred_logits"], taps, cast)
        if taps is not None:
            taps.update(dbg_feat4=feat4, dbg_dtok=dtok, dbg_stok=stok)
        depth, seg = self.depth_decoder(feat4, depths[-1], dtok, stok, (H, W), self.decoder_cast)
        out["pred_depth"] = depths + [depth]
        out["pred_seg"] = seg
        return out
