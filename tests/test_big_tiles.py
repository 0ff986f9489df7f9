"""Parity of the implicit-GEMM kernels AT THE TILE SHAPES THE STEP DISPATCHES TO.

csrc/igemm.hip picks its tiles from the problem size: `big = M >= 256*512` selects igemm_dma_kernel<256,160,8,1,3,*> and
<256,128,4,2,3,*>, the weight-gradient split count follows M, and tests/test_hip_kernels.py only has M <= ~2 000.  Every case
here is a layer of the real step at batch 8, 480x640 (M = 153 600 ... 2 457 600 output pixels), forward + data gradient +
weight gradient through the C ABI, against plain fp32 torch.nn.functional.conv2d arithmetic on the CPU (NOT against this
library's own fp32 kernels).  Inputs are bf16-rounded, so the only differences are the bf16 rounding of the outputs
(relative rms 2^-8/sqrt(12) = 1.1e-3) and fp32 summation order.

Tolerances: bf16 outputs 3e-3 relative L2 and 2 % of the largest reference magnitude element-wise (a misplaced tile or a
dropped tail row is 100 % off); fp32 weight gradients 3e-4 relative L2.
"""
import pytest
import torch
import torch.nn.functional as F

from gw_depth_amd import hip

pytestmark = pytest.mark.gpu
TOL_BF16, TOL_WGRAD, TOL_F32 = 3e-3, 3e-4, 3e-5


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    hip.set_library(None)
    torch.set_num_threads(16)
    return hip.library()


def rnd(*shape, seed, scale=1.0, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def nchw(t):
    return t.float().permute(0, 3, 1, 2).contiguous()


def close(got, want, tol, what):
    got, want = got.float().cpu(), want.float()
    err = float((got.double() - want.double()).norm() / (want.double().norm() + 1e-30))
    assert err < tol, "%s: relative L2 error %.3e >= %.1e" % (what, err, tol)
    worst = float((got - want).abs().max() / (want.abs().max() + 1e-30))
    assert worst < (0.02 if tol >= 1e-3 else 50 * tol), "%s: worst element off by %.3e of the largest magnitude" % (what, worst)


def act_ref(v, act, act_scale):
    if act == hip.ACT_RELU:
        return torch.relu(v)
    if act == hip.ACT_GELU:
        return F.gelu(v)
    if act == hip.ACT_ELU:
        return F.elu(v)
    if act == hip.ACT_SIGMOID:
        return torch.sigmoid(v) * act_scale
    return v


# name, (B, Hi, Wi), Cin, Cout, K, stride, pad, upsample_to, extras, tiles the dispatcher picks (documentation)
CASES = [
    ("layer4_conv2_512_512_3x3_s2", (8, 30, 40), 512, 512, 3, 2, 1, None, dict(shift=True, act=hip.ACT_RELU),
     "ResNet layer4 block 0: data gradient by parity class on 64 x 64 tiles (2 400 pixels per class: ragged last tile)"),
    ("layer3_conv2_256_256_3x3_s2", (8, 60, 80), 256, 256, 3, 2, 1, None, dict(shift=True, act=hip.ACT_RELU),
     "ResNet layer3 block 0: data gradient by parity class on 128 x 128 tiles"),
    ("odd_map_3x3_s2", (2, 31, 41), 128, 128, 3, 2, 1, None, {},
     "odd map: not a whole number of parity classes - the general transposed gather stays"),
    ("pooled_64_64_3x3_60x80", (8, 60, 80), 64, 64, 3, 1, 1, None, {},
     "N = 64 on a 1/8 map: 300 tiles of 128 rows would be a round and a tail - igemm_dma<64,64,..,BK 64> (600 workgroups, one round); halo variant not eligible"),
    ("pyramid2_layer_160_160_3x3", (8, 120, 160), 160, 160, 3, 1, 1, None, {},
     "fwd igemm_dma<256,160,8,1,3,0,...,HALO> (8 x 32 pixel patches, halo staged once per channel block), dgrad <..,1,...,HALO>, wgrad_taps_kernel: THE roofline kernel of bench.py"),
    ("persist_ragged_tail_605_tiles", (8, 121, 160), 160, 160, 3, 1, 1, None, {},
     "H = 121 is not a whole number of 8-row patches: the plain igemm_dma<256,160> (one tile per workgroup, ragged last tile), not the halo variant"),
    ("persist_short_tail_525_tiles", (7, 120, 160), 160, 160, 3, 1, 1, None, {},
     "B = 7: halo variant on 525 patches (one round of 512 + 13)"),
    ("pyramid2_lastconv_800_320_3x3", (8, 120, 160), 800, 320, 3, 1, 1, None, {},
     "fwd halo variant, two column tiles, 25 channel blocks x 9 taps; dgrad 320->800 halo variant (3000 tiles); wgrad_taps_kernel 50 tiles x 5 splits"),
    ("pyramid2_firstconv_80_160_3x3", (8, 120, 160), 80, 160, 3, 1, 1, None, {},
     "Cin = 80: igemm_dma<256,160,8,1,3,0> with the channel tail (was the register-staged kernel); dgrad 160->80 <256,128,4,2,3,1>; wgrad_dma<160,128>"),
    ("pyramid2_firstconv_80_80_3x3", (8, 120, 160), 80, 80, 3, 1, 1, None, {},
     "Cin = 80: igemm_dma<256,128,4,2,3,*> with the zero-page channel tail (27 K tiles: 9 taps x (32 + 32 + 16|zero)), forward and data gradient"),
    ("pyramid2_lastconv_320_80_1x1", (8, 120, 160), 320, 80, 1, 1, 0, None, {},
     "fwd <256,128,4,2,3,0> with 80 of 128 columns; dgrad 80->320 register-staged; wgrad <128,128> plain GEMM"),
    ("class3_mlp_fc1_64_128_gelu", (8, 120, 160), 64, 128, 1, 1, 0, None, dict(shift=True, act=hip.ACT_GELU, z=True),
     "fwd <256,128,4,2,3,0> + GELU epilogue with pre-activation copy; dgrad 128->64 <128,64,2,2,4,1>; wgrad <64,64> plain"),
    ("class3_qkv_64_192_ragged", (1, 414 * 8, 49), 64, 192, 1, 1, 0, None, dict(shift=True),
     "M = 162 288 (not a multiple of 256): <256,128> with a half-empty second column tile and a ragged last row tile"),
    ("layer1_conv3_64_256_1x1_bn_res_relu", (8, 120, 160), 64, 256, 1, 1, 0, None, dict(shift=True, scale=True, residual=True, act=hip.ACT_RELU),
     "backbone Bottleneck conv3: folded FrozenBN scale/shift + residual + ReLU in the <256,128> epilogue"),
    ("layer1_conv2_64_64_3x3_bn_relu", (8, 120, 160), 64, 64, 3, 1, 1, None, dict(shift=True, scale=True, act=hip.ACT_RELU),
     "fwd <128,64,2,2,4,0> at M = 153 600; wgrad <64,64,2,2,4,1>"),
    ("layer2_conv2_128_128_3x3_s2", (8, 120, 160), 128, 128, 3, 2, 1, None, dict(shift=True, act=hip.ACT_RELU),
     "stride 2: forward gather 0, data gradient through the general gather (mode 2) at M = 153 600 input pixels"),
    ("decoder_upconv1_64_64_up2_elu", (8, 120, 160), 64, 64, 3, 1, 1, (240, 320), dict(act=hip.ACT_ELU),
     "nearest-upsample fused into the gather (mode 2), M = 614 400; dgrad at the virtual size + footprint sum"),
    ("decoder_upconv2_64_32_up2_elu", (4, 240, 320), 64, 32, 3, 1, 1, (480, 640), dict(act=hip.ACT_ELU),
     "fwd <128,32,4,1,4,2>, M = 1 228 800; wgrad <32,128,1,4,4,0>"),
    ("decoder_conv2_32_32_3x3_elu", (4, 480, 640), 32, 32, 3, 1, 1, None, dict(act=hip.ACT_ELU),
     "fwd <128,32,4,1,4,0>, dgrad <..,1>, wgrad <32,128,1,4,4,1> at full resolution"),
    ("refpoint_diffusion_16_16_3x3_residual", (8, 441, 40), 16, 16, 3, 1, 1, None, dict(residual=True),
     "the same layer's data gradient with the skip gradient added in the tile kernel's epilogue (ops._ConvFn fan-out)"),
    ("decoder_conv2_32_32_3x3_residual_elu", (2, 480, 640), 32, 32, 3, 1, 1, None, dict(shift=True, residual=True, act=hip.ACT_ELU),
     "tconv_fwd<32,32> with shift + residual + ELU in the epilogue"),
    ("refpoint_diffusion_16_16_3x3_bias", (8, 441, 40), 16, 16, 3, 1, 1, None, dict(shift=True),
     "the 16-head score map of the 1/32 stage (multiscale_transformerr.py:299-302), 40 columns = a ragged second tile: halo-tiled kernels "
     "tconv_fwd<16,16> (forward, data gradient) and tconv_wgrad<16,16>, half-empty channel tiles"),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_dispatch_size_conv_against_fp32_torch(dev, case):
    name, (B, Hi, Wi), Cin, Cout, K, s, p, virt, ex, _ = case
    x = rnd(B, Hi, Wi, Cin, seed=11)
    w = rnd(Cout, K, K, Cin, seed=12, scale=(K * K * Cin) ** -0.5)
    scale = (torch.rand(Cout, generator=torch.Generator().manual_seed(13)) + 0.5) if ex.get("scale") else None
    shift = rnd(Cout, seed=14, dtype=torch.float32) if ex.get("shift") else None
    act, act_scale = ex.get("act", hip.ACT_NONE), ex.get("act_scale", 1.0)
    if virt is None:
        Ho, Wo = (Hi + 2 * p - K) // s + 1, (Wi + 2 * p - K) // s + 1
        gather, vv = hip.GATHER_CONV, (0, 0)
    else:
        Ho, Wo = virt
        gather, vv = hip.GATHER_UPSAMPLED, virt
    res = rnd(B, Ho, Wo, Cout, seed=15) if ex.get("residual") else None
    dims = (B, Hi, Wi, Cin, Ho, Wo, Cout, K, K)

    # ---- fp32 torch reference (the reference's own arithmetic: nn.Conv2d / F.interpolate(nearest) / FrozenBN affine)
    wk = w.float() if scale is None else (w.float() * scale[:, None, None, None]).to(torch.bfloat16).float()   # gwd_weight_prep folds, then rounds
    xin = nchw(x)
    if virt is not None:
        xin = F.interpolate(xin, size=virt, mode="nearest")
    pre = F.conv2d(xin, wk.permute(0, 3, 1, 2), None, stride=s, padding=p)
    if shift is not None:
        pre = pre + shift[None, :, None, None]
    if res is not None:
        pre = pre + nchw(res)
    y_ref = act_ref(pre, act, act_scale).permute(0, 2, 3, 1)

    cu = lambda t: None if t is None else t.cuda()
    wdev = torch.empty(Cout, K, K, Cin, dtype=torch.bfloat16, device="cuda")
    wt = torch.empty(Cin, K, K, Cout, dtype=torch.bfloat16, device="cuda")
    dev.weight_prep(w.float().cuda(), cu(scale), wdev, wt, Cout, K * K, Cin, hip.BF16)
    y = torch.full((B, Ho, Wo, Cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    z = torch.full_like(y, float("nan")) if ex.get("z") else None
    dev.conv_forward(x.cuda(), wdev, y, dims, z=z, shift=cu(shift), residual=cu(res), stride=s, pad=p, gather=gather, virt=vv,
                     act=act, act_scale=act_scale)
    torch.cuda.synchronize()
    close(y, y_ref, TOL_BF16, name + " forward")
    if z is not None:
        close(z, pre.permute(0, 2, 3, 1), TOL_BF16, name + " pre-activation copy")
    del y, z, pre

    # ---- data gradient: conv_transpose of the reference == the transposed-gather launch ops._ConvFn.backward makes
    gy = rnd(B, Ho, Wo, Cout, seed=16)
    Hd, Wd = (Hi, Wi) if virt is None else virt
    gx_ref = torch.nn.grad.conv2d_input((B, Cin, Hd, Wd), wk.permute(0, 3, 1, 2), nchw(gy), stride=s, padding=p).permute(0, 2, 3, 1)
    gx = torch.full((B, Hd, Wd, Cin), float("nan"), dtype=torch.bfloat16, device="cuda")
    dev.conv_forward(gy.cuda(), wt, gx, (B, Ho, Wo, Cout, Hd, Wd, Cin, K, K), stride=s, pad=p, gather=hip.GATHER_TRANSPOSED)
    torch.cuda.synchronize()
    close(gx, gx_ref, TOL_BF16, name + " data gradient")
    if s == 2 and K == 3 and virt is None:
        # the same launch with the producer's ReLU backward as the epilogue gate (what a Bottleneck's data gradient carries): on even maps
        # this is the parity-class kernel (dma_tile<..., PAR>), each row stored at its own pixel
        prod = rnd(B, Hd, Wd, Cin, seed=17)
        gxg = torch.full((B, Hd, Wd, Cin), float("nan"), dtype=torch.bfloat16, device="cuda")
        dev.conv_forward(gy.cuda(), wt, gxg, (B, Ho, Wo, Cout, Hd, Wd, Cin, K, K), stride=s, pad=p, gather=hip.GATHER_TRANSPOSED,
                         gate=prod.cuda(), gate_act=hip.ACT_RELU)
        torch.cuda.synchronize()
        close(gxg, gx_ref * (prod.float() > 0), TOL_BF16, name + " gated data gradient")
        del gxg
    del gx, gx_ref

    # ---- weight gradient (fp32 atomics into a zeroed buffer, real split count for this M; the FrozenBN scale multiplies it)
    dw_ref = torch.nn.grad.conv2d_weight(xin, (Cout, Cin, K, K), nchw(gy), stride=s, padding=p).permute(0, 2, 3, 1)
    if scale is not None:
        dw_ref = dw_ref * scale[:, None, None, None]
    dw = torch.zeros(Cout, K, K, Cin, device="cuda")
    dev.conv_wgrad(x.cuda(), gy.cuda(), dw, dims, stride=s, pad=p, gather=gather, virt=vv, scale=cu(scale))
    torch.cuda.synchronize()
    close(dw, dw_ref, TOL_WGRAD, name + " weight gradient")


def test_grouped_wgrad_launch_at_dispatch_sizes(dev):
    """gwd_conv_wgrad_batch: the plain-GEMM weight gradients of the step travel 32 to a call and run as grouped launches
    (igemm_wgrad_group_kernel<128,128> / <64,64>); here four real token-GEMM shapes of the 1/4-resolution stage in one call."""
    shapes = [(153600, 64, 128), (153600, 128, 64), (162288, 64, 192), (153600, 320, 80)]
    jobs, refs = [], []
    for i, (M, Kc, N) in enumerate(shapes):
        x, gy = rnd(M, 1, 1, Kc, seed=30 + i), rnd(M, 1, 1, N, seed=40 + i)
        dw = torch.zeros(N, 1, 1, Kc, device="cuda")
        jobs.append((x.cuda(), gy.cuda(), dw, (M, 1, 1, Kc, 1, 1, N, 1, 1), {}))
        refs.append(gy.float().view(M, N).t() @ x.float().view(M, Kc))
    dev.conv_wgrad_batch(jobs)
    torch.cuda.synchronize()
    for (M, Kc, N), job, ref in zip(shapes, jobs, refs):
        close(job[2].view(N, Kc), ref, TOL_WGRAD, "grouped wgrad %dx%dx%d" % (M, Kc, N))


def test_dominant_kernel_in_exact_fp32_mode(dev):
    """The fp32 parity mode at the same dispatch size (igemm_fwd_kernel<float,128,160>, v_mfma_f32_32x32x2_f32)."""
    B, H, W, C = 8, 120, 160, 160
    x, w = rnd(B, H, W, C, seed=21, dtype=torch.float32), rnd(C, 3, 3, C, seed=22, scale=(9 * C) ** -0.5, dtype=torch.float32)
    y_ref = F.conv2d(nchw(x), w.permute(0, 3, 1, 2), None, padding=1).permute(0, 2, 3, 1)
    y = torch.full((B, H, W, C), float("nan"), device="cuda")
    dev.conv_forward(x.cuda(), w.cuda(), y, (B, H, W, C, H, W, C, 3, 3), stride=1, pad=1)
    torch.cuda.synchronize()
    close(y, y_ref, TOL_F32, "fp32 160->160 forward")


# ---------------------------------------------------------------------------------------------------------------------------------
# ConvLn: convolution -> LayerNorm over channels [-> GELU] [+ skip] in ONE launch (gwd_conv_desc.ln_*, dma_tile<..., LN>), against
# fp32 torch arithmetic (F.conv2d + F.layer_norm + F.gelu: what points_sample.py:12-43 runs), at the sizes the step dispatches.
LN_CASES = [
    # name, (B, H, W), Cin, Cout (row pitch), C (real channels), gelu, residual, kernel
    ("pyr2_160_160_gelu", (8, 120, 160), 160, 160, 160, True, False, "<256,160,8,1,3,0,..,2,LN>: PyrBlock conv1 at 1/4 resolution"),
    ("pyr2_160_160_skip", (8, 120, 160), 160, 160, 160, False, True, "<256,160,..,0,LN> + residual after the normalisation: PyrBlock conv2"),
    ("pyr2_80_160_gelu_tail", (8, 120, 160), 80, 160, 160, True, False, "<256,160,..,TAIL,2,LN>: firstconv[2], Cin = 80"),
    ("pyr2_ragged_rows", (9, 119, 161), 160, 160, 160, True, True, "M = 172 431: <256,160,8,1,3> with a ragged last tile, GELU and skip together"),
    ("pyr1_64_64_padded_60", (8, 60, 80), 64, 64, 60, True, False, "<128,64,4,1,4,0,..,LN>: 60 real channels in 64-wide rows, padding written as zeros"),
    ("pyr1_64_64_padded_skip", (8, 60, 80), 64, 64, 60, False, True, "the same with the skip"),
    ("pyr1_32_32_padded_30", (8, 60, 80), 32, 32, 30, True, False, "<128,32,4,1,4,0,..,LN>: 30 real channels"),
]


@pytest.mark.parametrize("case", LN_CASES, ids=[c[0] for c in LN_CASES])
def test_fused_conv_layernorm_epilogue_against_fp32_torch(dev, case):
    name, (B, H, W), Cin, Np, C, gelu, with_res, _ = case
    x = rnd(B, H, W, Cin, seed=51)
    w = rnd(Np, 3, 3, Cin, seed=52, scale=(9 * Cin) ** -0.5)
    if C < Np:
        w[C:] = 0                                     # zero-padded rows of the kernel-side weight copy (ops._PadConvFn)
    gamma = (torch.rand(C, generator=torch.Generator().manual_seed(53)) + 0.5)
    beta = rnd(C, seed=54, dtype=torch.float32) * 0.3
    res = rnd(B, H, W, Np, seed=55) if with_res else None
    if res is not None and C < Np:
        res[..., C:] = 0
    conv = F.conv2d(nchw(x), w.float().permute(0, 3, 1, 2), None, padding=1).permute(0, 2, 3, 1)          # (B,H,W,Np) fp32
    ln = F.layer_norm(conv[..., :C], (C,), gamma, beta, 1e-5)
    if gelu:
        ln = F.gelu(ln)
    y_ref = torch.zeros(B, H, W, Np)
    y_ref[..., :C] = ln
    if res is not None:
        y_ref = y_ref + res.float()
    mean_ref = conv[..., :C].mean(-1).reshape(-1)
    rstd_ref = (conv[..., :C].var(-1, unbiased=False) + 1e-5).rsqrt().reshape(-1)

    rows = B * H * W
    y = torch.full((B, H, W, Np), float("nan"), dtype=torch.bfloat16, device="cuda")
    z = torch.full_like(y, float("nan"))
    mean = torch.full((rows,), float("nan"), device="cuda")
    rstd = torch.full((rows,), float("nan"), device="cuda")
    ok = dev.conv_forward(x.cuda(), w.cuda(), y, (B, H, W, Cin, H, W, Np, 3, 3), z=z, scale=gamma.cuda(), shift=beta.cuda(),
                          residual=None if res is None else res.cuda(), stride=1, pad=1, act=hip.ACT_GELU if gelu else hip.ACT_NONE,
                          ln=(mean, rstd, C))
    assert ok is not False, name + ": the library has no fused kernel for a shape the step uses"
    torch.cuda.synchronize()
    close(y, y_ref, TOL_BF16, name + " output")
    close(z, conv, TOL_BF16, name + " convolution copy")
    close(mean, mean_ref, 2e-4, name + " row means")
    close(rstd, rstd_ref, 2e-4, name + " row rstd")
    if C < Np:
        assert float(y[..., C:].float().abs().max()) == 0.0, name + ": padding channels must come out as zeros"


def test_conv_ln_autograd_node_equals_the_two_kernel_path(dev):
    """ops.conv_ln (one forward launch) against ops.conv2d + ops.layer_norm (two): same outputs to bf16 rounding of the intermediate,
    same gradients for input, weight, gamma, beta and the skip - forward values differ only because the fused epilogue normalises the
    fp32 accumulators while the pair normalises their bf16 copy."""
    from gw_depth_amd import ops
    B, H, W, Cc = 2, 60, 80, 160
    x0 = rnd(B, H, W, Cc, seed=61).cuda()
    w0 = rnd(Cc, 3, 3, Cc, seed=62, scale=(9 * Cc) ** -0.5, dtype=torch.float32).cuda()
    g0 = (torch.rand(Cc, generator=torch.Generator().manual_seed(63)) + 0.5).cuda()
    b0 = (rnd(Cc, seed=64, dtype=torch.float32) * 0.2).cuda()
    r0 = rnd(B, H, W, Cc, seed=65).cuda()
    gy = rnd(B, H, W, Cc, seed=66).cuda()
    out = []
    for fused in (True, False):
        x, r = x0.clone().requires_grad_(True), r0.clone().requires_grad_(True)
        w, g, b = (t.clone().requires_grad_(True) for t in (w0, g0, b0))
        for gelu, res in ((True, None), (False, r)):
            if fused:
                y = ops.conv_ln(x if gelu else h, w, g, b, 1, gelu=gelu, residual=res)
            else:
                y = ops.layer_norm(ops.conv2d(x if gelu else h, w, pad=1), g, b, gelu, residual=res)
            h = y
        with ops.COLSUMS, ops.WGRADS:
            y.backward(gy)
        torch.cuda.synchronize()
        out.append((y.detach(), x.grad, w.grad, g.grad, b.grad, r.grad))
    for name, a, c in zip(("output", "d input", "d weight", "d gamma", "d beta", "d skip"), out[0], out[1]):
        close(a, c.float().cpu(), 1e-2, "conv_ln vs conv2d + layer_norm: " + name)
