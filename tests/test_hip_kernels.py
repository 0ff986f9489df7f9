"""GPU parity of every C-ABI entry point against plain fp32 PyTorch math (tests/fake_device.py runs
the same tensor-level calls on the CPU).  fp32 kernels use the exact-fp32 MFMA: tolerance 2e-5
relative L2; bf16 kernels: 1.5e-2 (bf16 has 8 significant bits; accumulation is fp32)."""
import pytest
import torch

from gw_depth_amd import hip
from tests.fake_device import FakeDevice

pytestmark = pytest.mark.gpu
TOL = {torch.float32: 2e-5, torch.bfloat16: 1.5e-2}


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    hip.set_library(None)
    return hip.library()


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-20))


def rnd(*shape, dtype=torch.float32, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


CONV_CASES = [
    # name, B, Hi, Wi, Cin, Cout, K, stride, pad, virt, extras
    ("3x3_64_64", 2, 20, 24, 64, 64, 3, 1, 1, None, dict(shift=True, act=hip.ACT_RELU)),
    ("1x1_256_64_res", 2, 12, 10, 256, 64, 1, 1, 0, None, dict(shift=True, residual=True, act=hip.ACT_RELU)),
    ("3x3_s2_128_160", 2, 17, 21, 128, 160, 3, 2, 1, None, dict(act=hip.ACT_GELU, z=True, shift=True)),
    ("1x1_s2_64_256", 1, 16, 16, 64, 256, 1, 2, 0, None, dict()),
    ("7x7_s2_rgb", 2, 32, 40, 3, 64, 7, 2, 3, None, dict(shift=True, act=hip.ACT_RELU)),
    ("3x3_30_60_odd", 2, 12, 16, 30, 60, 3, 1, 1, None, dict()),
    ("3x3_300_120_odd", 1, 12, 16, 300, 120, 3, 1, 1, None, dict()),
    ("3x3_60_60_half", 2, 13, 17, 60, 60, 3, 1, 1, None, dict()),
    ("3x3_120_30_half", 1, 12, 16, 120, 30, 3, 1, 1, None, dict()),
    ("up2_64_64_elu", 2, 10, 12, 64, 64, 3, 1, 1, (20, 24), dict(act=hip.ACT_ELU)),
    ("up_size_64_32", 1, 9, 11, 64, 32, 3, 1, 1, (24, 32), dict(act=hip.ACT_ELU)),
    ("3x3_32_1_sig10", 2, 24, 32, 32, 1, 3, 1, 1, None, dict(act=hip.ACT_SIGMOID, act_scale=10.0)),
    ("3x3_32_2", 2, 24, 32, 32, 2, 3, 1, 1, None, dict()),
    ("3x3_32_2_ragged", 3, 37, 45, 32, 2, 3, 1, 1, None, dict(shift=True)),
    ("3x3_32_1_ragged", 1, 50, 19, 32, 1, 3, 1, 1, None, dict(act=hip.ACT_SIGMOID, act_scale=10.0)),
    ("3x3_800_320", 1, 12, 16, 800, 320, 3, 1, 1, None, dict()),
    ("lin_300x256_768", 300, 1, 1, 256, 768, 1, 1, 0, None, dict(shift=True)),
    ("lin_800x2048_256", 800, 1, 1, 2048, 256, 1, 1, 0, None, dict(shift=True)),
    ("lin_1176x64_128_gelu", 1176, 1, 1, 64, 128, 1, 1, 0, None, dict(shift=True, act=hip.ACT_GELU, z=True)),
    ("lin_5x384_384", 5, 1, 1, 384, 384, 1, 1, 0, None, dict()),
    # long reductions on few pixels: gemm_ksplit_kernel with a gather (ResNet layer4, the 1/32 and 1/64 pyramid levels)
    ("3x3_512_512_l4", 8, 15, 20, 512, 512, 3, 1, 1, None, dict(shift=True, act=hip.ACT_RELU)),
    ("3x3_s2_512_512_l4", 2, 30, 40, 512, 512, 3, 2, 1, None, dict(shift=True, act=hip.ACT_RELU)),
    ("3x3_160_160_p32", 8, 15, 20, 160, 160, 3, 1, 1, None, dict(shift=True)),
    ("3x3_160_160_p64_ragged", 3, 7, 10, 160, 160, 3, 1, 1, None, dict(residual=True)),
    ("3x3_128_72_ragged_n", 1, 9, 11, 128, 72, 3, 1, 1, None, dict(act=hip.ACT_GELU, z=True, shift=True)),
]


def conv_out(Hi, K, s, p):
    return (Hi + 2 * p - K) // s + 1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_forward_dgrad_wgrad(dev, case, dtype):
    name, B, Hi, Wi, Cin, Cout, K, s, p, virt, ex = case
    fake = FakeDevice()
    if virt is None:
        Ho, Wo = conv_out(Hi, K, s, p), conv_out(Wi, K, s, p)
        gather, vv = hip.GATHER_CONV, (0, 0)
    else:
        Ho, Wo = virt
        gather, vv = hip.GATHER_UPSAMPLED, virt
    dims = (B, Hi, Wi, Cin, Ho, Wo, Cout, K, K)
    x = rnd(B, Hi, Wi, Cin, dtype=dtype, seed=1)
    w = rnd(Cout, K, K, Cin, dtype=dtype, seed=2, scale=(K * K * Cin) ** -0.5)
    shift = rnd(Cout, seed=3) if ex.get("shift") else None
    res = rnd(B, Ho, Wo, Cout, dtype=dtype, seed=4) if ex.get("residual") else None
    act, act_scale = ex.get("act", hip.ACT_NONE), ex.get("act_scale", 1.0)
    kw = dict(shift=shift, residual=res, stride=s, pad=p, gather=gather, virt=vv, act=act, act_scale=act_scale)

    y_ref = torch.empty(B, Ho, Wo, Cout, dtype=dtype)
    z_ref = torch.empty_like(y_ref) if ex.get("z") else None
    fake.conv_forward(x, w, y_ref, dims, z=z_ref, **kw)

    cu = lambda t: None if t is None else t.cuda()
    y = torch.full((B, Ho, Wo, Cout), float("nan"), dtype=dtype, device="cuda")
    z = torch.empty_like(y) if ex.get("z") else None
    kwc = dict(kw, shift=cu(shift), residual=cu(res))
    dev.conv_forward(x.cuda(), w.cuda(), y, dims, z=z, **kwc)
    torch.cuda.synchronize()
    assert rel(y, y_ref) < TOL[dtype], "forward"
    if z is not None:
        assert rel(z, z_ref) < TOL[dtype], "pre-activation copy"

    # data gradient through the transposed gather (plain and strided); upsampled handled at virtual size
    gy = rnd(B, Ho, Wo, Cout, dtype=dtype, seed=5)
    wt_ref = torch.empty(Cin, K, K, Cout, dtype=dtype)
    fake.weight_prep(w.float(), None, None, wt_ref, Cout, K * K, Cin, 0)
    wt = torch.empty(Cin, K, K, Cout, dtype=dtype, device="cuda")
    dev.weight_prep(w.float().cuda(), None, None, wt, Cout, K * K, Cin, hip.dtype_code(wt))
    assert rel(wt, wt_ref) < 1e-6
    Hd, Wd = (Hi, Wi) if virt is None else virt
    ddims = (B, Ho, Wo, Cout, Hd, Wd, Cin, K, K)
    gx_ref = torch.empty(B, Hd, Wd, Cin, dtype=dtype)
    fake.conv_forward(gy, wt_ref, gx_ref, ddims, stride=s, pad=p, gather=hip.GATHER_TRANSPOSED)
    gx = torch.full((B, Hd, Wd, Cin), float("nan"), dtype=dtype, device="cuda")
    dev.conv_forward(gy.cuda(), wt, gx, ddims, stride=s, pad=p, gather=hip.GATHER_TRANSPOSED)
    torch.cuda.synchronize()
    assert rel(gx, gx_ref) < TOL[dtype], "dgrad"

    # weight gradient (fp32 accumulate, atomics)
    dw_ref = torch.zeros(Cout, K, K, Cin)
    fake.conv_wgrad(x, gy, dw_ref, dims, stride=s, pad=p, gather=gather, virt=vv)
    dw = torch.zeros(Cout, K, K, Cin, device="cuda")
    dev.conv_wgrad(x.cuda(), gy.cuda(), dw, dims, stride=s, pad=p, gather=gather, virt=vv)
    torch.cuda.synchronize()
    assert rel(dw, dw_ref) < TOL[dtype], "wgrad"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2400, 256, 256, hip.ACT_NONE, True), (2400, 256, 2048, hip.ACT_RELU, False), (800, 2048, 256, hip.ACT_NONE, True),
                                   (37, 40, 24, hip.ACT_RELU, True), (15360, 64, 128, hip.ACT_NONE, True)])
def test_linear_epilogue_dropout_multiplier_and_skip(dev, shape, dtype):
    """y = act(x W^T + b) * mult + residual in the GEMM epilogue (x + dropout(sublayer(x)), transformer.py:149-162) on every
    forward kernel (LDS-DMA 64x64 / 256x128 tiles, the register-staged fp32 / odd-width kernels, vector and scalar epilogues)."""
    M, K, N, act, with_res = shape
    fake = FakeDevice()
    x, w = rnd(M, 1, 1, K, dtype=dtype, seed=1), rnd(N, 1, 1, K, dtype=dtype, seed=2, scale=K ** -0.5)
    b = rnd(N, seed=3)
    g = torch.Generator().manual_seed(4)
    mult = ((torch.rand(M, 1, 1, N, generator=g) > 0.1).float() / 0.9).to(dtype)
    res = rnd(M, 1, 1, N, dtype=dtype, seed=5) if with_res else None
    dims = (M, 1, 1, K, 1, 1, N, 1, 1)
    y_r = torch.empty(M, 1, 1, N, dtype=dtype)
    fake.conv_forward(x, w, y_r, dims, shift=b, residual=res, act=act, mult=mult)
    y = torch.full((M, 1, 1, N), float("nan"), dtype=dtype, device="cuda")
    dev.conv_forward(x.cuda(), w.cuda(), y, dims, shift=b.cuda(), residual=None if res is None else res.cuda(), act=act, mult=mult.cuda())
    torch.cuda.synchronize()
    assert rel(y, y_r) < TOL[dtype]


@pytest.mark.parametrize("act", [hip.ACT_NONE, hip.ACT_RELU])
def test_linear_with_dropout_multiplier_autograd(dev, act):
    """ops.linear(..., residual=r, mult=m) forward AND backward (x, weight, bias, skip) against torch autograd of
    act(x W^T + b) * m + r, fp32 (exact-fp32 MFMA)."""
    from gw_depth_amd import ops
    M, K, N = 600, 256, 512
    x, w, b, r = (rnd(*sh, seed=i).cuda().requires_grad_(True) for i, sh in enumerate([(2, M // 2, K), (N, K), (N,), (2, M // 2, N)], start=1))
    g = torch.Generator().manual_seed(9)
    m = ((torch.rand(2, M // 2, N, generator=g) > 0.1).float() / 0.9).cuda()
    go = rnd(2, M // 2, N, seed=7).cuda()
    skip = act == hip.ACT_NONE                     # the product's two uses: out-projection / linear2 (+ skip), linear1 + ReLU (no skip)
    y = ops.linear(x, w, b, act, residual=r if skip else None, mult=m) + (0 if skip else r)
    gx, gw, gb, gr = torch.autograd.grad(y, [x, w, b, r], go)
    x2, w2, b2, r2 = (t.detach().clone().requires_grad_(True) for t in (x, w, b, r))
    pre = x2 @ w2.t() + b2
    y2 = (torch.relu(pre) if act == hip.ACT_RELU else pre) * m + r2
    ex, ew, eb, er = torch.autograd.grad(y2, [x2, w2, b2, r2], go)
    torch.cuda.synchronize()
    for got, want in ((y, y2), (gx, ex), (gw, ew), (gb, eb), (gr, er)):
        assert rel(got, want) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,C,gelu,affine", [(300, 256, False, True), (1000, 64, False, True), (77, 160, True, True),
                                                (513, 30, True, True), (64, 512, False, True), (40, 320, False, False),
                                                (777, 60, True, True), (130, 300, True, True), (65, 120, False, True), (33, 6, False, True)])
def test_layernorm(dev, rows, C, gelu, affine, dtype):
    fake = FakeDevice()
    x = rnd(rows, C, dtype=dtype, seed=1, scale=2.0)
    ga = (1 + 0.1 * rnd(C, seed=2)) if affine else None
    be = 0.1 * rnd(C, seed=3) if affine else None
    gy = rnd(rows, C, dtype=dtype, seed=4)
    y_r, m_r, r_r = torch.empty_like(x), torch.empty(rows), torch.empty(rows)
    fake.layernorm_forward(x, ga, be, y_r, m_r, r_r, rows, C, gelu)
    gx_r, dg_r, db_r = torch.empty_like(x), torch.zeros(C), torch.zeros(C)
    fake.layernorm_backward(gy, x, ga, be, m_r, r_r, gx_r, dg_r if affine else None, db_r if affine else None, rows, C, gelu)
    cu = lambda t: None if t is None else t.cuda()
    y, m, r = torch.empty_like(x).cuda(), torch.empty(rows).cuda(), torch.empty(rows).cuda()
    dev.layernorm_forward(x.cuda(), cu(ga), cu(be), y, m, r, rows, C, gelu)
    gx, dg, db = torch.empty_like(x).cuda(), torch.zeros(C).cuda(), torch.zeros(C).cuda()
    dev.layernorm_backward(gy.cuda(), x.cuda(), cu(ga), cu(be), m, r, gx, dg if affine else None, db if affine else None,
                           rows, C, gelu)
    torch.cuda.synchronize()
    assert rel(y, y_r) < TOL[dtype] and rel(m, m_r) < 1e-5 and rel(r, r_r) < 1e-5
    assert rel(gx, gx_r) < TOL[dtype]
    if affine:
        assert rel(dg, dg_r) < TOL[dtype] and rel(db, db_r) < TOL[dtype]
    gsk = rnd(rows, C, dtype=dtype, seed=6)                      # second gradient of x (fan-out), added inside the kernel
    gx2, dg2, db2 = torch.empty_like(x).cuda(), torch.zeros(C).cuda(), torch.zeros(C).cuda()
    fused = dev.layernorm_backward(gy.cuda(), x.cuda(), cu(ga), cu(be), m, r, gx2, dg2 if affine else None, db2 if affine else None,
                                   rows, C, gelu, gskip=gsk.cuda())
    want = gx_r.float() + gsk.float()
    assert rel(gx2.float() + (0 if fused else gsk.cuda().float()), want) < TOL[dtype]
    assert fused == (C % (4 if dtype == torch.bfloat16 else 2) == 0)
    res = rnd(rows, C, dtype=dtype, seed=5)                      # skip connection added in the same kernel
    y2_r = torch.empty_like(x)
    fake.layernorm_forward(x, ga, be, y2_r, m_r, r_r, rows, C, gelu, residual=res)
    y2 = torch.empty_like(x).cuda()
    dev.layernorm_forward(x.cuda(), cu(ga), cu(be), y2, m, r, rows, C, gelu, residual=res.cuda())
    assert rel(y2, y2_r) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,L", [(64 * 16 * 49, 49), (1000, 40), (999, 24), (512, 300), (100, 100), (7, 1000)])
def test_softmax(dev, rows, L, dtype):
    fake = FakeDevice()
    x = rnd(rows, L, dtype=dtype, seed=1, scale=3.0)
    x[0, : L // 2] = float("-inf")        # key-padding style -inf entries
    gy = rnd(rows, L, dtype=dtype, seed=2)
    y_r, gx_r = torch.empty_like(x), torch.empty_like(x)
    fake.softmax_forward(x, y_r, rows, L)
    fake.softmax_backward(gy, y_r, gx_r, rows, L)
    y, gx = torch.empty_like(x).cuda(), torch.empty_like(x).cuda()
    dev.softmax_forward(x.cuda(), y, rows, L)
    dev.softmax_backward(gy.cuda(), y_r.cuda(), gx, rows, L)
    torch.cuda.synchronize()
    assert rel(y, y_r) < TOL[dtype] and rel(gx, gx_r) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act,scale", [(hip.ACT_RELU, 1.0), (hip.ACT_GELU, 1.0), (hip.ACT_ELU, 1.0),
                                       (hip.ACT_SIGMOID, 10.0), (hip.ACT_NONE, 1.0)])
def test_act_backward_and_colsum(dev, act, scale, dtype):
    fake = FakeDevice()
    rows, C = 1234, 96
    gy = rnd(rows, C, dtype=dtype, seed=1)
    pre = rnd(rows, C, seed=2)
    ref = pre if act == hip.ACT_GELU else (scale * {hip.ACT_RELU: torch.relu, hip.ACT_ELU: torch.nn.functional.elu,
                                                    hip.ACT_SIGMOID: torch.sigmoid, hip.ACT_NONE: lambda t: t}.get(act, lambda t: t)(pre))
    ref = ref.to(dtype)
    ch = rnd(C, seed=3)
    gx_r, gx = torch.empty_like(gy), torch.empty_like(gy).cuda()
    fake.act_backward(gy, ref, gx_r, ch, rows, C, act, scale)
    dev.act_backward(gy.cuda(), ref.cuda(), gx, ch.cuda(), rows, C, act, scale)
    s_r, s = torch.zeros(C), torch.zeros(C).cuda()
    fake.colsum(gy, s_r, rows, C)
    dev.colsum(gy.cuda(), s, rows, C)
    torch.cuda.synchronize()
    assert rel(gx, gx_r) < TOL[dtype] and rel(s, s_r) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("log_err", [True, False])
@pytest.mark.parametrize("hw", [(30, 40), (15, 20), (120, 160), (96, 128)])
def test_silog_and_segce(dev, hw, log_err, dtype):
    fake = FakeDevice()
    B, H, W = 2, 96, 128
    h, w = hw if hw[0] <= H else (H, W)
    g = torch.Generator().manual_seed(5)
    gt = torch.rand(B, H, W, generator=g) * 9 + 0.5
    gt[torch.rand(B, H, W, generator=g) < 0.1] = 0.0
    pred = (torch.rand(B, h, w, generator=g) * 0.9 + 0.05).to(dtype)
    gl = torch.tensor([0.7])
    s_r, s = torch.zeros(3, dtype=torch.float64), torch.zeros(3, dtype=torch.float64).cuda()
    fake.silog_sums(pred, gt, s_r, B, h, w, H, W, log_err)
    dev.silog_sums(pred.cuda(), gt.cuda(), s, B, h, w, H, W, log_err)
    gp_r, gp = torch.empty_like(pred), torch.empty_like(pred).cuda()
    fake.silog_backward(pred, gt, s_r, gl, 0.25, 0.85, gp_r, B, h, w, H, W, log_err)
    dev.silog_backward(pred.cuda(), gt.cuda(), s, gl.cuda(), 0.25, 0.85, gp, B, h, w, H, W, log_err)
    torch.cuda.synchronize()
    assert s.cpu()[2] == s_r[2]                       # valid-pixel count: exact
    assert rel(s, s_r) < 1e-5 and rel(gp, gp_r) < TOL[dtype]

    P = B * h * w
    logits = rnd(P, 2, dtype=dtype, seed=9, scale=2.0)
    tgt = (torch.rand(P, generator=g) < 0.5).long()
    c_r, c = torch.zeros(1, dtype=torch.float64), torch.zeros(1, dtype=torch.float64).cuda()
    fake.seg_ce_sum(logits, tgt, c_r, P)
    dev.seg_ce_sum(logits.cuda(), tgt.cuda(), c, P)
    q_r, q = torch.empty_like(logits), torch.empty_like(logits).cuda()
    fake.seg_ce_backward(logits, tgt, gl, 2.0, q_r, P)
    dev.seg_ce_backward(logits.cuda(), tgt.cuda(), gl.cuda(), 2.0, q, P)
    torch.cuda.synchronize()
    assert rel(c, c_r) < 1e-5 and rel(q, q_r) < TOL[dtype]


def test_sqnorm_adamw(dev):
    fake = FakeDevice()
    n = 1_000_003
    p, g = rnd(n, seed=1), rnd(n, seed=2, scale=3.0)
    m, v = rnd(n, seed=3, scale=0.1), rnd(n, seed=4).abs() * 0.01
    outs = []
    for lib, to in ((fake, lambda t: t.clone()), (dev, lambda t: t.cuda())):
        P, G, M, V = to(p), to(g), to(m), to(v)
        P16 = torch.empty(n, dtype=torch.bfloat16, device=P.device)
        sq = torch.zeros(1, dtype=torch.float64, device=P.device)
        lib.sqnorm(G, sq, n)
        lib.adamw_step(P, G, M, V, P16, sq, n, 1e-4, 0.9, 0.999, 1e-8, 1e-4, 1 - 0.9 ** 3, 1 - 0.999 ** 3, 0.1, 0.5)
        outs.append((sq.cpu(), P.cpu(), M.cpu(), V.cpu(), P16.float().cpu()))
    torch.cuda.synchronize()
    for a, b in zip(outs[1], outs[0]):
        assert rel(a, b) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode", [hip.RESAMPLE_BILINEAR_AC, hip.RESAMPLE_NEAREST])
@pytest.mark.parametrize("shape", [(2, 7, 10, 120, 160, 32), (2, 15, 20, 30, 40, 1), (1, 60, 80, 120, 160, 24), (2, 7, 10, 60, 80, 60),
                                   (2, 1, 1, 16, 16, 60), (2, 6, 8, 12, 16, 64), (1, 9, 11, 24, 32, 8), (2, 30, 40, 30, 40, 16)])
def test_resample(dev, shape, mode, dtype):
    fake = FakeDevice()
    B, Hs, Ws, Ho, Wo, C = shape
    x = rnd(B, Hs, Ws, C, dtype=dtype, seed=1)
    gy = rnd(B, Ho, Wo, C, dtype=dtype, seed=2)
    y_r, gx_r = torch.empty(B, Ho, Wo, C, dtype=dtype), torch.empty(B, Hs, Ws, C, dtype=dtype)
    fake.resample_forward(x, y_r, B, Hs, Ws, Ho, Wo, C, mode)
    fake.resample_backward(gy, gx_r, B, Hs, Ws, Ho, Wo, C, mode)
    y, gx = torch.empty_like(y_r).cuda(), torch.empty_like(gx_r).cuda()
    dev.resample_forward(x.cuda(), y, B, Hs, Ws, Ho, Wo, C, mode)
    dev.resample_backward(gy.cuda(), gx, B, Hs, Ws, Ho, Wo, C, mode)
    torch.cuda.synchronize()
    assert rel(y, y_r) < TOL[dtype] and rel(gx, gx_r) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 120, 160, 32, 16), (2, 16, 16, 60, 2), (1, 30, 41, 8, 4), (2, 17, 19, 16, 8)])
def test_avgpool(dev, shape, dtype):
    fake = FakeDevice()
    B, H, W, C, k = shape
    x = rnd(B, H, W, C, dtype=dtype, seed=1)
    gy = rnd(B, H // k, W // k, C, dtype=dtype, seed=2)
    y_r, gx_r = torch.empty(B, H // k, W // k, C, dtype=dtype), torch.empty(B, H, W, C, dtype=dtype)
    fake.avgpool_forward(x, y_r, B, H, W, C, k)
    fake.avgpool_backward(gy, gx_r, B, H, W, C, k)
    y, gx = torch.empty_like(y_r).cuda(), torch.empty_like(gx_r).cuda()
    dev.avgpool_forward(x.cuda(), y, B, H, W, C, k)
    dev.avgpool_backward(gy.cuda(), gx, B, H, W, C, k)
    torch.cuda.synchronize()
    assert rel(y, y_r) < TOL[dtype] and rel(gx, gx_r) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("hd,nwin,wpi,shifted", [(4, 37, 1, False), (4, 24, 12, True), (8, 20, 4, True), (16, 9, 9, True),
                                                  (32, 6, 3, True), (32, 5, 1, False), (8, 2100, 4, True)])
@pytest.mark.parametrize("table", [False, True], ids=["dense_bias", "bias_table"])
def test_window_attention(dev, hd, nwin, wpi, shifted, dtype, table):
    """table: the (169, heads) relative-position PARAMETER gathered through relative_position_index inside the kernels, the
    gradient accumulated into a table-shaped buffer (the product path); dense: a (heads, 49, 49) bias."""
    from gw_depth_amd.model import relative_position_index
    fake = FakeDevice()
    H = 16
    qkv = rnd(nwin, 49, 3, H, hd, dtype=dtype, seed=1)
    bias = rnd(169, H, seed=2, scale=0.5) if table else rnd(H, 49, 49, seed=2, scale=0.5)
    ridx = relative_position_index().reshape(-1).to(torch.int32) if table else None
    g = torch.Generator().manual_seed(7)
    region = torch.randint(0, 3, (wpi, 49), generator=g, dtype=torch.int32) if shifted else None
    go = rnd(nwin, 49, H, hd, dtype=dtype, seed=3)
    scale = hd ** -0.5
    o_r = torch.empty(nwin, 49, H, hd, dtype=dtype)
    fake.winattn_forward(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], o_r, bias, region, wpi, scale, rel_index=ridx)
    g_r, db_r = torch.empty_like(qkv), torch.zeros_like(bias)
    fake.winattn_backward(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], go, g_r[:, :, 0], g_r[:, :, 1], g_r[:, :, 2], bias, db_r,
                          region, wpi, scale, rel_index=ridx)
    Q = qkv.cuda()
    rg = None if region is None else region.cuda()
    rl = None if ridx is None else ridx.cuda()
    o = torch.full_like(o_r, float("nan")).cuda()
    dev.winattn_forward(Q[:, :, 0], Q[:, :, 1], Q[:, :, 2], o, bias.cuda(), rg, wpi, scale, rel_index=rl)
    G, db = torch.full_like(g_r, float("nan")).cuda(), torch.zeros_like(bias).cuda()
    dev.winattn_backward(Q[:, :, 0], Q[:, :, 1], Q[:, :, 2], go.cuda(), G[:, :, 0], G[:, :, 1], G[:, :, 2], bias.cuda(), db,
                         rg, wpi, scale, rel_index=rl)
    torch.cuda.synchronize()
    assert rel(o, o_r) < TOL[dtype], "forward"
    for idx, name in enumerate(("dq", "dk", "dv")):
        assert rel(G[:, :, idx], g_r[:, :, idx]) < TOL[dtype], name
    assert rel(db, db_r) < TOL[dtype], "dbias"
    if table:       # the head-major scratch form of the table gradient (dbias_head_major): what ops._winattn_table_grad uses on the device
        G2, db_hm = torch.full_like(g_r, float("nan")).cuda(), torch.zeros(H, 169).cuda()
        dev.winattn_backward(Q[:, :, 0], Q[:, :, 1], Q[:, :, 2], go.cuda(), G2[:, :, 0], G2[:, :, 1], G2[:, :, 2], bias.cuda(), db_hm,
                             rg, wpi, scale, rel_index=rl, head_major=True)
        assert rel(db_hm.t(), db_r) < TOL[dtype], "dbias, head-major"
        assert rel(G2, g_r) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("e,nwin", [(12, 40), (16, 9), (24, 3), (12, 3000)])
def test_token_attention(dev, e, nwin, dtype):
    fake = FakeDevice()
    H = 16
    q = rnd(nwin, 49, H, 4, dtype=dtype, seed=1)
    kv = rnd(nwin, 49, 2, H, e, dtype=dtype, seed=2)          # k and v as strided views of one buffer
    go = rnd(nwin, 49, H, 4, dtype=dtype, seed=3)
    scale = 0.5
    o_r = torch.empty_like(q)
    fake.tokattn_forward(q, kv[:, :, 0], kv[:, :, 1], o_r, scale)
    gq_r, gkv_r = torch.empty_like(q), torch.empty_like(kv)
    fake.tokattn_backward(q, kv[:, :, 0], kv[:, :, 1], go, gq_r, gkv_r[:, :, 0], gkv_r[:, :, 1], scale)
    Q, KV = q.cuda(), kv.cuda()
    o = torch.full_like(o_r, float("nan")).cuda()
    dev.tokattn_forward(Q, KV[:, :, 0], KV[:, :, 1], o, scale)
    gq, gkv = torch.full_like(gq_r, float("nan")).cuda(), torch.full_like(gkv_r, float("nan")).cuda()
    dev.tokattn_backward(Q, KV[:, :, 0], KV[:, :, 1], go.cuda(), gq, gkv[:, :, 0], gkv[:, :, 1], scale)
    torch.cuda.synchronize()
    assert rel(o, o_r) < TOL[dtype] and rel(gq, gq_r) < TOL[dtype] and rel(gkv, gkv_r) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("H,W,shift,S", [(15, 20, 3, 100), (15, 20, 0, 100), (21, 21, 3, 64), (7, 9, 6, 256), (15, 20, 3, 300)])
def test_point_sample_in_the_padded_rolled_frame(dev, H, W, shift, S, dtype):
    """ops.point_sample(frame=...) == nearest sampling of torch.roll(F.pad(map), -shift): forward and map gradient (S > 256 builds the frame)."""
    from gw_depth_amd import ops
    B, C, WS = 2, 64, 7
    Hf, Wf = (H + WS - 1) // WS * WS, (W + WS - 1) // WS * WS
    fmap = rnd(B, H, W, C, dtype=dtype, seed=1)
    g = torch.Generator().manual_seed(2)
    coords = torch.rand(B, S, 2, generator=g) * 2.4 - 1.2                      # some points outside the frame
    # exact pixel centres and points on the frame's far rows / columns (padding) as well
    coords[:, :8, 0] = (2 * torch.arange(8).float() % Wf + 1) / Wf - 1
    coords[:, :8, 1] = (2 * (Hf - 1 - torch.arange(8).float() % Hf) + 1) / Hf - 1
    gout = torch.randn(B, S, C, generator=g)
    fake = FakeDevice()
    ref = torch.empty(B, S, C)
    fake.point_sample_framed_forward(fmap, coords, ref, B, H, W, C, S, (Hf, Wf, shift))
    gref = torch.empty(B, H, W, C, dtype=dtype)
    fake.point_sample_framed_backward(gout, coords, gref, B, H, W, C, S, (Hf, Wf, shift))
    x = fmap.cuda().requires_grad_(True)
    out = ops.point_sample(x, coords.cuda(), nearest=True, frame=(Hf, Wf, shift))
    (gx,) = torch.autograd.grad(out, [x], gout.cuda())
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), ref)
    assert rel(gx, gref) < TOL[dtype]
    assert (ref.abs().sum(-1) == 0).any() and (ref.abs().sum(-1) > 0).any()         # both padding / outside hits and real samples occurred


@pytest.mark.parametrize("Cout,Cin", [(64, 64), (32, 64), (160, 96), (1, 3)])
def test_upsample_taps_collapse_and_fold(dev, Cout, Cin):
    """gwd_upsample_taps_collapse / _fold against the element-wise formulas (3x3 over a 2x nearest-upsampled map == 4x4 / stride 2 taps)."""
    fake = FakeDevice()
    w = rnd(Cout, 3, 3, Cin, seed=1)
    for dtype in (torch.float32, torch.bfloat16):
        wk_r = torch.empty(Cin, 4, 4, Cout)
        fake.upsample_taps_collapse(w, wk_r)
        wk = torch.full((Cin, 4, 4, Cout), float("nan"), dtype=dtype, device="cuda")
        dev.upsample_taps_collapse(w.cuda(), wk)
        torch.cuda.synchronize()
        assert rel(wk, wk_r.to(dtype)) < (1e-6 if dtype == torch.float32 else 4e-3)
    D, dw0 = rnd(Cin, 4, 4, Cout, seed=2), rnd(Cout, 3, 3, Cin, seed=3)
    dw_r = dw0.clone()
    fake.upsample_taps_fold(D, dw_r)
    dw = dw0.clone().cuda()
    dev.upsample_taps_fold(D.cuda(), dw)
    torch.cuda.synchronize()
    assert rel(dw, dw_r) < 1e-6
    with pytest.raises(ValueError):
        dev.upsample_taps_fold(D.cuda(), dw0.cuda().permute(0, 2, 1, 3))


def test_packed_qkv_gradient_link_of_the_1_32_stage(dev, monkeypatch):
    """ops.GradLink: window_attention_qkv's backward hands its packed gradient (k / v slots) to ref_scores' backward, which writes the q slot -
    same gradients as the two zero-filled tensors autograd used to add."""
    from gw_depth_amd import model as M, ops
    torch.manual_seed(0)
    att = M.WindowAttention(512).cuda()
    for p in att.parameters():
        torch.nn.init.normal_(p, std=0.05)
    xw = (torch.randn(18, 49, 512, device="cuda") * 0.5).bfloat16().requires_grad_(True)        # 2 images x 9 windows
    x_ref = (torch.randn(2, 40, 512, device="cuda") * 0.5).bfloat16().requires_grad_(True)
    wgt = torch.randn(18, 49, 512, device="cuda")
    ps = [p for p in att.parameters() if p.requires_grad]

    def run():
        y = att(xw, x_ref, None)
        return y, torch.autograd.grad((y.float() * wgt).sum(), [xw, x_ref] + ps, allow_unused=True)

    y1, g_link = run()
    monkeypatch.setattr(ops, "GradLink", lambda: None)
    y2, g_two = run()
    torch.cuda.synchronize()
    assert torch.equal(y1, y2)
    n = 0
    for a, b in zip(g_link, g_two):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.isfinite(a).all() and rel(a, b) < 1e-3
            n += 1
    assert n >= 6


@pytest.mark.parametrize("e,nwin", [(12, 40), (16, 9), (24, 3), (12, 3000), (16, 1)])
def test_token_attention_pair_is_two_single_calls(dev, e, nwin):
    """Both class tokens in one launch (gwd_tokattn_pair_*): outputs and query gradients per token, k / v gradients summed."""
    fake = FakeDevice()
    H, dtype, scale = 16, torch.bfloat16, 0.5
    q, q2 = rnd(nwin, 49, H, 4, dtype=dtype, seed=1), rnd(nwin, 49, H, 4, dtype=dtype, seed=5)
    kv = rnd(nwin, 49, 2, H, e, dtype=dtype, seed=2)
    go, go2 = rnd(nwin, 49, H, 4, dtype=dtype, seed=3), rnd(nwin, 49, H, 4, dtype=dtype, seed=4)
    o_r, o2_r, gq_r, gq2_r = (torch.empty(nwin, 49, H, 4) for _ in range(4))
    gkv_r = torch.empty(nwin, 49, 2, H, e)
    fake.tokattn_pair_forward(q, q2, kv[:, :, 0], kv[:, :, 1], o_r, o2_r, scale)
    fake.tokattn_pair_backward(q, q2, kv[:, :, 0], kv[:, :, 1], go, go2, gq_r, gq2_r, gkv_r[:, :, 0], gkv_r[:, :, 1], scale)
    Q, Q2, KV = q.cuda(), q2.cuda(), kv.cuda()
    nan = lambda *shape: torch.full(shape, float("nan"), dtype=dtype, device="cuda")
    o, o2, gq, gq2, gkv = nan(nwin, 49, H, 4), nan(nwin, 49, H, 4), nan(nwin, 49, H, 4), nan(nwin, 49, H, 4), nan(nwin, 49, 2, H, e)
    dev.tokattn_pair_forward(Q, Q2, KV[:, :, 0], KV[:, :, 1], o, o2, scale)
    dev.tokattn_pair_backward(Q, Q2, KV[:, :, 0], KV[:, :, 1], go.cuda(), go2.cuda(), gq, gq2, gkv[:, :, 0], gkv[:, :, 1], scale)
    torch.cuda.synchronize()
    for name, a, b in (("o", o, o_r), ("o2", o2, o2_r), ("gq", gq, gq_r), ("gq2", gq2, gq2_r), ("gkv", gkv, gkv_r)):
        assert rel(a, b) < TOL[dtype], name
    # and against the single-call kernels themselves: the per-token halves are the same arithmetic
    o_s = nan(nwin, 49, H, 4)
    dev.tokattn_forward(Q2, KV[:, :, 0], KV[:, :, 1], o_s, scale)
    torch.cuda.synchronize()
    assert torch.equal(o_s, o2)
    # fp32 is not covered: the binding raises, ops.token_attention_pair issues the two single calls instead
    with pytest.raises(RuntimeError):
        dev.tokattn_pair_forward(Q.float(), Q2.float(), KV[:, :, 0].float(), KV[:, :, 1].float(), o.float(), o2.float(), scale)


def test_token_attention_pair_autograd_matches_two_nodes(dev):
    from gw_depth_amd import ops
    nwin, H, e = 24, 16, 16
    mk = lambda *s, seed: rnd(*s, dtype=torch.bfloat16, seed=seed).cuda().requires_grad_(True)
    q, q2, k, v = mk(nwin, 49, H, 4, seed=1), mk(nwin, 49, H, 4, seed=2), mk(nwin, 49, H, e, seed=3), mk(nwin, 49, H, e, seed=4)
    w1, w2 = rnd(nwin, 49, H * 4, dtype=torch.bfloat16, seed=5).cuda(), rnd(nwin, 49, H * 4, dtype=torch.bfloat16, seed=6).cuda()
    a, b = ops.token_attention_pair(q, q2, k, v, 0.5)
    g_pair = torch.autograd.grad((a.float() * w1).sum() + (b.float() * w2).sum(), [q, q2, k, v])
    a1, b1 = ops.token_attention(q, k, v, 0.5), ops.token_attention(q2, k, v, 0.5)
    g_two = torch.autograd.grad((a1.float() * w1).sum() + (b1.float() * w2).sum(), [q, q2, k, v])
    assert torch.equal(a, a1) and torch.equal(b, b1)
    for x, y in zip(g_pair, g_two):
        assert rel(x, y) < TOL[torch.bfloat16]
    # fp32 operands take the two-call route
    a32, b32 = ops.token_attention_pair(q.float(), q2.float(), k.float(), v.float(), 0.5)
    assert rel(a32, a) < TOL[torch.bfloat16] and rel(b32, b) < TOL[torch.bfloat16]


@pytest.mark.parametrize("layers,B,Q,sizes", [(6, 8, 100, [7] * 8), (2, 3, 100, [1, 12, 5]), (1, 2, 37, [37, 20]), (3, 2, 1000, [30, 64])])
def test_device_lsap_matches_scipy(dev, layers, B, Q, sizes):
    from scipy.optimize import linear_sum_assignment
    g = torch.Generator().manual_seed(11)
    sumT = sum(sizes)
    cost = torch.rand(layers, B, Q, sumT, generator=g) * 5 - 1
    cost[0, 0] = (cost[0, 0] * 4).round() / 4 + torch.rand(Q, sumT, generator=g) * 1e-4      # near-degenerate block
    off = [0]
    for s in sizes:
        off.append(off[-1] + s)
    out = torch.full((layers, sumT), -1, dtype=torch.int32, device="cuda")
    dev.lsap(cost.cuda(), torch.tensor(off, dtype=torch.int32, device="cuda"), out, max(sizes))
    torch.cuda.synchronize()
    out = out.cpu()
    for l in range(layers):
        for b in range(B):
            c = cost[l, b, :, off[b]:off[b + 1]].double().numpy()
            qi, ti = linear_sum_assignment(c)
            want = torch.empty(sizes[b], dtype=torch.int32)
            want[torch.as_tensor(ti)] = torch.as_tensor(qi, dtype=torch.int32)
            assert torch.equal(out[l, off[b]:off[b + 1]], want), (l, b)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("geom", [(2, 30, 40, 0), (2, 30, 40, 3), (1, 15, 20, 3)])
def test_window_map_multi_equals_single_maps(dev, geom, dtype):
    """gwd_window_map_multi: the three maps of a class-token Swin block (features, depth tokens, seg tokens) in one launch, both
    directions, the reverse with residual streams (one of them absent) - bit for bit what three gwd_window_map calls give; and through
    ops.window_gather_multi / window_scatter_multi the gradients of the single-map autograd nodes."""
    from gw_depth_amd import ops
    B, H, W, shift = geom
    Cs = [128, 64, 64]
    Hp, Wp = (H + 6) // 7 * 7, (W + 6) // 7 * 7
    nwin = B * (Hp // 7) * (Wp // 7)
    xs = [rnd(B, H, W, c, dtype=dtype, seed=3 + i).cuda() for i, c in enumerate(Cs)]
    wins = [torch.full((nwin, 49, c), 7.0, dtype=dtype).cuda() for c in Cs]
    dev.window_map_multi(xs, wins, B, H, W, Cs, shift, True)
    for x, w, c in zip(xs, wins, Cs):
        ref = torch.empty_like(w)
        dev.window_map(x, ref, B, H, W, c, shift, True)
        assert torch.equal(w, ref)
    gs = [rnd(nwin, 49, c, dtype=dtype, seed=13 + i).cuda() for i, c in enumerate(Cs)]
    ress = [rnd(B, H, W, Cs[0], dtype=dtype, seed=23).cuda(), None, rnd(B, H, W, Cs[2], dtype=dtype, seed=25).cuda()]
    outs = [torch.empty(B, H, W, c, dtype=dtype).cuda() for c in Cs]
    dev.window_map_multi(gs, outs, B, H, W, Cs, shift, False, residuals=ress)
    for g, o, r, c in zip(gs, outs, ress, Cs):
        ref = torch.empty_like(o)
        dev.window_map(g, ref, B, H, W, c, shift, False, residual=r)
        assert torch.equal(o, ref)
    # autograd nodes
    leaves = [x.clone().requires_grad_(True) for x in xs]
    rl = [r.clone().requires_grad_(True) for r in (ress[0], xs[1], ress[2])]
    multi = ops.window_scatter_multi(list(ops.window_gather_multi(leaves, shift)), B, H, W, shift, rl)
    sum((m.float() * (i + 1)).sum() for i, m in enumerate(multi)).backward()
    leaves1 = [x.clone().requires_grad_(True) for x in xs]
    rl1 = [r.detach().clone().requires_grad_(True) for r in rl]
    single = [ops.window_scatter(ops.window_gather(x, shift), B, H, W, shift, residual=r) for x, r in zip(leaves1, rl1)]
    sum((m.float() * (i + 1)).sum() for i, m in enumerate(single)).backward()
    for a_, b_ in zip(multi, single):
        assert torch.equal(a_, b_)
    for a_, b_ in zip(leaves + rl, leaves1 + rl1):
        assert torch.equal(a_.grad, b_.grad)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 30, 40, 32, 0), (2, 30, 40, 32, 3), (1, 14, 21, 8, 3), (2, 15, 20, 128, 3), (3, 12, 16, 4, 0)])
def test_window_map(dev, shape, dtype):
    """Index remapping copy: bit-exact in both directions, padded slots zero, and scatter(gather(x)) == x."""
    B, H, W, C, shift = shape
    if C * (2 if dtype == torch.bfloat16 else 4) % 16:
        pytest.skip("channel vector not 16-byte")
    fake = FakeDevice()
    Hp, Wp = (H + 6) // 7 * 7, (W + 6) // 7 * 7
    nwin = B * (Hp // 7) * (Wp // 7)
    x = rnd(B, H, W, C, dtype=dtype, seed=3)
    win_r = torch.empty(nwin, 49, C, dtype=dtype)
    fake.window_map(x, win_r, B, H, W, C, shift, True)
    win = torch.full((nwin, 49, C), 7.0, dtype=dtype).cuda()
    dev.window_map(x.cuda(), win, B, H, W, C, shift, True)
    assert torch.equal(win.cpu(), win_r)
    g = rnd(nwin, 49, C, dtype=dtype, seed=4)
    back_r = torch.empty(B, H, W, C, dtype=dtype)
    fake.window_map(g, back_r, B, H, W, C, shift, False)
    back = torch.empty(B, H, W, C, dtype=dtype).cuda()
    dev.window_map(g.cuda(), back, B, H, W, C, shift, False)
    assert torch.equal(back.cpu(), back_r)
    rt = torch.empty(B, H, W, C, dtype=dtype).cuda()
    dev.window_map(win, rt, B, H, W, C, shift, False)
    assert torch.equal(rt.cpu(), x)
    res = rnd(B, H, W, C, dtype=dtype, seed=6)                    # reverse + residual stream in one pass
    fused_r = torch.empty(B, H, W, C, dtype=dtype)
    fake.window_map(g, fused_r, B, H, W, C, shift, False, residual=res)
    fused = torch.empty(B, H, W, C, dtype=dtype).cuda()
    dev.window_map(g.cuda(), fused, B, H, W, C, shift, False, residual=res.cuda())
    assert torch.equal(fused.cpu(), fused_r)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(8, 17640, 16, 32), (2, 441, 16, 32), (3, 100, 16, 7), (2, 50, 32, 64), (1, 5, 8, 4)])
def test_inorm_gelu(dev, shape, dtype):
    B, L, C, S = shape
    if (C * (2 if dtype == torch.bfloat16 else 4)) % 16:
        pytest.skip("channel vector not 16-byte")
    fake = FakeDevice()
    a, u = rnd(B, L, C, dtype=dtype, seed=5), (rnd(B, L, C, dtype=dtype, seed=6) * 0.7 + 1.5).to(dtype)
    gy = rnd(B, L, C, dtype=dtype, seed=7)
    y_r, du_r = torch.empty(B, L, C), torch.empty(B, L, C)
    stat_r = torch.empty(B, C, 2)
    fake.inorm_gelu_forward(a, u, y_r, None, stat_r, B, L, C, S, 1e-5)
    fake.inorm_gelu_backward(gy, u, stat_r, None, du_r, B, L, C, S)
    y, du = torch.empty(B, L, C, dtype=dtype).cuda(), torch.empty(B, L, C, dtype=dtype).cuda()
    part = torch.full((B, S, C, 2), float("nan")).cuda()
    stat = torch.full((B, C, 2), float("nan")).cuda()
    dev.inorm_gelu_forward(a.cuda(), u.cuda(), y, part, stat, B, L, C, S, 1e-5)
    assert rel(stat, stat_r) < 2e-5
    part.fill_(float("nan"))
    dev.inorm_gelu_backward(gy.cuda(), u.cuda(), stat, part, du, B, L, C, S)
    torch.cuda.synchronize()
    assert rel(y, y_r) < TOL[dtype] and rel(du, du_r) < TOL[dtype]


def test_weight_cache_batch_refresh(dev):
    """One gwd_weight_prep_batch launch must reproduce the per-weight gwd_weight_prep copies after the masters change."""
    from gw_depth_amd import ops
    torch.manual_seed(3)
    shapes = [(64, 3, 3, 32), (160, 3, 3, 160), (256, 1, 1, 64), (10, 1, 1, 7), (30, 3, 3, 60), (2048, 1, 1, 512)]
    ws = [torch.randn(s, device="cuda") for s in shapes]
    rss = [None, torch.rand(160, device="cuda") + 0.5, None, None, torch.rand(30, device="cuda") + 0.5, None]
    cache = ops.WeightCache([(w.data_ptr(), w.data_ptr() + w.numel() * 4) for w in ws])
    cache.begin_pass()
    for w, rs in zip(ws, rss):
        assert cache.get(w, rs, "t") is not None
        if rs is not None:
            assert cache.get(w, rs, "fwd") is not None
    assert cache.get(torch.randn(8, 1, 1, 8, device="cuda"), None, "t") is None      # not parameter storage: never cached
    cache.end_pass()
    assert cache.get(ws[0], None, "t") is None                 # inactive outside a pass
    for w in ws:
        w.mul_(1.5).add_(0.25)
    cache.begin_pass()                                           # table upload + ONE launch
    assert cache.n_jobs == len(ws)
    for w, rs in zip(ws, rss):
        N, C = w.shape[0], w.shape[-1]
        taps = w.numel() // (N * C)
        ref_t = torch.empty((C,) + tuple(w.shape[1:-1]) + (N,), dtype=torch.bfloat16, device="cuda")
        ref_f = torch.empty(w.shape, dtype=torch.bfloat16, device="cuda")
        dev.weight_prep(w, rs, ref_f, ref_t, N, taps, C, hip.BF16)
        assert torch.equal(cache.get(w, rs, "t"), ref_t)
        if rs is not None:
            assert torch.equal(cache.get(w, rs, "fwd"), ref_f)
    cache.end_pass()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("nearest", [False, True])
@pytest.mark.parametrize("shape", [(2, 21, 21, 64, 40), (3, 15, 20, 7, 5), (2, 60, 80, 128, 30), (1, 12, 16, 1, 9)])
def test_point_sample(dev, shape, nearest, dtype):
    """grid_sample at points (incl. out-of-image and duplicate points) on a pixel-major map; fp32 output."""
    B, H, W, C, S = shape
    fake = FakeDevice()
    fmap = rnd(B, H, W, C, dtype=dtype, seed=8)
    g = torch.Generator().manual_seed(S + H)
    coords = torch.rand(B, S, 2, generator=g) * 2.4 - 1.2
    coords[:, 1] = coords[:, 0]                                   # a duplicate point: colliding scatter in the backward
    mode = 1 if nearest else 0
    out_r = torch.empty(B, S, C)
    fake.point_sample_forward(fmap, coords, out_r, B, H, W, C, S, mode)
    out = torch.empty(B, S, C).cuda()
    dev.point_sample_forward(fmap.cuda(), coords.cuda(), out, B, H, W, C, S, mode)
    assert rel(out, out_r) < 1e-6
    gout = rnd(B, S, C, seed=9)
    gm_r = torch.zeros(B, H, W, C, dtype=dtype)
    fake.point_sample_backward(gout, coords, gm_r, B, H, W, C, S, mode)
    gm = torch.zeros(B, H, W, C, dtype=dtype).cuda()
    dev.point_sample_backward(gout.cuda(), coords.cuda(), gm, B, H, W, C, S, mode)
    assert rel(gm, gm_r) < TOL[dtype]
    gm2 = torch.full((B, H, W, C), 7.0, dtype=dtype).cuda()      # gather form: writes every element, needs no zeroing
    assert dev.point_sample_backward_gather(gout.cuda(), coords.cuda(), gm2, B, H, W, C, S, mode)      # any C (scalar channels when not vector-sized)
    assert rel(gm2, gm_r) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,C,act", [(1000, 64, hip.ACT_GELU), (77, 256, hip.ACT_RELU), (513, 32, hip.ACT_ELU), (40, 1024, hip.ACT_GELU), (2403, 1024, hip.ACT_RELU),
                                          (9999, 96, hip.ACT_GELU)])
def test_act_backward_colsum(dev, rows, C, act, dtype):
    fake = FakeDevice()
    gy, ref = rnd(rows, C, dtype=dtype, seed=11), rnd(rows, C, dtype=dtype, seed=12)
    gx_r, db_r = torch.empty(rows, C, dtype=dtype), torch.full((C,), 0.5)
    fake.act_backward_colsum(gy, ref, gx_r, db_r, rows, C, act, 1.0)
    gx, db = torch.empty(rows, C, dtype=dtype).cuda(), torch.full((C,), 0.5).cuda()
    assert dev.act_backward_colsum(gy.cuda(), ref.cuda(), gx, db, rows, C, act, 1.0)
    assert rel(gx, gx_r) < TOL[dtype] and rel(db, db_r) < TOL[dtype]
    assert dev.act_backward_colsum(gy.cuda()[:, :6].contiguous(), ref.cuda()[:, :6].contiguous(), gx[:, :6].contiguous(), db[:6].contiguous(),
                                   rows, 6, act, 1.0) is False
    # with the dropout multiplier of the layer (gx = act'(ref) * (gy * mult)), also behind no activation at all
    mult = ((torch.rand(rows, C, generator=torch.Generator().manual_seed(13)) > 0.1).float() / 0.9).to(dtype)
    for a in (act, hip.ACT_NONE):
        gx_r, db_r = torch.empty(rows, C, dtype=dtype), torch.zeros(C)
        fake.act_backward_colsum(gy, ref, gx_r, db_r, rows, C, a, 1.0, mult=mult)
        gx, db = torch.empty(rows, C, dtype=dtype).cuda(), torch.zeros(C).cuda()
        assert dev.act_backward_colsum(gy.cuda(), ref.cuda(), gx, db, rows, C, a, 1.0, mult=mult.cuda())
        assert rel(gx, gx_r) < TOL[dtype] and rel(db, db_r) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode", [hip.RESAMPLE_BILINEAR_AC, hip.RESAMPLE_NEAREST])
@pytest.mark.parametrize("shape", [(2, 7, 10, 120, 160, 32), (2, 3, 5, 60, 80, 8), (1, 60, 80, 120, 160, 24), (2, 15, 20, 17, 33, 16), (1, 1, 1, 9, 7, 8),
                                   (2, 3, 5, 60, 80, 60), (1, 4, 4, 9, 9, 6)])
def test_resample_backward_separable(dev, shape, mode, dtype):
    """Two-pass (x then y) backward == single-pass gather, through the fp32 scratch."""
    B, Hs, Ws, Ho, Wo, C = shape
    fake = FakeDevice()
    gy = rnd(B, Ho, Wo, C, dtype=dtype, seed=21)
    gx_r = torch.empty(B, Hs, Ws, C, dtype=dtype)
    fake.resample_backward(gy, gx_r, B, Hs, Ws, Ho, Wo, C, mode)
    gx = torch.full((B, Hs, Ws, C), float("nan"), dtype=dtype).cuda()
    tmp = torch.full((B, Ho, Ws, C), float("nan")).cuda()
    ok = dev.resample_backward_sep(gy.cuda(), tmp, gx, B, Hs, Ws, Ho, Wo, C, mode)
    if C % 4:
        assert ok is False
        return
    assert ok and rel(gx, gx_r) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 8, 100, 300), (3, 4, 37, 65), (1, 2, 5, 7), (2, 2, 9, 1200)])
def test_attention_softmax_scale_and_key_mask(dev, shape, dtype):
    B, H, L, S = shape
    fake = FakeDevice()
    x = rnd(B, H, L, S, dtype=dtype, seed=31, scale=3.0)
    mask = torch.zeros(B, S, dtype=torch.uint8)
    mask[-1, S // 2:] = 1
    gy = rnd(B, H, L, S, dtype=dtype, seed=32)
    y_r, gx_r = torch.empty(B, H, L, S, dtype=dtype), torch.empty(B, H, L, S, dtype=dtype)
    fake.softmax_masked_forward(x, mask, y_r, B * H * L, S, H * L, 0.17677669)
    fake.softmax_scaled_backward(gy, y_r, gx_r, B * H * L, S, 0.17677669)
    y, gx = torch.empty_like(y_r).cuda(), torch.empty_like(gx_r).cuda()
    dev.softmax_masked_forward(x.cuda(), mask.cuda(), y, B * H * L, S, H * L, 0.17677669)
    dev.softmax_scaled_backward(gy.cuda(), y_r.cuda(), gx, B * H * L, S, 0.17677669)
    assert rel(y, y_r) < TOL[dtype] and rel(gx, gx_r) < TOL[dtype]
    assert float(y[-1, :, :, S // 2:].abs().max()) == 0.0                      # masked keys get exactly zero weight
    y2 = torch.empty_like(y)
    dev.softmax_masked_forward(x.cuda(), None, y2, B * H * L, S, 1, 1.0)        # no mask, unit scale == plain softmax
    y3 = torch.empty_like(y)
    dev.softmax_forward(x.cuda(), y3, B * H * L, S)
    assert torch.equal(y2, y3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_colsum_batch(dev, dtype):
    """gwd_colsum_batch == one gwd_colsum per job (accumulating into a non-zero out), 1..16 jobs of mixed shapes, two of
    them sharing one output; a non-vector channel count is refused (-4) before anything is launched."""
    shapes = [(300, 256), (2400, 256), (153600, 64), (5, 8), (19200, 136), (800, 1024), (1, 512), (4800, 128),
              (77, 24), (9600, 320), (1200, 512), (64, 64), (100, 256), (38400, 160), (333, 40), (2048, 1024)]
    for n in (1, 3, 16):
        jobs, refs = [], []
        for i, (rows, C) in enumerate(shapes[:n]):
            g = rnd(rows, C, dtype=dtype, seed=i).cuda()
            out = rnd(C, seed=100 + i).cuda()
            refs.append(out.clone() + g.float().sum(0))
            jobs.append((g, out, rows, C))
        if n == 16:                                           # two layers sharing one bias (a weight used twice)
            jobs[12] = (jobs[12][0], jobs[0][1], jobs[12][2], jobs[12][3])
            refs[0] = refs[0] + jobs[12][0].float().sum(0)
        assert all(dev.colsum_batchable(j[0], j[3]) for j in jobs)
        dev.colsum_batch(jobs)
        torch.cuda.synchronize()
        for i, (j, r) in enumerate(zip(jobs, refs)):
            if n == 16 and i == 12:
                continue
            assert rel(j[1], r) < (2e-5 if dtype == torch.float32 else 2e-3), (n, i)
    assert not dev.colsum_batchable(torch.empty(3, 30, dtype=dtype), 30)
    with pytest.raises(RuntimeError):
        dev.colsum_batch([(torch.zeros(3, 30, dtype=dtype).cuda(), torch.zeros(30).cuda(), 3, 30)])


def test_conv_wgrad_batch_equals_single_calls(dev):
    """gwd_conv_wgrad_batch: 52 mixed jobs (plain GEMMs, 3x3 and strided convolutions on all three grouped tile shapes - more than
    one group's worth of the first -, a shared gradient buffer) give exactly what 52 gwd_conv_wgrad calls give up to the order of the
    fp32 atomics."""
    dt = torch.bfloat16
    # (B, H, W, Cin, Cout, k[, stride]): every grouped class - 128 x 128 / 64 x 64 / 32 x 128 tiles x plain GEMM / same-size 3x3 / strided
    specs = [(2400, 1, 1, 256, 256, 1), (800, 1, 1, 256, 2048, 1), (300, 1, 1, 2048, 256, 1), (19200, 1, 1, 64, 128, 1),
             (2, 12, 16, 64, 64, 3), (4800, 1, 1, 128, 24, 1), (153600, 1, 1, 64, 64, 1), (1176, 1, 1, 64, 192, 1),
             (2, 30, 40, 128, 128, 3), (2, 30, 40, 256, 128, 3, 2), (2, 24, 32, 64, 64, 3, 2), (2, 30, 40, 128, 16, 3), (2, 30, 40, 256, 512, 1, 2)]
    jobs, refs = [], []
    for i in range(52):
        B, H, W, Ci, Co, K = specs[i % len(specs)][:6]
        st = specs[i % len(specs)][6] if len(specs[i % len(specs)]) > 6 else 1
        Ho, Wo = conv_out(H, K, st, K // 2), conv_out(W, K, st, K // 2)
        x = rnd(B, H, W, Ci, dtype=dt, seed=i).cuda()
        gy = rnd(B, Ho, Wo, Co, dtype=dt, seed=50 + i).cuda()
        dims = (B, H, W, Ci, Ho, Wo, Co, K, K)
        kw = dict(stride=st, pad=K // 2)
        dw = rnd(Co, K, K, Ci, seed=200 + i).cuda()
        ref = dw.clone()
        dev.conv_wgrad(x, gy, ref, dims, **kw)
        jobs.append((x, gy, dw, dims, kw))
        refs.append(ref)
    twin = len(specs)                                            # job `twin` (same shape as job 0) accumulates into job 0's buffer
    jobs[twin] = jobs[twin][:2] + (jobs[0][2],) + jobs[twin][3:]
    dev.conv_wgrad(jobs[twin][0], jobs[twin][1], refs[0], jobs[twin][3], **jobs[twin][4])
    dev.conv_wgrad_batch(jobs)
    torch.cuda.synchronize()
    for i, (j, r) in enumerate(zip(jobs, refs)):
        if i == twin:
            continue
        assert rel(j[2], r) < 1e-5, i


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("spec", [(2, 12, 16, 64, 64, 3, 1), (2, 12, 16, 256, 128, 1, 1), (2, 12, 16, 128, 128, 3, 2), (300, 1, 1, 72, 40, 1, 1)])
def test_conv_wgrad_row_scale(dev, dtype, spec):
    """d->scale in gwd_conv_wgrad: dw[n] += scale[n] * grad[n] (gradient of the unscaled weight of a BN-folded layer),
    single call and batched call, DMA and register-staged kernels."""
    B, H, W, Ci, Co, K, stride = spec
    pad = K // 2
    Ho, Wo = conv_out(H, K, stride, pad), conv_out(W, K, stride, pad)
    x = rnd(B, H, W, Ci, dtype=dtype, seed=1)
    gy = rnd(B, Ho, Wo, Co, dtype=dtype, seed=2)
    sc = rnd(Co, seed=3) + 1.5
    dims = (B, H, W, Ci, Ho, Wo, Co, K, K)
    ref = rnd(Co, K, K, Ci, seed=4)
    dw1, dw2 = ref.clone().cuda(), ref.clone().cuda()
    FakeDevice().conv_wgrad(x, gy, ref, dims, stride=stride, pad=pad, scale=sc)
    dev.conv_wgrad(x.cuda(), gy.cuda(), dw1, dims, stride=stride, pad=pad, scale=sc.cuda())
    dev.conv_wgrad_batch([(x.cuda(), gy.cuda(), dw2, dims, dict(stride=stride, pad=pad, scale=sc.cuda()))])
    assert rel(dw1, ref) < TOL[dtype] and rel(dw2, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(8, 9, 40, 16, 32), (2, 4, 40, 16, 32), (1, 1, 7, 16, 8), (3, 2, 128, 4, 64)])
def test_ref_point_attention_kernels(dev, shape, dtype):
    """gwd_ref_scores_* / gwd_ref_mix_* (the line-point-guided query rewrite of the 1/32 stage) vs the reference's einsums in
    fp32 torch (tests/fake_device.py): q read in place from the packed qkv projection, gradients incl. the token sums."""
    B, nwin, R, H, hd = shape
    fake = FakeDevice()
    C, T = H * hd, nwin * 49
    qkv = rnd(B * nwin, 49, 3, H, hd, dtype=dtype, seed=1)
    ref_k, ref_v = rnd(B, R, C, dtype=dtype, seed=2), rnd(B, R, C, dtype=dtype, seed=3)
    g_ra, g_q = rnd(B, T, R, H, dtype=dtype, seed=4), rnd(B, T, C, dtype=dtype, seed=5)
    scale = hd ** -0.5
    ra_r = torch.empty(B, T, R, H, dtype=dtype)
    fake.ref_scores_forward(qkv[:, :, 0], ref_k, ra_r, B, nwin, scale)
    dq_r, dk_r = torch.zeros_like(qkv), torch.empty(B, R, C)
    fake.ref_scores_backward(qkv[:, :, 0], ref_k, g_ra, dq_r[:, :, 0], dk_r, B, nwin, scale)
    qn_r, att_r = torch.empty(B, T, C, dtype=dtype), torch.empty(B, T, R, H, dtype=dtype)
    fake.ref_mix_forward(ra_r, ref_v, qn_r, att_r, H)
    dra_r, dv_r = torch.empty(B, T, R, H, dtype=dtype), torch.empty(B, R, C)
    fake.ref_mix_backward(att_r, ref_v, g_q, dra_r, dv_r, H)

    Q, K, V = qkv.cuda(), ref_k.cuda(), ref_v.cuda()
    ra = torch.full((B, T, R, H), float("nan"), dtype=dtype, device="cuda")
    dev.ref_scores_forward(Q[:, :, 0], K, ra, B, nwin, scale)
    dq, dk = torch.zeros_like(Q), torch.full((B, R, C), float("nan"), device="cuda")
    dev.ref_scores_backward(Q[:, :, 0], K, g_ra.cuda(), dq[:, :, 0], dk, B, nwin, scale)
    qn, att = torch.full((B, T, C), float("nan"), dtype=dtype, device="cuda"), torch.full((B, T, R, H), float("nan"), dtype=dtype, device="cuda")
    dev.ref_mix_forward(ra_r.cuda(), V, qn, att, H)
    dra, dv = torch.full((B, T, R, H), float("nan"), dtype=dtype, device="cuda"), torch.full((B, R, C), float("nan"), device="cuda")
    dev.ref_mix_backward(att_r.cuda(), V, g_q.cuda(), dra, dv, H)
    torch.cuda.synchronize()
    tol = TOL[dtype]
    assert rel(ra, ra_r) < tol and rel(dq[:, :, 0], dq_r[:, :, 0]) < tol and rel(dk, dk_r) < tol
    assert float(dq[:, :, 1:].abs().max()) == 0.0                      # only the q slot is written
    assert rel(qn, qn_r) < tol and rel(att, att_r) < tol and rel(dra, dra_r) < 2 * tol and rel(dv, dv_r) < tol


MHA_SHAPES = [
    # B, H, L, S, key mask, dropout multipliers, packed q|k projection
    (2, 8, 300, 300, True, True, True),       # DETR encoder self-attention (C2: 15 x 20 tokens)
    (2, 8, 100, 100, False, True, True),      # decoder self-attention
    (2, 8, 100, 300, True, True, False),      # decoder cross-attention
    (1, 8, 37, 65, True, False, False),       # ragged tiles, S not a multiple of 4
    (2, 4, 12, 12, False, False, True),       # 96 x 128 golden cases: 3 x 4 tokens
    (1, 2, 5, 321, True, True, False),        # odd S with dropout (scalar multiplier loads)
    (1, 8, 100, 1200, True, False, False),    # C5: 960 x 1280 -> 30 x 40 keys
]


@pytest.mark.parametrize("shape", MHA_SHAPES, ids=["%dx%dx%dx%d%s%s%s" % (s[0], s[1], s[2], s[3], "m" * s[4], "d" * s[5], "p" * s[6]) for s in MHA_SHAPES])
def test_mha_flash_forward_backward(dev, shape):
    """gwd_mha_flash_forward / _backward (bf16, matrix cores, no L x S tensor) vs fp32 torch math of
    multi_head_attention.py:329-375 on the same bf16-rounded operands: merged output, log-sum-exp, dq / dk / dv; key-padding
    masks, dropout multipliers, q | k read (and their gradients written) in place in a packed (B, L, 2E) projection."""
    B, H, L, S, masked, drop, packed = shape
    E, dt = 32 * H, torch.bfloat16
    scale = 32 ** -0.5
    qk = rnd(B, L, 2 * E, dtype=dt, seed=1) if packed else None
    q = qk[..., :E] if packed else rnd(B, L, E, dtype=dt, seed=1)
    k = qk[..., E:] if packed else rnd(B, S, E, dtype=dt, seed=2)
    v = rnd(B, S, E, dtype=dt, seed=3)
    go = rnd(B, L, E, dtype=dt, seed=5)
    kpm = None
    if masked:
        kpm = torch.zeros(B, S, dtype=torch.uint8)
        kpm[0, S // 2:] = 1
        kpm[-1, 3] = 1
    mult = None
    if drop:
        g = torch.Generator().manual_seed(4)
        mult = ((torch.rand(B, H, L, S, generator=g) > 0.1).float() / 0.9).to(dt)
    # reference, fp32 autograd
    qf, kf, vf = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    heads = lambda t, n: t.reshape(B, n, H, 32).transpose(1, 2)
    sc = heads(qf, L) @ heads(kf, S).transpose(-2, -1) * scale
    if kpm is not None:
        sc = sc.masked_fill(kpm.bool()[:, None, None, :], float("-inf"))
    P = sc.softmax(-1)
    lse_r = torch.logsumexp(sc, -1)
    o_r = ((P * mult.float() if mult is not None else P) @ heads(vf, S)).transpose(1, 2).reshape(B, L, E)
    o_r.backward(go.float())
    cu = lambda t: None if t is None else t.cuda()
    if packed:
        QK = qk.cuda()
        qc, kc = QK[..., :E], QK[..., E:]
        GQK = torch.full_like(QK, float("nan"))
        gq, gk = GQK[..., :E], GQK[..., E:]
    else:
        qc, kc = q.cuda(), k.cuda()
        gq, gk = torch.full_like(qc, float("nan")), torch.full_like(kc, float("nan"))
    vc = v.cuda()
    out = torch.full((B, L, E), float("nan"), dtype=dt, device="cuda")
    lse = torch.full((B, H, L), float("nan"), device="cuda")
    dev.mha_flash_forward(qc, kc, vc, cu(kpm), cu(mult), out, lse, H, scale)
    gv, delta = torch.full_like(vc, float("nan")), torch.empty_like(lse)
    dev.mha_flash_backward(qc, kc, vc, go.cuda(), out, cu(kpm), cu(mult), lse, delta, gq, gk, gv, H, scale)
    torch.cuda.synchronize()
    tol = TOL[dt]
    assert rel(out, o_r.detach()) < tol, "forward"
    assert float((lse.cpu() - lse_r.detach()).abs().max()) < 2e-2, "log-sum-exp"
    assert rel(gq, qf.grad) < tol, "dq"
    assert rel(gk, kf.grad) < tol, "dk"
    assert rel(gv, vf.grad) < tol, "dv"
    with pytest.raises(RuntimeError):                                   # fp32: not covered, the caller keeps the unfused path
        dev.mha_flash_forward(qc.float().contiguous(), kc.float().contiguous(), vc.float(), None, None, out.float(), lse, H, scale)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(8, 19200, 80), (2, 4800, 30), (1, 37, 7), (3, 1000, 256)])
def test_anchor_depth(dev, shape, dtype):
    """gwd_anchor_depth_forward / _backward vs torch math (pred = sum_r att * anchor; datt, accumulated danchor)."""
    B, P, R = shape
    fake = FakeDevice()
    att = torch.softmax(rnd(B, P, R, seed=1), -1).to(dtype)
    anchor = rnd(B, R, seed=2) + 3.0
    g = rnd(B, P, seed=3)
    pred_r, datt_r, dan_r = torch.empty(B, P), torch.empty(B, P, R, dtype=dtype), rnd(B, R, seed=4)
    dan = dan_r.clone().cuda()
    fake.anchor_depth_forward(att, anchor, pred_r, B, P, R)
    fake.anchor_depth_backward(att, anchor, g, datt_r, dan_r, B, P, R)
    pred, datt = torch.empty(B, P).cuda(), torch.empty(B, P, R, dtype=dtype).cuda()
    dev.anchor_depth_forward(att.cuda(), anchor.cuda(), pred, B, P, R)
    dev.anchor_depth_backward(att.cuda(), anchor.cuda(), g.cuda(), datt, dan, B, P, R)
    torch.cuda.synchronize()
    assert rel(pred, pred_r) < 2e-5 and rel(datt, datt_r) < TOL[dtype] and rel(dan, dan_r) < 2e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,C,ld,gelu", [(777, 60, 64, True), (4800, 30, 32, True), (130, 120, 128, False), (65, 60, 64, False)])
def test_layernorm_with_row_pitch_writes_zero_padding(dev, rows, C, ld, gelu, dtype):
    """LayerNorm over the C real channels of rows that are ld wide: padding never read (poisoned here), written as zeros."""
    fake = FakeDevice()
    x = rnd(rows, ld, dtype=dtype, seed=1, scale=2.0)
    x[:, C:] = float("nan")
    res = rnd(rows, ld, dtype=dtype, seed=5)
    gy = rnd(rows, ld, dtype=dtype, seed=4)
    gy[:, C:] = float("nan")
    ga, be = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    y_r, m_r, r_r = torch.empty_like(x), torch.empty(rows), torch.empty(rows)
    fake.layernorm_forward(x, ga, be, y_r, m_r, r_r, rows, C, gelu, residual=res, ld=ld)
    gx_r, dg_r, db_r = torch.empty_like(x), torch.zeros(C), torch.zeros(C)
    fake.layernorm_backward(gy, x, ga, be, m_r, r_r, gx_r, dg_r, db_r, rows, C, gelu, ld=ld)
    y, m, r = torch.full_like(x, 7.0).cuda(), torch.empty(rows).cuda(), torch.empty(rows).cuda()
    dev.layernorm_forward(x.cuda(), ga.cuda(), be.cuda(), y, m, r, rows, C, gelu, residual=res.cuda(), ld=ld)
    gx, dg, db = torch.full_like(x, 7.0).cuda(), torch.zeros(C).cuda(), torch.zeros(C).cuda()
    dev.layernorm_backward(gy.cuda(), x.cuda(), ga.cuda(), be.cuda(), m, r, gx, dg, db, rows, C, gelu, ld=ld)
    torch.cuda.synchronize()
    assert float(y[:, C:].abs().max()) == 0.0 and float(gx[:, C:].abs().max()) == 0.0
    assert rel(y[:, :C], y_r[:, :C]) < TOL[dtype] and rel(m, m_r) < 1e-5 and rel(r, r_r) < 1e-5
    assert rel(gx[:, :C], gx_r[:, :C]) < TOL[dtype] and rel(dg, dg_r) < TOL[dtype] and rel(db, db_r) < TOL[dtype]


def test_unpad_add_batch(dev):
    torch.manual_seed(5)
    specs = [(60, 9, 1, 60, 64, 64), (120, 9, 5, 60, 64, 128), (30, 1, 1, 120, 128, 32), (30, 9, 1, 30, 32, 32), (7, 3, 2, 5, 8, 7)] * 6   # 30 jobs: two launches
    jobs, want = [], []
    for N, taps, G, Cg, Cgp, Np in specs:
        src = torch.randn(Np, taps, G * Cgp, device="cuda")
        dst = torch.randn(N, taps, G * Cg, device="cuda")
        want.append(dst + src.view(Np, taps, G, Cgp)[:N, :, :, :Cg].reshape(N, taps, G * Cg))
        jobs.append((src, dst.view(-1), N, taps, G, Cg, Cgp))
    dev.unpad_add_batch(jobs)
    torch.cuda.synchronize()
    for (src, dst, N, taps, G, Cg, Cgp), w in zip(jobs, want):
        assert torch.equal(dst.view_as(w), w)


def test_padded_weight_copies_refreshed_by_the_batch_launch(dev):
    """WeightCache with a padding geometry: first pass = torch-built zero-padded copies, later passes = ONE gwd_weight_prep_batch
    launch that rewrites the real entries of the same buffers (padding stays zero)."""
    from gw_depth_amd import ops
    torch.manual_seed(4)
    cases = [((60, 3, 3, 60), (64, 60, 64)), ((120, 3, 3, 300), (128, 60, 64)), ((30, 1, 1, 120), (32, 120, 128)), ((30, 3, 3, 30), (32, 30, 32))]
    ws = [torch.randn(s, device="cuda") for s, _ in cases]
    plain = torch.randn(64, 3, 3, 32, device="cuda")
    cache = ops.WeightCache([(w.data_ptr(), w.data_ptr() + w.numel() * 4) for w in ws + [plain]])
    cache.begin_pass()
    first = [(cache.get(w, None, "fwd", g), cache.get(w, None, "t", g)) for w, (_, g) in zip(ws, cases)]
    assert cache.get(plain, None, "t") is not None
    cache.end_pass()
    for w, (_, g), (f, t) in zip(ws, cases, first):
        assert torch.equal(f, ops.padded_weight(w, g, "fwd", torch.bfloat16)) and torch.equal(t, ops.padded_weight(w, g, "t", torch.bfloat16))
    for w in ws + [plain]:
        w.mul_(-0.75).add_(0.125)
    cache.begin_pass()
    for w, (_, g), (f, t) in zip(ws, cases, first):
        assert cache.get(w, None, "fwd", g) is f and cache.get(w, None, "t", g) is t          # same buffers, refreshed in place
        assert torch.equal(f, ops.padded_weight(w, g, "fwd", torch.bfloat16)) and torch.equal(t, ops.padded_weight(w, g, "t", torch.bfloat16))
    cache.end_pass()


@pytest.mark.parametrize("spec", [(2, 20, 24, 60, 60, 3, (64, 60, 64), 1), (1, 20, 24, 300, 120, 3, (128, 60, 64), 5), (2, 20, 24, 120, 30, 1, (32, 120, 128), 1),
                                  (8, 60, 80, 60, 60, 3, (64, 60, 64), 1)])
def test_padded_conv_equals_the_plain_conv(dev, spec):
    """ops.conv2d_padded on zero-padded activations == ops.conv2d on the real channels: output (padding zero), input gradient
    (padding zero), weight gradient in the parameter's own shape."""
    from gw_depth_amd import ops
    B, H, W, Ci, Co, K, geom, G = spec
    Np, Cg, Cgp = geom
    torch.manual_seed(6)
    w = (torch.randn(Co, K, K, Ci, device="cuda") * (K * K * Ci) ** -0.5).requires_grad_(True)
    x = torch.randn(B, H, W, Ci, device="cuda").bfloat16()
    xp = torch.zeros(B, H, W, G * Cgp, device="cuda", dtype=torch.bfloat16)
    xp.view(B, H, W, G, Cgp)[..., :Cg] = x.view(B, H, W, G, Cg)
    x.requires_grad_(True)
    xp.requires_grad_(True)
    gy = torch.randn(B, H, W, Co, device="cuda").bfloat16()
    y = ops.conv2d(x, w, pad=K // 2)
    y.backward(gy)
    gw0, gx0 = w.grad.clone(), x.grad.clone()
    w.grad = None
    yp = ops.conv2d_padded(xp, w, K // 2, geom)
    gyp = torch.zeros(B, H, W, Np, device="cuda", dtype=torch.bfloat16)
    gyp[..., :Co] = gy
    yp.backward(gyp)
    torch.cuda.synchronize()
    assert yp.shape[-1] == Np and float(yp[..., Co:].abs().max()) == 0.0
    assert rel(yp[..., :Co], y) < 2e-3                            # same products, another summation order
    gxp = xp.grad.view(B, H, W, G, Cgp)
    assert float(gxp[..., Cg:].abs().max()) == 0.0 and rel(gxp[..., :Cg].reshape(B, H, W, Ci), gx0) < 2e-3
    assert w.grad.shape == gw0.shape and rel(w.grad, gw0) < 2e-3


@pytest.mark.parametrize("shape", [(2, 96, 128), (1, 97, 131), (3, 50, 33), (8, 480, 640), (1, 7, 9)])
def test_fused_stem_equals_conv_bn_relu_maxpool(dev, shape):
    """gwd_stem_forward (conv 7x7 s2 p3 + folded BN + ReLU + max-pool 3x3 s2 p1 in one kernel) against the same chain in torch on
    the same bf16-rounded operands; ragged sizes exercise the conv and the pool padding on every border."""
    B, H, W = shape
    fake = FakeDevice()
    torch.manual_seed(7)
    w = torch.randn(64, 7, 7, 3) * 0.1
    scale, shift = torch.rand(64) + 0.5, torch.randn(64) * 0.2
    x = torch.randn(B, H, W, 3).bfloat16()
    Hp, Wp = hip.stem_out(H), hip.stem_out(W)
    pk_r, y_r = torch.empty(hip.STEM_PACKED_ELEMS, dtype=torch.bfloat16), torch.empty(B, Hp, Wp, 64, dtype=torch.bfloat16)
    fake.stem_pack(w, scale, pk_r)
    fake.stem_forward(x, pk_r, shift, y_r)
    pk, y = torch.empty(hip.STEM_PACKED_ELEMS, dtype=torch.bfloat16, device="cuda"), torch.full((B, Hp, Wp, 64), -1.0, dtype=torch.bfloat16, device="cuda")
    dev.stem_pack(w.cuda(), scale.cuda(), pk)
    dev.stem_forward(x.cuda(), pk, shift.cuda(), y)
    torch.cuda.synchronize()
    assert torch.equal(pk.cpu(), pk_r)
    assert float(y.min()) >= 0.0 and rel(y, y_r) < 5e-3
    assert float((y.float().cpu() - y_r.float()).abs().max()) < 0.05 * float(y_r.float().abs().max())


@pytest.mark.parametrize("shape", [(8, 480, 640, 120, 160, 16), (8, 480, 640, 15, 20, 128), (2, 96, 128, 3, 4, 64), (3, 100, 131, 25, 33, 32), (1, 960, 1280, 240, 320, 16)])
@pytest.mark.parametrize("normalize", [False, True])
def test_mask_levels_and_sine_position_embedding(dev, shape, normalize):
    """gwd_pos_sine (level mask by nearest resize, cumulative counts, sin / cos embedding) against the torch chain of the reference
    (backbone.py:81-88, position_encoding.py:28-48) on ragged per-image paddings."""
    from gw_depth_amd import ops
    from gw_depth_amd.model import pos_sine
    import torch.nn.functional as F
    B, H, W, h, w, npf = shape
    pad = torch.zeros(B, H, W, dtype=torch.bool)
    g = torch.Generator().manual_seed(B * H + w)
    for b in range(1, B):                                         # image 0 unpadded, the others padded on the right / bottom
        pad[b, int(H * (0.5 + 0.5 * torch.rand(1, generator=g))):, :] = True
        pad[b, :, int(W * (0.5 + 0.5 * torch.rand(1, generator=g))):] = True
    m_ref = F.interpolate(pad[None].float(), size=(h, w)).to(torch.bool)[0]
    p_ref = pos_sine(m_ref, npf, normalize)                       # torch path (no counts attached)
    (m,) = ops.mask_levels(pad.cuda(), [(h, w)])
    p = pos_sine(m, npf, normalize)
    torch.cuda.synchronize()
    assert m.dtype == torch.bool and torch.equal(m.cpu(), m_ref)
    assert p.shape == p_ref.shape and float((p.cpu() - p_ref).abs().max()) < 2e-5


@pytest.mark.parametrize("shape", [(2, 30, 40, 64, (16, 8, 4, 2)), (8, 120, 160, 160, (16, 8, 4, 2)), (1, 17, 23, 8, (4, 2))])
def test_pyramid_concat_equals_cat_of_upsampled_branches(dev, shape):
    """ops.pyramid_concat (up-sampling kernels write their channel slice of the concat, backward reads the slices in place) ==
    torch.cat of the separately up-sampled branches, forward bit for bit, gradients to the same tolerance as the resample tests."""
    from gw_depth_amd import ops
    B, H, W, C, pools = shape
    torch.manual_seed(9)
    x = torch.randn(B, H, W, C, device="cuda").bfloat16().requires_grad_(True)
    ys = [torch.randn(B, max(H // k, 1), max(W // k, 1), C, device="cuda").bfloat16().requires_grad_(True) for k in pools]
    ref = torch.cat([x] + [ops.upsample_bilinear_ac(y, (H, W)) for y in ys], dim=-1)
    g = torch.randn_like(ref)
    ref.backward(g)
    want = [t.grad.clone() for t in [x] + ys]
    for t in [x] + ys:
        t.grad = None
    out = ops.pyramid_concat(x, ys)
    out.backward(g)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    for t, wv in zip([x] + ys, want):
        assert torch.equal(t.grad, wv)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(8, 120, 160, 160), (2, 60, 80, 64), (1, 16, 16, 8), (3, 37, 50, 24)])
def test_psp_pools_equal_four_average_pools(dev, shape, dtype):
    """ops.psp_pools (one pass forward, one pass backward incl. the gradient of the map itself read from a wider concat gradient)
    against four F.avg_pool2d calls (floor semantics at ragged sizes) and autograd's sum of their gradients."""
    from gw_depth_amd import ops
    import torch.nn.functional as F
    B, H, W, C = shape
    torch.manual_seed(10)
    x = torch.randn(B, H, W, C, device="cuda").to(dtype).requires_grad_(True)
    xp, pooled = ops.psp_pools(x, (16, 8, 4, 2))
    xr = x.detach().float().permute(0, 3, 1, 2).requires_grad_(True)
    ref = [F.avg_pool2d(xr, k, k) for k in (16, 8, 4, 2)]
    wide = torch.randn(B, H, W, 3 * C, device="cuda").to(dtype)                     # the concat's gradient: x's share is a channel slice
    gs = [torch.randn_like(p) for p in pooled]
    torch.autograd.backward([xp] + pooled, [wide[..., :C]] + gs)
    torch.autograd.backward(ref, [g.float().permute(0, 3, 1, 2) for g in gs])
    want = xr.grad.permute(0, 2, 3, 1) + wide[..., :C].float()
    torch.cuda.synchronize()
    assert torch.equal(xp, x)
    for p, r in zip(pooled, ref):
        assert p.shape == r.permute(0, 2, 3, 1).shape and rel(p, r.permute(0, 2, 3, 1)) < TOL[dtype] * 0.3
    assert rel(x.grad, want) < TOL[dtype] * 0.3


@pytest.mark.parametrize("sizes", [[7] * 8, [1, 12, 0, 5], [64, 3]])
def test_fused_set_criterion_equals_the_torch_formulation(dev, sizes, monkeypatch):
    """gwd_match_cost + gwd_lsap + gwd_set_losses_* (one autograd node) against forward_packed's torch formulation (criteria.FUSED_SETLOSS = False):
    same assignment, same 2 x layers loss terms, same gradients w.r.t. logits and lines - ragged target counts incl. an empty image."""
    from gw_depth_amd.criteria import HungarianMatcherLine as HungarianMatcher, SetCriterion, pack_targets
    torch.manual_seed(11)
    L_, B, Q = 6, len(sizes), 100
    crit = SetCriterion(1, {}, 0.1, ["lines_labels", "lines"], HungarianMatcher(1.0, 5.0)).cuda()
    targets = [{"labels": torch.zeros(n, dtype=torch.int64, device="cuda"), "lines": torch.rand(n, 6, device="cuda")} for n in sizes]
    packed = pack_targets(targets, "cuda")
    logits0, lines0 = torch.randn(L_, B, Q, 2, device="cuda"), torch.rand(L_, B, Q, 6, device="cuda")
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setattr("gw_depth_amd.criteria.FUSED_SETLOSS", mode == "1")
        lg, ln = logits0.clone().requires_grad_(True), lines0.clone().requires_grad_(True)
        outs = {"pred_logits": lg[0], "pred_lines": ln[0], "aux_outputs": [{"pred_logits": lg[i], "pred_lines": ln[i]} for i in range(1, L_)]}
        losses = crit.forward_packed(outs, packed)
        total = sum(v * (1.0 + 0.1 * i) for i, (k, v) in enumerate(sorted(losses.items())))
        total.backward()
        torch.cuda.synchronize()
        res[mode] = ({k: float(v) for k, v in losses.items()}, lg.grad.clone(), ln.grad.clone(), crit.last_query_of_target.clone())
    a, b = res["1"], res["0"]
    assert torch.equal(a[3], b[3])
    assert set(a[0]) == set(b[0]) and len(a[0]) == 2 * L_
    for k in a[0]:
        assert abs(a[0][k] - b[0][k]) <= 1e-5 * max(1.0, abs(b[0][k])), (k, a[0][k], b[0][k])
    assert rel(a[1], b[1]) < 1e-5 and rel(a[2], b[2]) < 1e-5


GATE_CASES = [
    # B, H, W, Cin(gy channels), Cout(gx channels), K, stride of the forward layer, residual
    (2, 12, 16, 64, 256, 1, 1, True),            # 1x1 data gradient + skip gradient (Bottleneck conv1 with the fan-out)
    (2, 12, 16, 128, 128, 3, 1, False),          # 3x3 transposed gather
    (2, 12, 16, 128, 128, 3, 2, False),          # 3x3 stride 2 (Bottleneck conv2 of a layer's first block)
    (8, 120, 160, 160, 160, 3, 1, False),        # the 256 x 160 tiles
    (2, 12, 16, 30, 60, 3, 1, False),            # odd widths: register-staged kernel, scalar epilogue
    (2, 256, 320, 32, 64, 3, 1, True),           # halo-tile kernel (>= 131 072 pixels, 64 <- 32 channels)
    (2, 64, 80, 1, 32, 3, 1, False),             # thin data gradient (depth head)
    (3, 37, 45, 2, 32, 3, 1, False),             # thin data gradient (seg head), ragged tiles
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("gate_act", [hip.ACT_RELU, hip.ACT_ELU])
@pytest.mark.parametrize("case", GATE_CASES, ids=["%dx%dx%dx%d_%d_k%ds%d%s" % (c[0], c[1], c[2], c[3], c[4], c[5], c[6], "r" * c[7]) for c in GATE_CASES])
def test_conv_data_gradient_with_activation_gate(dev, case, gate_act, dtype):
    """gwd_conv_desc.gate: the epilogue's last step multiplies by act'(.) of the activation whose output is `gate` - a data-gradient
    launch then returns the gradient w.r.t. the producer's pre-activation value (every kernel family that can run a data gradient)."""
    B, H, W, Cg, Cx, K, stride, with_res = case
    if dtype == torch.float32 and B * H * W > 40000:
        pytest.skip("big maps in bf16 only")
    pad = K // 2
    Ho, Wo = conv_out(H, K, stride, pad), conv_out(W, K, stride, pad)
    gy = rnd(B, Ho, Wo, Cg, dtype=dtype, seed=1)
    wt = rnd(Cx, K, K, Cg, dtype=dtype, seed=2, scale=0.1)          # the transposed (data-gradient) weight layout
    xin = rnd(B, H, W, Cx, dtype=dtype, seed=3)                     # the layer's input = output of the producer's activation
    if gate_act == hip.ACT_RELU:
        xin = xin.clamp_min(0)
    else:
        xin = torch.where(xin > 0, xin, torch.expm1(xin.float()).to(dtype))
    res = rnd(B, H, W, Cx, dtype=dtype, seed=4) if with_res else None
    dims = (B, Ho, Wo, Cg, H, W, Cx, K, K)
    want = torch.empty(B, H, W, Cx)
    FakeDevice().conv_forward(gy, wt, want, dims, stride=stride, pad=pad, gather=hip.GATHER_TRANSPOSED, residual=res, gate=xin, gate_act=gate_act)
    got = torch.empty(B, H, W, Cx, dtype=dtype, device="cuda")
    dev.conv_forward(gy.cuda(), wt.cuda(), got, dims, stride=stride, pad=pad, gather=hip.GATHER_TRANSPOSED,
                     residual=None if res is None else res.cuda(), gate=xin.cuda(), gate_act=gate_act)
    torch.cuda.synchronize()
    assert rel(got, want) < TOL[dtype]
    if gate_act == hip.ACT_RELU:                                    # exactly zero wherever the producer's output was zero
        assert torch.equal(got.cpu() == 0, (xin <= 0) | (got.cpu() == 0)) and bool((got.cpu()[xin <= 0] == 0).all())


def test_conv_gate_rejects_what_it_cannot_do(dev):
    x, w, y = (torch.zeros(s, device="cuda", dtype=torch.bfloat16) for s in ((1, 8, 8, 32), (32, 1, 1, 32), (1, 8, 8, 32)))
    dims = (1, 8, 8, 32, 8, 8, 32, 1, 1)
    with pytest.raises(RuntimeError):
        dev.conv_forward(x, w, y, dims, gate=x, gate_act=hip.ACT_SIGMOID)
    with pytest.raises(RuntimeError):
        dev.conv_forward(x, w, y, dims, gate=x, gate_act=hip.ACT_RELU, mult=x)
    # a GELU gate (from the producer's pre-activation tensor) exists in the LDS-DMA tile kernels only: elsewhere the call says so (False)
    xf, wf, yf = x.float(), w.float(), y.float()
    assert dev.conv_forward(xf, wf, yf, dims, gate=xf, gate_act=hip.ACT_GELU) is False


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,cin,hidden", [(11760, 384, 768), (2400, 64, 128), (777, 64, 128), (3528, 512, 1024)])
def test_mlp_gelu_backward_in_the_data_gradient_epilogue(dev, rows, cin, hidden, dtype):
    """layers.Mlp with GELU's backward as the gate of fc2's data-gradient GEMM (gwd_conv_desc.gate = fc1's pre-activation tensor,
    gate_act GELU; bf16: dma_tile<..., GATE = 2>, fp32: the call refuses and the gate runs as a pass of its own) against the same MLP
    with the separate activation-backward pass: same output, same gradients for the input, both weights and both biases."""
    from gw_depth_amd import layers
    torch.manual_seed(3)
    mlp = layers.Mlp(cin, hidden).cuda()
    with torch.no_grad():
        mlp.fc1.bias.normal_(0, 0.1)
        mlp.fc2.bias.normal_(0, 0.1)
        mlp.fc1.weight.mul_(3.0)
    x0 = torch.randn(rows, cin, device="cuda").to(dtype)
    res0 = torch.randn(rows, cin, device="cuda").to(dtype)
    got = {}
    for mode in (True, False):
        layers.GELU_GATE = mode
        try:
            for p_ in mlp.parameters():
                p_.grad = None
            x = x0.clone().requires_grad_(True)
            r = res0.clone().requires_grad_(True)
            out = mlp(x, residual=r)
            (out.float() * torch.linspace(-1, 1, cin, device="cuda")).sum().backward()
            torch.cuda.synchronize()
            got[mode] = [out.detach().float(), x.grad.float(), r.grad.float()] + [p_.grad.float() for p_ in mlp.parameters()]
        finally:
            layers.GELU_GATE = True
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for a_, b_ in zip(got[True], got[False]):
        assert rel(a_, b_) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", [hip.ACT_RELU, hip.ACT_ELU])
def test_deferred_activation_backward_equals_the_separate_pass(dev, act, dtype):
    """ops.conv2d(defer=True) -> ops.conv2d(in_gate=act) (+ the fan-out skip, + an upsampled consumer) against the same chain with
    every layer running its own activation backward: same outputs, same gradients for the input and all weights."""
    from gw_depth_amd import ops
    torch.manual_seed(12)
    B, H, W, C = 2, 24, 32, 64
    x0 = torch.randn(B, H, W, C, device="cuda").to(dtype)
    ws = [(torch.randn(C, k, k, C, device="cuda") * (0.3 / k)) for k in (1, 3, 3)]
    b0 = torch.randn(C, device="cuda") * 0.1
    res = {}
    for mode in (True, False):
        x = x0.clone().requires_grad_(True)
        w = [t.clone().requires_grad_(True) for t in ws]
        b = b0.clone().requires_grad_(True)
        g = act if mode else hip.ACT_NONE
        h = ops.conv2d(x, w[0], b, act=act, defer=mode)                             # producer with a bias: its gradient moves to the column-sum queue
        h2, h = ops.conv2d(h, w[1], pad=1, act=act, in_gate=g, defer=mode, fanout=True)       # consumer + producer, input used twice
        out = ops.conv2d(h2, w[2], pad=1, in_gate=g, upsample_to=(2 * H, 2 * W))    # upsampled consumer: gate after the footprint sum
        loss = (out.float() ** 2).mean() + (h.float() * 0.37).sum() * 1e-3           # second consumer of h behind the fan-out
        loss.backward()
        torch.cuda.synchronize()
        res[mode] = [out.detach().float(), x.grad.float(), b.grad.float()] + [t.grad.float() for t in w]
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for a, r in zip(res[True], res[False]):
        assert rel(a, r) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,C,skip", [(614400 // 64, 64, False), (777, 64, True), (130, 160, True), (65, 24, False), (33, 7, False)])
def test_layernorm_backward_applies_the_elu_gate_of_its_input(dev, rows, C, skip, dtype):
    """GWD_LN_ELU_INPUT: gx = (LN backward + gskip) * elu'(x), x = the ELU output the norm was applied to (C = 7: no vector kernel,
    the binding reports it and ops falls back to a separate pass)."""
    x = rnd(rows, C, dtype=dtype, seed=1)
    x = torch.where(x > 0, x, torch.expm1(x.float()).to(dtype))
    gy, ga, be = rnd(rows, C, dtype=dtype, seed=2), rnd(C, seed=3) + 1.0, rnd(C, seed=4)
    gs = rnd(rows, C, dtype=dtype, seed=5) if skip else None
    fake = FakeDevice()
    y, mean, rstd = torch.empty(rows, C), torch.empty(rows), torch.empty(rows)
    fake.layernorm_forward(x, ga, be, y, mean, rstd, rows, C, False)
    want, dg_r, db_r = torch.empty(rows, C), torch.zeros(C), torch.zeros(C)
    fake.layernorm_backward(gy, x, ga, be, mean, rstd, want, dg_r, db_r, rows, C, False, gskip=gs, elu_input=True)
    got, dg, db = torch.empty(rows, C, dtype=dtype, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    done = dev.layernorm_backward(gy.cuda(), x.cuda(), ga.cuda(), be.cuda(), mean.cuda(), rstd.cuda(), got, dg, db, rows, C, False,
                                  gskip=None if gs is None else gs.cuda(), elu_input=True)
    torch.cuda.synchronize()
    if C == 7:
        assert done is False
        return
    assert done is True
    assert rel(got, want) < TOL[dtype] and rel(dg, dg_r) < TOL[dtype] * 2 and rel(db, db_r) < TOL[dtype] * 2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("gate_act", [hip.ACT_RELU, hip.ACT_ELU])
@pytest.mark.parametrize("shape", [(2, 12, 16, 24, 32, 64), (1, 9, 11, 24, 32, 32), (2, 30, 40, 60, 80, 12)])
def test_nearest_upsample_backward_with_activation_gate(dev, shape, gate_act, dtype):
    B, Hs, Ws, Ho, Wo, C = shape
    gy = rnd(B, Ho, Wo, C, dtype=dtype, seed=1)
    src = rnd(B, Hs, Ws, C, dtype=dtype, seed=2)
    src = src.clamp_min(0) if gate_act == hip.ACT_RELU else torch.where(src > 0, src, torch.expm1(src.float()).to(dtype))
    want = torch.empty(B, Hs, Ws, C)
    FakeDevice().resample_backward(gy, want, B, Hs, Ws, Ho, Wo, C, hip.RESAMPLE_NEAREST, gate=src, gate_act=gate_act)
    got = torch.empty(B, Hs, Ws, C, dtype=dtype, device="cuda")
    assert dev.resample_backward(gy.cuda(), got, B, Hs, Ws, Ho, Wo, C, hip.RESAMPLE_NEAREST, gate=src.cuda(), gate_act=gate_act) is True
    torch.cuda.synchronize()
    assert rel(got, want) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_deferred_elu_chain_of_the_decoder_equals_the_separate_passes(dev, dtype):
    """upconv (ELU, defer) -> LayerNorm(in_gate) -> conv (ELU, defer) -> upconv(in_gate; ELU, defer) -> conv(in_gate; ELU, defer) ->
    thin head(in_gate): the decoder branch of model.DensePrediction with and without the deferred activation backward."""
    from gw_depth_amd import ops
    torch.manual_seed(13)
    B, H, W, C = 2, 20, 24, 64
    x0 = torch.randn(B, H, W, C, device="cuda").to(dtype)
    ws = [torch.randn(co, 3, 3, ci, device="cuda") * (0.5 / (3 * ci ** 0.5)) * 3 for co, ci in ((64, 64), (64, 64), (32, 64), (32, 32), (1, 32))]
    ga, be = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
    res = {}
    for mode in (True, False):
        x = x0.clone().requires_grad_(True)
        w = [t.clone().requires_grad_(True) for t in ws]
        g_, b_ = ga.clone().requires_grad_(True), be.clone().requires_grad_(True)
        g = hip.ACT_ELU if mode else hip.ACT_NONE
        u1 = ops.layer_norm(ops.conv2d(x, w[0], pad=1, act=hip.ACT_ELU, upsample_to=(2 * H, 2 * W), defer=mode), g_, b_, in_gate=g)
        c1 = ops.conv2d(u1, w[1], pad=1, act=hip.ACT_ELU, defer=mode)
        u2 = ops.conv2d(c1, w[2], pad=1, act=hip.ACT_ELU, upsample_to=(4 * H, 4 * W), in_gate=g, defer=mode)
        c2 = ops.conv2d(u2, w[3], pad=1, act=hip.ACT_ELU, in_gate=g, defer=mode)
        out = ops.conv2d(c2, w[4], pad=1, act=hip.ACT_SIGMOID, act_scale=10.0, in_gate=g)
        (out.float() ** 2).mean().backward()
        torch.cuda.synchronize()
        res[mode] = [out.detach().float(), x.grad.float(), g_.grad.float(), b_.grad.float()] + [t.grad.float() for t in w]
    tol = 1e-5 if dtype == torch.float32 else 3e-2
    for a, r in zip(res[True], res[False]):
        assert rel(a, r) < tol


KSPLIT_CASES = [
    # rows, K, N, extras  (K >= 1024, 1x1: gemm_ksplit_kernel)
    (800, 2048, 256, dict(shift=True, residual=True, mult=True)),                  # FFN linear2: dropout + skip in the epilogue
    (2400, 2048, 256, dict(shift=True)),
    (777, 1024, 40, dict(shift=True, act=hip.ACT_RELU, residual=True)),            # ragged rows, a partial column tile
    (2400, 2048, 512, dict(scale=True, shift=True, act=hip.ACT_RELU, gate=hip.ACT_RELU)),
    (9600, 1024, 256, dict(shift=True, act=hip.ACT_GELU, z=True)),
    (130, 1056, 72, dict(gather=hip.GATHER_TRANSPOSED, residual=True, gate=hip.ACT_ELU)),
]


@pytest.mark.parametrize("case", KSPLIT_CASES, ids=["%dx%dx%d_%s" % (c[0], c[1], c[2], "_".join(sorted(c[3]))) for c in KSPLIT_CASES])
def test_long_reduction_gemm_split_over_waves(dev, case, monkeypatch):
    """gemm_ksplit_kernel (1x1 layers with K >= 1024 on few rows) against fp32 math."""
    M, K, N, ex = case
    dt = torch.bfloat16
    x, w = rnd(M, 1, 1, K, dtype=dt, seed=1), rnd(N, 1, 1, K, dtype=dt, seed=2, scale=K ** -0.5)
    kw, kw_gpu = {}, {}
    for name, shape in (("scale", (N,)), ("shift", (N,))):
        if ex.get(name):
            kw[name] = rnd(*shape, seed=3 + len(name)) * 0.5 + (1.0 if name == "scale" else 0.0)
    for name in ("residual", "mult"):
        if ex.get(name):
            kw[name] = rnd(M, 1, 1, N, dtype=dt, seed=7 + len(name))
    if ex.get("gate"):
        g = rnd(M, 1, 1, N, dtype=dt, seed=11)
        kw["gate"], kw["gate_act"] = (g.clamp_min(0) if ex["gate"] == hip.ACT_RELU else torch.where(g > 0, g, torch.expm1(g.float()).to(dt))), ex["gate"]
    for k in ("act", "gather"):
        if k in ex:
            kw[k] = ex[k]
    dims = (M, 1, 1, K, 1, 1, N, 1, 1)
    want, zw = torch.empty(M, 1, 1, N), (torch.empty(M, 1, 1, N) if ex.get("z") else None)
    FakeDevice().conv_forward(x, w, want, dims, z=zw, **kw)
    got = torch.empty(M, 1, 1, N, dtype=dt, device="cuda")
    zg = torch.empty(M, 1, 1, N, dtype=dt, device="cuda") if ex.get("z") else None
    dev.conv_forward(x.cuda(), w.cuda(), got, dims, z=zg, **{k: (v.cuda() if torch.is_tensor(v) else v) for k, v in kw.items()})
    torch.cuda.synchronize()
    assert rel(got, want) < TOL[dt]
    if zg is not None:
        assert rel(zg, zw) < TOL[dt]


ACTK_TILES = [
    # rows, K, N -> tile family of the LDS-DMA dispatcher
    (131072, 64, 128),      # 256 x 128 (big)
    (131072, 32, 160),      # 256 x 160 (big, 160-wide: activation-free variant or the run-time switch only)
    (40000, 64, 128),       # 128 x 128
    (9600, 64, 256),        # 64 x 64 quarter tiles
    (40000, 128, 64),       # 128 x 64
    (40000, 64, 32),        # 128 x 32
]


@pytest.mark.parametrize("act", [hip.ACT_NONE, hip.ACT_RELU, hip.ACT_GELU, hip.ACT_ELU, hip.ACT_SIGMOID])
@pytest.mark.parametrize("tile", ACTK_TILES, ids=["%dx%dx%d" % t for t in ACTK_TILES])
def test_compile_time_activation_variants_of_every_tile_family(dev, tile, act):
    """igemm_dma_kernel<..., ACTK>: none / ReLU / GELU (+ pre-activation copy) as compile-time epilogues, ELU / sigmoid through the
    run-time switch - every tile family the dispatcher can pick, plain and transposed gather, with shift and skip."""
    M, K, N = tile
    dt = torch.bfloat16
    x, w = rnd(M, 1, 1, K, dtype=dt, seed=1), rnd(N, 1, 1, K, dtype=dt, seed=2, scale=K ** -0.5)
    shift, res = rnd(N, seed=3) * 0.3, rnd(M, 1, 1, N, dtype=dt, seed=4)
    dims = (M, 1, 1, K, 1, 1, N, 1, 1)
    for gather in (hip.GATHER_CONV, hip.GATHER_TRANSPOSED):
        want = torch.empty(M, 1, 1, N)
        zw = torch.empty(M, 1, 1, N) if act == hip.ACT_GELU else None
        FakeDevice().conv_forward(x, w, want, dims, z=zw, shift=shift, residual=res, act=act, gather=gather)
        got = torch.empty(M, 1, 1, N, dtype=dt, device="cuda")
        zg = torch.empty(M, 1, 1, N, dtype=dt, device="cuda") if zw is not None else None
        dev.conv_forward(x.cuda(), w.cuda(), got, dims, z=zg, shift=shift.cuda(), residual=res.cuda(), act=act, gather=gather)
        torch.cuda.synchronize()
        assert rel(got, want) < TOL[dt], gather
        if zg is not None:
            assert rel(zg, zw) < TOL[dt], gather


# ---------------------------------------------------------------------------------------------------------------------------------
# gwd_bmm: the library's own strided batched GEMM (point-head affinity maps, fp32-mode attention products) against torch.matmul
BMM_CASES = [
    # name, a shape, b shape, trans_b, dtype, tol
    ("affinity_1/4", (8, 19200, 64), (8, 80, 64), True, torch.bfloat16, 1.5e-2),
    ("affinity_1/8_padded", (8, 4800, 128), (8, 32, 128), True, torch.bfloat16, 1.5e-2),
    ("attention_scores_fp32", (8, 8, 300, 32), (8, 8, 300, 32), True, torch.float32, 2e-5),
    ("attention_mix_fp32", (8, 8, 100, 300), (8, 8, 300, 32), False, torch.float32, 2e-5),
    ("ragged_unaligned_fp32", (3, 77, 30), (3, 45, 30), True, torch.float32, 2e-5),
    ("ragged_unaligned_bf16", (2, 3, 131, 27), (2, 3, 27, 19), False, torch.bfloat16, 1.5e-2),
    ("broadcast_b", (4, 2, 70, 64), (1, 2, 64, 40), False, torch.float32, 2e-5),
    ("anchor_column", (8, 4800, 300), (8, 300, 1), False, torch.float32, 2e-5),
]


@pytest.mark.parametrize("case", BMM_CASES, ids=[c[0] for c in BMM_CASES])
def test_own_batched_gemm_forward_and_gradients(case):
    from gw_depth_amd import ops
    name, ash, bsh, trans_b, dt, tol = case
    g = torch.Generator().manual_seed(7)
    a0 = (torch.randn(*ash, generator=g) * 0.5).to(dt)
    b0 = (torch.randn(*bsh, generator=g) * 0.5).to(dt)
    alpha = 0.37
    ar, br = a0.double().requires_grad_(True), b0.double().requires_grad_(True)
    ref = alpha * (ar @ (br.transpose(-1, -2) if trans_b else br))
    gy = torch.randn(ref.shape, generator=g).to(dt)
    ref.backward(gy.double())
    a, b = a0.cuda().requires_grad_(True), b0.cuda().requires_grad_(True)
    y = (ops.matmul_nt if trans_b else ops.matmul_nn)(a, b, alpha=alpha)
    assert y.shape == ref.shape and y.dtype == dt
    y.backward(gy.cuda())
    torch.cuda.synchronize()

    def err(got, want):
        return float((got.double().cpu() - want).norm() / (want.norm() + 1e-30))
    assert err(y.detach(), ref.detach()) < tol, (name, "forward", err(y.detach(), ref.detach()))
    assert err(a.grad, ar.grad) < tol, (name, "d a", err(a.grad, ar.grad))
    assert err(b.grad, br.grad) < tol, (name, "d b", err(b.grad, br.grad))


def test_own_batched_gemm_on_strided_head_views():
    """The fp32 attention path hands over (B, H, L, hd) views of packed (B, L, 2E) projections: no copies, unit inner stride only."""
    from gw_depth_amd import ops
    B, L, S, H, hd = 2, 50, 70, 8, 32
    E = H * hd
    g = torch.Generator().manual_seed(9)
    qk = torch.randn(B, L, 2 * E, generator=g).cuda()
    kk = torch.randn(B, S, 2 * E, generator=g).cuda()
    q = qk[..., :E].reshape(B, L, H, hd).transpose(1, 2)
    k = kk[..., E:].reshape(B, S, H, hd).transpose(1, 2)
    got = ops.matmul_nt(q, k)
    want = q.double() @ k.double().transpose(-1, -2)
    assert float((got.double() - want).norm() / want.norm()) < 2e-5
